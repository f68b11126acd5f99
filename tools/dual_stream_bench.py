#!/usr/bin/env python3
"""Experiment: one batch as S concurrent sub-batches, each on its own HIP stream (own experiment object, shared weights), against
the whole batch on one stream.  Samples are independent, so results are identical; what changes is that the
latency-bound small-map launches of one sub-batch run beside the other's instead of leaving most CUs idle.

    python3 tools/dual_stream_bench.py [--config adm256|celebahq256] [--streams 2] [--graph] [--timesteps 10]
"""
import argparse
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="adm256")
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--timesteps", type=int, default=10)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    S = args.streams
    mk = lambda batch: argparse.Namespace(tiny=False, batch=batch, timesteps=args.timesteps, dry_run=False, dtype=args.dtype)
    full_b = 16 if args.config == "adm256" else 8
    whole = bench.WORKLOADS[args.config](mk(full_b), dev, bench.PRECISIONS[args.dtype])
    parts = [bench.WORKLOADS[args.config](mk(full_b // S), dev, bench.PRECISIONS[args.dtype]) for _ in range(S)]
    for w in [whole] + parts:
        w.exp.use_graphs = args.graph
    x = whole.inputs(1, 1, 0)[0]
    xs = list(x.chunk(S))
    streams = [torch.cuda.Stream() for _ in range(S)]
    if args.config == "celebahq256":          # the bound constraint is per batch shape: each part binds its own slice of y
        pass

    def run_whole():
        return whole.run(x)

    def run_parts():
        outs = [None] * S
        def work(i):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                outs[i] = parts[i].run(xs[i].contiguous())
        th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        return torch.cat(outs)

    a = run_whole(); b = run_parts()                     # warm-up (graph capture, workspaces)
    torch.cuda.synchronize()
    if args.config == "adm256":
        print("identical results:", bool(torch.equal(a, b)))
    for r in range(args.rounds):
        for name, fn in (("whole batch, 1 stream", run_whole), (f"{S} sub-batches, {S} streams", run_parts)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"round {r}: {name:28s} {1e3 * dt / args.timesteps:8.2f} ms per timestep  ({full_b / dt * args.timesteps / (100 if args.config == 'celebahq256' else 50):.2f} images/s at the full schedule)", flush=True)


if __name__ == "__main__":
    main()
