#!/usr/bin/env python3
"""Does running the top-resolution ResBlocks sub-batch by sub-batch keep the activations in the 256 MB Infinity Cache?

A chain conv -> 3 x [GN+SiLU, conv, GN+SiLU, conv(+residual)] on [B, H, H, C] bf16 tensors, whole batch at once vs in sub-batches
(each sub-batch runs the WHOLE chain before the next one starts), each variant captured in one hipGraph and replayed.
Same arithmetic either way (GroupNorm and the convolutions are per sample).

    python tools/chain_bench.py [--res 256] [--ch 256] [--batch 16]
"""
import argparse
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--ch", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--subs", default="16,8,4,2,1")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    H, C, B = args.res, args.ch, args.batch
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    nconv = 1 + 2 * args.blocks
    pws = [ops.pack_conv(torch.randn(C, C, 3, 3) / math.sqrt(9 * C), torch.zeros(C), dt, dev) for _ in range(nconv)]
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)

    def chain(xs):
        h = ops.conv2d(xs, pws[0])
        for i in range(args.blocks):
            a = ops.groupnorm(h, gam, bet, groups=32, eps=1e-5, silu=True)
            a = ops.conv2d(a, pws[1 + 2 * i])
            a = ops.groupnorm(a, gam, bet, groups=32, eps=1e-5, silu=True)
            h = ops.conv2d(a, pws[2 + 2 * i], res=h)
        return h

    fl = 2.0 * B * H * H * C * C * 9 * nconv
    ref = None
    for sb in [int(s) for s in args.subs.split(",")]:
        if sb > B or B % sb:
            continue
        def run():
            return [chain(x[i:i + sb]) for i in range(0, B, sb)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                outs = run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = run()
        g.replay()
        torch.cuda.synchronize()
        full = torch.cat(outs, 0)
        if ref is None:
            ref = full.clone()
        same = torch.equal(full, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(f"{C}ch @{H}^2 B={B} in sub-batches of {sb:2d}: {ms:8.3f} ms  {fl / ms / 1e9:6.0f} TFLOP/s (convs only counted)  bit-equal to whole batch: {same}", flush=True)
        del g, outs, full


if __name__ == "__main__":
    main()
