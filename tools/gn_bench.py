#!/usr/bin/env python3
"""Micro-benchmark of nlc_groupnorm (+SiLU) on the ADM-256 shapes (B=16): achieved GB/s against the algorithmic
traffic of the two-pass design (2 reads + 1 write of the activation), HIP events on the launch stream.

    python tools/gn_bench.py [--reps 20]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

SILU = True
SHAPES = [  # (H, C0, C1)
    (256, 256, 0), (256, 512, 0), (256, 256, 256), (128, 256, 0), (128, 512, 0), (128, 512, 256), (64, 512, 0), (64, 1024, 0),
    (32, 512, 0), (32, 1024, 0), (16, 1024, 0), (16, 1024, 1024), (8, 1024, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-silu", action="store_true", help="normalise only (memory-side rate of the same access pattern)")
    ap.add_argument("--prestats", action="store_true", help="statistics 'ride along' (zeros attached): times the apply launch only (it derives its coefficients from the totals), 1 read + 1 write")
    args = ap.parse_args()
    global SILU
    SILU = not args.no_silu
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tot_ms = tot_b = 0.0
    for H, c0, c1 in SHAPES:
        x0 = torch.randn(args.batch, H, H, c0, device=dev).to(dt)
        x1 = torch.randn(args.batch, H, H, c1, device=dev).to(dt) if c1 else None
        C = c0 + c1
        if args.prestats:
            x0._nlc_stats = torch.zeros(args.batch, c0 // 8, 4, device=dev, dtype=torch.int64)       # totals (zeros: timing only)
            if x1 is not None:
                x1._nlc_stats = torch.zeros(args.batch, c1 // 8, 4, device=dev, dtype=torch.int64)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        for _ in range(3):
            ops.groupnorm(x0, g, b, groups=32, eps=1e-5, silu=SILU, x1=x1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.groupnorm(x0, g, b, groups=32, eps=1e-5, silu=SILU, x1=x1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        nbytes = (2.0 if args.prestats else 3.0) * args.batch * H * H * C * x0.element_size()
        tot_ms += ms; tot_b += nbytes
        print(f"GN {C:5d}ch ({c0}+{c1}) @{H:3d}^2  {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.0f} GB/s", flush=True)
    print(f"total {tot_ms:.2f} ms, {tot_b / tot_ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
