#!/usr/bin/env python3
"""Micro-benchmark of nlc_conv2d on the ADM-256 hot shapes (B=16), HIP events on the launch stream.

    python tools/conv_bench.py [--reps 20]
"""
import argparse
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

SHAPES = [  # (H, Cin, Cout, k, note); a trailing "ups" in the note = fused nearest-2x upsample (H is the INPUT size)
    (256, 256, 256, 3, "256->256 @256^2 (31% of FLOPs)"),
    (256, 512, 256, 3, "512->256 @256^2"),
    (128, 256, 256, 3, "256->256 @128^2"),
    (64, 512, 512, 3, "512->512 @64^2"),
    (32, 512, 512, 3, "512->512 @32^2"),
    (16, 1024, 1024, 3, "1024->1024 @16^2"),
    (8, 1024, 1024, 3, "1024->1024 @8^2"),
    (256, 768, 256, 1, "1x1 768->256 @256^2"),
    (32, 512, 1536, 1, "qkv 512->1536 @32^2"),
    (128, 256, 256, 3, "256->256 @128^2->256^2 ups"),
    (64, 512, 512, 3, "512->512 @64^2->128^2 ups"),
    (8, 1024, 3072, 1, "qkv 1024->3072 @8^2"),
    (8, 1024, 1024, 1, "proj 1024->1024 @8^2"),
    (16, 1024, 3072, 1, "qkv 1024->3072 @16^2"),
    (16, 1024, 1024, 1, "proj 1024->1024 @16^2"),
    (256, 256, 6, 3, "out 256->6 @256^2"),
    (16, 512, 1024, 3, "512->1024 @16^2"),
    (16, 2048, 1024, 3, "2048->1024 @16^2"),
    (16, 1024, 512, 3, "1024->512 @16^2"),
    (16, 256, 1024, 3, "256->1024 @16^2"),
]


# the small-map launches of cfg 4 (CelebA-HQ-256 simple UNet, use --batch 8) and of ADM-256's 8x8 / 16x16 levels: few tiles, long K
SMALL = [
    (8, 512, 512, 3, "512->512 @8^2"),
    (8, 1024, 512, 3, "1024->512 @8^2"),
    (16, 512, 512, 3, "512->512 @16^2"),
    (16, 1024, 512, 3, "1024->512 @16^2"),
    (32, 256, 256, 3, "256->256 @32^2"),
    (32, 512, 256, 3, "512->256 @32^2"),
    (16, 512, 1536, 1, "qkv 512->1536 @16^2"),
    (16, 512, 512, 1, "proj 512->512 @16^2"),
    (8, 1024, 512, 1, "nin 1024->512 @8^2"),
    (8, 1024, 1024, 3, "1024->1024 @8^2"),
    (8, 2048, 1024, 3, "2048->1024 @8^2"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="adm", choices=["adm", "small"], help="shape list: the ADM-256 hot shapes, or the small-map launches")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", type=int, default=-1, help="run only SHAPES[i] (for rocprofv3 --pmc passes)")
    ap.add_argument("--first", type=int, default=0, help="skip SHAPES[:first]")
    ap.add_argument("--tuning", default="0", help="comma list of nlc_conv_desc.tuning values to A/B, interleaved in this one process")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--policy", default="auto", choices=sorted(ops.CONV_POLICIES), help="nlc_conv_desc.policy for every launch")
    ap.add_argument("--res", action="store_true", help="with a residual input (the second conv of a ResBlock)")
    ap.add_argument("--zeros", action="store_true", help="all-zero activations and weights: same instruction stream, least energy per MFMA "
                    "(what the clock does to the rate: MI355X_MICROARCH.md 'DVFS give-back')")
    ap.add_argument("--no-stats", action="store_true", help="launch without ride-along GroupNorm statistics (what the epilogue's statistics work costs)")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison with the no_halo dispatch (counter passes: no other kernel in the trace)")
    ap.add_argument("--shape", action="append", default=[], help="extra shape H,Cin,Cout,k (replaces the list; may repeat)")
    args = ap.parse_args()
    global SHAPES
    if args.set == "small":
        SHAPES = SMALL
    if args.shape:
        SHAPES = [tuple(int(v) for v in sh.split(",")) + (f"custom {sh}",) for sh in args.shape]
    if args.only >= 0:
        SHAPES = SHAPES[args.only:args.only + 1]
    elif args.first:
        SHAPES = SHAPES[args.first:]
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ops.CONV_POLICY = args.policy
    for H, cin, cout, k, note in SHAPES:
        x = torch.randn(args.batch, H, H, cin, device=dev).to(dt)
        w = torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)
        if args.zeros:
            x.zero_(); w.zero_()
        pw = ops.pack_conv(w, torch.zeros(cout), dt, dev)
        ups = note.endswith("ups")
        Ho = 2 * H if ups else H
        res = torch.randn(args.batch, Ho, Ho, cout, device=dev).to(dt) if args.res else None
        tunings = [int(t) for t in args.tuning.split(",")]
        st = not args.no_stats
        for _ in range(3):
            got = ops.conv2d(x, pw, upsample2x=ups, res=res, emit_stats=st)
        torch.cuda.synchronize()
        if not args.zeros and not args.no_check and k == 3 and args.policy == "auto":
            # a schedule experiment that breaks the result must not pass as a timing: the same launch through the other 3x3 kernel
            ops.CONV_POLICY = "no_halo"
            ref = ops.conv2d(x, pw, upsample2x=ups, res=res, emit_stats=False).float()
            ops.CONV_POLICY = args.policy
            err, sc = (got.float() - ref).abs().max().item(), ref.abs().max().item()
            assert err <= 2e-2 * sc, f"{note}: result differs from the no_halo dispatch by {err:.3e} (scale {sc:.3e})"
            del ref
        fl = 2.0 * args.batch * H * H * cout * cin * k * k * (4 if ups else 1)
        times = {t: [] for t in tunings}
        for _ in range(args.rounds):                      # variants interleaved round by round in ONE process
            for t in tunings:
                ops.CONV_TUNING = t
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    ops.conv2d(x, pw, upsample2x=ups, res=res, emit_stats=st)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / args.reps)
        ops.CONV_TUNING = 0
        for t in tunings:
            v = sorted(times[t])
            med, mn = v[len(v) // 2], v[0]
            print(f"{note:32s} tuning={t}  median {med * 1e3:9.1f} us {fl / med / 1e9:7.0f} TFLOP/s   min {mn * 1e3:9.1f} us {fl / mn / 1e9:7.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
