#!/usr/bin/env python3
"""GroupNorm + SiLU of a ResBlock input and its 1x1 skip projection: the two separate passes (nlc_groupnorm_prestats + nlc_conv2d) against
the one launch that does both from a single read (nlc_conv_desc.norm_out) + its coefficient launch.  Interleaved in one process.

    python tools/gn_pw_bench.py [B H C0 C1 Cout] ...      (default: the ADM-256 256x256 and 128x128 output-block shapes)"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

shapes = [(16, 256, 256, 256, 256), (16, 128, 256, 256, 256), (8, 256, 128, 0, 128), (200, 32, 256, 256, 256)]
if len(sys.argv) >= 6:
    shapes = [tuple(int(v) for v in sys.argv[1:6])]
dev = torch.device("cuda:0")
dt = torch.bfloat16
for B, H, C0, C1, Cout in shapes:
    C = C0 + C1
    src = torch.randn(B, H, H, 64, device=dev).to(dt)
    def prod(c):
        return ops.conv2d(src, ops.pack_conv(torch.randn(c, 64, 3, 3) / 24, torch.zeros(c), dt, dev))
    h0 = prod(C0)
    h1 = prod(C1) if C1 else None
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    pw = ops.pack_conv(torch.randn(Cout, C, 1, 1) / math.sqrt(C), torch.zeros(Cout), dt, dev)
    ok = ops.conv2d(h0, pw, x1=h1, query_norm_out=True)
    def plain():
        ops.groupnorm(h0, gamma, beta, groups=32, eps=1e-5, silu=True, x1=h1)
        ops.conv2d(h0, pw, x1=h1)
    def fused():
        coef = ops.groupnorm_coef(h0, gamma, beta, groups=32, eps=1e-5, x1=h1)
        ops.conv2d(h0, pw, x1=h1, gn_coef=coef, gn_act=1, norm_out=True)
    variants = [("separate", plain)] + ([("one read", fused)] if ok else [])
    times = {n: [] for n, _ in variants}
    for n, f in variants:
        f(); f()
    torch.cuda.synchronize()
    for _ in range(5):
        for n, f in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 10 * 1e3)
    line = f"B={B} {H}x{H} {C0}+{C1} -> {Cout}: " + "   ".join(f"{n} {sorted(t)[len(t) // 2]:8.1f} us" for n, t in times.items())
    print(line + ("" if ok else "   (norm_out not supported for this launch)"), flush=True)
