#!/usr/bin/env python3
"""Micro-benchmark of nlc_conv_first (3 -> 256 channels, NCHW f32 state -> NHWC bf16) on the ADM-256 shape.

    python tools/first_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, H, Cout in ((16, 256, 256), (8, 256, 128), (200, 32, 128)):        # ADM-256, CelebA-HQ-256 (cfg 4), EDM-32 (cfg 3)
    x = torch.randn(B, 3, H, H, device=dev)
    w = torch.randn(Cout, 9, 3, device=dev) * 0.2          # [Cout][KH*KW][Cin]
    b = torch.zeros(Cout, device=dev)
    sc = torch.ones(B, device=dev)
    for _ in range(3):
        ops.conv_first(x, w, b, torch.bfloat16, sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv_first(x, w, b, torch.bfloat16, sc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"conv_first 3->{Cout} @{H}^2 B={B}: {us:.1f} us  ({B * H * H * Cout * 2 / us / 1e6:.2f} TB/s written)")
