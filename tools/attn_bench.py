#!/usr/bin/env python3
"""Micro-benchmark of nlc_attention on the ADM-256 shapes (B=16, D=64)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

for T, H in ((1024, 8), (256, 16), (64, 16)):
    qkv = torch.randn(16, T, 3 * H * 64, device="cuda:0").to(torch.bfloat16)
    for _ in range(3):
        ops.attention(qkv, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, H)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 4.0 * 16 * H * T * T * 64
    print(f"attention T={T:5d} heads={H:2d} D=64  {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
