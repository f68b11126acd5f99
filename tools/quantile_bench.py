#!/usr/bin/env python3
"""Micro-benchmark of nlc_dynamic_threshold (exact per-sample quantile by radix select) on the ADM-256 state: [16, 3 x 256 x 256] f32.

    python tools/quantile_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

x = torch.randn(16, 3 * 256 * 256, device="cuda:0")
for _ in range(3):
    ops.dynamic_threshold(x, 0.995, 1e9)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.dynamic_threshold(x, 0.995, 1e9)
e1.record()
torch.cuda.synchronize()
print("dynamic_threshold [16, 196608]: %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
