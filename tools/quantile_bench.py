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
for single in (True, False):
    for _ in range(3):
        ops.dynamic_threshold(x, 0.995, 1e9, single_workgroup=single)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.dynamic_threshold(x, 0.995, 1e9, single_workgroup=single)
    e1.record()
    torch.cuda.synchronize()
    print("dynamic_threshold [16, 196608] %s: %.1f us" % ("one workgroup per sample, one launch " if single else "G workgroups per sample, 10 launches", e0.elapsed_time(e1) / 50 * 1e3))
