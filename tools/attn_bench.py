#!/usr/bin/env python3
"""Micro-benchmark of nlc_attention on the ADM-256 shapes (B=16, D=64).

    python tools/attn_bench.py                 # all three shapes, natural logits and log2 logits (what 16-bit models feed)
    python tools/attn_bench.py 1024 8 [reps [D [B]]]   # one shape, log2 logits (rocprofv3 --pmc passes); D / B: other head widths / batches
Inputs carry the logit scale the networks produce (q and k each scaled by ch^-1/4, unit-variance activations)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import ops  # noqa: E402

shapes = ((1024, 8), (256, 16), (64, 16))
reps = 20
if len(sys.argv) >= 3:
    shapes = ((int(sys.argv[1]), int(sys.argv[2])),)
    reps = int(sys.argv[3]) if len(sys.argv) >= 4 else 20
one_shape = len(sys.argv) >= 3
D = int(sys.argv[4]) if len(sys.argv) >= 5 else 64
B = int(sys.argv[5]) if len(sys.argv) >= 6 else 16
for T, H in shapes:
    for base2 in ((True,) if one_shape else (False, True)):
        qkv = torch.randn(B, T, 3, H, D, device="cuda:0")
        qkv[:, :, :2] *= D ** -0.25
        if base2:
            qkv[:, :, 0] *= ops.LOG2E
        qkv = qkv.reshape(B, T, 3 * H * D).to(torch.bfloat16)
        out = [ops.attention(qkv, H, base2=base2) for _ in range(3)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.attention(qkv, H, base2=base2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 4.0 * B * H * T * T * D
        print(f"attention T={T:5d} heads={H:2d} D={D} B={B} {'log2 logits   ' if base2 else 'natural logits'} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
