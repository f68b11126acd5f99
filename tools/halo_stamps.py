#!/usr/bin/env python3
"""Print the stamped phase timeline of the halo conv kernel (see tools/halo_stamps.sh)."""
import ctypes
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffusion_nlc_amd import _ext, ops  # noqa: E402

H, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 256)
dev = torch.device("cuda:0")
x = torch.randn(16, H, H, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
pw = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
for _ in range(5):
    ops.conv2d(x, pw)
torch.cuda.synchronize()
lib = _ext.load()
buf = (ctypes.c_ulonglong * 64)()
lib.nlc_debug_halo_stamps.argtypes = [ctypes.c_void_p]
rc = lib.nlc_debug_halo_stamps(buf)
st = [[buf[w_ * 8 + q] for q in range(8)] for w_ in range(8)]
t0 = min(s[0] for s in st)
names = ["start", "dma issued", "reads1 issued", "mfma1 issued", "reads2 issued", "mfma2 issued", "vmcnt done", "barrier done"]
print("rc", rc, " (s_memtime ticks relative to the earliest wave's step start)")
print("wave " + " ".join(f"{n:>14s}" for n in names))
for w_ in range(8):
    print(f"{w_:4d} " + " ".join(f"{st[w_][q] - t0:14d}" for q in range(8)))
