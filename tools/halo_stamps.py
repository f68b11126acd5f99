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
with_res = len(sys.argv) >= 5 and sys.argv[4] == "res"
dev = torch.device("cuda:0")
x = torch.randn(16, H, H, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
pw = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
res = torch.randn(16, H, H, cout, device=dev).to(torch.bfloat16) if with_res else None
import time
t_end = time.time() + 2.5                       # >= 2 s of back-to-back launches so the clock settles under load
while time.time() < t_end:
    for _ in range(20):
        ops.conv2d(x, pw, res=res)
    torch.cuda.synchronize()
torch.cuda.synchronize()
lib = _ext.load()
buf = (ctypes.c_ulonglong * 340)()
lib.nlc_debug_halo_stamps.argtypes = [ctypes.c_void_p]
rc = lib.nlc_debug_halo_stamps(buf)
st = [[buf[w_ * 8 + q] for q in range(8)] for w_ in range(8)]
t0 = min(s[0] for s in st)
names = ["start", "dma issued", "reads1 issued", "mfma1 issued", "reads2 issued", "mfma2 issued", "vmcnt done", "barrier done"]
clk = [buf[64 + i] for i in range(4)]
if clk[3] > clk[1]:
    print(f"in-kernel clock of workgroup 0 over its whole tile loop: {(clk[2] - clk[0]) / (clk[3] - clk[1]) * 100:.0f} MHz "
          f"(ds_memtime / ds_memrealtime x 100 MHz; loop = {(clk[3] - clk[1]) / 100:.1f} us)")
tt = [[buf[68 + w_ * 4 + q] for q in range(4)] for w_ in range(8)]
print("tile boundary (cycles): epilogue of tile 1, whole tile 2 = epilogue(1) start -> epilogue(2) start, epilogue of tile 2")
for w_ in range(8):
    print(f"  wave {w_}: epilogue1 {tt[w_][1] - tt[w_][0]:6d}   tile2 {tt[w_][2] - tt[w_][0]:7d}   epilogue2 {tt[w_][3] - tt[w_][2]:6d}")
ep = [[buf[100 + w_ * 6 + q] for q in range(6)] for w_ in range(8)]
print("inside the epilogue of tile 1 (cycles from its start): cadd loads issued, pixel loop + stores done, stats done, acc re-initialised")
for w_ in range(8):
    print(f"  wave {w_}: " + " ".join(f"{ep[w_][q] - ep[w_][0]:7d}" for q in range(1, 5)))
sp = [[buf[148 + w_ * 12 + q] for q in range(10)] for w_ in range(8)]
print("k-step durations of (tile 1, channel block 1), cycles, taps 0..8, then their sum:")
for w_ in range(8):
    d = [sp[w_][q + 1] - sp[w_][q] for q in range(9)]
    print(f"  wave {w_}: " + " ".join(f"{v:6d}" for v in d) + f"   sum {sum(d):7d}")
cbs = [[buf[244 + w_ * 12 + q] for q in range(12)] for w_ in range(8)]
ncb = cin // 64
print("tile 1: cycles per channel block (start of cb c -> start of cb c+1; the last one -> start of the epilogue), then the epilogue:")
for w_ in range(8):
    t = cbs[w_][:ncb] + [cbs[w_][10]]
    print(f"  wave {w_}: " + " ".join(f"{t[i + 1] - t[i]:7d}" for i in range(ncb)) + f"   epilogue {cbs[w_][11] - cbs[w_][10]:6d}")
print("rc", rc, " (s_memtime ticks relative to the earliest wave's step start)")
print("wave " + " ".join(f"{n:>14s}" for n in names))
for w_ in range(8):
    print(f"{w_:4d} " + " ".join(f"{st[w_][q] - t0:14d}" for q in range(8)))
