#!/usr/bin/env python3
"""Diagnostic: timeline of the workgroups of conv_small_kernel (s_memtime / s_memrealtime stamps in a patched copy of
conv_small.hip at its `/*@stamp:N ...*/` markers; nothing executes in the shipped source).

    python tools/small_stamps.py build
    python tools/small_stamps.py run [B H W C]      # default 16 8 8 1024; a GroupNorm+FiLM+SiLU of the input is fused in
"""
import ctypes
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "diffusion-nlc_amd" / "csrc"
OUT = ROOT / "diffusion-nlc_amd" / "libnlc_hip_sstamp.so"
NSLOT = 12
MAXWG = 2048


def patched_source():
    s = (SRC / "conv_small.hip").read_text()
    names = {}
    s = s.replace("namespace {\n",
                  "__device__ unsigned long long g_sstamps[%d * %d];\n"
                  "extern \"C\" int nlc_debug_read_sstamps(void* dst, int bytes) {\n"
                  "    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sstamps), bytes, 0, hipMemcpyDeviceToHost); }\n"
                  "namespace {\n" % (MAXWG, NSLOT), 1)
    s = s.replace("/*@stamp:begin*/",
                  "unsigned long long st[%d] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};\n"
                  "    auto now = [&]() { return (unsigned long long)__builtin_amdgcn_s_memtime(); };\n"
                  "    const unsigned long long st0 = now(); st[10] = (unsigned long long)__builtin_amdgcn_s_memrealtime();\n" % NSLOT, 1)
    s = s.replace("/*@stamp:end*/", "st[9] = st[7] != 0 ? 1 : 0; st[11] = now() - st0; if (threadIdx.x == 0 && L < %d) for (int k = 0; k < %d; ++k) g_sstamps[L * %d + k] = st[k];" % (MAXWG, NSLOT, NSLOT))

    def rep(m):
        names[int(m.group(1))] = m.group(2).strip()
        return "st[%s] = now() - st0;" % m.group(1)
    s = re.sub(r"/\*@stamp:(\d+) ([^*]*)\*/", rep, s)
    return s, names


def build():
    tmp = ROOT / "gpurun_out" / "stamps"
    tmp.mkdir(parents=True, exist_ok=True)
    src = SRC / "conv_small_stamp_tmp.hip"
    text, _ = patched_source()
    src.write_text(text)
    try:
        obj = tmp / "conv_small_stamp.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{ROOT / 'include'}", f"-I{SRC}",
                               "-c", str(src), "-o", str(obj)])
        objs = [str(p) for p in sorted((SRC / "obj").glob("*.o")) if p.name != "conv_small.o"]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + objs + [str(obj)])
    finally:
        src.unlink()
    print("built", OUT)


def run(B=16, H=8, W=8, Cc=1024):
    os.environ["NLC_HIP_LIB"] = str(OUT)
    sys.path.insert(0, str(ROOT))
    import math
    import numpy as np
    import torch
    from diffusion_nlc_amd import _ext, ops
    lib = _ext.load()
    _, names = patched_source()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, H, W, 64, generator=g).to(dev).bfloat16()
    x = ops.conv2d(z, ops.pack_conv(torch.randn(Cc, 64, 1, 1, generator=g) / 8, torch.zeros(Cc), torch.bfloat16, dev))
    pw = ops.pack_conv(torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9), torch.zeros(Cc), torch.bfloat16, dev)
    gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    ss = (torch.randn(B, 2 * Cc, generator=g) * 0.2).to(dev)
    spec = ops.gn_in_spec(x, gam, bet, groups=32, eps=1e-5, silu=True, scale=ss[:, :Cc], shift=ss[:, Cc:])
    assert spec is not None and ops.conv2d(x, pw, query_gn_in=True)
    big = torch.empty(96 << 20, device=dev)                    # evict the weights from L2 / MALL between launches (as in the network)
    ts = []
    for _ in range(6):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(x, pw, gn_in=spec, res=x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"== {B} x {H}x{W} x {Cc} -> {Cc}")
    print("launch (events, cold L2): %s us" % ", ".join(f"{t:.1f}" for t in ts))
    buf = np.zeros(MAXWG * NSLOT, dtype=np.uint64)
    assert lib.nlc_debug_read_sstamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    a = buf.reshape(MAXWG, NSLOT).astype(np.float64)
    a = a[a[:, 11] > 0]
    print("workgroups:", len(a), " owners:", int(a[:, 9].sum()))
    t0 = (a[:, 10] - a[:, 10].min()) * 10.0
    print("entry skew: median %.0f ns, max %.0f ns" % (np.median(t0), t0.max()))
    own = a[a[:, 9] > 0]
    for i in sorted(names):
        v = a[:, i] if i < 7 else own[:, i]
        v = v[v > 0]
        if len(v):
            print(f"   {names[i]:34s} median {np.median(v):9.0f} cycles  (min {v.min():.0f}, max {v.max():.0f}){'   [owners]' if i >= 7 else ''}")
    print(f"   {'end (all workgroups)':34s} median {np.median(a[:, 11]):9.0f} cycles  (max {a[:, 11].max():.0f})")
    print(f"   {'end (owners)':34s} median {np.median(own[:, 11]):9.0f} cycles  (max {own[:, 11].max():.0f})")
    end_ns = t0 + a[:, 11] / 2.1                               # ~2.1 GHz: rough wall position of each workgroup's end
    print("   last workgroup ends ~%.1f us after the first one entered (at 2.1 GHz)" % (end_ns.max() / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(*[int(v) for v in sys.argv[2:6]])
