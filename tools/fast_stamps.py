#!/usr/bin/env python3
"""Diagnostic: timeline of one workgroup of conv_fast_kernel on a small-map launch (s_memtime / s_memrealtime stamps in a patched
copy of conv_fast.hip; nothing in the shipped source).

    python tools/fast_stamps.py build
    python tools/fast_stamps.py run [H Cin Cout batch]      # default 8 512 512 8 (cfg 4's 8x8 level: 16 tiles x 4 splits)

Per workgroup (wave 0, lane 0): cycles from kernel entry to: tap tables done | first DMA landed + barrier | k-loop done | partial sums
stored and drained | arrival counted | (last arriver) partials read back | end; plus the entry time in 100 MHz ticks (start skew).
"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "diffusion-nlc_amd" / "csrc"
OUT = ROOT / "diffusion-nlc_amd" / "libnlc_hip_fstamp.so"
NSLOT = 12
MAXWG = 4096


def patched_source() -> str:
    s = (SRC / "conv_fast.hip").read_text()

    def sub(old, new):
        nonlocal s
        assert s.count(old) >= 1, old
        s = s.replace(old, new, 1)

    sub("namespace {\n",
        "__device__ unsigned long long g_fstamps[%d * %d];\n"
        "extern \"C\" int nlc_debug_read_fstamps(void* dst, int bytes) {\n"
        "    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fstamps), bytes, 0, hipMemcpyDeviceToHost); }\n"
        "namespace {\n" % (MAXWG, NSLOT))
    sub("    const int nblk = p.MT * p.NT;\n    int bid = blockIdx.x;",
        "    unsigned long long st[%d] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};\n"
        "    auto now = [&]() { return (unsigned long long)__builtin_amdgcn_s_memtime(); };\n"
        "    const unsigned long long st0 = now(); st[10] = (unsigned long long)__builtin_amdgcn_s_memrealtime();\n"
        "    auto flush = [&]() { if (threadIdx.x == 0) { const int w = blockIdx.y * gridDim.x + blockIdx.x; if (w < %d) for (int k = 0; k < %d; ++k) g_fstamps[w * %d + k] = st[k]; } };\n"
        "    const int nblk = p.MT * p.NT;\n    int bid = blockIdx.x;" % (NSLOT, MAXWG, NSLOT, NSLOT))
    sub("    const int64_t wrow = (int64_t)TAPS * p.Cin_pad * ES;",
        "    asm volatile(\"\" :: \"v\"(pix[3][TAPS - 1]), \"v\"(vmask[3]));\n    st[0] = now() - st0;\n"
        "    const int64_t wrow = (int64_t)TAPS * p.Cin_pad * ES;")
    sub("    int kt = 0, cur = 0;", "    st[1] = now() - st0;\n    int kt = 0, cur = 0;")
    sub("    const int n = n0 + wn * 64 + fq * 16;\n    const bool all16 = n + 16 <= p.Cout;",
        "    st[2] = now() - st0;\n    const int n = n0 + wn * 64 + fq * 16;\n    const bool all16 = n + 16 <= p.Cout;")
    sub("        __syncthreads();                                             // ... and every lane's, before the arrival is counted",
        "        __syncthreads();                                             // ... and every lane's, before the arrival is counted\n"
        "        st[3] = now() - st0;")
    sub("        __syncthreads();\n        if (!s_last) return;",
        "        __syncthreads();\n        st[4] = now() - st0;\n        if (!s_last) { flush(); return; }")
    sub("    if (n >= p.Cout) return;\n",
        "    asm volatile(\"\" :: \"v\"(acc[3][3][3]), \"v\"(acc[0][0][0]));\n    st[5] = now() - st0; st[9] = 1;\n"
        "    if (n >= p.Cout) { flush(); return; }\n")
    sub("    Stat16 st16;                                                // ride-along GroupNorm statistics: this lane's 16 channels, 4 pixels",
        "    st[7] = now() - st0;\n    Stat16 st16;                                                // ride-along GroupNorm statistics: this lane's 16 channels, 4 pixels")
    sub("            done = true;\n", "            st[8] = now() - st0;\n            done = true;\n")
    # kernel end: last closing brace of the kernel = before 'template <typename T, int TAPS, int STAGES, bool X3 = false>\nint launch_fast'
    sub("}\n\ntemplate <typename T, int TAPS, int STAGES, bool X3 = false>\nint launch_fast",
        "    st[6] = now() - st0;\n    flush();\n}\n\ntemplate <typename T, int TAPS, int STAGES, bool X3 = false>\nint launch_fast")
    return s


def build():
    tmp = ROOT / "gpurun_out" / "stamps"
    tmp.mkdir(parents=True, exist_ok=True)
    src = SRC / "conv_fast_stamp_tmp.hip"
    src.write_text(patched_source())
    try:
        obj = tmp / "conv_fast_stamp.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{ROOT / 'include'}", f"-I{SRC}",
                               "-c", str(src), "-o", str(obj)])
        objs = [str(p) for p in sorted((SRC / "obj").glob("*.o")) if p.name != "conv_fast.o"]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + objs + [str(obj)])
    finally:
        src.unlink()
    print("built", OUT)


def run(H=8, cin=512, cout=512, batch=8, k=3):
    os.environ["NLC_HIP_LIB"] = str(OUT)
    sys.path.insert(0, str(ROOT))
    import math
    import numpy as np
    import torch
    from diffusion_nlc_amd import _ext, ops
    lib = _ext.load()
    dev = torch.device("cuda:0")
    x = torch.randn(batch, H, H, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)
    pw = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, dev)
    big = torch.empty(64 << 20, device=dev)                    # evict the weights from L2 between launches (as in the network)
    ts = []
    for _ in range(6):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(x, pw, allow_split=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("launch (events, cold L2): %s us" % ", ".join(f"{t:.1f}" for t in ts))
    buf = np.zeros(MAXWG * NSLOT, dtype=np.uint64)
    assert lib.nlc_debug_read_fstamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    a = buf.reshape(MAXWG, NSLOT).astype(np.float64)
    a = a[a[:, 2] > 0]
    if not len(a):
        print("no stamped workgroups (the launch did not take conv_fast)"); return
    print("workgroups:", len(a), " last arrivers:", int(a[:, 9].sum()))
    t0 = (a[:, 10] - a[:, 10].min()) * 10.0                   # ns
    print("entry skew: median %.0f ns, max %.0f ns" % (np.median(t0), t0.max()))
    names = ["tap tables done", "first DMA landed + barrier", "k-loop done", "partials stored + drained", "arrival counted"]
    for i, nm in enumerate(names):
        v = a[:, i]
        print(f"   {nm:30s} median {np.median(v):9.0f} cycles  (min {v.min():.0f}, max {v.max():.0f})")
    la = a[a[:, 9] > 0]
    if len(la):
        print(f"   {'last arriver: k-loop done':30s} median {np.median(la[:, 2]):9.0f} cycles")
        print(f"   {'last arriver: stored + drained':30s} median {np.median(la[:, 3]):9.0f} cycles")
        print(f"   {'last arriver: arrival counted':30s} median {np.median(la[:, 4]):9.0f} cycles   (incl. its agent-scope acquire)")
        print(f"   {'last arriver: partials read':30s} median {np.median(la[:, 5]):9.0f} cycles")
        print(f"   {'last arriver: bias ready':30s} median {np.median(la[:, 7]):9.0f} cycles")
        print(f"   {'last arriver: rows stored':30s} median {np.median(la[:, 8]):9.0f} cycles")
        print(f"   {'last arriver: end':30s} median {np.median(la[:, 6]):9.0f} cycles  (max {la[:, 6].max():.0f})")
        end_ns = (la[:, 10] - a[:, 10].min()) * 10.0
        print("   last arrivers enter at (ns after the first workgroup): median %.0f" % np.median(end_ns))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(*[int(v) for v in sys.argv[2:7]])
