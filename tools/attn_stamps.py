#!/usr/bin/env python3
"""Diagnostic: where a wave of attn_d64_kernel spends a 64-key tile (s_memtime stamps in a patched copy of attention.hip; nothing
executes in the shipped source).  Per wave the cycles of four segments are summed over all tiles:
  S    top of the tile -> the tile maximum is known (8 K-fragment reads, 8 S^T MFMAs, 16 v_max3 + the half exchange: the first VALU
       instruction that reads the scores waits for the MFMAs)
  EXP  -> all 32 exponentials issued (+ the rare offset path)
  PV   -> the 12 O^T / row-sum MFMAs issued (16 transposed V reads, 16 converts)
  SYNC -> past the tile's counted DMA wait and barrier

    python tools/attn_stamps.py build
    python tools/attn_stamps.py run [T H B]      # default 1024 8 16
"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "diffusion-nlc_amd" / "csrc"
OUT = ROOT / "diffusion-nlc_amd" / "libnlc_hip_astamp.so"
NSLOT = 12
MAXW = 8192


def patched_source():
    s = (SRC / "attention.hip").read_text()

    def sub(old, new, count=1):
        nonlocal s
        assert s.count(old) >= 1, old
        s = s.replace(old, new, count)

    sub("namespace {\n",
        "__device__ unsigned long long g_astamps[%d * %d];\n"
        "extern \"C\" int nlc_debug_read_astamps(void* dst, int bytes) {\n"
        "    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_astamps), bytes, 0, hipMemcpyDeviceToHost); }\n"
        "namespace {\n" % (MAXW, NSLOT))
    # only the shipped instantiation pattern is patched: the d64 kernel's tile loop
    sub("    constexpr int PP = 8 / W;                              // DMA pieces (8 rows x 128 B) of K, and of V, per wave and tile\n",
        "    constexpr int PP = 8 / W;                              // DMA pieces (8 rows x 128 B) of K, and of V, per wave and tile\n"
        "    const unsigned long long k_entry = (unsigned long long)__builtin_amdgcn_s_memtime(), r_entry = (unsigned long long)__builtin_amdgcn_s_memrealtime();\n")
    sub("    for (int t = 0; t < ntiles; ++t) {\n        const int st = t & (F_NST - 1);",
        "    unsigned long long acc_s = 0, acc_e = 0, acc_p = 0, acc_y = 0;\n"
        "    auto now = [&]() { return (unsigned long long)__builtin_amdgcn_s_memtime(); };\n"
        "    const unsigned long long k_begin = now();\n"
        "    for (int t = 0; t < ntiles; ++t) {\n        unsigned long long t0 = now();\n        const int st = t & (F_NST - 1);")
    sub("        const float mnew = fmaxf(mrun_, BASE2 ? tm : tm * L2E);",
        "        asm volatile(\"\" :: \"v\"(tm));\n        { const unsigned long long t1 = now(); acc_s += t1 - t0; t0 = t1; }\n"
        "        const float mnew = fmaxf(mrun_, BASE2 ? tm : tm * L2E);")
    sub("        // ---- O^T += V^T P^T and l += 1^T P^T : P^T k-steps come straight from the score registers",
        "        asm volatile(\"\" :: \"v\"(sA[0][15]), \"v\"(sB[0][15]));\n        { const unsigned long long t1 = now(); acc_e += t1 - t0; t0 = t1; }\n"
        "        // ---- O^T += V^T P^T and l += 1^T P^T : P^T k-steps come straight from the score registers")
    sub("        // own DMA pieces of tile t + 1 landed (the pieces of later tiles may stay in flight) ...",
        "        { const unsigned long long t1 = now(); acc_p += t1 - t0; t0 = t1; }\n"
        "        // own DMA pieces of tile t + 1 landed (the pieces of later tiles may stay in flight) ...")
    sub("        __syncthreads();                                       // ... and everybody's are published; tile t's stage is free\n    }",
        "        __syncthreads();                                       // ... and everybody's are published; tile t's stage is free\n"
        "        { const unsigned long long t1 = now(); acc_y += t1 - t0; t0 = t1; }\n    }\n"
        "    const unsigned long long k_loop_end = now();\n"
        % ())
    # the kernel's last statement: the store loop's closing braces, then the kernel's
    sub("                *reinterpret_cast<uint2*>(op + db * 32 + 8 * g) = make_uint2(pk.x, pk.y);\n            }\n    }\n}",
        "                *reinterpret_cast<uint2*>(op + db * 32 + 8 * g) = make_uint2(pk.x, pk.y);\n            }\n    }\n"
        "    const unsigned long long k_stores_issued = now();\n"
        "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
        "    if (lane == 0) { const int w = blockIdx.x * W + wave; if (w < %d) { unsigned long long* o = g_astamps + w * %d;\n"
        "        o[0] = acc_s; o[1] = acc_e; o[2] = acc_p; o[3] = acc_y; o[4] = k_loop_end - k_begin; o[5] = (unsigned long long)ntiles; o[6] = (unsigned long long)wave;\n"
        "        o[7] = k_begin - k_entry; o[8] = k_stores_issued - k_loop_end; o[9] = now() - k_stores_issued; o[10] = r_entry; o[11] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } }\n}"
        % (MAXW, NSLOT))
    return s


def build():
    tmp = ROOT / "gpurun_out" / "stamps"
    tmp.mkdir(parents=True, exist_ok=True)
    src = SRC / "attention_stamp_tmp.hip"
    src.write_text(patched_source())
    try:
        obj = tmp / "attention_stamp.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{ROOT / 'include'}", f"-I{SRC}",
                               "-c", str(src), "-o", str(obj)])
        objs = [str(p) for p in sorted((SRC / "obj").glob("*.o")) if p.name != "attention.o"]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + objs + [str(obj)])
    finally:
        src.unlink()
    print("built", OUT)


def run(T=1024, H=8, B=16):
    os.environ["NLC_HIP_LIB"] = str(OUT)
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    from diffusion_nlc_amd import _ext, ops
    lib = _ext.load()
    D = 64
    qkv = torch.randn(B, T, 3, H, D, device="cuda:0")
    qkv[:, :, :2] *= D ** -0.25
    qkv[:, :, 0] *= ops.LOG2E
    qkv = qkv.reshape(B, T, 3 * H * D).to(torch.bfloat16)
    for _ in range(3):
        ops.attention(qkv, H, base2=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(qkv, H, base2=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"== T={T} heads={H} B={B}: {e0.elapsed_time(e1) * 100:.1f} us per launch WITH the stamps")
    buf = np.zeros(MAXW * NSLOT, dtype=np.uint64)
    assert lib.nlc_debug_read_astamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    a = buf.reshape(MAXW, NSLOT).astype(np.float64)
    a = a[a[:, 5] > 0]
    nt = a[0, 5]
    print(f"waves stamped: {len(a)}, tiles per wave: {int(nt)}")
    names = ["S    (K reads, 8 S^T MFMAs, tile maximum)", "EXP  (32 exponentials)", "PV   (16 converts, 16 V^T reads, 12 MFMAs)", "SYNC (DMA wait + barrier)"]
    tot = a[:, :4].sum(axis=1)
    for i, nm in enumerate(names):
        v = a[:, i] / nt
        print(f"   {nm:46s} median {np.median(v):7.0f} cycles per tile  ({100 * np.median(a[:, i] / tot):4.1f} % of the loop)   older half {np.median(v[a[:, 6] < 4]):7.0f}  younger half {np.median(v[a[:, 6] >= 4]):7.0f}")
    print(f"   {'tile loop':46s} median {np.median(tot / nt):7.0f} cycles per tile; whole loop {np.median(a[:, 4]):.0f} cycles")
    print(f"   prologue (entry -> first tile: Q loads, first K / V DMA, barrier)   median {np.median(a[:, 7]):7.0f} cycles")
    print(f"   epilogue (1 / l, 8 x 8-byte stores per lane issued)                 median {np.median(a[:, 8]):7.0f} cycles;  until they are acknowledged: + {np.median(a[:, 9]):.0f}")
    r0 = a[:, 10].min()
    print(f"   wave entry after the first wave's: median {np.median(a[:, 10] - r0) * 10:.0f} ns, max {(a[:, 10] - r0).max() * 10:.0f} ns;  last wave ends {(a[:, 11] - r0).max() * 10:.0f} ns after the first one entered")
    cyc = a[:, 7] + a[:, 4] + a[:, 8] + a[:, 9]
    ns = (a[:, 11] - a[:, 10]) * 10.0
    print(f"   in-kernel clock (a wave's s_memtime cycles / its s_memrealtime span): median {np.median(cyc / ns):.3f} GHz;  a wave lives {np.median(cyc):.0f} cycles = {np.median(ns) / 1e3:.1f} us")
    life = ns / 1e3
    print("   wave lifetime percentiles (us): " + ", ".join(f"p{q}={np.percentile(life, q):.1f}" for q in (1, 10, 50, 90, 99, 100)))
    wg = (np.arange(len(a)) // 8)
    wl = np.array([life[wg == i].max() for i in range(wg.max() + 1)])
    print("   per workgroup (slowest wave), by workgroup id mod 8 (= XCD under round-robin placement): " +
          ", ".join(f"{np.median(wl[np.arange(len(wl)) % 8 == x]):.1f}" for x in range(8)))
    mfma = 20 * 32 * nt * len(a) / 1024.0                      # matrix cycles per SIMD over the launch (4096 waves on 1024 SIMDs)
    span = (a[:, 11].max() - r0) * 10.0 * np.median(cyc / ns)  # launch span in shader cycles at the in-kernel clock
    print(f"   matrix cycles per SIMD {mfma:.0f} of a launch span of {span:.0f} shader cycles -> matrix pipe busy {100 * mfma / span:.1f} % (at the in-kernel clock)")
    print("   matrix work of one wave and tile: 20 MFMAs x 32 = 640 cycles; VALU issue: 32 x 8 (v_exp_f32) + ~62 x 4 = ~500 cycles")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(*[int(v) for v in sys.argv[2:5]])
