#!/usr/bin/env python3
"""Diagnostic: where a tile of conv_halo_kernel<bf16> spends its cycles (s_memtime stamps), WITHOUT stamp code in the shipped source.

    python tools/halo_stamps.py build      # here or on the GPU box: patched copy of conv_halo.hip -> diffusion-nlc_amd/libnlc_hip_stamp.so
    python tools/halo_stamps.py run [H Cin Cout]   # on the GPU: launches the layer, prints per-phase cycle shares

The patch adds, for waves 0 and 4 (the two waves of SIMD 0) of every workgroup, running sums of
  k-loop | epilogue: residual + rows stored | statistics | wait for the next tile's bias + accumulator init | first k-step after an
  epilogue | a mid-tile k-step (tap 4 of the first channel block)
into a __device__ array that no kernel code reads, plus an extern "C" reader.  Stamp values never reach an output element.
"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "diffusion-nlc_amd" / "csrc"
OUT = ROOT / "diffusion-nlc_amd" / ("libnlc_hip_stamp_nostore.so" if os.environ.get("STAMP_NOSTORE") else "libnlc_hip_stamp%s.so" % os.environ.get("STAMP_KT", ""))
NSLOT = 16


def patched_source() -> str:
    s = (SRC / "conv_halo.hip").read_text()

    def sub(old, new, count=1):
        nonlocal s
        assert s.count(old) >= 1, old
        s = s.replace(old, new, count)

    sub("namespace {\n\n__device__ uint4 g_zero_page_h",
        "__device__ unsigned long long g_stamps[256 * 2 * %d];\n"
        "extern \"C\" int nlc_debug_read_stamps(void* dst, int bytes) {\n"
        "    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost); }\n"
        "extern \"C\" int nlc_debug_clear_stamps() { static unsigned long long z[256 * 2 * %d];\n"
        "    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z), 0, hipMemcpyHostToDevice); }\n"
        "namespace {\n\n__device__ uint4 g_zero_page_h" % (NSLOT, NSLOT))
    # accumulators + helper right before the epilogue lambda
    sub("    auto epilogue = [&](const TileH& t, const TileH& nx, bool has_next) {",
        "    unsigned long long st_acc[%d] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;\n"
        "    auto now = [&]() { return (unsigned long long)__builtin_amdgcn_s_memtime(); };\n"
        "    auto epilogue = [&](const TileH& t, const TileH& nx, bool has_next) {\n"
        "        st_t1 = now(); st_acc[0] += st_t1 - st_t0;\n" % NSLOT)
    # end of the epilogue (after the accumulators of the next tile are initialised)
    sub("        post_ep = early ? 2 : 0;\n    };",
        "        st_t0 = now(); st_acc[1] += st_t0 - st_t1; st_acc[7] += 1;\n        post_ep = early ? 2 : 0;\n    };")
    # per-step stamps: first step of a tile (kt == 0) and tap 4 of the first channel block
    sub("                __syncthreads();\n                bcur = bnext;\n                ++kt;",
        "                __syncthreads();\n                if constexpr (tap == 0) { if (kt == 0) { st_t1 = now(); st_acc[4] += st_t1 - st_t0; } }\n"
        "                if constexpr (tap == 2) { if (kt == 2) st_acc[6] += now() - st_t1; }\n"
        "                if constexpr (tap == 3) { if (kt == 3) st_t1 = now(); }\n"
        "                if constexpr (tap == 4) { if (kt == 4) st_acc[5] += now() - st_t1; }\n"
        "                bcur = bnext;\n                ++kt;")
    # inside step 0 of a tile (plain branch): after the DMA issue, after the second cluster's issue, after the counted wait, after the barrier
    sub("                issue_dma();\n                if constexpr (FIRST && tap < 8) {",
        "                if constexpr (tap == 0) { if (kt == 0) st_acc[8] += now() - st_t0; }\n"
        "                issue_dma();\n                if constexpr (FIRST && tap < 8) {")
    sub("                xf_mid();\n                load_frags(fa0, fb0, nast, bnext",
        "                xf_mid();\n"
        "                if constexpr (tap == 0) { if (kt == 0) st_acc[9] += now() - st_t0; }\n"
        "                load_frags(fa0, fb0, nast, bnext")
    # the CF instantiation leaves its epilogue early
    sub("                    if (has_stats) emit_stats(st16, t, n);\n                    return;",
        "                    if (has_stats) emit_stats(st16, t, n);\n"
        "                    st_t0 = now(); st_acc[1] += st_t0 - st_t1; st_acc[7] += 1; STEPS_EP return;")
    sub("                // retire weights kt+2; instructions younger than them may stay in flight:",
        "                if constexpr (tap == 0) { if (kt == 0) st_acc[10] += now() - st_t0; }\n"
        "                // retire weights kt+2; instructions younger than them may stay in flight:")
    sub("                __syncthreads();\n                if constexpr (tap == 0) { if (kt == 0) { st_t1 = now();",
        "                if constexpr (tap == 0) { if (kt == 0) st_acc[11] += now() - st_t0; }\n"
        "                __syncthreads();\n                if constexpr (tap == 0) { if (kt == 0) { st_t1 = now();")
    # loop top (after the next tile's decode) and right before the first cluster of step 0
    sub("        int kt = 0;\n        const int c_end = t_cb1(cur), nkt = t_nk(cur);",
        "        int kt = 0;\n        const int c_end = t_cb1(cur), nkt = t_nk(cur);\n        st_acc[12] += now() - st_t0;")
    sub("                load_frags(fa1, fb1, hs, bcur, tap_c, K1{});      // this step's second half\n                mma16(fa0, fb0);",
        "                load_frags(fa1, fb1, hs, bcur, tap_c, K1{});      // this step's second half\n"
        "                if constexpr (tap == 0) { if (kt == 0) st_acc[13] += now() - st_t0; }\n                mma16(fa0, fb0);")
    if os.environ.get("STAMP_NOSTORE"):
        sub("                    *reinterpret_cast<uint4*>(op) = pk0;\n                    *reinterpret_cast<uint4*>(op + 8) = pk1;",
            "                    if (p.B < 0) { *reinterpret_cast<uint4*>(op) = pk0;\n                    *reinterpret_cast<uint4*>(op + 8) = pk1; }")
    if os.environ.get("STAMP_STEPS"):
        # per-k-step histogram (36 steps of a 256-channel tile): running sums in 2 KiB of LDS behind the kernel's own, lane 0 of waves 0 / 4
        sub("constexpr int HALO_LDS = 2 * A_STAGE + NBST * B_STAGE + SCRATCH + 2 * COEF_STAGE;",
            "constexpr int HALO_LDS_K = 2 * A_STAGE + NBST * B_STAGE + SCRATCH + 2 * COEF_STAGE;\nconstexpr int HALO_LDS = HALO_LDS_K + 2048;")
        sub("    auto epilogue = [&](const TileH& t, const TileH& nx, bool has_next) {",
            "    unsigned long long* st_steps = reinterpret_cast<unsigned long long*>(smem + HALO_LDS_K) + (wave >> 2) * 64;\n"
            "    if ((wave & 3) == 0 && lane < 64) st_steps[lane] = 0;\n    unsigned long long st_prev = 0;\n"
            "    auto epilogue = [&](const TileH& t, const TileH& nx, bool has_next) {")
        sub("    constexpr int wdist = 3;", "    st_prev = now();\n    constexpr int wdist = 3;")
        sub("                bcur = bnext;\n                ++kt;",
            "                if ((wave & 3) == 0 && lane == 0 && kt < 64) { unsigned long long t = now(); st_steps[kt] += t - st_prev; st_prev = t; }\n"
            "                bcur = bnext;\n                ++kt;")
        sub("        st_t0 = now(); st_acc[1] += st_t0 - st_t1; st_acc[7] += 1;\n",
            "        st_t0 = now(); st_acc[1] += st_t0 - st_t1; st_acc[7] += 1;\n"
            "        if ((wave & 3) == 0 && lane == 0) { st_steps[63] += st_t0 - st_prev; st_prev = st_t0; }\n")
        s = s.replace("STEPS_EP", "if ((wave & 3) == 0 && lane == 0) { st_steps[63] += st_t0 - st_prev; st_prev = st_t0; }")
        sub("    dma_wait_h<0>();          // the redundant tail fetches",
            "    dma_wait_h<0>();          // the redundant tail fetches\n"
            "    if ((wave == 0 || wave == 4) && lane < 64) g_steps[(blockIdx.x * 2 + (wave >> 2)) * 64 + lane] = st_steps[lane];")
        sub("__device__ unsigned long long g_stamps[",
            "__device__ unsigned long long g_steps[256 * 2 * 64];\n"
            "extern \"C\" int nlc_debug_read_steps(void* dst, int bytes) {\n"
            "    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_steps), bytes, 0, hipMemcpyDeviceToHost); }\n"
            "__device__ unsigned long long g_stamps[")
    skt = int(os.environ.get("STAMP_KT", "0"))
    if skt:
        # the in-step stamps (slots 8-11, 13) on k-step STAMP_KT (a multiple of 9: tap 0 of a later channel block) instead of step 0,
        # measured from the barrier that ended the step before it (slot 14 = that whole step)
        s = s.replace("if constexpr (tap == 0) { if (kt == 0) st_acc[", "if constexpr (tap == %d) { if (kt == %d) st_acc[" % (skt % 9, skt))
        sub("                bcur = bnext;\n                ++kt;",
            "                if constexpr (tap == %d) { if (kt == %d) st_t0 = now(); }\n"
            "                if constexpr (tap == %d) { if (kt == %d) st_acc[14] += now() - st_t0; }\n"
            "                bcur = bnext;\n                ++kt;" % ((skt - 1) % 9, skt - 1, skt % 9, skt))
    s = s.replace("STEPS_EP", "")
    # start-of-kernel time + final write-out
    sub("    constexpr int wdist = 3;", "    st_t0 = now();\n    constexpr int wdist = 3;")
    sub("    dma_wait_h<0>();          // the redundant tail fetches",
        "    dma_wait_h<0>();          // the redundant tail fetches\n"
        "    if ((wave == 0 || wave == 4) && lane == 0) {\n"
        "        for (int k = 0; k < %d; ++k) g_stamps[(blockIdx.x * 2 + (wave >> 2)) * %d + k] = st_acc[k];\n    }" % (NSLOT, NSLOT))
    return s


def build():
    tmp = ROOT / "gpurun_out" / "stamps"
    tmp.mkdir(parents=True, exist_ok=True)
    src = SRC / "conv_halo_stamp_tmp.hip"
    src.write_text(patched_source())
    try:
        obj = tmp / "conv_halo_stamp.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{ROOT / 'include'}", f"-I{SRC}",
                               "-c", str(src), "-o", str(obj)])
        objs = [str(p) for p in sorted((SRC / "obj").glob("*.o")) if p.name != "conv_halo.o"]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + objs + [str(obj)])
    finally:
        src.unlink()
    print("built", OUT)


def run(H=256, cin=256, cout=256, res=False, x3=False):
    os.environ["NLC_HIP_LIB"] = str(OUT)
    sys.path.insert(0, str(ROOT))
    import math
    import numpy as np
    import torch
    from diffusion_nlc_amd import _ext, ops
    lib = _ext.load()          # the ctypes handle of the patched library (NLC_HIP_LIB)
    dev = torch.device("cuda:0")
    dt = torch.float32 if x3 else torch.bfloat16             # --x3: the f32 tensors / split-f16 math instantiation
    x = torch.randn(16, H, H, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    pw = ops.pack_conv(w, torch.zeros(cout), dt, dev, math="f16x3" if x3 else "native")
    r = torch.randn(16, H, H, cout, device=dev).to(dt) if res else None
    for _ in range(5):
        ops.conv2d(x, pw, res=r)
    torch.cuda.synchronize()
    lib.nlc_debug_clear_stamps()
    reps = 10
    for _ in range(reps):
        ops.conv2d(x, pw, res=r)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * NSLOT, dtype=np.uint64)
    rc = lib.nlc_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    a = buf.reshape(256, 2, NSLOT).astype(np.float64)      # (last launch only: every launch overwrites)
    names = ["k-loop", "epilogue (all of it)", "-", "-", "first k-step after epilogue", "k-step 4 of the tile", "k-steps 1 + 2 of the tile", "(tiles)", "step 0: first cluster issued", "step 0: + DMA issued", "step 0: + second cluster issued", "step 0: + counted wait", "loop top (next tile decoded)", "step 0: second-half fragments read", "step STAMP_KT, whole"]
    if os.environ.get("STAMP_STEPS"):
        sb = np.zeros(256 * 2 * 64, dtype=np.uint64)
        assert lib.nlc_debug_read_steps(sb.ctypes.data_as(ctypes.c_void_p), sb.nbytes) == 0
        sb = sb.reshape(256, 2, 64).astype(np.float64)
        nk = cin // 64 * 9
        for wv in range(2):
            per = np.median(sb[:, wv, :] / np.maximum(a[:, wv, 7:8], 1), axis=0)
            print(f"wave {wv * 4}: cycles per k-step (median over workgroups, mean over tiles; step = barrier to barrier; 'ep' = end of the last step to the end of the epilogue)")
            for c in range(nk // 9):
                print("   block %d: " % c + " ".join(f"{per[c * 9 + t]:6.0f}" for t in range(9)))
            print(f"   ep: {per[63]:6.0f}   sum of steps {per[:nk].sum():.0f}")
    for wv in range(2):
        tiles = a[:, wv, 7]
        print(f"wave {wv * 4}: tiles per workgroup {tiles.mean():.1f}")
        for k, nm in enumerate(names):
            if k in (2, 3, 7):
                continue
            per = a[:, wv, k] / np.maximum(tiles, 1)
            print(f"   {nm:32s} {np.median(per):10.0f} cycles per tile (min {per.min():.0f}, max {per.max():.0f})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        a = [int(v) for v in sys.argv[2:5]]
        run(*a, res="--res" in sys.argv, x3="--x3" in sys.argv)
