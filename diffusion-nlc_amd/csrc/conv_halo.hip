// 3x3 stride-1 pad-1 convolution with the input HALO tile resident in LDS (gfx950).
//
// The plain implicit-GEMM kernels (conv_fast.hip / conv_igemm.hip) stage the A operand once per tap,
// i.e. the same input pixels 9 times; measured on the 256->256 @256^2 layer that kernel is staging-
// bound (1.40 ms with the MFMAs removed vs 1.19 ms with the staging removed, 1.66 ms together).
// Here a workgroup owns a 16x16 output patch of one image x 128 output channels and, per 64-channel
// block, brings the 18x18 input halo into LDS ONCE; the nine taps then read shifted windows of it.
// Only the weights are staged per tap.  LDS-DMA bytes per MFMA drop 3.1x.
//
//   512 threads = 8 waves as 4 (M) x 2 (N); each wave 64 pixels (4 patch rows) x 64 cout = 4x4 MFMA tiles.
//   LDS: 2 halo stages (328 rows x 128 B, 324 used) + 4 weight stages (128 x 128 B) + DMA scratch = 154 KiB.
//   Halo of channel block cb+1 is fetched while block cb computes; weights run THREE k-steps ahead so that
//   the fragments of the next k-step can be read from LDS while the current one is still in the MFMA pipe
//   (register double buffering across the barrier - with one barrier per k-step and all waves in lockstep
//   the MFMA pipe otherwise idles through every fragment-read burst).  One counted s_waitcnt vmcnt(N) +
//   barrier per k-step.  All DMA is issued from inline asm (see conv_fast.hip for why).
//   Rows are XOR-swizzled chunk ^ (row & 7): conflict-free ds_read_b128 for 16 consecutive rows at ANY
//   alignment, which the tap shifts need (brute-forced).
#include "common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

__device__ uint4 g_zero_page_h[8];

constexpr int HT = 512;                         // threads
constexpr int PATCH = 16;                       // output patch edge
constexpr int HALO = PATCH + 2;                 // 18
constexpr int HALO_ROWS = HALO * HALO;          // 324
constexpr int A_INSTR = 41;                     // DMA wave-instructions per halo (8 rows each): 328 rows
constexpr int A_STAGE = A_INSTR * 8 * KB_BYTES; // 41 KiB
constexpr int B_STAGE = BN * KB_BYTES;          // 16 KiB
constexpr int NBST = 4;                         // weight stages (3 steps ahead)
constexpr int NA = 6, NB = 2;                   // DMA wave-instructions per wave: per halo / per weight tile
constexpr int SCRATCH = 8 * 8 * KB_BYTES;       // landing zone of the padding DMA instructions (8 KiB)
constexpr int HALO_LDS = 2 * A_STAGE + NBST * B_STAGE + SCRATCH;   // 157,696 B
constexpr int EPI_LD = BN + 4;

template <typename T> struct MmaH;
template <> struct MmaH<bf16_raw> {
    static constexpr int KBE = KB_BYTES / 2;
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct MmaH<float> {
    static constexpr int KBE = KB_BYTES / 4;
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int hoff(int row, int chunk) { return row * KB_BYTES + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16h(const void* gptr, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(lds_base)
                 : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait_h() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T>
__global__ __launch_bounds__(HT, 2) void conv_halo_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = ElemTraits<T>::kPerChunk;
    constexpr int KBE = MmaH<T>::KBE;
    constexpr int ES = (int)sizeof(T);

    const int tiles_x = p.Win / PATCH, tiles_y = p.Hin / PATCH;
    const int MTH = p.B * tiles_y * tiles_x;
    const int nblk = MTH * p.NT;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / p.NT, nt = bid - mt * p.NT;
    const int tb = mt / (tiles_y * tiles_x);
    const int trem = mt - tb * tiles_y * tiles_x;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * PATCH, x0 = tx * PATCH, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lrow = lane >> 3;                      // row within a DMA wave-instruction (8 rows x 128 B)
    const int lslot = lane & 7;                      // LDS 16-byte slot this lane's DMA lands in
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ldsB = lds0 + 2 * A_STAGE;
    const unsigned ldsScratch = ldsB + NBST * B_STAGE + wave * 8 * KB_BYTES;
    char* smemB = smem + 2 * A_STAGE;
    const int ncb = p.Cin_pad / KBE;
    const int nk = ncb * 9;

    // ---- this lane's (up to) 6 halo rows: DMA instruction q = wave + 8 j covers LDS rows 8q .. 8q+7; q >= 41 is padding
    int hpix[NA], hchunk[NA];
    unsigned hvalid = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int R = (wave + 8 * j) * 8 + lrow;
        const int hy = R / HALO, hx = R - hy * HALO;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = R < HALO_ROWS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        hpix[j] = ok ? (tb * p.Hin + iy) * p.Win + ix : 0;
        hchunk[j] = lslot ^ (R & 7);                 // source-side swizzle
        hvalid |= (ok ? 1u : 0u) << j;
    }
    const int64_t wrow = (int64_t)9 * p.Cin_pad * ES;
    const char* zero = reinterpret_cast<const char*>(g_zero_page_h);

    auto issue_A = [&](int cb) {
        const unsigned base = lds0 + (cb & 1) * A_STAGE + wave * 8 * KB_BYTES;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int cch = cb * KBE + hchunk[j] * PER;
            const char* src; int C, ch;
            if (cch < p.C0) { src = p.x0; C = p.C0; ch = cch; } else { src = p.x1; C = p.C1; ch = cch - p.C0; }
            const bool ok = ((hvalid >> j) & 1u) && cch < p.Ctot;
            const char* ptr = ok ? src + ((int64_t)hpix[j] * C + ch) * ES : zero;
            const bool real = (wave + 8 * j) < A_INSTR;                       // wave-uniform
            glds16h(ptr, real ? base + j * 64 * KB_BYTES : ldsScratch);      // every wave issues exactly NA instructions
        }
    };
    // weights of k-step kt (= cb*9 + tap) into a B stage; DMA instruction q = wave + 8 j covers rows 8q..8q+7
    auto issue_B = [&](int kt, int bstage) {
        const int cb = kt / 9, tap = kt - cb * 9;
        const unsigned base = ldsB + bstage * B_STAGE + wave * 8 * KB_BYTES;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = (wave + 8 * j) * 8 + lrow;
            const int gchunk = lslot ^ (row & 7);
            const char* ptr = p.w + (int64_t)(n0 + row) * wrow + ((int64_t)tap * p.Cin_pad + cb * KBE + gchunk * PER) * ES;
            glds16h(ptr, base + j * 64 * KB_BYTES);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int a_lane = wm * 4 * HALO + fr;           // halo row of (patch row wm*4, patch col fr) for tap (0,0)
    const int b_lane = wn * 64 + fr;

    // Per-lane LDS byte offsets of every fragment this lane will ever read, computed ONCE: the halo rows
    // (patch row + r, col + s) for the 6 x 3 (row, column) shifts and both k halves, and the weight rows.
    // With the taps unrolled at compile time the k-loop then carries no address arithmetic beyond one
    // add of the stage base per read (measured: the runtime-tap version spent more VALU issue cycles on
    // addresses than the MFMAs took).
    int aoff[6][3][2], boff[2];
#pragma unroll
    for (int yy = 0; yy < 6; ++yy)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) aoff[yy][s][kk] = hoff(a_lane + yy * HALO + s, kk * 4 + fq);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) boff[kk] = hoff(b_lane, kk * 4 + fq);     // + j*16 rows = + j*2048 bytes (same row & 7)

    // fragment sets (register double buffer)
    uint4 fa0[4], fb0[4], fa1[4], fb1[4];
    auto load_frags = [&](uint4 (&fa)[4], uint4 (&fb)[4], int astage, int bstage, auto tap_c, auto kk_c) {
        constexpr int tap = decltype(tap_c)::value, kk = decltype(kk_c)::value;
        constexpr int r = tap / 3, s = tap % 3;
        const char* As = smem + astage * A_STAGE;
        const char* Bs = smemB + bstage * B_STAGE + boff[kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 16 * KB_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + aoff[i + r][s][kk]);
    };
    auto mma16 = [&](const uint4 (&fa)[4], const uint4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) MmaH<T>::run(fa[i], fb[j], acc[i][j]);
    };

    // ---- prologue: halo of block 0, weights of steps 0..2
    issue_A(0);
    issue_B(0, 0);
    issue_B(1, 1);
    issue_B(2, 2);
    dma_wait_h<NB>();                                // halo 0 + weights 0,1 landed (weights 2 may fly)
    __syncthreads();
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    load_frags(fa0, fb0, 0, 0, K0{}, K0{});

    int kt = 0, bcur = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more_cb = cb + 1 < ncb;
        auto step = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            const int bnext = (bcur + 1) & 3;
            // Straight-line body (no data-dependent branches, so the compiler can software-pipeline it):
            // past the end the weight DMA simply re-fetches the last tile into a free stage and the
            // fragment prefetch reads stale-but-valid LDS; neither result is used.
            issue_B(min(kt + 3, nk - 1), (bcur + 3) & 3);
            if constexpr (tap == 0) { if (more_cb) issue_A(cb + 1); }      // AFTER the weights: they are needed first
            // kk = 0: read this step's second half while the first half is in the MFMA pipe
            // sched_barrier: keep "issue the NEXT fragments' LDS reads, THEN run the current MFMA cluster" - left
            // alone hipcc sinks each read next to its first use and the LDS latency is exposed twice per step
            // with every wave of the workgroup in the same phase (SQ_WAIT_ANY 51 %, MFMA busy 35 %).
            load_frags(fa1, fb1, cb & 1, bcur, tap_c, K1{});
            mma16(fa0, fb0);
            // kk = 1: read the NEXT step's first half (its weights were published by the previous barrier;
            //         the next halo, if tap == 8, landed by the end of tap 2)
            {
                constexpr int ntap = tap == 8 ? 0 : tap + 1;
                const int nast = tap == 8 ? (cb + 1) & 1 : cb & 1;
                load_frags(fa0, fb0, nast, bnext, std::integral_constant<int, ntap>{}, K0{});
            }
            mma16(fa1, fb1);
            // retire weights kt+2; instructions younger than them may stay in flight:
            // this step's weights kt+3 (NB) and a halo issued at this step (tap 0) or the previous one (tap 1)
            if constexpr (tap <= 1) { if (more_cb) dma_wait_h<NB + NA>(); else dma_wait_h<NB>(); }
            else dma_wait_h<NB>();
            __syncthreads();
            bcur = bnext;
            ++kt;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    }

    dma_wait_h<0>();          // the redundant tail fetches

    // ---- epilogue: ONE pass - all 8 waves drop their 64x64 accumulators into an f32 LDS image
    //      [256][EPI_LD] (132 KiB; every stage is free now), then all 512 threads apply bias / embedding /
    //      residual / activation and store 16-byte chunks of contiguous channels.
    const int HWo = p.Hin * p.Win;
    float* epi = reinterpret_cast<float*>(smem);
    const bool vec_ok = (p.Cout % PER) == 0;
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    epi[(wm * 64 + i * 16 + fq * 4 + reg) * EPI_LD + wn * 64 + j * 16 + fr] = acc[i][j][reg];
        __syncthreads();
        // each thread owns ONE 16-byte column chunk (cc) for 256*CPR/HT rows: per-column terms are loaded once,
        // the row loop is unrolled so the LDS reads / residual loads / stores of several rows are in flight together
        constexpr int CPR = BN / PER;
        constexpr int RSTEP = HT / CPR, NIT = 256 / RSTEP;
        const int cc = tid % CPR, row0 = tid / CPR;
        const int n = n0 + cc * PER;
        if (n < p.Cout) {
            const bool full = vec_ok && (n + PER <= p.Cout);
            float cbias[PER], cemb[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const bool okc = n + k < p.Cout;
                cbias[k] = (p.bias && okc) ? p.bias[n + k] : 0.f;
                cemb[k] = (p.emb && okc) ? p.emb[(int64_t)tb * p.emb_stride + n + k] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = row0 + it * RSTEP;                    // row within the 256-pixel patch
                const int py = row >> 4, px = row & 15;
                const int64_t m = ((int64_t)tb * p.Hin + y0 + py) * p.Win + x0 + px;
                float v[PER];
#pragma unroll
                for (int k = 0; k < PER; k += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(epi + row * EPI_LD + cc * PER + k);
                    v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
                }
                float rr[PER];
#pragma unroll
                for (int k = 0; k < PER; ++k) rr[k] = 0.f;
                if (p.res) {
                    const T* rp = reinterpret_cast<const T*>(p.res) + m * p.Cout + n;
                    if (full) chunk_to_f32<T>(*reinterpret_cast<const uint4*>(rp), rr);
                    else {
#pragma unroll
                        for (int k = 0; k < PER; ++k) rr[k] = (n + k < p.Cout) ? ElemTraits<T>::load(rp + k) : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    float x = v[k];
                    if (p.bias) x += cbias[k];
                    if (p.emb) x += cemb[k];
                    if (p.res) x += rr[k];
                    v[k] = apply_act(x * p.out_scale, p.act);
                }
                if (p.out_mode == NLC_OUT_NHWC) {
                    T* op = reinterpret_cast<T*>(p.out) + m * p.Cout + n;
                    if (full) *reinterpret_cast<uint4*>(op) = f32_to_chunk<T>(v);
                    else {
#pragma unroll
                        for (int k = 0; k < PER; ++k) if (n + k < p.Cout) ElemTraits<T>::store(op + k, v[k]);
                    }
                } else {
                    const int64_t rem = (int64_t)(y0 + py) * p.Win + x0 + px;
#pragma unroll
                    for (int k = 0; k < PER; ++k)
                        if (n + k < p.Cout) reinterpret_cast<float*>(p.out)[((int64_t)tb * p.Cout + n + k) * HWo + rem] = v[k];
                }
            }
        }
    }
}

template <typename T>
int launch_halo(const KParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        attr_set = true;
    }
    const int grid = p.B * (p.Hin / PATCH) * (p.Win / PATCH) * p.NT;
    hipLaunchKernelGGL((conv_halo_kernel<T>), dim3(grid), dim3(HT), HALO_LDS, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(halo): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

}  // namespace

// 3x3 / stride 1 / pad 1 / no upsample, H and W multiples of 16, enough tiles to fill the chip.
// NLC_CONV_HALO=0 disables, =1 forces (for eligible shapes) regardless of the tile count.
int nlc_conv_halo_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    static const char* force = getenv("NLC_CONV_HALO");
    if (force && force[0] == '0') return NLC_EUNSUPPORTED;
    if (!(p.KH == 3 && p.KW == 3 && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1 && !p.ups)) return NLC_EUNSUPPORTED;
    if (p.Hin % PATCH || p.Win % PATCH || p.Hout != p.Hin || p.Wout != p.Win) return NLC_EUNSUPPORTED;
    if ((int64_t)p.B * p.Hin * p.Win >= (1ll << 31)) return NLC_EUNSUPPORTED;
    const int blocks = p.B * (p.Hin / PATCH) * (p.Win / PATCH) * p.NT;
    if (!(force && force[0] == '1') && blocks < 256) return NLC_EUNSUPPORTED;
    return dtype == NLC_BF16 ? launch_halo<bf16_raw>(p, stream) : launch_halo<float>(p, stream);
}
