// Pointwise (1x1, stride 1, unpadded) convolutions with many output tiles: the skip projections of the ResBlocks, the attention
// qkv / proj layers, the networks' "nin" shortcuts.  They are HBM-bound GEMMs, C[M][N] = X[M][K] W[N][K]^T with short K (2 ... 12 k-blocks
// of 64 channels), and conv_fast.hip runs them latency-bound: a workgroup keeps ONE k-step (16 KiB of activations) in flight, two
// workgroups per CU -> 32 KiB of activation bytes in flight per CU, half of what a CU's share of HBM needs behind a ~2 us miss
// (bandwidth-delay: 31 GB/s x 2 us = 62 KiB; MI355X_MICROARCH.md "72 KiB in flight per CU hide most of an HBM miss"): 4.3-4.4 TB/s
// where the streaming kernels reach 5.8, plus a prologue (first DMA round trip) and an epilogue bubble per 128-pixel tile.  The
// four-stage / one-workgroup-per-CU form of conv_fast has the bytes in flight but half the waves, and measured 40-65 % slower.
//
// This kernel: TWO persistent 256-thread workgroups per CU, each walking a contiguous range of (128 pixel x 128 channel) tiles - the
// tile, wave layout (4 waves = 2 (M) x 2 (N), 64 px x 64 cout each), fragment layout and register-direct epilogue of conv_fast.hip
// (weights = MFMA A operand, weight rows permuted at DMA time: a lane ends up with 16 consecutive channels of one pixel).  What
// changes is the staging: the activation stream (HBM) runs through a ring of THREE 16-KiB LDS stages and the weight stream (L2 hits)
// through two, by LDS-DMA with counted vmcnt waits, and both rings run straight across tile boundaries: per workgroup two
// activation k-steps are in flight the whole launch (64 KiB per CU), there is no per-tile prologue, and while one workgroup of a CU
// is in its epilogue the other is in its k-loop.  (A first version with ONE 512-thread workgroup per CU and 256-pixel tiles had the
// same bytes in flight and was 5-10 % SLOWER than conv_fast: its eight waves leave every barrier together, so all of them sit in the
// epilogue at the same time with the matrix pipe idle.)  2 x 80 KiB = all 160 KiB of LDS.  The N-tiles of one pixel tile are
// consecutive in a workgroup's range, so the second read of the activation tile comes from the XCD's L2.
// Supported (dispatch falls back to conv_fast otherwise): 16-bit, NHWC output, whole 128-channel N-tiles, both input segments
// multiples of 64 channels, no embedding / activation / upsampled residual, >= 2048 tiles (static ranges: the imbalance is one tile).
#include "common.h"
#include "conv_params.h"

namespace {

constexpr int PW_BM = 128, PW_BN = 128, PW_THREADS = 256, PW_NSA = 3, PW_NSB = 2;
constexpr int PW_A_BYTES = PW_BM * KB_BYTES;              // 16 KiB per activation stage
constexpr int PW_B_BYTES = PW_BN * KB_BYTES;              // 16 KiB per weight stage
constexpr int PW_LDS = PW_NSA * PW_A_BYTES + PW_NSB * PW_B_BYTES;      // 80 KiB: two workgroups per CU
constexpr int PW_PCS = 4;                                 // LDS-DMA wave-instructions per wave, k-step and stream (4 activation, 4 weight pieces)

__device__ __forceinline__ int pw_lds_off(int row, int chunk) { return row * KB_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void pw_glds16(const void* gptr, unsigned lds_base) {
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
__device__ __forceinline__ void pw_glds16_s(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_base), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void pw_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T>
__global__ __launch_bounds__(PW_THREADS, 2) void conv_pw_kernel(const KParams p, int tiles_total, int NTn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = 8, ES = 2, KBE = 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = tid >> 3;                              // 0..31: LDS row (mod 32) this lane's DMA lands in
    const int gc = (tid & 7) ^ ((lr >> 1) & 7);           // global 16-byte chunk it fetches (source-side swizzle; rows lr + 32 i share it)
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nk = p.Cin_pad / KBE;
    const int HWo = p.Hout * p.Wout;

    // this workgroup's contiguous tile range; tile t = (pixel tile t / NTn, channel tile t % NTn)
    const int G = gridDim.x;
    const int t0 = (int)(((int64_t)tiles_total * blockIdx.x) / G), t1 = (int)(((int64_t)tiles_total * (blockIdx.x + 1)) / G);
    const int total = (t1 - t0) * nk;                     // k-steps of this workgroup
    if (total <= 0) return;

    const int64_t wrow = (int64_t)p.Cin_pad * ES;         // bytes per packed weight row (one tap)
    // LDS row R of the weight tile receives output channel (R & 64) + ((R & 15) >> 2) * 16 + ((R >> 4) & 3) * 4 + (R & 3) (conv_fast.hip)
    unsigned woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = lr + 32 * i;
        const int ch = (R & 64) + ((R & 15) >> 2) * 16 + ((R >> 4) & 3) * 4 + (R & 3);
        woff[i] = (unsigned)((int64_t)ch * wrow + (int64_t)gc * PER * ES);
    }
    // The two input segments' base pointers and widths, pinned in SGPRs: left to itself the compiler turns the per-lane choice between
    // the kernel arguments (p.x0, p.C0) and (p.x1, p.C1) into a per-lane LOAD from the kernarg segment - a vector-memory load followed by
    // s_waitcnt vmcnt(0) in front of every DMA, which drains the prefetch ring at every k-step.
    const char* x0p = p.x0; const char* x1p = p.x1;
    int c0w = p.C0, c1w = p.C1;
    asm volatile("" : "+s"(x0p), "+s"(x1p), "+s"(c0w), "+s"(c1w));
    // ---- issue cursors, across tile boundaries: the activation stream runs two k-steps ahead of the compute cursor, the weight
    //      stream one
    int at = t0, as_ = 0;                                 // tile and k-block of the next activation step to issue
    // this lane's 4 activation rows of that tile (clamped to M - 1: rows beyond M are fetched from the last pixel and never stored -
    // no zero page, no branch) as byte addresses of the lane's chunk in k-block 0 of each input segment: a k-step adds 128 bytes
    const char* ap0[4]; const char* ap1[4];
    const int nk0 = c0w / KBE;                            // k-blocks of the first segment
    auto a_setup = [&](int tile) {
        const int mt = tile / NTn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = min(mt * PW_BM + lr + 32 * i, p.M - 1);
            ap0[i] = x0p + (m * c0w + gc * PER) * ES;
            ap1[i] = x1p + (m * c1w + gc * PER) * ES;     // (never dereferenced when C1 == 0: every k-block is in the first segment)
        }
    };
    auto issue_a = [&](int stage) {
        const unsigned a_base = lds0 + stage * PW_A_BYTES + wave * 8 * KB_BYTES;
        const bool first = as_ < nk0;                     // wave-uniform: a k-block never straddles the segments (C0 % 64 == 0: dispatch)
        const int koff = (first ? as_ : as_ - nk0) * KB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_glds16((first ? ap0[i] : ap1[i]) + koff, a_base + i * 32 * KB_BYTES);
        if (++as_ == nk) { as_ = 0; ++at; if (at < t1) a_setup(at); }
    };
    int bt = t0, bs = 0;                                  // tile and k-block of the next weight step to issue
    const char* bw = p.w + (int64_t)(t0 % NTn) * PW_BN * wrow;      // weight rows of that tile's channel tile (wave-uniform)
    auto issue_b = [&](int stage) {
        const unsigned b_base = lds0 + PW_NSA * PW_A_BYTES + stage * PW_B_BYTES + wave * 8 * KB_BYTES;
        const char* wb = bw + bs * KB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_glds16_s(woff[i], wb, b_base + i * 32 * KB_BYTES);
        if (++bs == nk) { bs = 0; ++bt; bw = p.w + (int64_t)(bt % NTn) * PW_BN * wrow; }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    // All sixteen fragment reads of a k-step are issued up front (64 VGPRs) and the 32 MFMAs follow in k order: the first half's
    // products start as soon as its eight fragments have landed (counted lgkmcnt) while the second half's reads are still in flight.
    // The eight waves of the workgroup leave every barrier together, so the two waves of a SIMD would otherwise stall on the same
    // LDS reads at the same time and then contend for the matrix pipe at the same time.
    auto compute = [&](int sa, int sb) {
        const char* As = smem + sa * PW_A_BYTES;
        const char* Bs = smem + PW_NSA * PW_A_BYTES + sb * PW_B_BYTES;
        uint4 fa[2][4], fb[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[kk][j] = *reinterpret_cast<const uint4*>(Bs + pw_lds_off(wn * 64 + j * 16 + fr, chunk));
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = *reinterpret_cast<const uint4*>(As + pw_lds_off(wm * 64 + i * 16 + fr, chunk));
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);       // 16 DS reads first ...
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mfma16<T>::run(fb[kk][j], fa[kk][i], acc[i][j]);      // D[channel][pixel]
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);       // ... then the 32 MFMAs
    };
    // sum over the 16 pixel lanes of a DPP row: an all-reduce (conv_fast.hip)
    auto row16_sum = [](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
    const bool whole = (HWo % PW_BM) == 0;                // a 128-pixel tile lies inside one image: one statistics atomic per wave and tile
    const float sc = p.out_scale;

    // ---- epilogue of tile `tile` straight from registers: lane (fr, fq) of wave (wm, wn) holds, for the 4 pixels
    //      m0 + wm*64 + i*16 + fr, the 16 consecutive channels n0 + wn*64 + fq*16 + [0, 16).  Returns nothing; issues, per wave,
    //      exactly 8 row stores (full tiles) + 1 or 4 statistics atomics - the counted wait behind it relies on that.
    auto epilogue = [&](int tile) {
        const int mt = tile / NTn, nt = tile - mt * NTn;
        const int m0 = mt * PW_BM, n = nt * PW_BN + wn * 64 + fq * 16;
        float cb[16];
        if (p.bias) {
            const float4* bp4 = reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 b4 = bp4[q]; cb[q * 4] = b4.x; cb[q * 4 + 1] = b4.y; cb[q * 4 + 2] = b4.z; cb[q * 4 + 3] = b4.w; }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) cb[k] = 0.f;
        }
        uint4 rq0[4], rq1[4], pka[4], pkb[4];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = min(m0 + wm * 64 + i * 16 + fr, p.M - 1);
                const T* rp = reinterpret_cast<const T*>(p.res) + (int64_t)m * p.Cout + n;
                rq0[i] = *reinterpret_cast<const uint4*>(rp);
                rq1[i] = *reinterpret_cast<const uint4*>(rp + 8);
            }
        }
        Stat16 st16;
        st16.zero();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) v[j * 4 + reg] = acc[i][j][reg] + cb[j * 4 + reg];
            if (has_res) {
                float rr[16];
                chunk_to_f32<T>(rq0[i], rr); chunk_to_f32<T>(rq1[i], rr + 8);
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] += rr[k];
            }
            if (sc != 1.0f) {
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] *= sc;
            }
            pka[i] = f32_to_chunk<T>(v); pkb[i] = f32_to_chunk<T>(v + 8);
            if (has_stats) {                         // of the STORED (rounded) values - what the GroupNorm that follows reads
                const int m = m0 + wm * 64 + i * 16;                              // first pixel of this 16-row (wave-row-uniform)
                if (!whole) st16.zero();
                if (m < p.M) { st16.add_chunk<T>(0, pka[i]); st16.add_chunk<T>(1, pkb[i]); }
                if (!whole) {                        // rows of 16 consecutive pixels never straddle images (dispatch: M, HWo multiples of 16)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st16.s[k] = row16_sum(st16.s[k]); st16.q[k] = row16_sum(st16.q[k]); }
                    if (m < p.M) st16.emit_row(p.stats, p.div_hwo.div(m), p.Cout, n, p.stats_gran, fr);
                }
            }
        }
        if (has_stats && whole) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { st16.s[k] = row16_sum(st16.s[k]); st16.q[k] = row16_sum(st16.q[k]); }
            st16.emit_row(p.stats, m0 / HWo, p.Cout, n, p.stats_gran, fr);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + fr;
            if (m >= p.M) continue;
            T* op = reinterpret_cast<T*>(p.out) + (int64_t)m * p.Cout + n;
            *reinterpret_cast<uint4*>(op) = pka[i];
            *reinterpret_cast<uint4*>(op + 8) = pkb[i];
        }
    };

    // ---- prologue: activations of steps 0 and 1 and weights of step 0 in flight, step 0 landed
    a_setup(t0);
    issue_b(0);
    issue_a(0);
    if (total > 1) { issue_a(1); pw_wait<PW_PCS>(); } else pw_wait<0>();
    __syncthreads();
    zero_acc();
    int ct = t0, cs = 0;                                  // compute cursor
    int sa = 0, sb = 0;                                   // stages of the current step
    for (int g = 0; g < total; ++g) {
        const bool more1 = g + 1 < total, more2 = g + 2 < total;      // workgroup-uniform
        // weights of step g + 1 first, then activations of step g + 2: their stages were last read during step g - 1, and every wave
        // has passed that step's trailing barrier.  In issue order this wave now has in flight: [A(g+1)] B(g+1) A(g+2).
        if (more1) issue_b(sb ^ 1);
        if (more2) { int s2 = sa + 2; if (s2 >= PW_NSA) s2 -= PW_NSA; issue_a(s2); }
        compute(sa, sb);
        if (++cs == nk) {
            const bool full = (ct / NTn) * PW_BM + PW_BM <= p.M;                  // workgroup-uniform: every lane stores its 4 rows
            epilogue(ct);
            zero_acc();
            cs = 0; ++ct;
            // A(g+1) and B(g+1) must have landed.  Behind them this wave issued, in order: A(g+2) (if any), the epilogue's 8 row stores
            // and its statistics atomics (the epilogue's loads have completed: their values were used) - full tiles only; a partial
            // tile (the last pixel tile of a launch) drains everything
            if (full && more2) {
                if (!has_stats) pw_wait<PW_PCS + 8>();
                else if (whole) pw_wait<PW_PCS + 8 + 1>();
                else pw_wait<PW_PCS + 8 + 4>();
            } else {
                pw_wait<0>();
            }
        } else {
            if (more2) pw_wait<PW_PCS>(); else pw_wait<0>();
        }
        __syncthreads();
        if (++sa == PW_NSA) sa = 0;
        sb ^= 1;
    }
}

template <typename T>
int launch_pw(const KParams& p, int tiles_total, int NTn, int grid, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
    });
    hipLaunchKernelGGL((conv_pw_kernel<T>), dim3(grid), dim3(PW_THREADS), PW_LDS, stream, p, tiles_total, NTn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(pointwise): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

}  // namespace

// 1 if the persistent pointwise kernel takes this launch (conv_params.h)
int nlc_conv_pw_ok(const KParams& p, int dtype) {
    if (!nlc_is16(dtype) || (p.tuning & 32768)) return 0;                       // tuning bit 15: never (A/B against conv_fast)
    if (p.policy == NLC_CONV_GENERIC) return 0;
    if (!(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && !p.ups)) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win) return 0;
    if (p.out_mode != NLC_OUT_NHWC || p.act != NLC_ACT_NONE || p.emb || p.res_ups || p.gn_coef || p.ksplit > 1) return 0;
    if ((p.Cout % PW_BN) != 0 || (p.C0 % 64) != 0 || (p.C1 % 64) != 0 || p.Cin_pad != p.C0 + p.C1) return 0;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) != 0) return 0;
    const int HWo = p.Hout * p.Wout;
    if ((p.M & 15) != 0 || (HWo & 15) != 0) return 0;                           // 16-pixel rows never straddle images (statistics)
    if (p.Cin_pad > 768) return 0;        // longer K: the launch is matrix-bound, not latency-bound, and conv_fast's leaner k-loop wins
                                          // (1024 -> 512 @64x64, B = 16: 99 vs 92 us)
    const int64_t tiles = (int64_t)cdiv(p.M, PW_BM) * (p.Cout / PW_BN);
    return tiles >= 2048 ? 1 : 0;                                               // static tile ranges: >= 4 tiles per workgroup
}

int nlc_conv_pw_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!nlc_conv_pw_ok(p, dtype)) return NLC_EUNSUPPORTED;
    static DeviceOnce once;
    const int ncu = once.ncu[nlc_device_once(once, [] {})];
    const int NTn = p.Cout / PW_BN;
    const int tiles = cdiv(p.M, PW_BM) * NTn;
    const int grid = tiles < 2 * ncu ? tiles : 2 * ncu;                         // two persistent workgroups per CU
    if (dtype == NLC_BF16) return launch_pw<bf16_raw>(p, tiles, NTn, grid, stream);
    return launch_pw<f16_raw>(p, tiles, NTn, grid, stream);
}
