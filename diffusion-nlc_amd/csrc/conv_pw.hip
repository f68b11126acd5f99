// Pointwise (1x1, stride 1, unpadded) convolutions with many output tiles: the skip projections of the ResBlocks, the attention
// qkv / proj layers, the networks' "nin" shortcuts.  They are GEMMs C[M][N] = X[M][K] W[N][K]^T with short K (2 ... 12 k-blocks of 64
// channels) that move far more bytes per flop than the 3x3 layers, and conv_fast.hip - one workgroup per 128 x 128 tile, every tile
// with its own prologue, first DMA round trip and epilogue bubble - runs them at 3.2-4.4 TB/s of algorithmic traffic.
//
// What bounds them was measured, not assumed (profiles/r04_summary.md section 5): a pure LDS-DMA stream of the same tensor reaches
// 6.1 TB/s in EITHER access shape (128-byte column slices of 128 pixel rows, or whole rows) with two 16-KiB stages in flight per
// workgroup, 5.3 TB/s with the output write stream beside it (tools/dma_shape_probe.hip) - so neither the access shape nor the depth
// of the prefetch ring is the limit; the L2 is (tools/pw_pmc.sh: busy 96 % of the launch, 50 M requests, HBM read latency a relaxed
// 1 200 cycles): the activation tile once per channel tile, the SAME weight k-blocks again for every pixel tile, 32-byte write requests.
// Two kernels here, both persistent, TWO 256-thread workgroups per CU (a first version with ONE 512-thread workgroup per CU and
// 256-pixel tiles was 5-10 % SLOWER than conv_fast: its eight waves leave every barrier together, so all of them sit in the epilogue
// at the same time), both walking the channel tiles of a pixel tile side by side in one XCD (cowalk, below):
//  * conv_pwr_kernel (K <= 512, >= 1536 tiles): weights register-resident, all LDS a five-stage activation ring - see its comment;
//  * conv_pw_kernel (any K <= 768, >= 768 tiles): the tile, wave layout (4 waves = 2 (M) x 2 (N), 64 px x 64 cout each), fragment
//    layout and register-direct epilogue of conv_fast.hip (weights = MFMA A operand, weight rows permuted at DMA time: a lane ends up
//    with 16 consecutive channels of one pixel); the activation stream runs through a ring of THREE 16-KiB LDS stages and the weight
//    stream (L2 hits) through two, by LDS-DMA with counted vmcnt waits, and both rings run straight across tile boundaries: no per-tile
//    prologue, and while one workgroup of a CU is in its epilogue the other is in its k-loop.  2 x 80 KiB = all 160 KiB of LDS.
// Supported (dispatch falls back to conv_fast otherwise): 16-bit, NHWC output, whole 128-channel N-tiles, both input segments
// multiples of 64 channels, no embedding / activation / upsampled residual, >= 768 tiles (static ranges: the imbalance is one tile).
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
__global__ __launch_bounds__(PW_THREADS, 2) void conv_pw_kernel(const KParams p, int tiles_total, int NTn, int cowalk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = 8, ES = 2, KBE = 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = tid >> 3;                              // 0..31: LDS row (mod 32) this lane's DMA lands in
    const int gc = (tid & 7) ^ ((lr >> 1) & 7);           // global 16-byte chunk it fetches (source-side swizzle; rows lr + 32 i share it)
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nk = p.Cin_pad / KBE;
    const int HWo = p.Hout * p.Wout;

    // this workgroup's tiles: t0, t0 + tstep, ... (cnt of them); tile t = (pixel tile t / NTn, channel tile t % NTn).
    // cowalk (launches with 2..S channel tiles, S = workgroups per XCD): the NTn workgroups of a group sit in the SAME XCD (workgroup w runs on
    // XCD w % 8), hold one channel tile each for the whole launch and walk the same contiguous range of pixel tiles side by side, so an
    // activation tile comes from HBM once and the group's other reads of it hit that XCD's L2 while it is still there (measured before:
    // with the channel tiles of a pixel tile visited one after the other by one workgroup, FETCH_SIZE was 2.1 x the activation bytes -
    // 64 workgroups x 128 KiB between the two visits is twice the L2).  Otherwise: one contiguous range of tiles per workgroup.
    const int G = gridDim.x;
    int t0, cnt, tstep;
    if (cowalk) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = G >> 3;
        const int nt = slot % NTn, grp = (slot / NTn) * 8 + xcd, NG = (S / NTn) * 8, MT = tiles_total / NTn;
        const int mb = (int)(((int64_t)MT * grp) / NG), me = (int)(((int64_t)MT * (grp + 1)) / NG);
        t0 = mb * NTn + nt; cnt = me - mb; tstep = NTn;
    } else {
        t0 = (int)(((int64_t)tiles_total * blockIdx.x) / G);
        cnt = (int)(((int64_t)tiles_total * (blockIdx.x + 1)) / G) - t0; tstep = 1;
    }
    const int total = cnt * nk;                           // k-steps of this workgroup
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
    int at = t0, as_ = 0, a_left = cnt;                   // tile and k-block of the next activation step to issue; tiles left to issue
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
        if (++as_ == nk) { as_ = 0; at += tstep; if (--a_left > 0) a_setup(at); }
    };
    int bt = t0, bs = 0;                                  // tile and k-block of the next weight step to issue
    const char* bw = p.w + (int64_t)(t0 % NTn) * PW_BN * wrow;      // weight rows of that tile's channel tile (wave-uniform)
    auto issue_b = [&](int stage) {
        const unsigned b_base = lds0 + PW_NSA * PW_A_BYTES + stage * PW_B_BYTES + wave * 8 * KB_BYTES;
        const char* wb = bw + bs * KB_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_glds16_s(woff[i], wb, b_base + i * 32 * KB_BYTES);
        if (++bs == nk) { bs = 0; bt += tstep; bw = p.w + (int64_t)(bt % NTn) * PW_BN * wrow; }
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
            cs = 0; ct += tstep;
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
int launch_pw(const KParams& p, int tiles_total, int NTn, int grid, int cowalk, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pw_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
    });
    hipLaunchKernelGGL((conv_pw_kernel<T>), dim3(grid), dim3(PW_THREADS), PW_LDS, stream, p, tiles_total, NTn, cowalk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(pointwise): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Register-resident weights (K <= 512).  Counters of the kernel above on 512 -> 256 @256x256, B = 16 (tools/pw_pmc.sh,
// profiles/r04f_pw_pmc.json): HBM is NOT the limit - 1.3 GB read at an L2-side read latency of 1 200 cycles, 40 % of the streaming
// rate - the L2 is: busy 96 % of the launch, 50 M requests of which 33 M are 128-byte reads (the activation tile once per channel
// tile + the SAME weight k-blocks again for every pixel tile: 4.2 GB through L2 for 1.07 GB of activations) and 17 M are 32-byte
// writes.  Time tracks the L2 request count: activation stream alone 202 us, + stores 342, + weight stream 400, + MFMA 450.
// Here a workgroup keeps ONE channel tile for the whole launch (the side-by-side walk above), so its weights - 32 output channels x K per
// wave - are loaded ONCE into registers as MFMA A-operand fragments (128 VGPRs at K = 512): no weight stream, no weight fragment reads
// from LDS, and all 80 KiB of LDS are a five-stage activation ring (4 x 16 KiB in flight per workgroup).  Tile = 64 pixels x 128
// channels, a stage = 64 pixels x 128 input channels (256-byte row pieces, chunk index XOR-swizzled with the pixel's low 4 bits at
// DMA time so the fragment reads of 16 pixels hit 16 different 16-byte bank groups); every wave reads the whole stage (B operand) and
// multiplies it with its own 32 channels.  A lane ends up with 8 consecutive channels of a pixel: 16-byte stores, 64 bytes per pixel
// and wave (half the write requests).
//
// NORM (nlc_conv_desc.norm_out): the ResBlock whose skip projection this is also needs act(GroupNorm(x)) of the SAME input for its first
// 3x3 - a separate pass that reads x once more.  Here the stage that just fed the MFMAs is read from LDS a second time, normalised
// (y = a[b][c] x + b[b][c] from the nlc_groupnorm_coef table of the tile's image, kept in 4 KiB of LDS and reloaded when the image
// changes; + SiLU) and written to norm_out [M][C0 + C1]: thread (chunk column tid & 15, pixel row tid >> 4) handles its 8 channels of
// pixels row, row + 16, +32, +48 - whole 256-byte row pieces per wave-instruction on both the LDS and the global side.  Of a group's
// NTn workgroups the one with kb % NTn == nt does stage kb.  One stage less in the ring (4 x 16 + 4 KiB: two workgroups per CU).
constexpr int PR_BM = 64, PR_ROW = 256, PR_STAGE = PR_BM * PR_ROW;
constexpr int PR_COEF_BYTES = 4096;                       // float2 [512]
template <bool NORM> struct PrRing {
    static constexpr int NS = NORM ? 4 : 5, AH = NS - 1, LDS = NS * PR_STAGE + (NORM ? PR_COEF_BYTES : 0);
};

template <typename T, int NKB, bool NORM>
__global__ __launch_bounds__(PW_THREADS, 2) void conv_pwr_kernel(const KParams p, int MT, int NTn) {
    constexpr int PR_NS = PrRing<NORM>::NS, PR_AH = PrRing<NORM>::AH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int HWo = p.Hout * p.Wout;
    // group of NTn workgroups in one XCD, one channel tile each, the same pixel tiles side by side (conv_pw_kernel, cowalk)
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = G >> 3;
    const int nt = slot % NTn, grp = (slot / NTn) * 8 + xcd, NG = (S / NTn) * 8;
    const int mb = (int)(((int64_t)MT * grp) / NG), cnt = (int)(((int64_t)MT * (grp + 1)) / NG) - mb;
    if (cnt <= 0) return;
    const int total = cnt * NKB;
    const int n0 = nt * PW_BN + wave * 32;                 // this wave's 32 output channels

    // ---- weights: lane (fr, fq) holds, for channel block jb and 32-wide k slice q, k = 32 q + 8 fq ... + 7 of output channel
    //      n0 + (fr >> 2) * 8 + jb * 4 + (fr & 3) - the row permutation that leaves a lane with 8 consecutive channels of D
    uint4 wr[NKB][4][2];
    {
        const int64_t wrow = (int64_t)p.Cin_pad * 2;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const char* wp = p.w + (int64_t)(n0 + (fr >> 2) * 8 + jb * 4 + (fr & 3)) * wrow + fq * 16;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wr[kb][kk][jb] = *reinterpret_cast<const uint4*>(wp + (kb * 4 + kk) * 64);
        }
    }
    float cb[8];
    {
        const int n = n0 + fq * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) cb[k] = p.bias ? p.bias[n + k] : 0.f;
    }
    // Everything loaded so far is consumed HERE, in front of the loop: the compiler's wait-count pass would otherwise put its
    // s_waitcnt vmcnt(0) for these loads in front of their first use INSIDE the loop, where it drains the DMA ring at every step.
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
                asm volatile("" : "+v"(wr[kb][kk][jb].x), "+v"(wr[kb][kk][jb].y), "+v"(wr[kb][kk][jb].z), "+v"(wr[kb][kk][jb].w));
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(cb[k]));

    const char* x0p = p.x0; const char* x1p = p.x1;
    int c0w = p.C0, c1w = p.C1;
    asm volatile("" : "+s"(x0p), "+s"(x1p), "+s"(c0w), "+s"(c1w));
    const int nk0 = c0w >> 7;                              // 128-channel blocks of the first segment
    // ---- DMA: wave-instruction i of wave w fills LDS rows (pixels) (4 i + w) * 4 + (lane >> 4), 16 lanes per 256-byte row; the lane
    //      in chunk slot (lane & 15) fetches global chunk slot ^ (row & 15); (row & 15) = 4 w + (lane >> 4) for every i
    const int dr = wave * 4 + (lane >> 4);
    const int gch = (lane & 15) ^ dr;
    unsigned ao0[4], ao1[4];                               // byte offsets of this lane's 4 rows of the tile in k-block 0 of each segment
    auto a_setup = [&](int mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = min(mt * PR_BM + i * 16 + dr, p.M - 1);      // rows beyond M: the last pixel, never stored
            ao0[i] = (unsigned)((m * c0w + gch * 8) * 2);
            ao1[i] = (unsigned)((m * c1w + gch * 8) * 2);
        }
    };
    int at = mb, as_ = 0, a_left = cnt;
    auto issue = [&](int stage) {
        const unsigned base = lds0 + stage * PR_STAGE + wave * 4 * PR_ROW;
        const bool first = as_ < nk0;                      // wave-uniform
        const unsigned koff = (unsigned)((first ? as_ : as_ - nk0) * PR_ROW);
#pragma unroll
        for (int i = 0; i < 4; ++i) pw_glds16_s((first ? ao0[i] : ao1[i]) + koff, first ? x0p : x1p, base + i * 16 * PR_ROW);
        if (++as_ == NKB) { as_ = 0; ++at; if (--a_left > 0) a_setup(at); }
    };

    f32x4_t acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    };
    auto row16_sum = [](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
    const bool whole = (HWo % PR_BM) == 0;
    const float sc = p.out_scale;
    // ---- epilogue: lane (fr, fq) holds, for the 4 pixels m0 + 16 pb + fr, channels n0 + 8 fq + [0, 8): per wave exactly 4 row
    //      stores (full tiles) + 1 or 4 statistics atomics
    auto epilogue = [&](int mt) {
        const int m0 = mt * PR_BM, n = n0 + fq * 8;
        uint4 rq[4], pk[4];
        if (has_res) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int m = min(m0 + pb * 16 + fr, p.M - 1);
                rq[pb] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.Cout + n);
            }
        }
        Stat16 st;
        st.zero();
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            float v[8];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jb * 4 + r] = acc[pb][jb][r] + cb[jb * 4 + r];
            if (has_res) {
                float rr[8];
                chunk_to_f32<T>(rq[pb], rr);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += rr[k];
            }
            if (sc != 1.0f) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= sc;
            }
            pk[pb] = f32_to_chunk<T>(v);
            if (has_stats) {
                const int m = m0 + pb * 16;
                if (!whole) st.zero();
                if (m < p.M) st.add_chunk<T>(0, pk[pb]);
                if (!whole) {
                    st.s[0] = row16_sum(st.s[0]); st.s[1] = row16_sum(st.s[1]); st.q[0] = row16_sum(st.q[0]); st.q[1] = row16_sum(st.q[1]);
                    if (m < p.M) st.emit_row8(p.stats, p.div_hwo.div(m), p.Cout, n, p.stats_gran, fr);
                }
            }
        }
        if (has_stats && whole) {
            st.s[0] = row16_sum(st.s[0]); st.s[1] = row16_sum(st.s[1]); st.q[0] = row16_sum(st.q[0]); st.q[1] = row16_sum(st.q[1]);
            st.emit_row8(p.stats, m0 / HWo, p.Cout, n, p.stats_gran, fr);
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int m = m0 + pb * 16 + fr;
            if (m >= p.M) continue;
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.Cout + n) = pk[pb];
        }
    };

    // ---- NORM: coefficient table of the current image in LDS, and the normalisation of one stage
    [[maybe_unused]] char* coef_s = smem + PR_NS * PR_STAGE;
    [[maybe_unused]] int img = -1;
    [[maybe_unused]] const int Ctot = c0w + c1w;
    [[maybe_unused]] auto load_coefs = [&](int b) {          // workgroup-uniform; every thread moves 16 bytes (two channels)
        __syncthreads();                                      // nobody is still reading the previous image's table
        if (tid * 2 < Ctot) {
            const uint4 v = *reinterpret_cast<const uint4*>(p.gn_coef + ((int64_t)b * Ctot + tid * 2) * 2);
            *reinterpret_cast<uint4*>(coef_s + tid * 16) = v;
        }
        __syncthreads();
    };
    [[maybe_unused]] auto normalise = [&](const char* As, int mt, int kb) {
        const int cc = tid & 15, pr = tid >> 4;
        const int ch = kb * 128 + cc * 8;
        float ca[8], cb[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(coef_s + (ch + 2 * q) * 8);      // a0 b0 a1 b1
            ca[2 * q] = v.x; cb[2 * q] = v.y; ca[2 * q + 1] = v.z; cb[2 * q + 1] = v.w;
        }
        const bool silu = p.gn_act == NLC_ACT_SILU;
#pragma unroll 1                                                  // (register pressure: K = 512 holds 128 weight registers)
        for (int i = 0; i < 4; ++i) {
            const int row = pr + 16 * i, m = mt * PR_BM + row;
            const uint4 xv = *reinterpret_cast<const uint4*>(As + row * PR_ROW + ((cc ^ pr) << 4));
            float v[8];
            chunk_to_f32<T>(xv, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = fmaf(ca[k], v[k], cb[k]); if (silu) v[k] = silu_f(v[k]); }
            const uint4 pk = f32_to_chunk<T>(v);
            if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.norm_out) + (int64_t)m * Ctot + ch) = pk;
        }
    };

    // ---- prologue: PR_AH stages in flight, stage 0 landed
    a_setup(mb);
    if (total >= PR_AH) {
#pragma unroll
        for (int d = 0; d < PR_AH; ++d) issue(d);
        pw_wait<PW_PCS * (PR_AH - 1)>();
    } else {
        for (int d = 0; d < total; ++d) issue(d);
        pw_wait<0>();
    }
    __syncthreads();
    zero_acc();
    int stage = 0, g = 0;
    for (int ti = 0; ti < cnt; ++ti) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb, ++g) {
            const bool more = g + PR_AH < total;           // workgroup-uniform
            if constexpr (NORM) {
                if (kb == 0) {                              // a tile lies inside one image (dispatch: HWo % 64 == 0)
                    const int b = p.div_hwo.div((mb + ti) * PR_BM);
                    if (b != img) { load_coefs(b); img = b; }
                }
            }
            // the stage read during step g - 1 (every wave is past that step's barrier) receives step g + PR_AH
            if (more) issue(stage == 0 ? PR_NS - 1 : stage - 1);
            const char* As = smem + stage * PR_STAGE;
            // fragments of k slice kk + 1 are requested before the 8 MFMAs of slice kk (pinned: left alone the compiler reads two
            // fragments, waits, multiplies, reads two more - the LDS round trip in line with the matrix pipe)
            uint4 f[2][4];
            if constexpr (NORM && NKB == 4) {                // (no register to spare for the second fragment set: measured level anyway)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        const uint4 fx = *reinterpret_cast<const uint4*>(As + (pb * 16 + fr) * PR_ROW + (((kk * 4 + fq) ^ fr) << 4));
                        Mfma16<T>::run(wr[kb][kk][0], fx, acc[pb][0]);
                        Mfma16<T>::run(wr[kb][kk][1], fx, acc[pb][1]);
                    }
            } else {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) f[0][pb] = *reinterpret_cast<const uint4*>(As + (pb * 16 + fr) * PR_ROW + ((fq ^ fr) << 4));
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)
                        f[(kk + 1) & 1][pb] = *reinterpret_cast<const uint4*>(As + (pb * 16 + fr) * PR_ROW + ((((kk + 1) * 4 + fq) ^ fr) << 4));
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    Mfma16<T>::run(wr[kb][kk][0], f[kk & 1][pb], acc[pb][0]);      // D[channel][pixel]
                    Mfma16<T>::run(wr[kb][kk][1], f[kk & 1][pb], acc[pb][1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            }
            const bool full = (mb + ti) * PR_BM + PR_BM <= p.M;
            bool mine = false;                              // workgroup-uniform: this workgroup normalises this stage (4 more stores)
            if constexpr (NORM) {
                mine = (kb % NTn) == nt;
                if (mine) normalise(As, mb + ti, kb);
            }
            if (kb == NKB - 1) {
                epilogue(mb + ti);
                zero_acc();
                // step g + 1 must have landed; behind it this wave issued steps g + 2 .. g + PR_AH, then (NORM) this stage's 4 row
                // stores, then the 4 row stores and the statistics atomics (full tiles; a partial tile - the last of a launch - drains
                // everything).  Older stores than step g + 1's DMA are simply waited for.
                if (full && more) {
                    if (mine) {
                        if (!has_stats) pw_wait<PW_PCS * (PR_AH - 1) + 8>();
                        else if (whole) pw_wait<PW_PCS * (PR_AH - 1) + 8 + 1>();
                        else pw_wait<PW_PCS * (PR_AH - 1) + 8 + 4>();
                    } else {
                        if (!has_stats) pw_wait<PW_PCS * (PR_AH - 1) + 4>();
                        else if (whole) pw_wait<PW_PCS * (PR_AH - 1) + 4 + 1>();
                        else pw_wait<PW_PCS * (PR_AH - 1) + 4 + 4>();
                    }
                } else {
                    pw_wait<0>();
                }
            } else {
                if (more && full) { if (mine) pw_wait<PW_PCS * (PR_AH - 1) + 4>(); else pw_wait<PW_PCS * (PR_AH - 1)>(); }
                else pw_wait<0>();
            }
            __syncthreads();
            if (++stage == PR_NS) stage = 0;
        }
    }
}

template <typename T, int NKB, bool NORM>
int launch_pwr(const KParams& p, int MT, int NTn, int grid, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pwr_kernel<T, NKB, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PrRing<NORM>::LDS);
    });
    hipLaunchKernelGGL((conv_pwr_kernel<T, NKB, NORM>), dim3(grid), dim3(PW_THREADS), PrRing<NORM>::LDS, stream, p, MT, NTn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(pointwise, resident weights): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}
template <typename T>
int launch_pwr_k(const KParams& p, int MT, int NTn, int grid, hipStream_t stream) {
    if (p.norm_out) {
        switch (p.Cin_pad / 128) {
            case 1: return launch_pwr<T, 1, true>(p, MT, NTn, grid, stream);
            case 2: return launch_pwr<T, 2, true>(p, MT, NTn, grid, stream);
            case 3: return launch_pwr<T, 3, true>(p, MT, NTn, grid, stream);
            default: return launch_pwr<T, 4, true>(p, MT, NTn, grid, stream);
        }
    }
    switch (p.Cin_pad / 128) {
        case 1: return launch_pwr<T, 1, false>(p, MT, NTn, grid, stream);
        case 2: return launch_pwr<T, 2, false>(p, MT, NTn, grid, stream);
        case 3: return launch_pwr<T, 3, false>(p, MT, NTn, grid, stream);
        default: return launch_pwr<T, 4, false>(p, MT, NTn, grid, stream);
    }
}

// the resident-weights form: K <= 512 in whole 128-channel blocks per segment, whole groups of channel tiles per XCD, 32-bit offsets
bool pwr_ok(const KParams& p, int grid) {
    if (p.tuning & 65536) return false;                                         // tuning bit 16: never (A/B against the streaming form)
    if ((p.C0 & 127) != 0 || (p.C1 & 127) != 0 || p.Cin_pad > 512) return false;
    // every workgroup first loads its channel tile's weights (128 KiB from L2): launches of fewer than ~1 500 tiles stay with the streaming
    // form (512 -> 256 @64x64, B = 16, 1 024 tiles: 31.1 us streaming, 34.9 resident, 35.0 conv_fast; 256 -> 512 @16x16, B = 200, 1 600: 34.4 / 32.0 / 38.6)
    if ((int64_t)cdiv(p.M, PW_BM) * (p.Cout / PW_BN) < 1536) return false;
    const int NTn = p.Cout / PW_BN, S = grid >> 3;
    if ((grid & 7) != 0 || S < NTn || (S % NTn) != 0) return false;
    if ((int64_t)p.M * p.C0 * 2 >= (1ll << 32) || (int64_t)p.M * p.C1 * 2 >= (1ll << 32)) return false;
    return true;
}

}  // namespace

// 1 if the persistent pointwise kernel takes this launch (conv_params.h)
int nlc_conv_pw_ok(const KParams& p, int dtype) {
    if (!nlc_is16(dtype) || (p.tuning & 32768)) return 0;                       // tuning bit 15: never (A/B against conv_fast)
    if (p.policy == NLC_CONV_GENERIC) return 0;
    if (!(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && !p.ups)) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win) return 0;
    if (p.out_mode != NLC_OUT_NHWC || p.act != NLC_ACT_NONE || p.emb || p.res_ups || (p.gn_coef && !p.norm_out) || p.ksplit > 1) return 0;
    if ((p.Cout % PW_BN) != 0 || (p.C0 % 64) != 0 || (p.C1 % 64) != 0 || p.Cin_pad != p.C0 + p.C1) return 0;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15) != 0) return 0;
    const int HWo = p.Hout * p.Wout;
    if ((p.M & 15) != 0 || (HWo & 15) != 0) return 0;                           // 16-pixel rows never straddle images (statistics)
    if (p.Cin_pad > 768) return 0;        // longer K: the launch is matrix-bound, not latency-bound, and conv_fast's leaner k-loop wins
                                          // (1024 -> 512 @64x64, B = 16: 99 vs 92 us)
    const int64_t tiles = (int64_t)cdiv(p.M, PW_BM) * (p.Cout / PW_BN);
    return tiles >= 768 ? 1 : 0;          // static tile ranges; below, conv_fast's one-tile workgroups are level or ahead (32x32, B = 16: 24 us both)
}

static int pw_grid(const KParams& p) {
    static DeviceOnce once;
    const int ncu = once.ncu[nlc_device_once(once, [] {})];
    const int tiles = cdiv(p.M, PW_BM) * (p.Cout / PW_BN);
    return tiles < 2 * ncu ? tiles : 2 * ncu;                                   // two persistent workgroups per CU
}

// nlc_conv_desc.norm_out: the resident-weights form, tiles inside one image, the coefficient table of one image in 4 KiB of LDS
int nlc_conv_pw_norm_ok(const KParams& pin, int dtype) {
    KParams p = pin;
    if (!p.norm_out) p.norm_out = reinterpret_cast<char*>(16);                  // (query before the buffer exists)
    if (!p.gn_coef) p.gn_coef = reinterpret_cast<const float*>(16);
    if (!nlc_conv_pw_ok(p, dtype) || !pwr_ok(p, pw_grid(p))) return 0;
    if (((p.Hout * p.Wout) % PR_BM) != 0 || (p.C0 + p.C1) * 8 > PR_COEF_BYTES) return 0;
    return 1;
}

int nlc_conv_pw_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!nlc_conv_pw_ok(p, dtype)) return NLC_EUNSUPPORTED;
    if (p.norm_out && !nlc_conv_pw_norm_ok(p, dtype)) return NLC_EUNSUPPORTED;
    static DeviceOnce once;
    const int ncu = once.ncu[nlc_device_once(once, [] {})];
    const int NTn = p.Cout / PW_BN;
    const int tiles = cdiv(p.M, PW_BM) * NTn;
    const int grid = tiles < 2 * ncu ? tiles : 2 * ncu;                         // two persistent workgroups per CU
    if (pwr_ok(p, grid)) {
        const int MT = cdiv(p.M, PR_BM);
        return dtype == NLC_BF16 ? launch_pwr_k<bf16_raw>(p, MT, NTn, grid, stream) : launch_pwr_k<f16_raw>(p, MT, NTn, grid, stream);
    }
    // side-by-side walk of the channel tiles (kernel comment): whole groups of NTn workgroups per XCD; tuning bit 17 = never (A/B)
    const int S = grid >> 3;
    const int cowalk = (NTn > 1 && (grid & 7) == 0 && S >= NTn && (S % NTn) == 0 && !(p.tuning & 131072)) ? 1 : 0;
    if (dtype == NLC_BF16) return launch_pw<bf16_raw>(p, tiles, NTn, grid, cowalk, stream);
    return launch_pw<f16_raw>(p, tiles, NTn, grid, cowalk, stream);
}
