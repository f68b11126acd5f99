// Fast path of nlc_conv2d: stride-1 3x3 (pad 1) and 1x1 (pad 0) convolutions, the shapes that carry
// >99 % of the sampling FLOPs.  Same tile geometry and LDS image as conv_igemm.hip (128x128 tile,
// 4 waves x (4x4) 16x16 MFMA tiles, 128-byte k-blocks, XOR-swizzled rows) but
//   * global -> LDS staging by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass;
//     the swizzle is applied on the per-lane SOURCE address (the DMA destination is lane-linear) and
//     out-of-image taps read a zero page, so padding costs nothing.  The DMA is issued from inline
//     asm: hipcc drains vmcnt(0) in front of every ds_read it cannot prove disjoint from a pending
//     LDS-DMA (i.e. all of them), which serialises load and compute; here the DMA of k-step k+1 is
//     invisible to the compiler, flies under the MFMAs of step k and is retired by ONE hand-placed
//     s_waitcnt vmcnt(0) in front of the step's barrier (every wave waits for its own DMA, then the
//     barrier publishes all of them - cdna_hip_programming.md §5.7 item 1);
//   * the tap loop is fully unrolled inside the channel-block loop: per-row input pixel indices and
//     validity bits for all taps are computed once per tile, a k-step's address work is one
//     multiply-add per row;
//   * the epilogue runs straight from registers: the MFMA operands are swapped (weights = A) and the weight rows are
//     permuted at DMA time so that a lane ends up with 16 CONSECUTIVE output channels of one pixel - bias / embedding /
//     residual / scale / activation, 16-byte residual loads and output stores, no LDS transpose and no barrier
//     (the short-K 1x1 tiles used to spend a quarter of their time in the LDS-staged epilogue);
//   * optionally the GroupNorm statistics of the output ride along (see conv_halo.hip).
#include "common.h"
#include "conv_params.h"
#include <utility>

namespace {

__device__ uint4 g_zero_page[8];      // 128 zero bytes: source of every out-of-image / out-of-channel chunk (and of a missing bias)

template <typename T> using Mma = Mfma16<T>;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * KB_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

// One LDS-DMA wave-instruction: lane l's 16 bytes at gptr land at LDS byte address lds_base + 16*l.
// lds_base must be wave-uniform (it goes through M0, saved and restored around the instruction).
__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_base) {
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
// SGPR-base form: address = sbase (wave-uniform) + voff (per lane, 32-bit)
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_base) {
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
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


template <int N> __device__ __forceinline__ void dma_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// STAGES = 2: two workgroups per CU hide each other's DMA latency - the choice wherever a launch has more workgroups than CUs
// (STAGES = 3 / 4 measured 10-45 % slower on every such shape of tools/conv_bench.py, profiles/r02_summary.md: eight resident waves
// matter more than a deeper pipeline).  STAGES = 4: launches with AT MOST one workgroup per CU (the 8x8 ... 32x32 levels at small
// batch: 16-256 workgroups) have nobody to overlap with, and with one k-step in flight every k-step pays a whole memory latency
// (measured 2.3 us per k-step, 42 us for 18 k-steps of 512 -> 512 @8x8, B = 8); three k-steps in flight behind a counted wait.
// X3 (T = float, NLC_MATH_F16X3): f32 tensors, split-f16 matrix math (conv_halo.hip has the scheme).  The weights arrive packed as
// (hi, lo) halves, so the two weight reads of a k-step ARE the two operand halves; the activation tile is staged once per tap here
// and read by two waves only, so its fragments are split in registers (conv_params.h: f16x3_split4), 64 VALU per 48 MFMAs.
template <typename T, int TAPS, int STAGES, bool X3 = false>
__global__ __launch_bounds__(NTHREADS, STAGES == 2 ? 2 : 1) void conv_fast_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = ElemTraits<T>::kPerChunk;
    constexpr int KBE = Mma<T>::KBE;
    constexpr int ES = (int)sizeof(T);
    constexpr int KW = TAPS == 9 ? 3 : 1;

    const int nblk = p.MT * p.NT;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / p.NT, nt = bid - mt * p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    if constexpr (X3 && std::is_same<T, float>::value) f16x3_enter();   // f32 -> f16 conversions saturate (conv_params.h)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = tid >> 3;                            // LDS row (mod 32) this lane's DMA lands in
    const int gc = (tid & 7) ^ ((lr >> 1) & 7);         // global 16-byte chunk it fetches (source-side swizzle)
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS byte address of smem

    const int HWo = p.Hout * p.Wout;
    const int HL = p.ups ? 2 * p.Hin : p.Hin, WL = p.ups ? 2 * p.Win : p.Win;
    const int ncb_all = p.Cin_pad / KBE;
    // split-K: blockIdx.y owns the channel blocks [cb0, cb1)
    const int cb0 = (int)(((int64_t)ncb_all * blockIdx.y) / p.ksplit), cb1 = (int)(((int64_t)ncb_all * (blockIdx.y + 1)) / p.ksplit);

    // ---- per-row tap tables: input pixel index (within the whole tensor) and validity bit per tap.  Launch-invariant divisors go
    //      through host-computed multiply-shift constants, and a tap is (row r, column s): three row bases + three column offsets and
    //      their validity per output pixel instead of nine independent (y, x, four compares, multiply) chains - the tables cost
    //      7.3 k cycles of a 60 k-cycle small-map launch before the first DMA could be issued (tools/fast_stamps.py).
    int pix[4][TAPS];
    unsigned vmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lr + 32 * i;
        const bool mv = m < p.M;
        const int mm = mv ? m : 0;
        if constexpr (TAPS == 1) {
            if (!p.ups) {                            // unpadded stride-1 1x1: input pixel = output pixel, no (image, y, x) needed
                pix[i][0] = mm;
                vmask[i] = mv ? 1u : 0u;
                continue;
            }
        }
        const int b = p.div_hwo.div(mm);
        const int rem = mm - b * HWo;
        const int oy = p.div_wo.div(rem), ox = rem - oy * p.Wout;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        const int sh = p.ups ? 1 : 0;
        int rowbase[KW]; bool oky[KW]; int col[KW]; bool okx[KW];
#pragma unroll
        for (int r = 0; r < KW; ++r) {
            const int iy = iy0 + r, ix = ix0 + r;
            oky[r] = mv && (unsigned)iy < (unsigned)HL;
            okx[r] = (unsigned)ix < (unsigned)WL;
            rowbase[r] = (b * p.Hin + (iy >> sh)) * p.Win;
            col[r] = ix >> sh;
        }
        unsigned vm = 0;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const bool ok = oky[t / KW] && okx[t % KW];
            pix[i][t] = ok ? rowbase[t / KW] + col[t % KW] : 0;
            vm |= (ok ? 1u : 0u) << t;
        }
        vmask[i] = vm;
    }
    const int64_t wrow = (int64_t)TAPS * p.Cin_pad * ES;
    // LDS row R = lr + 32 i of the weight tile receives output channel (R & 64) + ((R & 15) >> 2) * 16 + ((R >> 4) & 3) * 4
    // + (R & 3): with the MFMA operands swapped a lane then holds 16 consecutive channels of one pixel (conv_halo.hip)
    unsigned woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = lr + 32 * i;
        const int ch = (R & 64) + ((R & 15) >> 2) * 16 + ((R >> 4) & 3) * 4 + (R & 3);
        woff[i] = (unsigned)((int64_t)ch * wrow + (int64_t)gc * PER * ES);
    }
    const char* wtile = p.w + (int64_t)n0 * wrow;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    // issue the LDS-DMA of k-step (cb, tap) into `stage` (8 wave-instructions per wave: 4 A, 4 B)
    auto stage_step = [&](int stage, int cb, auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        const unsigned a_base = lds0 + stage * STAGE_BYTES + wave * 8 * KB_BYTES;
        const unsigned b_base = a_base + BM * KB_BYTES;
        const int cch = cb * KBE + gc * PER;
        const char* src; int C, ch;
        if (cch < p.C0) { src = p.x0; C = p.C0; ch = cch; } else { src = p.x1; C = p.C1; ch = cch - p.C0; }
        const bool cvalid = cch < p.Ctot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // rows beyond M are never stored: their LDS rows may hold anything, so a 32-row piece that lies entirely beyond M is
            // not fetched at all (the M = 16 linears of the embedding MLPs spent half their DMA traffic on zero-page rows)
            if (TAPS == 1 && STAGES == 2 && m0 + 32 * i >= p.M) continue;        // workgroup-uniform (the counted waits of deeper pipelines assume 8 pieces)
            const bool ok = cvalid && ((vmask[i] >> tap) & 1u);
            const char* ptr = ok ? src + ((int64_t)pix[i][tap] * C + ch) * ES : zero;
            glds16(ptr, a_base + i * 32 * KB_BYTES);
        }
        const char* wb = wtile + ((int64_t)tap * p.Cin_pad + cb * KBE) * ES;          // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_s(woff[i], wb, b_base + i * 32 * KB_BYTES);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // M-tiles (16 rows) of this wave that hold real rows: 4 except in the last M-tile of a launch - and in the M = 16 linears of
    // the embedding MLPs, where 7 of 8 M-tiles are padding and the exact-f32 MFMAs (1/16 of the bf16 rate) were the bound
    // (1x1 instantiations only: in the 3x3 ones the second copy of the nine-tap body cost the 8x8 level 28 % - 53 -> 68 us)
    const int nlive = TAPS == 1 ? min(4, max(0, (p.M - m0 - wm * 64 + 15) >> 4)) : 4;        // wave-uniform
    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE_BYTES;
        const char* Bs = As + BM * KB_BYTES;
        if constexpr (X3 && std::is_same<T, float>::value) {
            // chunk fq holds channels 4 fq .. 4 fq + 3 of the 32-channel block, chunk 4 + fq channels 16 + 4 fq ..: together the eight
            // k-values of one lane; the packed weights hold exactly those eight as hi halves in chunk fq, lo halves in chunk 4 + fq
            uint4 wh[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wh[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, fq));
                wl[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, 4 + fq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (TAPS == 1 && i >= nlive) break;
                const uint4 x0 = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, fq));
                const uint4 x1 = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, 4 + fq));
                uint2 h0, l0, h1, l1;
                f16x3_split4(x0, h0, l0);
                f16x3_split4(x1, h1, l1);
                const uint4 ah = make_uint4(h0.x, h0.y, h1.x, h1.y), al = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mfma_f16(wh[j], ah, acc[i][j]);
                    mfma_f16(wl[j], ah, acc[i][j]);
                    mfma_f16(wh[j], al, acc[i][j]);
                }
            }
        } else
        if (TAPS != 1 || nlive == 4) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fa[4], fb[4];
                const int chunk = kk * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, chunk));
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, chunk));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<T>::run(fb[j], fa[i], acc[i][j]);   // D[channel][pixel]
            }
        } else if (TAPS == 1 && nlive > 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fb[4];
                const int chunk = kk * 4 + fq;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, chunk));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i >= nlive) break;
                    const uint4 fa = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, chunk));
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<T>::run(fb[j], fa, acc[i][j]);
                }
            }
        }
    };

    // ---- main loop: k-step (cb, tap), taps innermost and unrolled.  Per step: issue the DMA of the step STAGES - 1 ahead into
    //      the stage that was read in the previous step (every wave has passed that step's trailing barrier), compute this
    //      stage, wait until the NEXT step's DMA has landed (8 wave-instructions per step and wave, in issue order: all but the
    //      newest 8 (STAGES - 2) may stay in flight; the last steps of a tile simply drain), barrier.
    constexpr int AHEAD = STAGES - 1;
    const int nsteps = (cb1 - cb0) * TAPS;
    auto issue_ahead = [&](int stage, int cb, auto tap_c, auto d_c) {       // step (cb, tap) + d
        constexpr int t2 = decltype(tap_c)::value + decltype(d_c)::value;
        stage_step(stage, cb + t2 / TAPS, std::integral_constant<int, t2 % TAPS>{});
    };
    // prologue: steps 0 .. AHEAD-1
    [&]<int... d>(std::integer_sequence<int, d...>) {
        ((d < nsteps ? issue_ahead(d, cb0, std::integral_constant<int, 0>{}, std::integral_constant<int, d>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, AHEAD>{});
    // The bias slice of this lane is fetched here, behind the prologue's DMA, and pinned into registers before the k-loop: loaded in
    // the epilogue it cost every workgroup one exposed memory round trip (~2 k cycles of a 55 k-cycle small-map launch).  Plain
    // loads + an empty asm that consumes them: the compiler waits for them there itself (vmcnt(0) - once, in the prologue).
    float cbias[16];
    const bool bias_early = p.bias && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) && n0 + BN <= p.Cout;      // workgroup-uniform
    if (bias_early) {
        const float4* bp4 = reinterpret_cast<const float4*>(p.bias + n0 + (wave & 1) * 64 + (lane >> 4) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 b4 = bp4[q]; cbias[q * 4] = b4.x; cbias[q * 4 + 1] = b4.y; cbias[q * 4 + 2] = b4.z; cbias[q * 4 + 3] = b4.w; }
    }
    if (nsteps >= AHEAD) dma_wait_n<8 * (AHEAD - 1)>(); else dma_wait_all();
    __syncthreads();
    if (bias_early) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(cbias[k]));
    }
    int kt = 0, cur = 0;
    for (int cb = cb0; cb < cb1; ++cb) {
        auto body = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            int nst = cur + AHEAD; if (nst >= STAGES) nst -= STAGES;
            const bool more = kt + AHEAD < nsteps;               // workgroup-uniform
            if (more) issue_ahead(nst, cb, tap_c, std::integral_constant<int, AHEAD>{});
            compute(cur);
            if constexpr (STAGES == 2) dma_wait_all();
            else { if (more) dma_wait_n<8 * (AHEAD - 1)>(); else dma_wait_all(); }
            __syncthreads();
            ++kt;
            if (++cur == STAGES) cur = 0;
        };
        body(std::integral_constant<int, 0>{});
        if constexpr (TAPS == 9) {
            body(std::integral_constant<int, 1>{}); body(std::integral_constant<int, 2>{});
            body(std::integral_constant<int, 3>{}); body(std::integral_constant<int, 4>{});
            body(std::integral_constant<int, 5>{}); body(std::integral_constant<int, 6>{});
            body(std::integral_constant<int, 7>{}); body(std::integral_constant<int, 8>{});
        }
    }

    // ---- after the k-loop lane (fr, fq) of wave (wm, wn) holds, for the 4 pixels m0 + wm*64 + i*16 + fr, the 16
    //      consecutive output channels n0 + wn*64 + fq*16 + [0, 16): acc[i][j][reg] -> channel offset j*4 + reg
    const int n = n0 + wn * 64 + fq * 16;
    const bool all16 = n + 16 <= p.Cout;

    // ---- split-K: every workgroup of a tile writes its accumulators as raw f32 partial sums, laid out
    //      [split][tile][wave][accumulator register 0..63][lane] (a wave-instruction moves 256 contiguous bytes); the workgroup that
    //      ARRIVES LAST at the tile (one self-resetting counter per tile) reads all of the tile's partials back in split order - a
    //      fixed summation order whoever is last - and runs the normal epilogue below.  The partials cross workgroups on different
    //      XCDs (one L2 each) as sc1 (write-through) stores, drained, barrier, one lane's agent-scope atomic add; no release fence
    //      (it writes back the XCD's whole dirty L2) and no separate reduce launch.  conv_halo.hip spells out the ordering argument.
    //      Unlike that kernel this one runs TWO workgroups per CU, which is outside the geometry the sc1-loads-instead-of-acquire
    //      hand-off is measured for, so the last arriver ALSO issues the agent-scope acquire (invalidates this CU's L1; ~2 us on a
    //      25-50 us launch) before its loads.  tuning bit 10 adds the release on the producer side as well (stress test).
    if (p.ksplit > 1) {
        __shared__ int s_last;
        const int ks = p.ksplit;
        // workspace = [1024 arrival counters][partial sums]: the counters sit in FRONT so that no launch's partials ever cover them.
        // Partials move as 16-byte transactions, [split][tile][wave][accumulator tile 0..15][lane][4]: one f32x4 accumulator per
        // instruction, 1 KiB contiguous per wave-instruction - a quarter of the vector-memory instructions of the dword form (a
        // last arriver read 4 x 128 KiB in 2048 wave-instructions: 9 us of a 35 us launch, tools/fast_stamps.py).  Same cache policy
        // as before on both sides (sc1: write-through stores, L1-bypassing loads), written as inline asm because the atomic
        // builtins stop at 8 bytes; the loads are waited for explicitly (the compiler does not see them).
        auto pbase = [&](int sp) { return p.partial + 1024 + ((((int64_t)sp * nblk + bid) * 4 + wave) * 64) * 64 + lane * 4; };
        {
            float* pp = pbase(blockIdx.y);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // (s_nop 1: a store of more than 8 bytes needs wait states before a VALU write of its data registers (two on this target) - the
                    //  hazard recognizer inserts it for the compiler's own stores but cannot see into an asm statement; without it
                    //  the next tile's v_accvgpr_read overwrote dword 0 of the data for the last lanes still being read)
                    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" :: "v"(pp + i * 1024), "v"(acc[i][j]), "n"(j * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this lane's partial stores have been acknowledged ...
        __syncthreads();                                             // ... and every lane's, before the arrival is counted
        if (tid == 0) {
            if (p.tuning & 1024) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            int* cnt = reinterpret_cast<int*>(p.partial) + bid;
            const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == ks - 1;
            if (last) {
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-resetting
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;                                         // workgroup-uniform
        // The number of splits is a compile-time constant inside each case: the loaded registers must reach the wait below untouched
        // (no select, no copy - the compiler does not know they are still being written), so there is no "if (split exists)" here.
        // Chunk c = (accumulator row i, round r of up to four splits): 4 x 4 ... 16 loads.  The loads of chunk c + 1 are in flight while
        // chunk c is waited for (counted vmcnt) and added - the read-back used to be one exposed memory round trip per chunk (4 per
        // tile at <= 4 splits, 8 above: 10 k of a 65 k-cycle launch, tools/fast_stamps.py), now about half of them.
        auto reduce = [&](auto ks_c) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr int R = (KS + 3) / 4, NC = 4 * R;
            // (two chunks in flight = 128 + 64 accumulator registers: the one-workgroup-per-CU instantiation has them - 256 VGPRs, no
            //  spills; the two-per-CU one is capped at 256 and spilled 4, possibly a register still being loaded into: one chunk there)
            constexpr bool PIPE = STAGES >= 4;
            f32x4_t t[2][4][4];
#pragma unroll
            for (int c = -1; c < NC; ++c) {                          // (every index below is a compile-time constant after unrolling)
                int nxt = 0;                                         // loads of chunk c + 1, issued before chunk c is waited for
                if (PIPE ? c + 1 < NC : c >= 0) {
                    const int cn = PIPE ? c + 1 : c, i = cn / R, s0 = (cn % R) * 4, nu = KS - s0 < 4 ? KS - s0 : 4;
                    nxt = PIPE ? nu * 4 : 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u >= nu) break;
                        const float* pp = pbase(s0 + u) + i * 4 * 256;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(t[PIPE ? (cn & 1) : 0][u][j]) : "v"(pp), "n"(j * 1024) : "memory");
                    }
                }
                if (c < 0) continue;
                const int i = c / R, s0 = (c % R) * 4, nu = KS - s0 < 4 ? KS - s0 : 4;
                auto& q = t[PIPE ? (c & 1) : 0];
                // (operands listed per split count: a register that no load of this chunk targets must not appear as "+v")
                if (nu == 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]) : "n"(nxt) : "memory");
                else if (nu == 2) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]),
                                               "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]) : "n"(nxt) : "memory");
                else if (nu == 3) asm volatile("s_waitcnt vmcnt(%12)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]),
                                               "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]),
                                               "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[2][3]) : "n"(nxt) : "memory");
                else asm volatile("s_waitcnt vmcnt(%16)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]),
                                  "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]), "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[2][3]),
                                  "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]), "+v"(q[3][3]) : "n"(nxt) : "memory");
                if (s0 == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u >= nu) break;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] += q[u][j][r];      // split order: 0, 1, 2, ...
                }
            }
        };
        switch (ks) {
            case 2: reduce(std::integral_constant<int, 2>{}); break;
            case 3: reduce(std::integral_constant<int, 3>{}); break;
            case 4: reduce(std::integral_constant<int, 4>{}); break;
            case 5: reduce(std::integral_constant<int, 5>{}); break;
            case 6: reduce(std::integral_constant<int, 6>{}); break;
            case 7: reduce(std::integral_constant<int, 7>{}); break;
            default: reduce(std::integral_constant<int, 8>{}); break;      // (nlc_conv_fast_ksplit: at most 8)
        }
    }
    if (n >= p.Cout) return;
    // Statistics per WAVE (its 64 consecutive pixels lie inside one image whenever the map is a whole multiple of 64 pixels - 8x8 maps and
    // up, split launches included: one reduction + one atomic instruction per wave) or, on other maps, per 16-pixel row (emit_pix: four
    // of each - what every split launch used to pay, 8x8 and 16x16 levels included)
    const bool pix_stats = (HWo % 64) != 0;
    const int wave_m = m0 + wm * 64;                             // first pixel of this wave (M % 64 == 0 with such maps: all or nothing)

    // ---- epilogue straight from registers
    const bool vec_ok = (p.Cout % PER) == 0;
    const bool full = vec_ok && all16;
    constexpr int NCH = 16 / PER;
    // branch-free loads (clamped index, zero page for a missing bias): a per-element "if (in range) load" compiles to
    // load; s_waitcnt vmcnt(0) pairs, one full memory latency each
    if (!bias_early) {
        const float* zf = reinterpret_cast<const float*>(g_zero_page);
        const float* bp = p.bias ? p.bias : zf;
        const int hb = p.bias ? 1 : 0, last = p.Cout - 1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float vb = bp[hb ? min(n + k, last) : k]; cbias[k] = (n + k < p.Cout) ? vb : 0.f; }
    }
    Stat16 st16;                                                // ride-along GroupNorm statistics: this lane's 16 channels, 4 pixels
    st16.zero();
    // sum over the 16 pixel lanes of a DPP row, VALU only (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): an all-reduce
    auto row16_sum = [](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    // split launches: a 128-pixel tile may straddle images, a 16-pixel row (the lanes fr = 0..15, consecutive m) cannot when both the
    // map and the launch are whole multiples of 16 pixels (launch-uniform; every network shape) - the row is then reduced and added to
    // its image's totals with one atomic instruction; otherwise every lane adds its own pixel
    const bool rows_ok = (p.M & 15) == 0 && (HWo & 15) == 0;
    auto emit_pix = [&](Stat16& st, int m) {
        const int b = p.div_hwo.div(m);
        if (rows_ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { st.s[k] = row16_sum(st.s[k]); st.q[k] = row16_sum(st.q[k]); }
            st.emit_row(p.stats, b, p.Cout, n, p.stats_gran, fr);
        } else {
            st.emit_lane(p.stats, b, p.Cout, n, p.stats_gran);
        }
    };
    bool done = false, stats_emitted = false;
    if constexpr (sizeof(T) == 2) {
        // hot path (bf16, whole 16-channel slice, NHWC, no per-image embedding): option switches hoisted out of the
        // element loops (see conv_halo.hip: the general path costs ~19 VALU per output element)
        if (full && p.out_mode == NLC_OUT_NHWC && !p.emb) {
            const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
            const float sc = p.out_scale;
            const int act = p.act;
            // all residual chunks are requested before the first row is stored (the compiler keeps a row's loads behind the previous
            // row's stores - it cannot see that `out` and `res` do not alias: four exposed memory latencies per tile)
            uint4 rq0[4], rq1[4], pka[4], pkb[4];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = min(m0 + wm * 64 + i * 16 + fr, p.M - 1);
                    const T* rp = reinterpret_cast<const T*>(p.res) + res_row_m(p, m) * p.Cout + n;
                    rq0[i] = *reinterpret_cast<const uint4*>(rp);
                    rq1[i] = *reinterpret_cast<const uint4*>(rp + 8);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + fr;
                if (m >= p.M) continue;
                float v[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) v[j * 4 + reg] = acc[i][j][reg] + cbias[j * 4 + reg];
                if (has_res) {
                    const uint4 r0 = rq0[i], r1 = rq1[i];
                    float rr[16];
                    chunk_to_f32<T>(r0, rr); chunk_to_f32<T>(r1, rr + 8);
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] += rr[k];
                }
                if (sc != 1.0f) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] *= sc;
                }
                if (act == NLC_ACT_SILU) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = silu_f(v[k]);
                } else if (act == NLC_ACT_GELU) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = gelu_erf(v[k]);
                }
                const uint4 pk0 = f32_to_chunk<T>(v), pk1 = f32_to_chunk<T>(v + 8);
                pka[i] = pk0; pkb[i] = pk1;
                if (has_stats) {         // of the STORED (rounded) values - what the GroupNorm that follows reads
                    if (pix_stats) st16.zero();
                    st16.add_chunk<T>(0, pk0); st16.add_chunk<T>(1, pk1);
                    if (pix_stats) emit_pix(st16, m);       // split launches: a tile may straddle images -> per 16-pixel row
                }
            }
            // The statistics atomics go out BEFORE the row stores: they are performed at the memory side (device scope across the
            // XCDs' L2s) and take a few microseconds to retire - issued last, behind the stores, that latency was the tail of every
            // small launch (+2 ... 3 us per conv_fast launch when the statistics became atomics; the kernel cannot end before they do).
            if (has_stats && !pix_stats) {   // dispatch guarantees Cout % 128 == 0 -> no lane was skipped
#pragma unroll
                for (int k = 0; k < 4; ++k) { st16.s[k] = row16_sum(st16.s[k]); st16.q[k] = row16_sum(st16.q[k]); }
                if (wave_m < p.M) st16.emit_row(p.stats, p.div_hwo.div(wave_m), p.Cout, n, p.stats_gran, fr);
                stats_emitted = true;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + fr;
                if (m >= p.M) continue;
                T* op = reinterpret_cast<T*>(p.out) + (int64_t)m * p.Cout + n;
                *reinterpret_cast<uint4*>(op) = pka[i];
                *reinterpret_cast<uint4*>(op + 8) = pkb[i];
            }
            done = true;
        }
    }
    float wsc[16];                                   // X3: power-of-two factor of each output channel's sum (exact); n + 15 < Cout_pad
    if constexpr (X3 && std::is_same<T, float>::value) {
#pragma unroll
        for (int k = 0; k < 16; ++k) wsc[k] = p.w_scale[n + k];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (done) break;
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
        const int b = p.div_hwo.div(m);
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if constexpr (X3 && std::is_same<T, float>::value) v[j * 4 + reg] = acc[i][j][reg] * wsc[j * 4 + reg] + cbias[j * 4 + reg];
                else v[j * 4 + reg] = acc[i][j][reg] + cbias[j * 4 + reg];
            }
        if (p.emb) {
            const float* ep = p.emb + (int64_t)b * p.emb_stride;
            float ev[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) ev[k] = ep[min(n + k, p.Cout - 1)];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += (n + k < p.Cout) ? ev[k] : 0.f;
        }
        if (p.res) {
            const T* rp = reinterpret_cast<const T*>(p.res) + res_row_m(p, m) * p.Cout + n;
            if (full) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float rr[PER];
                    chunk_to_f32<T>(*reinterpret_cast<const uint4*>(rp + c * PER), rr);
#pragma unroll
                    for (int k = 0; k < PER; ++k) v[c * PER + k] += rr[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) if (n + k < p.Cout) v[k] += ElemTraits<T>::load(rp + k);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = apply_act(v[k] * p.out_scale, p.act);
        if (p.out_mode == NLC_OUT_NHWC) {
            T* op = reinterpret_cast<T*>(p.out) + (int64_t)m * p.Cout + n;
            if (full) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint4 pk = f32_to_chunk<T>(v + c * PER);
                    *reinterpret_cast<uint4*>(op + c * PER) = pk;
                    if constexpr (sizeof(T) == 2) {
                        if (p.stats) {
                            if (pix_stats && c == 0) st16.zero();
                            st16.add_chunk<T>(c, pk);
                        }
                    }
                }
                if constexpr (sizeof(T) == 2) {
                    if (p.stats && pix_stats) emit_pix(st16, m);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) if (n + k < p.Cout) ElemTraits<T>::store(op + k, v[k]);
            }
        } else {                                             // NCHW f32 (last layer): consecutive lanes = consecutive pixels
            const int rem = m - b * HWo;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (n + k < p.Cout) reinterpret_cast<float*>(p.out)[((int64_t)b * p.Cout + n + k) * HWo + rem] = v[k];
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (p.stats && !pix_stats && !stats_emitted) {       // (general epilogue) Cout % 128 == 0 -> no lane was skipped
#pragma unroll
            for (int k = 0; k < 4; ++k) { st16.s[k] = row16_sum(st16.s[k]); st16.q[k] = row16_sum(st16.q[k]); }
            if (wave_m < p.M) st16.emit_row(p.stats, p.div_hwo.div(wave_m), p.Cout, n, p.stats_gran, fr);
        }
    }
}

template <typename T, int TAPS, int STAGES, bool X3 = false>
int launch_fast(const KParams& p, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fast_kernel<T, TAPS, STAGES, X3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  STAGES * STAGE_BYTES);
    });
    hipLaunchKernelGGL((conv_fast_kernel<T, TAPS, STAGES, X3>), dim3(p.MT * p.NT, p.ksplit), dim3(NTHREADS), STAGES * STAGE_BYTES, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(fast): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

}  // namespace

// Shapes of the fast path: 3x3 (any stride, any top/left padding: the far side is zero-filled by the tap tables, so
// the stride-2 Downsample convs of ADM / the sigma nets and the asymmetric-pad ones of the simple UNet qualify) and
// unpadded stride-1 1x1.
static bool fast_shape(const KParams& p) {
    const bool k3 = p.KH == 3 && p.KW == 3;
    const bool k1 = p.KH == 1 && p.KW == 1 && p.pad_t == 0 && p.pad_l == 0 && p.stride == 1;
    if (!(k3 || k1)) return false;
    if (p.ups && p.stride != 1) return false;
    return (int64_t)p.B * p.Hin * p.Win < (1ll << 31);
}

// Split-K policy (the caller opts in by passing a workspace; the f32 parity path never does): shapes with fewer output tiles than
// 2 per CU and a long K; at least 2 channel blocks (18 / 2 k-steps) per split, at most 8 splits, aiming at >= 2
// workgroups per CU (the 8x8 / 16x16 levels of ADM-256 at B = 16 have 64 / 256 tiles for 144-288 k-steps).
int nlc_conv_fast_ksplit(const KParams& p, int dtype) {
    if ((p.Cout & 3) || !fast_shape(p) || (p.tuning & 2048)) return 1;       // bit 11: no split-K anywhere (A/B, stress test)
    const bool k3 = p.KH == 3;
    const int tiles = p.MT * p.NT;
    const int ncb = p.Cin_pad / (nlc_is16(dtype) ? Mma<bf16_raw>::KBE : Mma<float>::KBE);
    const int target = (p.tuning & 16384) ? 256 : 512;           // workgroups aimed at (A/B: bit 14 = one per CU)
    if (tiles >= target) return 1;
    int s = cdiv(target, tiles);
    // >= 18 (3x3) / 8 (1x1) k-steps per split; a 3x3 launch that would otherwise put fewer than 128 workgroups on the chip goes down to
    // one channel block (9 k-steps) per split (512 -> 512 @8x8, B = 8: 16 tiles x 4 -> x 8, 25.2 -> 22.8 us with the 16-byte hand-off;
    // as a general rule it is slower: profiles/r03_summary.md section 5)
    const int min_cb = k3 ? (tiles * (ncb / 2) < 128 ? 1 : 2) : 8;                   // (one channel block per split, up to 16
    if (s > ncb / min_cb) s = ncb / min_cb;          //  splits: measured slower, profiles/r03_summary.md section 5)
    if (s > 8) s = 8;                                // (the read-back in the kernel is instantiated for 2 ... 8)
    return s < 2 ? 1 : s;
}

// GroupNorm statistics ride along on the fast path when the N-tiles are whole and the output is 16-bit NHWC: per tile from the conv
// epilogue if every 128-pixel tile lies inside one image, else (and from the last-arriving workgroup when K is split) per 16-pixel row
int nlc_conv_fast_stats_partials(const KParams& p, int dtype) {
    return (nlc_is16(dtype) && p.out_mode == NLC_OUT_NHWC && (p.Cout % BN) == 0 && fast_shape(p)) ? 1 : 0;
}

// workspace of a split launch: the arrival counters (4 KiB in front; tiles < 512) + ks x (whole 128 x 128 tiles) f32 partial sums
int64_t nlc_conv_fast_split_bytes(const KParams& p, int ks) {
    return ks > 1 ? (int64_t)ks * p.MT * p.NT * (BM * BN) * (int64_t)sizeof(float) + 4096 : 0;
}

// returns NLC_EUNSUPPORTED when the shape is not one the fast path handles (caller falls back)
int nlc_conv_fast_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!fast_shape(p)) return NLC_EUNSUPPORTED;
    const bool k3 = p.KH == 3;
    // deep pipeline for launches that leave every workgroup alone on its CU (tuning bit 12: never, bit 13: always - A/B)
    static DeviceOnce once;
    const int ncu = once.ncu[nlc_device_once(once, [] {})];
    const int64_t grid = (int64_t)p.MT * p.NT * p.ksplit;
    const bool deep = (p.tuning & 8192) ? true : (p.tuning & 4096) ? false : grid <= ncu;
    if (deep && nlc_is16(dtype)) {
        if (dtype == NLC_BF16) return k3 ? launch_fast<bf16_raw, 9, 4>(p, stream) : launch_fast<bf16_raw, 1, 4>(p, stream);
        return k3 ? launch_fast<f16_raw, 9, 4>(p, stream) : launch_fast<f16_raw, 1, 4>(p, stream);
    }
    if (dtype == NLC_BF16) return k3 ? launch_fast<bf16_raw, 9, 2>(p, stream) : launch_fast<bf16_raw, 1, 2>(p, stream);
    if (dtype == NLC_F16) return k3 ? launch_fast<f16_raw, 9, 2>(p, stream) : launch_fast<f16_raw, 1, 2>(p, stream);
    if (p.math == NLC_MATH_F16X3) return k3 ? launch_fast<float, 9, 2, true>(p, stream) : launch_fast<float, 1, 2, true>(p, stream);
    return k3 ? launch_fast<float, 9, 2>(p, stream) : launch_fast<float, 1, 2>(p, stream);
}
