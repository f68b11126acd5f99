// 3x3 / stride 1 / pad 1 convolutions on SMALL maps (8x8 ... 32x32), 16-bit: the launch-bound regime of every network here
// (ADM-256's 8x8 level, the three lowest levels of the CelebA-HQ UNet at batch 8) - with the GroupNorm (+FiLM) (+SiLU) that precedes
// the convolution in every ResBlock (src/unet_adm.py:182-185,206-211,248-252; src/unet_simple.py:117-124;
// src/edm_networks.py:185-192) applied to the input ON ITS WAY INTO LDS, from the ride-along totals of the producing launches.
//
// What the per-op launches cost there (tools/fast_stamps.py, profiles/r04j_fast_stamps_after.log; 512 -> 512 @8x8, B = 8: 51 k cycles):
// the normalisation is a launch of its own (10 us for 1-2 MB); conv_fast re-gathers its 128-pixel activation tile once PER TAP
// (32 KB of L2 -> LDS traffic per k-step); its k-loop is bound by the LATENCY of that stream (bytes in flight / ~2 k cycles, measured
// here with tools/small_stamps.py: 16 KB weight stages, two in flight = 16 B/clk = 1000 cycles per k-step for 512 cycles of matrix
// work); and the last-arriving workgroup of a tile reads ALL the tile's partial sums alone (8 x 64 KB through one CU: 15 k cycles).
// Here:
//   * a workgroup owns (BM = 128 or 256 consecutive pixels x 128 output channels x a k-slice of NB <= 4 64-channel blocks).  Its whole
//     input slice - the pixels WITH their one-pixel halo, image by image, zero padding included - is loaded ONCE through registers
//     (all loads of up to two blocks in flight together), normalised there (a, b per (image, channel) derived in the kernel from
//     the totals: conv_params.h, gn_group_from_totals_t - the arithmetic of nlc_groupnorm_prestats) and written to LDS as
//     [channel block][halo slot][128 bytes]; the nine taps read shifted windows of it.  L2 -> LDS traffic per k-step: the 16 KB
//     weight stage only;
//   * weights stream through a ring of 3-8 LDS stages by LDS-DMA (conv_fast.hip's SGPR-base form and row permutation), issued
//     BEFORE the input slice is touched, so they land under the normalisation prologue.  The ring takes what LDS the input slice
//     leaves: bytes in flight are what the k-loop's rate is made of (above).  256-pixel tiles (8 waves) halve the weight bytes per
//     MFMA - the choice wherever they still fill the chip;
//   * slot swizzle: chunk c of halo slot (yy, xx) sits at 16-byte position c ^ (xx & 7) of its 128-byte row and the halo row
//     pitch is even, so the 16 lanes of a ds_read_b128 group - two rows of 8 pixels on an 8-wide map, 16 consecutive pixels on
//     wider ones - cover all 64 banks once for every tap shift (conv_halo.hip's argument with the key taken from xx instead of
//     from the row index: xx is what the 16 lanes enumerate);
//   * K is split over workgroups (f32 partials as sc1 16-byte stores, conv_fast.hip's layout).  With 2 / 4 / 8 splits and a grid
//     that is resident at once (<= one workgroup per CU: LDS >= 96 KB forces that) the reduction is DISTRIBUTED: every workgroup of
//     a tile counts its arrival, waits until all ks have arrived (one lane polls, bounded), and then reduces and finishes 1 / ks of
//     the tile - 16 loads of 16 bytes per lane whatever ks is - in split order (deterministic).  A second counter per tile counts
//     departures; the last one to leave resets both (the workspace contract: counters are zero between launches).  Other split
//     counts / larger grids: the last arriver reduces alone, as in conv_fast.hip.  Hand-off: row 1 of the guide's table (sc1 stores,
//     every wave's vmcnt(0), barrier, one lane's agent-scope add; sc1 poll; barrier; sc1 loads; one workgroup per CU) - no acquire
//     (tuning bit 10: fenced variant for the stress test);
//   * workgroups that read the same weight slice (same channel tile and k-slice, different pixel tiles) are placed on one XCD.
// The kernel body is a device function: conv_small_kernel runs it once; resblock_small_kernel (below) runs a ResBlock's two
// convolutions in ONE launch behind a grid barrier - the A/B the round-4 verdict asked for (tools/resblock_bench.py).
#include "common.h"
#include "conv_params.h"
#include "conv_small.h"
#include <utility>

namespace {

__device__ uint4 g_zero_small[8];

constexpr int WST_BYTES = BN * KB_BYTES;                 // 16 KiB: one tap x one 64-channel block x 128 output channels
constexpr int SMALL_SPIN_MAX = 1 << 22;                  // polls of an arrival counter before a workgroup gives up waiting (~ seconds)

__device__ __forceinline__ int w_off(int row, int chunk) { return row * KB_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

// SGPR-base LDS-DMA (conv_fast.hip): address = sbase (wave-uniform) + voff (per lane), 16 bytes per lane to lds_base + 16 * lane
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

// LDS: [A planes: nb x slots x 128 B][weight ring: NWST x 16 KiB][coefficient table: nseg x nb*64 x (a, b) f32]
__host__ __device__ inline int small_a_bytes(const SmallGeom& g) { return g.nb * g.slots * KB_BYTES; }

// One convolution's share of this workgroup.  WM = wave rows (2: 128-pixel tile, 256 threads; 4: 256 pixels, 512 threads).
// Returns true in a workgroup that stored (part of) its tile's output; nobody returns early, so a caller can put a grid barrier behind.
template <typename T, int WM, int NWST>
__device__ __forceinline__ bool conv_small_body(const SmallParams& sp, char* smem, int L) {
    static_assert(sizeof(T) == 2, "16-bit storage only");
    constexpr int AHEAD = NWST - 1;
    constexpr int NTHR = WM * 2 * 64, NWAVE = WM * 2, BMT = WM * 64;
    constexpr int SG = NTHR / 8;                             // slot groups: thread t handles slots (t >> 3) + SG * it
    constexpr int NIT = 7;                                   // slots <= 7 * SG (host-checked)
    constexpr int WDMA = 128 / SG;                           // weight DMA wave-instructions per wave and stage (4 | 2)
    const KParams& p = sp.k;
    const SmallGeom& g = sp.geo;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    /*@stamp:begin*/
    // ---- work item: (pixel tile mt, channel tile nt, k-slice ksl).  Groups (nt, ksl) - the readers of one weight slice - are dealt
    //      to XCDs (workgroup L runs on XCD L % 8 as observed; speed only): L -> xcd = L & 7, idx = L >> 3 -> group idx / MT.
    const int G = p.NT * g.ks;
    int mt, grp;
    if ((G & 7) == 0) { const int xcd = L & 7, idx = L >> 3; grp = (idx / g.MT) * 8 + xcd; mt = idx % g.MT; }
    else { grp = L / g.MT; mt = L - grp * g.MT; }
    const int nt = grp % p.NT, ksl = grp / p.NT;
    const int tile = mt * p.NT + nt, nblk = g.MT * p.NT;
    const int m0 = mt * BMT, n0 = nt * BN;
    const int cb0 = ksl * g.nb;                              // first 64-channel block of this k-slice
    const int nsteps = g.nb * 9;

    char* Abase = smem;
    const unsigned w_lds = lds0 + small_a_bytes(g);
    char* Wbase = smem + small_a_bytes(g);
    float2* coefT = reinterpret_cast<float2*>(smem + small_a_bytes(g) + NWST * WST_BYTES);

    // ---- weight stream (conv_fast.hip): LDS row R = lr + SG i of a stage receives output channel (R & 64) + ((R & 15) >> 2) * 16
    //      + ((R >> 4) & 3) * 4 + (R & 3) - with the MFMA operands swapped a lane ends up with 16 consecutive channels of a pixel
    const int lr = tid >> 3;
    const int gcw = (tid & 7) ^ ((lr >> 1) & 7);
    const int64_t wrow = (int64_t)9 * p.Cin_pad * 2;
    unsigned woff[WDMA];
#pragma unroll
    for (int i = 0; i < WDMA; ++i) {
        const int R = lr + SG * i;
        const int ch = (R & 64) + ((R & 15) >> 2) * 16 + ((R >> 4) & 3) * 4 + (R & 3);
        woff[i] = (unsigned)((int64_t)ch * wrow + (int64_t)gcw * 16);
    }
    const char* wtile = p.w + (int64_t)n0 * wrow + (int64_t)cb0 * 128;
    auto issue_w = [&](int stage, int step) {                // step = cbl * 9 + tap
        const int cbl = step / 9, tap = step - cbl * 9;
        const char* wb = wtile + ((int64_t)tap * p.Cin_pad) * 2 + cbl * 128;         // wave-uniform
        const unsigned b_base = w_lds + stage * WST_BYTES + wave * 8 * KB_BYTES;
#pragma unroll
        for (int i = 0; i < WDMA; ++i) glds16_s(woff[i], wb, b_base + i * SG * KB_BYTES);
    };
#pragma unroll
    for (int d = 0; d < AHEAD; ++d)
        if (d < nsteps) issue_w(d, d);

    // ---- halo slots of this thread: slot s = sg + SG it holds halo position (seg, yy, xx); chunk c8 of every channel block
    const int c8 = tid & 7, sg = tid >> 3;
    const int H = p.Hin, W = p.Win, Wp = W + 2, segslots = (g.SR + 2) * Wp;
    const int R0 = m0 / W;                                   // first row of the tile in the stacked (B * H)-row image
    int spix[NIT];                                           // source pixel index, -1: zero padding, -2: no such slot
    int sdst[NIT];                                           // LDS byte offset inside a plane (swizzled)
    int sseg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = sg + SG * it;
        const int seg = g.div_segslots.div(s);
        const int r = s - seg * segslots;
        const int yy = g.div_wp.div(r), xx = r - yy * Wp;
        const int Rs = R0 + seg * g.SR;
        const int b = g.div_h.div(Rs), y0 = Rs - b * H;
        const int y = y0 + yy - 1, x = xx - 1;
        const bool inimg = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        spix[it] = s < g.slots ? (inimg ? (b * H + y) * W + x : -1) : -2;
        sdst[it] = s * KB_BYTES + ((c8 ^ (xx & 7)) << 4);
        sseg[it] = seg;
    }
    const bool norm = sp.gn.tot0 != nullptr;
    const bool silu = sp.gn.act == NLC_ACT_SILU;
    const int nch = g.nb * 64;
    auto load_block = [&](int cbl, uint4 (&v)[NIT]) {
        const int cch = (cb0 + cbl) * 64 + c8 * 8;
        const T* src; int C, ch;
        if (cch < p.C0) { src = reinterpret_cast<const T*>(p.x0); C = p.C0; ch = cch; }
        else { src = reinterpret_cast<const T*>(p.x1); C = p.C1; ch = cch - p.C0; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const T* ptr = spix[it] >= 0 ? src + ((int64_t)spix[it] * C + ch) : reinterpret_cast<const T*>(g_zero_small);
            v[it] = *reinterpret_cast<const uint4*>(ptr);
        }
    };
    auto store_block = [&](int cbl, const uint4 (&v)[NIT]) {
        char* plane = Abase + cbl * g.slots * KB_BYTES;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (spix[it] == -2) continue;
            uint4 o = v[it];
            if (norm && spix[it] >= 0) {
                const float4* cf = reinterpret_cast<const float4*>(coefT + sseg[it] * nch + cbl * 64 + c8 * 8);
                float f[8];
                chunk_to_f32<T>(v[it], f);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float4 c = cf[h];                   // a0 b0 a1 b1
                    float y0 = c.x * f[2 * h] + c.y, y1 = c.z * f[2 * h + 1] + c.w;
                    if (silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
                    f[2 * h] = y0; f[2 * h + 1] = y1;
                }
                o = f32_to_chunk<T>(f);
            }
            *reinterpret_cast<uint4*>(plane + sdst[it]) = o;
        }
    };
    // the first two blocks' loads go out BEFORE the coefficient chain (totals -> group statistics -> a, b: a dependent memory round
    // trip plus f64 arithmetic) and the bias / embedding loads, so that all these latencies overlap
    uint4 va[NIT], vb[NIT];
    load_block(0, va);
    if (g.nb > 1) load_block(1, vb);

    // bias (+ per-image embedding row) of this lane's 16 output channels
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int n = n0 + wn * 64 + fq * 16;
    const int wave_m = m0 + wm * 64;                          // the wave's 64 pixels lie in one image (H * W % 64 == 0)
    const int wave_b = p.div_hwo.div(wave_m);
    float cbias[16];
    {
        const float* zf = reinterpret_cast<const float*>(g_zero_small);
        const float4* bp4 = reinterpret_cast<const float4*>(p.bias ? p.bias + n : zf);
        const float4* ep4 = reinterpret_cast<const float4*>(p.emb ? p.emb + (int64_t)wave_b * p.emb_stride + n : zf);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float4 b4 = bp4[p.bias ? h : 0], e4 = ep4[p.emb ? h : 0];
            cbias[4 * h] = b4.x + e4.x; cbias[4 * h + 1] = b4.y + e4.y; cbias[4 * h + 2] = b4.z + e4.z; cbias[4 * h + 3] = b4.w + e4.w;
        }
    }
    /*@stamp:0 weight DMA issued, slot tables, first loads issued*/
    // ---- GroupNorm coefficients of the slice: (a, b) per (segment image, channel), computed here from the totals
    if (norm) {
        const GnIn& q = sp.gn;
        for (int e = tid; e < g.nseg * nch; e += NTHR) {
            const int seg = e / nch, cl = e - seg * nch;
            const int ch = cb0 * 64 + cl;
            const int b = g.div_h.div(R0 + seg * g.SR);
            float mean, rstd;
            gn_group_from_totals_t(q, b, q.div_gs.div(ch), mean, rstd);
            float aa = rstd * (q.gamma ? q.gamma[ch] : 1.f);
            float bb = (q.beta ? q.beta[ch] : 0.f) - mean * aa;
            if (q.scale) {
                const float sc = 1.f + q.scale[(int64_t)b * q.ss_stride + ch];
                const float sh = q.shift[(int64_t)b * q.ss_stride + ch];
                aa *= sc; bb = bb * sc + sh;
            }
            coefT[e] = float2{aa, bb};
        }
    }
    __syncthreads();
    /*@stamp:1 coefficient table*/

    // ---- the input slice: registers -> (normalise) -> LDS
    store_block(0, va);
    if (g.nb > 1) {
        if (g.nb > 2) load_block(2, va);
        store_block(1, vb);
        if (g.nb > 2) {
            if (g.nb > 3) load_block(3, vb);
            store_block(2, va);
            if (g.nb > 3) store_block(3, vb);
        }
    }
    /*@stamp:2 input slice in LDS*/
    dma_wait_all();                                           // (the compiler has waited for its own loads; this retires the weight DMA of steps 0 .. AHEAD-1)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(cbias[k]));

    /*@stamp:3 weights landed, barrier*/
    // ---- fragment addresses: pixel m0 + wm*64 + i*16 + fr -> halo slot of tap (0, 0); tap (dy, dx) adds dy * Wp + dx slots
    int aoff[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = wm * 64 + i * 16 + fr;
        const int ty = g.div_w.div(ml), x = ml - ty * W;
        const int seg = g.div_sr.div(ty), yy = ty - seg * g.SR;
        const int slot = seg * segslots + yy * Wp + x;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) aoff[i][dx] = (slot + dx) * KB_BYTES + ((fq ^ ((x + dx) & 7)) << 4);
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage, int cbl, auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr int dy = tap / 3, dx = tap % 3;
        const char* As = Abase + cbl * g.slots * KB_BYTES + dy * Wp * KB_BYTES;
        const char* Bs = Wbase + stage * WST_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + (aoff[i][dx] ^ (kk * 64)));
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + w_off(wn * 64 + j * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mfma16<T>::run(fb[j], fa[i], acc[i][j]);      // D[channel][pixel]
        }
    };

    // ---- k-loop: (channel block, tap), taps unrolled; the weight stage of step kt + AHEAD is issued into the stage read in step
    //      kt - 1 (every wave has passed that step's barrier); WDMA DMA wave-instructions per step and wave, counted waits
    int kt = 0, cur = 0;
    for (int cbl = 0; cbl < g.nb; ++cbl) {
        auto body = [&](auto tap_c) {
            int nst = cur + AHEAD; if (nst >= NWST) nst -= NWST;
            const bool more = kt + AHEAD < nsteps;            // workgroup-uniform
            if (more) issue_w(nst, kt + AHEAD);
            compute(cur, cbl, tap_c);
            if (more) dma_wait_n<WDMA * (AHEAD - 1)>(); else dma_wait_all();
            __syncthreads();
            ++kt;
            if (++cur == NWST) cur = 0;
        };
        body(std::integral_constant<int, 0>{}); body(std::integral_constant<int, 1>{}); body(std::integral_constant<int, 2>{});
        body(std::integral_constant<int, 3>{}); body(std::integral_constant<int, 4>{}); body(std::integral_constant<int, 5>{});
        body(std::integral_constant<int, 6>{}); body(std::integral_constant<int, 7>{}); body(std::integral_constant<int, 8>{});
    }
    /*@stamp:4 k-loop*/

    // ---- epilogue of one UNIT = (16-pixel row i of the wave, channel half h of the lane's 16): 8 channels of one pixel per lane, from
    //      the two accumulator tiles (i, 2h), (i, 2h + 1): bias (+ embedding row) + residual, * out_scale, activation, 16-byte store;
    //      ride-along statistics of the STORED values per half
    const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
    Stat16 st_h[2];
    st_h[0].zero(); st_h[1].zero();
    auto finish_unit = [&](int i, int h, const f32x4_t& a0, const f32x4_t& a1, const uint4& rq) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = a0[r] + (h ? cbias[8 + r] : cbias[r]);
            v[4 + r] = a1[r] + (h ? cbias[12 + r] : cbias[4 + r]);
        }
        if (has_res) {
            float rr[8];
            chunk_to_f32<T>(rq, rr);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += rr[k];
        }
        if (p.out_scale != 1.0f) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= p.out_scale;
        }
        if (p.act == NLC_ACT_SILU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = silu_f(v[k]);
        } else if (p.act == NLC_ACT_GELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = gelu_erf(v[k]);
        }
        const uint4 pk = f32_to_chunk<T>(v);
        if (has_stats) {
            Stat16 t; t.zero(); t.add_chunk<T>(0, pk);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                st_h[0].s[k] += h ? 0.f : t.s[k]; st_h[0].q[k] += h ? 0.f : t.q[k];
                st_h[1].s[k] += h ? t.s[k] : 0.f; st_h[1].q[k] += h ? t.q[k] : 0.f;
            }
        }
        T* op = reinterpret_cast<T*>(p.out) + (int64_t)(wave_m + i * 16 + fr) * p.Cout + n + h * 8;
        if (sp.out_sc1) {                                     // (resblock_small_kernel: h crosses a grid barrier to other CUs - write-through)
            const f32x4_t qa = __builtin_bit_cast(f32x4_t, pk);
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(op), "v"(qa) : "memory");
        } else {
            *reinterpret_cast<uint4*>(op) = pk;
        }
    };
    auto res_chunk = [&](int i, int h) {
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + (int64_t)(wave_m + i * 16 + fr) * p.Cout + n + h * 8);
    };
    auto row16_sum = [](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    auto emit_stats = [&](bool h0, bool h1) {                 // wave-uniform flags: which halves this workgroup finished
        if (!has_stats) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!(h ? h1 : h0)) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) { st_h[h].s[k] = row16_sum(st_h[h].s[k]); st_h[h].q[k] = row16_sum(st_h[h].q[k]); }
            st_h[h].emit_row8(p.stats, wave_b, p.Cout, n + h * 8, p.stats_gran, fr);
        }
    };
    auto finish_all = [&]() {                                 // the whole tile from acc[][]
        uint4 rq[4][2];
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { rq[i][0] = res_chunk(i, 0); rq[i][1] = res_chunk(i, 1); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            finish_unit(i, 0, acc[i][0], acc[i][1], rq[i][0]);
            finish_unit(i, 1, acc[i][2], acc[i][3], rq[i][1]);
        }
        emit_stats(true, true);
    };

    if (g.ks == 1) {
        /*@stamp:7 partials read back*/
        finish_all();
        /*@stamp:8 rows stored*/
        /*@stamp:end*/
        return true;
    }

    // ---- split-K hand-off: partials [split][tile][wave][accumulator tile 0..15][lane][4] as sc1 16-byte stores (conv_fast.hip)
    int* s_flag = reinterpret_cast<int*>(coefT);              // (the coefficient table is dead: every wave is past the k-loop's last barrier)
    auto pbase = [&](int s) { return p.partial + 1024 + ((((int64_t)s * nblk + tile) * NWAVE + wave) * 64) * 64 + lane * 4; };
    {
        float* pp = pbase(ksl);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" :: "v"(pp + i * 1024), "v"(acc[i][j]), "n"(j * 1024) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    /*@stamp:5 partials stored + drained*/
    int* cnt_arrive = reinterpret_cast<int*>(p.partial) + tile;
    int* cnt_depart = cnt_arrive + 512;
    bool owner = true;
    if (g.dist) {
        // distributed reduction: wait for all ks arrivals, then finish units [ksl * 8 / ks, (ksl + 1) * 8 / ks) of every wave
        if (tid == 0) {
            if (p.tuning & 1024) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(cnt_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.ks && ++spins < SMALL_SPIN_MAX) __builtin_amdgcn_s_sleep(2);
            // gave up (a peer never arrived: the grid was not resident at once): leave a mark that nlc_conv_desc.debug bit 0 reports
            if (spins >= SMALL_SPIN_MAX) __hip_atomic_store(reinterpret_cast<int*>(p.partial) + 1023, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.tuning & 1024) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        /*@stamp:6 arrival counted*/
        auto reduce_units = [&](auto ks_c) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr int U = 8 / KS;
            f32x4_t t[U][KS][2];
            uint4 rq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int unit = ksl * U + u, i = unit >> 1, h = unit & 1;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float* pp = pbase(s) + (i * 4 + 2 * h) * 256;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[u][s][0]) : "v"(pp) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=v"(t[u][s][1]) : "v"(pp) : "memory");
                }
                if (has_res) rq[u] = res_chunk(i, h);
            }
            // (every loaded register is named: none of them may be touched before the wait)
            if constexpr (KS == 2) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0][0][0]), "+v"(t[0][0][1]), "+v"(t[0][1][0]), "+v"(t[0][1][1]), "+v"(t[1][0][0]), "+v"(t[1][0][1]),
                             "+v"(t[1][1][0]), "+v"(t[1][1][1]), "+v"(t[2][0][0]), "+v"(t[2][0][1]), "+v"(t[2][1][0]), "+v"(t[2][1][1]),
                             "+v"(t[3][0][0]), "+v"(t[3][0][1]), "+v"(t[3][1][0]), "+v"(t[3][1][1]) :: "memory");
            } else if constexpr (KS == 4) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0][0][0]), "+v"(t[0][0][1]), "+v"(t[0][1][0]), "+v"(t[0][1][1]), "+v"(t[0][2][0]), "+v"(t[0][2][1]),
                             "+v"(t[0][3][0]), "+v"(t[0][3][1]), "+v"(t[1][0][0]), "+v"(t[1][0][1]), "+v"(t[1][1][0]), "+v"(t[1][1][1]),
                             "+v"(t[1][2][0]), "+v"(t[1][2][1]), "+v"(t[1][3][0]), "+v"(t[1][3][1]) :: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0][0][0]), "+v"(t[0][0][1]), "+v"(t[0][1][0]), "+v"(t[0][1][1]), "+v"(t[0][2][0]), "+v"(t[0][2][1]),
                             "+v"(t[0][3][0]), "+v"(t[0][3][1]), "+v"(t[0][4][0]), "+v"(t[0][4][1]), "+v"(t[0][5][0]), "+v"(t[0][5][1]),
                             "+v"(t[0][6][0]), "+v"(t[0][6][1]), "+v"(t[0][7][0]), "+v"(t[0][7][1]) :: "memory");
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int unit = ksl * U + u, i = unit >> 1, h = unit & 1;
                f32x4_t a0 = t[u][0][0], a1 = t[u][0][1];
#pragma unroll
                for (int s = 1; s < KS; ++s) { a0 += t[u][s][0]; a1 += t[u][s][1]; }      // split order: 0, 1, 2, ...
                finish_unit(i, h, a0, a1, rq[u]);
            }
            const int u0 = ksl * U;
            emit_stats(U > 1 || (u0 & 1) == 0, U > 1 || (u0 & 1) == 1);
        };
        if (g.ks == 2) reduce_units(std::integral_constant<int, 2>{});
        else if (g.ks == 4) reduce_units(std::integral_constant<int, 4>{});
        else reduce_units(std::integral_constant<int, 8>{});
        /*@stamp:7 partials read back*/
        // departure: the last workgroup to leave the tile re-zeroes both counters (nobody polls `arrive` any more: all ks have passed)
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(cnt_depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == g.ks - 1) {
                __hip_atomic_store(cnt_arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt_depart, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if constexpr (WM == 2) {
        // the last arriver reduces the whole tile (conv_fast.hip); 128-pixel tiles only (the host never picks the other form without
        // the distributed reduction: its 256-register budget has no room for this read-back)
        if (tid == 0) {
            if (p.tuning & 1024) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            const int old = __hip_atomic_fetch_add(cnt_arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == g.ks - 1;
            if (last) {
                __hip_atomic_store(cnt_arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-resetting
                if (p.tuning & 1024) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            *s_flag = last;
        }
        __syncthreads();
        owner = *s_flag != 0;                                 // workgroup-uniform
        /*@stamp:6 arrival counted*/
        if (owner) {
            auto reduce = [&](auto ks_c) {
                constexpr int KS = decltype(ks_c)::value;
                constexpr int R = (KS + 3) / 4, NC = 4 * R;
                // (two chunks in flight = 128 + 64 accumulator registers: the one-wave-per-SIMD form has them)
                constexpr bool PIPE = true;
                f32x4_t t[2][4][4];
#pragma unroll
                for (int c = -1; c < NC; ++c) {
                    int nxt = 0;
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
            switch (g.ks) {
                case 2: reduce(std::integral_constant<int, 2>{}); break;
                case 3: reduce(std::integral_constant<int, 3>{}); break;
                case 4: reduce(std::integral_constant<int, 4>{}); break;
                case 5: reduce(std::integral_constant<int, 5>{}); break;
                case 6: reduce(std::integral_constant<int, 6>{}); break;
                case 7: reduce(std::integral_constant<int, 7>{}); break;
                default: reduce(std::integral_constant<int, 8>{}); break;
            }
            /*@stamp:7 partials read back*/
            finish_all();
        }
    }
    /*@stamp:8 rows stored*/
    /*@stamp:end*/
    return owner;
}

template <typename T, int WM, int NWST>
__global__ __launch_bounds__(WM * 128, WM / 2) void conv_small_kernel(const SmallParams sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)conv_small_body<T, WM, NWST>(sp, smem, blockIdx.x);
}

// A ResBlock's two convolutions in ONE launch (the A/B of VERDICT r04 item 1; tools/resblock_bench.py): every workgroup runs its
// share of conv1 (h = conv3x3(act(GN(x))) + bias (+ emb), written write-through, totals of h added), all workgroups meet at a grid
// barrier, then its share of conv2 (out = conv3x3(act(GN(h) FiLM)) + bias + x).  The grid is resident at once (host-checked: at most
// one workgroup per CU, >= 96 KB of LDS each).  Hand-off of h and its totals through the barrier: sc1 stores (and memory-side
// atomics), every wave's vmcnt(0), workgroup barrier, ONE lane's agent-scope add; that lane polls (sc1 loads, bounded), then ONE
// agent-scope acquire (invalidates this CU's L1; h's lines were never in any L2: sc1 stores drop them), vmcnt(0), workgroup barrier,
// plain loads - the guide's consumer form.  `bar`: two words (arrive, depart), zero between launches (the last to leave resets them).
template <typename T, int WM, int NWST>
__global__ __launch_bounds__(WM * 128, WM / 2) void resblock_small_kernel(const SmallParams sp1, const SmallParams sp2, int* bar) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)conv_small_body<T, WM, NWST>(sp1, smem, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this lane's stores of h, its statistics atomics and counter resets are done
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = gridDim.x;
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n && ++spins < SMALL_SPIN_MAX) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == n - 1) {                                       // everyone has passed the poll: re-zero for the next launch
            __hip_atomic_store(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    (void)conv_small_body<T, WM, NWST>(sp2, smem, blockIdx.x);
}

template <typename T, int WM, int NWST>
int launch_resblock(const SmallParams& sp1, const SmallParams& sp2, int* bar, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_small_kernel<T, WM, NWST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int lds = small_a_bytes(sp1.geo) + NWST * WST_BYTES + sp1.geo.nseg * sp1.geo.nb * 64 * 8;
    hipLaunchKernelGGL((resblock_small_kernel<T, WM, NWST>), dim3(sp1.geo.MT * sp1.k.NT * sp1.geo.ks), dim3(WM * 128), lds, stream, sp1, sp2, bar);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_resblock_small: launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

template <typename T, int WM, int NWST>
int launch_small(const SmallParams& sp, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_kernel<T, WM, NWST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int lds = small_a_bytes(sp.geo) + NWST * WST_BYTES + sp.geo.nseg * sp.geo.nb * 64 * 8;
    hipLaunchKernelGGL((conv_small_kernel<T, WM, NWST>), dim3(sp.geo.MT * sp.k.NT * sp.geo.ks), dim3(WM * 128), lds, stream, sp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(small): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}

DeviceOnce g_small_dev;

// one tile height (wm = 2 | 4 wave rows): segments, slots, k-slice; false if the map does not decompose that way
bool small_geom_for(const KParams& p, int wm, int ncu, SmallGeom& g) {
    const int W = p.Win, H = p.Hin, BMT = wm * 64;
    if (p.M % BMT) return false;
    const int TR = BMT / W;
    int nseg, SR;
    if (TR <= H) { if (H % TR) return false; nseg = 1; SR = TR; }
    else { if (TR % H) return false; nseg = TR / H; SR = H; }
    if (nseg > 4) return false;
    g.wm = wm; g.MT = p.M / BMT;
    g.TR = TR; g.SR = SR; g.nseg = nseg;
    g.slots = nseg * (SR + 2) * (W + 2);
    if (g.slots > 7 * (wm * 16)) return false;
    if (g.MT * p.NT > 511) return false;                      // arrival + departure counters: 2 x 512 words (the last one: the give-up mark)
    const int ncb = p.Ctot / 64;
    // k-slice: NB in {4, 3, 2, 1} blocks of 64 channels (whole slices, at most 8 of them; the weight ring needs >= 3 stages beside the
    // input slice).  ONE round of workgroups (measured: a second round costs more than the fused normalisation saves): the NB that
    // puts the most workgroups on the chip without exceeding one per CU, the larger NB (fewer partial sums) on a tie
    int nb = 0;
    int64_t best = 0;
    const int forced = (p.tuning >> 24) & 3;                  // tuning bits 24-25: NB = 1 / 2 / 4 (A/B)
    for (int c = 4; c >= 1; --c) {                            // (3: the 384- / 768- / 1536-channel concatenations of the up paths)
        if (ncb % c || ncb / c > 8) continue;
        if (c * g.slots * KB_BYTES + 3 * WST_BYTES + nseg * c * 64 * 8 > 160 * 1024) continue;
        if (forced && c != (forced == 3 ? 4 : forced)) continue;
        const int64_t grid = (int64_t)g.MT * p.NT * (ncb / c);
        if (grid > ncu && !forced) continue;
        if (grid > best) { best = grid; nb = c; }
    }
    if (!nb) return false;
    g.nb = nb; g.ks = ncb / nb;
    const int left = 160 * 1024 - nb * g.slots * KB_BYTES - nseg * nb * 64 * 8;
    g.nwst = left >= 8 * WST_BYTES ? 8 : left >= 6 * WST_BYTES ? 6 : left >= 4 * WST_BYTES ? 4 : 3;
    if (g.nwst - 1 > nb * 9) g.nwst = 4;                      // (a ring deeper than the k-loop is long)
    g.div_w = FastDiv::make(W); g.div_wp = FastDiv::make(W + 2); g.div_h = FastDiv::make(H); g.div_sr = FastDiv::make(SR);
    g.div_segslots = FastDiv::make((SR + 2) * (W + 2));
    return true;
}

}  // namespace

// Geometry of the small-map kernel for this launch, or false.  Tiles of BM = 128 or 256 consecutive pixels of the [B][H][W] raster:
// W in {8, 16, 32} (the 16 lanes of a fragment row enumerate xx), either a whole number of image rows inside one image
// (TR = BM / W <= H, H % TR == 0) or a whole number of images (TR % H == 0, at most 4); whole tiles in M and N; whole 64-channel
// blocks from either source.  256-pixel tiles wherever they still put >= 192 workgroups on the chip (half the weight bytes per MFMA).
bool nlc_conv_small_geom(const KParams& p, int dtype, SmallGeom& g) {
    if (!nlc_is16(dtype) || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.ups || p.res_ups) return false;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.out_mode != NLC_OUT_NHWC || p.math != NLC_MATH_NATIVE) return false;
    if (p.gn_coef || p.norm_out) return false;
    const int W = p.Win, H = p.Hin;
    if (!(W == 8 || W == 16 || W == 32) || (H & 1) || ((H * W) % 64)) return false;       // (a wave's 64 pixels lie inside one image)
    if ((p.Cout % BN) || p.Cout_pad != p.Cout) return false;
    if ((p.Ctot % 64) || (p.C0 % 64) || p.Cin_pad != p.Ctot) return false;
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    if (p.emb && ((reinterpret_cast<uintptr_t>(p.emb) & 15) || (p.emb_stride & 3))) return false;
    SmallGeom g2{}, g4{};
    const int ncu = g_small_dev.ncu[nlc_device_once(g_small_dev, [] {})];
    const bool ok2 = !(p.tuning & (1 << 26)) && small_geom_for(p, 2, ncu, g2);                // tuning bit 26: 256-pixel tiles or nothing (A/B)
    bool ok4 = !(p.tuning & (1 << 22)) && small_geom_for(p, 4, ncu, g4);                      // tuning bit 22: 128-pixel tiles only (A/B)
    // the distributed reduction needs every workgroup resident at once: at most one per CU (LDS), so at most as many as CUs
    auto can_dist = [&](const SmallGeom& q) { return (q.ks == 2 || q.ks == 4 || q.ks == 8) && (int64_t)q.MT * p.NT * q.ks <= ncu; };
    // 256-pixel tiles (half the weight bytes per MFMA, twice the partial sums at equal LDS): only with the distributed reduction (or
    // no split; their register budget has no room for the last arriver's read-back), and - measured, tools/resblock_bench.py - only
    // where the 128-pixel form is left with a three-stage weight ring (four-block slices) AND splits K itself (unsplit it has no partial
    // sums at all: 256 ch @8x8, B = 200: 70.6 vs 74.9 us per ResBlock) and they still fill >= 3/4 of the chip
    ok4 = ok4 && (g4.ks == 1 || can_dist(g4));
    const bool use4 = ok4 && (!ok2 || (g2.nwst < 4 && g2.ks > 1 && (int64_t)g4.MT * p.NT * g4.ks * 4 >= ncu * 3));
    if (!use4 && !ok2) return false;
    g = use4 ? g4 : g2;
    g.dist = g.ks > 1 && can_dist(g) && (g.wm == 4 || !(p.tuning & (1 << 23)));                           // tuning bit 23: last-arriver form (A/B; 128-pixel tiles)
    return true;
}

int64_t nlc_conv_small_split_bytes(const KParams& p, const SmallGeom& g) {
    return g.ks > 1 ? (int64_t)g.ks * p.M * p.NT * BN * (int64_t)sizeof(float) + 4096 : 0;
}

int nlc_conv_small_dispatch(const SmallParams& sp, int dtype, hipStream_t stream) {
    const int wm = sp.geo.wm, st = sp.geo.nwst;
#define NLC_SMALL_CASE(WMV, STV) if (wm == WMV && st == STV) { NLC_SWITCH_16(dtype, return (launch_small<T16, WMV, STV>(sp, stream))); }
    NLC_SMALL_CASE(2, 3) NLC_SMALL_CASE(2, 4) NLC_SMALL_CASE(2, 6) NLC_SMALL_CASE(2, 8)
    NLC_SMALL_CASE(4, 3) NLC_SMALL_CASE(4, 4) NLC_SMALL_CASE(4, 6) NLC_SMALL_CASE(4, 8)
#undef NLC_SMALL_CASE
    return NLC_EUNSUPPORTED;
}

// Both convolutions of a ResBlock in one launch: same geometry for both (Cin == Cout, no projection), a grid that is resident at once.
// Instantiated for the geometries of the measured shapes only (tools/resblock_bench.py); anything else: NLC_EUNSUPPORTED.
int nlc_resblock_small_dispatch(const SmallParams& sp1, const SmallParams& sp2, int* bar, int dtype, hipStream_t stream) {
    const SmallGeom &a = sp1.geo, &b = sp2.geo;
    if (a.wm != b.wm || a.nwst != b.nwst || a.nb != b.nb || a.ks != b.ks || a.MT != b.MT || sp1.k.NT != sp2.k.NT || a.slots != b.slots || a.nseg != b.nseg)
        return NLC_EUNSUPPORTED;
    const int ncu = g_small_dev.ncu[nlc_device_once(g_small_dev, [] {})];
    if ((int64_t)a.MT * sp1.k.NT * a.ks > ncu) return NLC_EUNSUPPORTED;
#define NLC_RB_CASE(WMV, STV) if (a.wm == WMV && a.nwst == STV) { NLC_SWITCH_16(dtype, return (launch_resblock<T16, WMV, STV>(sp1, sp2, bar, stream))); }
    NLC_RB_CASE(2, 8) NLC_RB_CASE(2, 6) NLC_RB_CASE(4, 3)
#undef NLC_RB_CASE
    return NLC_EUNSUPPORTED;
}
