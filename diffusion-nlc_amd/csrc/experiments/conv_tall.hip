// 3x3 stride-1 pad-1 convolution, bf16, "tall" variant of the LDS-halo kernel (conv_halo.hip): a 16 x 16 pixel patch x 256 output
// channels per workgroup, 32-channel k-blocks, 8 waves.
//
// Why it was built: with ALL 256 output channels of a patch in one workgroup the input halo is fetched - and, with the GroupNorm
// prologue (template parameter GN, nlc_conv_desc.gn_coef), normalised - once per patch instead of once per 128-channel N-tile;
// the prologue of conv_halo.hip transformed every input element NT x 1.27 times and lost more in the convolutions than the
// separate GroupNorm pass costs.
// MEASURED (ADM-256, B = 16, one box, interleaved, profiles/r02_ab_tall.log): production dispatch 6.26 images/s; this kernel for
// every eligible launch, GroupNorm as a separate pass 6.14; this kernel WITH the GroupNorm prologue (apply pass gone) 6.17.
// The kernel itself is level with conv_halo per launch (1164-1181 vs 1170-1172 TFLOP/s on 256->256 @256^2) and ~2 % behind over
// the network; normalising each element 1.27 times instead of 2.5-5 times still costs the convolutions as much as the
// memory-bound apply pass: on a chip that holds its clock down under the matrix load the extra VALU + transcendental work is
// paid in clock whether or not it overlaps the MFMAs.  So: OPT-IN (NLC_CONV_FORCE_TALL, or tuning bit 3 under NLC_CONV_AUTO for
// whole-network A/B runs), parity-tested with and without the prologue (tests/test_ops_gpu.py), never in the production dispatch.
//
//   512 threads = 8 waves as 4 (M) x 2 (N): wave (wm, wn) owns patch rows 4 wm .. 4 wm + 3 (4 M-tiles of 16 pixels) x 128 channels
//   (8 N-tiles): 128 accumulator registers per lane at two waves per SIMD.  k-step = one tap x 32 channels = ONE
//   v_mfma_f32_16x16x32_bf16 k: 32 MFMAs per wave, 12 fragment reads (the next step's four pixel fragments into the other
//   register set + a ring of four weight fragments refilled 16 MFMAs ahead), 2 weight pieces + (taps 0-2 of a channel block)
//   1 halo piece of the next block, one counted vmcnt wait + barrier.  LDS: 2 halo stages (21 pieces of 16 rows x 64 B) +
//   4 weight stages (256 rows x 64 B) + scratch = 114 KiB.  Rows are 64 bytes, 16-byte chunk index XOR-swizzled with
//   (row >> 1) & 3 (conflict-free ds_read_b128 for 16 consecutive rows at any alignment).  Halo pieces go through
//   buffer_load ... lds: rows outside the image carry an out-of-range offset and the hardware writes zeros.
//   Persistent over an XCD-contiguous tile list, DMA streams run across tile boundaries, register-direct epilogue with the
//   MFMA operands swapped (a lane holds 2 x 16 consecutive channels of one pixel per M-tile), bias / embedding folded into the
//   accumulators' initial value, ride-along GroupNorm statistics - all as in conv_halo.hip / conv_wide.hip.
// Shapes: bf16, NHWC output, Cout % 256 == 0, H % 16 == 0, W % 16 == 0, C0 % 64 == 0 and C1 % 64 == 0 (two 32-channel blocks
// per loop trip), no activation.
#include "common.h"
#include "conv_params.h"

namespace {

__device__ uint4 g_zero_page_t[16];             // 256 zero bytes: stands in for a missing bias / embedding
constexpr unsigned OOB = 0x80000000u;           // halo rows outside the image: an offset beyond any descriptor -> the DMA writes zeros

constexpr int WT = 512;                         // threads = 8 waves
constexpr int PT = 16;                          // output patch edge
constexpr int HWD = PT + 2;                     // 18 x 18 halo
constexpr int HROWS = HWD * HWD;                // 324
constexpr int RB = 64;                          // bytes per LDS row = 32 bf16 channels
constexpr int KBE = 32;                         // channels per k-block
constexpr int BNT = 256;                        // output channels per tile
constexpr int A_PIECES = 21;                    // DMA wave-instructions per halo (16 rows each): 336 rows
constexpr int A_STAGE = A_PIECES * 1024;        // 21 KiB
constexpr int B_STAGE = BNT * RB;               // 16 KiB
constexpr int NBST = 4;                         // weight stages (3 steps ahead)
constexpr int NA = 3, NB = 2;                   // DMA wave-instructions per wave: per halo / per weight tile
constexpr int SCRATCH = 8 * 1024;               // landing zone of the padding DMA instructions (one piece per wave)
constexpr int COEF_STAGE = 1024;                // GroupNorm prologue: (a, b) of 128 input channels = four k-blocks = one DMA piece
constexpr int TALL_LDS = 2 * A_STAGE + NBST * B_STAGE + SCRATCH + 2 * COEF_STAGE;      // 118,784 B

__device__ __forceinline__ void wglds(const void* gptr, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_base))
                 : "memory");
}
__device__ __forceinline__ void wglds_s(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_base)), "s"(sbase)
                 : "memory");
}
// Buffer form: address = descriptor base + voff (per lane, 32 bits) + soff (uniform); a lane whose offset lies outside the
// descriptor's num_records gets ZEROS written to its LDS slot - the conv's zero padding costs no zero page, no select and no
// 64-bit per-lane address (ten 64-bit halo pointers per lane, in several live versions, spilled inside the k-loop).
__device__ __forceinline__ void wglds_buf(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_base)), "s"(rsrc), "s"(soff)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct TileT { int tb, y0, x0, n0; };

template <bool UPS, bool GN>
__global__ __launch_bounds__(WT, 2) void conv_tall_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = bf16_raw;
    constexpr int ES = 2, PER = 8;

    const int tiles_x = p.Wout / PT, tiles_y = p.Hout / PT;
    const int NT2 = p.Cout / BNT;
    const int nblk = p.B * tiles_y * tiles_x * NT2;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int gx = (G - xcd + 7) >> 3;
    const int cq = nblk >> 3, cr = nblk & 7;
    const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_len = cq + (xcd < cr ? 1 : 0);
    auto decode = [&](int tl) {
        const int id = chunk_start + tl;
        const int mt = id / NT2, nt = id - mt * NT2;
        const int tb = mt / (tiles_y * tiles_x);
        const int trem = mt - tb * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        return TileT{tb, ty * PT, tx * PT, nt * BNT};
    };
    int tl = wi;
    if (tl >= chunk_len) return;                     // workgroup-uniform
    TileT cur = decode(tl);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lrow = lane >> 2;                      // row within a DMA wave-instruction (16 rows x 64 B)
    const int lslot = lane & 3;                      // LDS 16-byte slot this lane's DMA lands in
    const int hchunk = lslot ^ ((lrow >> 1) & 3);    // source chunk (source-side swizzle; piece bases are multiples of 16 rows)
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ldsB = lds0 + 2 * A_STAGE;
    const unsigned ldsScratch = ldsB + NBST * B_STAGE + wave * 1024;
    const unsigned ldsCoef = ldsB + NBST * B_STAGE + SCRATCH;
    char* smemB = smem + 2 * A_STAGE;
    const char* smemCoef = smemB + NBST * B_STAGE + SCRATCH;
    const int ncb = p.Ctot / KBE;                    // dispatch: C0 % 64 == 0, C1 % 64 == 0 -> even
    const int nk = ncb * 9;
    const int cbs1 = p.C0 / KBE;                     // first k-block of the second input segment

    // ---- halo pieces: piece q = wave + 8 j covers LDS rows 16 q .. 16 q + 15; offsets computed where the piece is issued
    //      (conv_wide.hip: kept per tile they spilled inside the k-loop)
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x0), 0, (int)((int64_t)p.B * p.Hin * p.Win * p.C0 * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.C1 > 0 ? p.x1 : p.x0), 0,
                                                                          (int)((int64_t)p.B * p.Hin * p.Win * (p.C1 > 0 ? p.C1 : p.C0) * ES), 0x00020000);
    const int64_t wrow = (int64_t)9 * p.Cin_pad * ES;
    // halo pieces [J0, J0 + N) of k-block cb of tile t; this lane's halo row of piece j = (tid >> 2) + 128 j
    auto issue_A = [&](const TileT& t, int cb, int astage, int cstage, auto j0_c, auto n_c) {
        constexpr int J0 = decltype(j0_c)::value, N = decltype(n_c)::value;
        const unsigned base = lds0 + astage * A_STAGE + wave * 1024;
        const bool second = cb >= cbs1 && p.C1 > 0;                          // wave-uniform
        const unsigned soff = (unsigned)(second ? cb - cbs1 : cb) * (KBE * ES);
        const unsigned C = (unsigned)(second ? p.C1 : p.C0);
        int tv = tid;
        asm volatile("" : "+v"(tv));                 // opaque: keeps the (row -> hy, hx) divisions HERE instead of hoisted + spilled
        const int r0v = tv >> 2;
        const unsigned hoff = (unsigned)(((tv & 3) ^ ((tv >> 3) & 3)) * (PER * ES));
#pragma unroll
        for (int j = J0; j < J0 + N; ++j) {
            const bool real = (wave + 8 * j) < A_PIECES;                       // wave-uniform
            const int R = r0v + 128 * j;
            const int hy = R / HWD, hx = R - hy * HWD;
            const int iy = t.y0 + hy - 1, ix = t.x0 + hx - 1;
            const bool ok = R < HROWS && iy >= 0 && iy < p.Hout && ix >= 0 && ix < p.Wout;
            const int sy = UPS ? iy >> 1 : iy, sx = UPS ? ix >> 1 : ix;          // fused nearest-2x upsample (src/unet_adm.py:107-109)
            const unsigned pixel = (unsigned)((t.tb * p.Hin + sy) * p.Win + sx);
            const unsigned voff = ok ? pixel * C * ES + hoff : OOB;
            if constexpr (GN) {
                // GroupNorm prologue: wave 7's third piece is padding (q = 23 >= 21) - at the first k-block of every group of four
                // it fetches that group's 128 (a, b) pairs instead: no extra instruction in anybody's vmcnt arithmetic
                if (j == 2 && wave == 7 && (cb & 3) == 0) {
                    const char* cbase = reinterpret_cast<const char*>(p.gn_coef) + ((int64_t)t.tb * p.Ctot + cb * KBE) * 8;     // wave-uniform
                    unsigned lo;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lo));
                    wglds_s(lo, cbase, ldsCoef + cstage * COEF_STAGE);
                    continue;
                }
            }
            if (second) wglds_buf(voff, rs1, soff, real ? base + j * 8192 : ldsScratch);
            else wglds_buf(voff, rs0, soff, real ? base + j * 8192 : ldsScratch);
        }
    };
    // GroupNorm (+FiLM) (+SiLU) of this lane's 16 bytes of own halo piece j (8 channels of one pixel) of k-block cb of tile t, in
    // place.  Rows outside the image stay zero: the convolution pads the NORMALISED input.
    auto xform = [&](const TileT& t, int cb, int astage, int cstage, int j) {
        if ((wave + 8 * j) >= A_PIECES) return;                               // wave-uniform
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int R = (tv >> 2) + 128 * j;
        const int hy = R / HWD, hx = R - hy * HWD;
        const int iy = t.y0 + hy - 1, ix = t.x0 + hx - 1;
        const bool ok = R < HROWS && iy >= 0 && iy < p.Hout && ix >= 0 && ix < p.Wout;
        if (!ok) return;
        char* xp = smem + astage * A_STAGE + (wave + 8 * j) * 1024 + (tv & 63) * 16;
        const uint4 d = *reinterpret_cast<const uint4*>(xp);
        const int hch = (tv & 3) ^ ((tv >> 3) & 3);
        const float* cf = reinterpret_cast<const float*>(smemCoef + cstage * COEF_STAGE + (cb & 3) * 256) + hch * 16;
        const unsigned w[4] = {d.x, d.y, d.z, d.w};
        unsigned o[4];
        const bool silu = p.gn_act == NLC_ACT_SILU;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 ab = *reinterpret_cast<const float4*>(cf + q * 4);       // a0 b0 a1 b1
            float y0 = fmaf(__uint_as_float(w[q] << 16), ab.x, ab.y);
            float y1 = fmaf(__uint_as_float(w[q] & 0xffff0000u), ab.z, ab.w);
            if (silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
            o[q] = (unsigned)f32_to_bf16(y0) | ((unsigned)f32_to_bf16(y1) << 16);
        }
        *reinterpret_cast<uint4*>(xp) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    // weight pieces of this wave: q = wave, wave + 8; LDS row R = 16 q + lrow holds output channel
    // (R & 192) + (m >> 2) * 16 + ((R >> 4) & 3) * 4 + (m & 3), m = R & 15: with the MFMA operands swapped lane (fr, fq) of wave
    // (wm, wn) then ends up with the 16 consecutive channels wn * 128 + g * 64 + fq * 16 .. + 15 (g = 0, 1) of one pixel per M-tile
    unsigned woff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int R = (wave + 8 * i) * 16 + lrow;
        const int m = R & 15;
        const int ch = (R & 192) + (m >> 2) * 16 + ((R >> 4) & 3) * 4 + (m & 3);
        woff[i] = (unsigned)((int64_t)ch * wrow + (int64_t)hchunk * PER * ES);
    }
    auto issue_B = [&](int n0, int kt, int bstage) {
        const int cb = kt / 9, tap = kt - cb * 9;
        const unsigned base = ldsB + bstage * B_STAGE + wave * 1024;
        const char* sb = p.w + (int64_t)n0 * wrow + ((int64_t)tap * p.Cin_pad + cb * KBE) * ES;       // wave-uniform
#pragma unroll
        for (int i = 0; i < NB; ++i) wglds_s(woff[i], sb, base + i * 8192);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][8];
    // per-lane fragment offsets: halo row of M-tile t (patch row 4 wm + t, columns fr) for tap (r, s) is a_lane + k with
    // k = (t + r) * 18 + s; a_lane = wm * 72 + fr and 72 = 9 * 8, so the swizzle term ((row >> 1) & 3) depends on (fr + k) & 7 only
    const int a_lane = wm * 4 * HWD + fr;
    int aoffm[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) aoffm[m] = a_lane * RB + ((fq ^ (((fr + m) >> 1) & 3)) << 4);
    const int boff = (wn * 128 + fr) * RB + ((fq ^ ((fr >> 1) & 3)) << 4);         // + j * 1024 for N-tile j (16 rows)

    // registers: 128 accumulators + two sets of four pixel fragments (used by all N-tiles of a step, so they alternate) + a ring
    // of FOUR weight fragments: slot j & 3 is refilled with N-tile j + 4 (of this k-step, or j - 4 of the next one) right after
    // the four MFMAs of N-tile j have been issued, 16 MFMAs = 256+ cycles ahead of its use
    uint4 fa0[4], fa1[4], fb[4];
    auto load_fa = [&](uint4 (&fa)[4], int astage, auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr int r = tap / 3, s = tap % 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = (t + r) * HWD + s;                                   // compile-time
            fa[t] = *reinterpret_cast<const uint4*>(smem + astage * A_STAGE + aoffm[k & 7] + k * RB);
        }
    };
    auto load_fb = [&](int bstage, auto j_c) {
        constexpr int j = decltype(j_c)::value;
        fb[j & 3] = *reinterpret_cast<const uint4*>(smemB + bstage * B_STAGE + boff + j * 1024);
    };
    auto mma4 = [&](const uint4 (&fa)[4], auto j_c) {
        constexpr int j = decltype(j_c)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[j & 3]), __builtin_bit_cast(bf16x8_t, fa[t]),
                                                                acc[t][j], 0, 0, 0);
    };

    // ---- epilogue: lane (fr, fq) of wave (wm, wn) holds, for M-tile t, pixel (row 4 wm + t, col fr) x channels
    //      n0 + wn * 128 + (j >> 2) * 64 + fq * 16 + (j & 3) * 4 + reg  (acc[t][j][reg]).  bias + embedding are the accumulators'
    //      initial value.
    const float* zf = reinterpret_cast<const float*>(g_zero_page_t);
    auto init_acc = [&](const TileT& t) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n = t.n0 + wn * 128 + g * 64 + fq * 16;
            const float* bp = p.bias ? p.bias + n : zf;
            const float* ep = p.emb ? p.emb + (int64_t)t.tb * p.emb_stride + n : zf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(bp + q * 4);
                const float4 e4 = *reinterpret_cast<const float4*>(ep + q * 4);
                const f32x4_t c4 = f32x4_t{b4.x + e4.x, b4.y + e4.y, b4.z + e4.z, b4.w + e4.w};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt][g * 4 + q] = c4;
            }
        }
    };
    auto row16_sum = [&](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    auto epilogue = [&](const TileT& t, const TileT& nx) {
        const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
        const float sc = p.out_scale;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n = t.n0 + wn * 128 + g * 64 + fq * 16;
            float gsum[2] = {0.f, 0.f}, gsq[2] = {0.f, 0.f};              // this lane's two 8-channel chunks of group g over its 4 pixels
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int64_t m = ((int64_t)t.tb * p.Hout + t.y0 + wm * 4 + tt) * p.Wout + t.x0 + fr;
                T* op = reinterpret_cast<T*>(p.out) + m * p.Cout + n;
                const T* rp = reinterpret_cast<const T*>(p.res) + res_row(p, t.tb, t.y0 + wm * 4 + tt, t.x0 + fr) * p.Cout + n;
#pragma unroll
                for (int c = 0; c < 2; ++c) {                                 // 8 channels = N-tiles 4 g + 2 c, 4 g + 2 c + 1
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = acc[tt][g * 4 + 2 * c + (k >> 2)][k & 3];
                    if (has_res) {
                        float rr[8];
                        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(rp + c * 8), rr);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += rr[k];
                    }
                    if (sc != 1.0f) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= sc;
                    }
                    const uint4 pk = f32_to_chunk<T>(v);
                    *reinterpret_cast<uint4*>(op + c * 8) = pk;
                    if (has_stats) {                 // of the STORED (bf16-rounded) values
                        float sv[8];
                        chunk_to_f32<T>(pk, sv);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { gsum[c] += sv[k]; gsq[c] = fmaf(sv[k], sv[k], gsq[c]); }
                    }
                }
            }
            if (has_stats) {
                // reduce over the 16 pixel lanes of a DPP row in a fixed order; one partial per (patch, M-wave): [b][part][chunk][{sum, sumsq}]
                const float r4[4] = {row16_sum(gsum[0]), row16_sum(gsq[0]), row16_sum(gsum[1]), row16_sum(gsq[1])};
                if (fr == 0) {
                    const int part = ((t.y0 / PT) * tiles_x + t.x0 / PT) * 4 + wm;
                    float* dst = p.stats + (((int64_t)t.tb * p.stats_P + part) * (p.Cout >> 3) + (n >> 3)) * 2;
                    *reinterpret_cast<float4*>(dst) = float4{r4[0], r4[1], r4[2], r4[3]};
                }
            }
        }
        init_acc(nx);
    };

    // ---- prologue (first tile only): halo of k-block 0, weights of steps 0..2
    init_acc(cur);
    int cst = 0;                                     // coefficient stage of the halo in flight (GroupNorm prologue)
    issue_A(cur, 0, 0, cst, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    issue_B(cur.n0, 0, 0);
    issue_B(cur.n0, 1, 1);
    issue_B(cur.n0, 2, 2);
    wwait<NB>();                                     // halo 0 + weights 0, 1 landed (weights 2 may fly)
    __syncthreads();
    if constexpr (GN) {                              // first halo of the launch: normalise all own pieces at once
#pragma unroll
        for (int j = 0; j < NA; ++j) xform(cur, 0, 0, cst, j);
        __syncthreads();
    }
    load_fa(fa0, 0, std::integral_constant<int, 0>{});
    [&]<int... j>(std::integer_sequence<int, j...>) { (load_fb(0, std::integral_constant<int, j>{}), ...); }(std::make_integer_sequence<int, 4>{});

    constexpr int wdist = 3;
    int bcur = 0, hs = 0;                            // weight stage / halo stage of the current k-step (run across tiles)
    for (;;) {
        const bool has_next = tl + gx < chunk_len;
        const TileT nxt = has_next ? decode(tl + gx) : cur;
        int kt = 0;
        for (int cb = 0; cb < ncb; cb += 2) {
            // one k-step: `par` says which pixel-fragment set holds this step's operands (steps alternate; two k-blocks = 18 steps per trip)
            auto step = [&](int cbx, auto tap_c, auto par_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int par = decltype(par_c)::value;
                const bool last_cb = cbx + 1 == ncb;
                const bool more = !last_cb || has_next;          // a halo follows this one in the stream
                const int bnext = (bcur + 1) & 3;
                constexpr int ntap = tap == 8 ? 0 : tap + 1;
                const int nast = tap == 8 ? hs ^ 1 : hs;
                // fragments interleaved with this step's 32 MFMAs: the next step's four pixel fragments + eight weight fragments
                // (N-tiles 4-7 of this step, 0-3 of the next): 4 x [4 MFMA, 2 reads] + 4 x [4 MFMA, 1 read]
                if constexpr (par == 0) load_fa(fa1, nast, std::integral_constant<int, ntap>{});
                else load_fa(fa0, nast, std::integral_constant<int, ntap>{});
                [&]<int... j>(std::integer_sequence<int, j...>) {
                    ((par == 0 ? mma4(fa0, std::integral_constant<int, j>{}) : mma4(fa1, std::integral_constant<int, j>{}),
                      load_fb(j < 4 ? bcur : bnext, std::integral_constant<int, (j + 4) & 7>{})), ...);
                }(std::make_integer_sequence<int, 8>{});
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                {   // DMA: weights of step kt + 3, then (taps 0-2) one halo piece of the next k-block
                    const int k3 = kt + wdist;
                    const bool wrap = k3 >= nk;
                    issue_B(wrap ? nxt.n0 : cur.n0, wrap ? (has_next ? k3 - nk : nk - 1) : k3, (bcur + wdist) & 3);
                    if constexpr (tap < NA) {
                        if constexpr (GN && tap == 0) { if (more && (((last_cb ? 0 : cbx + 1) & 3) == 0)) cst ^= 1; }     // a new group of four k-blocks
                        if (more) issue_A(last_cb ? nxt : cur, last_cb ? 0 : cbx + 1, hs ^ 1, cst, std::integral_constant<int, tap>{}, std::integral_constant<int, 1>{});
                    }
                }
                if constexpr (GN && tap >= 5 && tap <= 7) {
                    // GroupNorm prologue: the next k-block's halo pieces (issued at taps 0-2, retired by every wave's wait at tap 4, the
                    // coefficient piece published by that step's barrier) are normalised in place, own piece tap - 5 per step; the
                    // barriers of these steps publish the result before tap 8 prefetches the next block's first fragments.  Hard
                    // scheduling fences: the block must not be drawn into the [MFMA, ds_read] groups around it.
                    if (more) {
                        __builtin_amdgcn_sched_barrier(0);
                        xform(last_cb ? nxt : cur, last_cb ? 0 : cbx + 1, hs ^ 1, cst, tap - 5);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // retire weights kt + 2; younger instructions may stay in flight: this step's weight pieces (NB) and the halo pieces of
                // this step and the previous one (issue order per step: weights, then the halo piece)
                if constexpr (tap == 0 || tap == NA) { if (more) wwait<NB + 1>(); else wwait<NB>(); }
                else if constexpr (tap >= 1 && tap < NA) { if (more) wwait<NB + 2>(); else wwait<NB>(); }
                else wwait<NB>();
                __syncthreads();
                bcur = bnext;
                ++kt;
                if constexpr (tap == 8) hs ^= 1;
            };
            using P0 = std::integral_constant<int, 0>;
            using P1 = std::integral_constant<int, 1>;
            step(cb, std::integral_constant<int, 0>{}, P0{}); step(cb, std::integral_constant<int, 1>{}, P1{}); step(cb, std::integral_constant<int, 2>{}, P0{});
            step(cb, std::integral_constant<int, 3>{}, P1{}); step(cb, std::integral_constant<int, 4>{}, P0{}); step(cb, std::integral_constant<int, 5>{}, P1{});
            step(cb, std::integral_constant<int, 6>{}, P0{}); step(cb, std::integral_constant<int, 7>{}, P1{}); step(cb, std::integral_constant<int, 8>{}, P0{});
            step(cb + 1, std::integral_constant<int, 0>{}, P1{}); step(cb + 1, std::integral_constant<int, 1>{}, P0{}); step(cb + 1, std::integral_constant<int, 2>{}, P1{});
            step(cb + 1, std::integral_constant<int, 3>{}, P0{}); step(cb + 1, std::integral_constant<int, 4>{}, P1{}); step(cb + 1, std::integral_constant<int, 5>{}, P0{});
            step(cb + 1, std::integral_constant<int, 6>{}, P1{}); step(cb + 1, std::integral_constant<int, 7>{}, P0{}); step(cb + 1, std::integral_constant<int, 8>{}, P1{});
        }
        epilogue(cur, nxt);                          // registers -> global, asynchronous stores; no LDS, no barrier
        if (!has_next) break;
        cur = nxt;
        tl += gx;
    }
    wwait<0>();                                      // the redundant tail fetches
}

bool tall_eligible(const KParams& p, int dtype) {
    // opt-in: NLC_CONV_FORCE_TALL, or tuning bit 3 under NLC_CONV_AUTO (A/B runs of whole networks)
    if (dtype != NLC_BF16 || !(p.policy == NLC_CONV_FORCE_TALL || (p.policy == NLC_CONV_AUTO && (p.tuning & 8)))) return false;
    if (!(p.KH == 3 && p.KW == 3 && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1)) return false;
    const int HL = p.ups ? 2 * p.Hin : p.Hin, WL = p.ups ? 2 * p.Win : p.Win;
    if (p.Hout % PT || p.Wout % PT || p.Hout != HL || p.Wout != WL) return false;
    if (p.C0 % 64 || p.C1 % 64 || p.Ctot / KBE > 256) return false;
    if (p.Cout % BNT || p.out_mode != NLC_OUT_NHWC || p.act != NLC_ACT_NONE) return false;
    if (p.gn_coef && (p.Ctot % 128 || (reinterpret_cast<uintptr_t>(p.gn_coef) & 15))) return false;     // whole coefficient pieces
    if ((int64_t)p.B * p.Hout * p.Wout >= (1ll << 31)) return false;
    if ((int64_t)p.B * p.Hin * p.Win * (p.C0 > p.C1 ? p.C0 : p.C1) * 2 >= (1ll << 31)) return false;         // 32-bit halo offsets, OOB marker 2^31
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    if (p.emb && ((reinterpret_cast<uintptr_t>(p.emb) & 15) || (p.emb_stride & 3))) return false;
    const int blocks = p.B * (p.Hout / PT) * (p.Wout / PT) * (p.Cout / BNT);
    return p.policy == NLC_CONV_FORCE_TALL || blocks >= 128;
}

}  // namespace

int nlc_conv_tall_stats_partials(const KParams& p, int dtype) {
    if (!tall_eligible(p, dtype)) return 0;
    return (p.Hout / PT) * (p.Wout / PT) * 4;
}

int nlc_conv_tall_prologue_ok(const KParams& p, int dtype) {
    KParams q = p;
    if (!q.gn_coef) q.gn_coef = reinterpret_cast<const float*>(uintptr_t(16));     // "would a table be accepted": alignment is the caller's
    return tall_eligible(q, dtype) && q.Ctot % 128 == 0 ? 1 : 0;
}

int nlc_conv_tall_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!tall_eligible(p, dtype)) return NLC_EUNSUPPORTED;
    static DeviceOnce once;
    const int slot = nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tall_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tall_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tall_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tall_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TALL_LDS);
    });
    const int ncu = once.ncu[slot];
    const int nblk = p.B * (p.Hout / PT) * (p.Wout / PT) * (p.Cout / BNT);
    const int grid = nblk < ncu ? nblk : ncu;        // one persistent workgroup per CU
    if (p.gn_coef) {
        if (p.ups) hipLaunchKernelGGL((conv_tall_kernel<true, true>), dim3(grid), dim3(WT), TALL_LDS, stream, p);
        else hipLaunchKernelGGL((conv_tall_kernel<false, true>), dim3(grid), dim3(WT), TALL_LDS, stream, p);
    } else {
        if (p.ups) hipLaunchKernelGGL((conv_tall_kernel<true, false>), dim3(grid), dim3(WT), TALL_LDS, stream, p);
        else hipLaunchKernelGGL((conv_tall_kernel<false, false>), dim3(grid), dim3(WT), TALL_LDS, stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(tall): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}
