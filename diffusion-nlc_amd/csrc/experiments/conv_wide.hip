// 3x3 stride-1 pad-1 convolution, bf16, "wide" variant of the LDS-halo kernel (conv_halo.hip) for the big launches:
// a 16 x 32 pixel patch (512 pixels) x 128 output channels per workgroup, 32-channel k-blocks, 8 waves.
//
// Why a second tile shape was tried: in conv_halo.hip (256 pixels x 128 cout) every k-step stages a 16 KiB weight tile for 256
// pixels' worth of MFMAs and reads 128 KiB of fragments from LDS per 1024 matrix cycles; here a weight tile serves 512 pixels
// (96 KiB of fragment reads per 1024 matrix cycles, 3.1 DMA pieces per SIMD instead of 5.3) and an 18 x 34 halo per 512 pixels
// is 1.20x the patch instead of 1.27x.  The price is 128 accumulator registers per lane at two waves per SIMD.
// MEASURED (tools/conv_bench.py, same box, interleaved): 1236 vs 1265 TFLOP/s on 256->256 @256^2, 1341 vs 1368 on 512->256 @256^2,
// 1071 vs 1195 on 256->256 @128^2, 1151 vs 1268 on 512->512 @64^2 (wide vs conv_halo): level at best.  Both kernels sit at the
// clock the chip holds under this matrix load (1.68 GHz of 2.4, profiles/r02_pmc_sq_conv256.json), so fewer LDS bytes and DMA
// issues per MFMA did not turn into time.  The kernel therefore stays OPT-IN (NLC_CONV_FORCE_WIDE); it is parity-tested
// (tests/test_ops_gpu.py::test_conv2d_wide_kernel) and kept as the second data point on the tile shape.
// (A first version with ONE wave per SIMD - 4 waves x 128 pixels x 128 channels, 256 accumulators - measured 17 % slower than
//  conv_halo: with no partner wave on the SIMD every DMA issue stalls the matrix pipe outright.)
//
//   512 threads = 8 waves as 4 (M) x 2 (N): wave (wm, wn) owns patch rows 4 wm .. 4 wm + 3 (8 M-tiles of 16 pixels) x 64 channels.
//   k-step = one tap x 32 channels = ONE v_mfma_f32_16x16x32_bf16 k: 32 MFMAs per wave, 12 fragment reads (a ring of four pixel
//   fragments refilled 16 MFMAs ahead + the next step's four weight fragments), 1 weight piece + (taps 0-4 of a channel block) 1 halo piece of the next block, one counted
//   vmcnt wait + barrier.  LDS: 2 halo stages (39 pieces of 16 rows x 64 B) + 4 weight stages (128 rows x 64 B) + scratch
//   = 118 KiB.  Rows are 64 bytes, 16-byte chunk index XOR-swizzled with (row >> 1) & 3: conflict-free ds_read_b128 for 16
//   consecutive rows at any alignment (brute-forced over all 64 alignments).  Halo pieces go through buffer_load ... lds: rows
//   outside the image carry an out-of-range offset and the hardware writes zeros (no zero page, 32-bit offsets).
//   Persistent over an XCD-contiguous tile list, DMA streams run across tile boundaries, register-direct epilogue with the
//   MFMA operands swapped (a lane holds 16 consecutive channels of one pixel per M-tile), bias / embedding folded into the
//   accumulators' initial value, ride-along GroupNorm statistics - all as in conv_halo.hip.
// Shapes: bf16, NHWC output, Cout % 128 == 0, H % 16 == 0, W % 32 == 0, C0 % 64 == 0 and C1 % 64 == 0 (two 32-channel blocks
// per loop trip), no activation; everything else stays on conv_halo / conv_fast.
#include "common.h"
#include "conv_params.h"

namespace {

__device__ uint4 g_zero_page_w[8];              // 128 zero bytes: stands in for a missing bias / embedding
constexpr unsigned OOB = 0x80000000u;           // halo rows outside the image: an offset beyond any descriptor -> the DMA writes zeros

constexpr int WT = 512;                         // threads = 8 waves
constexpr int PH = 16, PW = 32;                 // output patch
constexpr int HWD = PW + 2, HHT = PH + 2;       // 34 x 18 halo
constexpr int HROWS = HWD * HHT;                // 612
constexpr int RB = 64;                          // bytes per LDS row = 32 bf16 channels
constexpr int KBE = 32;                         // channels per k-block
constexpr int A_PIECES = 39;                    // DMA wave-instructions per halo (16 rows each): 624 rows
constexpr int A_STAGE = A_PIECES * 1024;        // 39 KiB
constexpr int B_STAGE = BN * RB;                // 8 KiB
constexpr int NBST = 4;                         // weight stages (3 steps ahead)
constexpr int NA = 5, NB = 1;                   // DMA wave-instructions per wave: per halo / per weight tile
constexpr int SCRATCH = 8 * 1024;               // landing zone of the padding DMA instructions (one piece per wave)
constexpr int WIDE_LDS = 2 * A_STAGE + NBST * B_STAGE + SCRATCH;      // 120,832 B

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

struct TileW { int tb, y0, x0, n0; };

template <bool UPS>
__global__ __launch_bounds__(WT, 2) void conv_wide_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = bf16_raw;
    constexpr int ES = 2, PER = 8;

    const int tiles_x = p.Wout / PW, tiles_y = p.Hout / PH;
    const int nblk = p.B * tiles_y * tiles_x * p.NT;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int gx = (G - xcd + 7) >> 3;
    const int cq = nblk >> 3, cr = nblk & 7;
    const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_len = cq + (xcd < cr ? 1 : 0);
    auto decode = [&](int tl) {
        const int id = chunk_start + tl;
        const int mt = id / p.NT, nt = id - mt * p.NT;
        const int tb = mt / (tiles_y * tiles_x);
        const int trem = mt - tb * tiles_y * tiles_x;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        return TileW{tb, ty * PH, tx * PW, nt * BN};
    };
    int tl = wi;
    if (tl >= chunk_len) return;                     // workgroup-uniform
    TileW cur = decode(tl);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lrow = lane >> 2;                      // row within a DMA wave-instruction (16 rows x 64 B)
    const int lslot = lane & 3;                      // LDS 16-byte slot this lane's DMA lands in
    const int hchunk = lslot ^ ((lrow >> 1) & 3);    // source chunk (source-side swizzle; piece bases are multiples of 16 rows)
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ldsB = lds0 + 2 * A_STAGE;
    const unsigned ldsScratch = ldsB + NBST * B_STAGE + wave * 1024;
    char* smemB = smem + 2 * A_STAGE;
    const int ncb = p.Ctot / KBE;                    // dispatch: C0 % 64 == 0, C1 % 64 == 0 -> even
    const int nk = ncb * 9;
    const int cbs1 = p.C0 / KBE;                     // first k-block of the second input segment

    // ---- halo pieces: piece q = wave + 8 j covers LDS rows 16 q .. 16 q + 15.  The 32-bit byte offset of this lane's row into the
    //      input segment's buffer descriptor is computed where the piece is issued (a dozen VALU per piece against 64 MFMAs per
    //      step) instead of being kept per tile: ten offsets per lane in two live versions were what spilled inside the k-loop.
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x0), 0, (int)((int64_t)p.B * p.Hin * p.Win * p.C0 * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.C1 > 0 ? p.x1 : p.x0), 0,
                                                                          (int)((int64_t)p.B * p.Hin * p.Win * (p.C1 > 0 ? p.C1 : p.C0) * ES), 0x00020000);
    // halo row of piece j of this lane = (tid >> 2) + 128 j
    const int64_t wrow = (int64_t)9 * p.Cin_pad * ES;
    // halo pieces [J0, J0 + N) of k-block cb of tile t
    auto issue_A = [&](const TileW& t, int cb, int astage, auto j0_c, auto n_c) {
        constexpr int J0 = decltype(j0_c)::value, N = decltype(n_c)::value;
        const unsigned base = lds0 + astage * A_STAGE + wave * 1024;
        const bool second = cb >= cbs1 && p.C1 > 0;                          // wave-uniform
        const unsigned soff = (unsigned)(second ? cb - cbs1 : cb) * (KBE * ES);
        const unsigned C = (unsigned)(second ? p.C1 : p.C0);
        int tv = tid;
        asm volatile("" : "+v"(tv));                 // opaque: keeps the (row -> hy, hx) divisions and the chunk term HERE instead of
        const int r0v = tv >> 2;                     // hoisted + spilled (a reload inside the k-loop drains the DMA queue: vmcnt(0))
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
            if (second) wglds_buf(voff, rs1, soff, real ? base + j * 8192 : ldsScratch);
            else wglds_buf(voff, rs0, soff, real ? base + j * 8192 : ldsScratch);
        }
    };
    // weight piece of this wave: LDS row R = 16 wave + lrow holds output channel (R & 64) + (m >> 2) * 16 + ((R >> 4) & 3) * 4 + (m & 3),
    // m = R & 15: with the MFMA operands swapped lane (fr, fq) of wave (wm, wn) then ends up with the 16 consecutive channels
    // wn * 64 + fq * 16 .. + 15 of one pixel per M-tile (as in conv_halo.hip)
    unsigned woff;
    {
        const int R = wave * 16 + lrow;
        const int m = R & 15;
        const int ch = (R & 64) + (m >> 2) * 16 + ((R >> 4) & 3) * 4 + (m & 3);
        woff = (unsigned)((int64_t)ch * wrow + (int64_t)hchunk * PER * ES);
    }
    auto issue_B = [&](int n0, int kt, int bstage) {
        const int cb = kt / 9, tap = kt - cb * 9;
        const unsigned base = ldsB + bstage * B_STAGE + wave * 1024;
        const char* sb = p.w + (int64_t)n0 * wrow + ((int64_t)tap * p.Cin_pad + cb * KBE) * ES;       // wave-uniform
        wglds_s(woff, sb, base);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[8][4];
    // per-lane fragment offsets: halo row of M-tile t (patch row 4 wave + (t >> 1), columns 16 (t & 1) + fr) for tap (r, s) is
    // a_lane + k with k = ((t >> 1) + r) * 34 + 16 (t & 1) + s; a_lane = wm * 136 + fr and 136 = 17 * 8, so the swizzle term
    // ((row >> 1) & 3) depends on (fr + k) & 7 only: 8 registers + the compile-time k * 64 in the immediate offset
    const int a_lane = wm * 4 * HWD + fr;
    int aoffm[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) aoffm[m] = a_lane * RB + ((fq ^ (((fr + m) >> 1) & 3)) << 4);
    const int boff = (wn * 64 + fr) * RB + ((fq ^ ((fr >> 1) & 3)) << 4);          // + j * 1024 for N-tile j (16 rows)

    // register budget at two waves per SIMD is 256: 128 accumulators + a ring of FOUR pixel fragments (16) + two sets of weight
    // fragments (2 x 16).  Slot t & 3 is refilled with M-tile t + 4 (of this k-step, or t - 4 of the next one) right after the
    // four MFMAs of M-tile t have been issued: 16 MFMAs = 256+ cycles ahead of its use.  The weight fragments are used by all
    // M-tiles of a step, so they alternate between two sets.
    uint4 fa[4], fb0[4], fb1[4];
    auto load_fb = [&](uint4 (&fb)[4], int bstage) {
        const char* Bs = smemB + bstage * B_STAGE + boff;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Bs + j * 1024);
    };
    auto load_fa = [&](int astage, auto tap_c, auto t_c) {
        constexpr int tap = decltype(tap_c)::value, t = decltype(t_c)::value;
        constexpr int r = tap / 3, s = tap % 3;
        constexpr int k = ((t >> 1) + r) * HWD + 16 * (t & 1) + s;
        fa[t & 3] = *reinterpret_cast<const uint4*>(smem + astage * A_STAGE + aoffm[k & 7] + k * RB);
    };
    auto mma4 = [&](const uint4 (&fb)[4], auto t_c) {
        constexpr int t = decltype(t_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[j]), __builtin_bit_cast(bf16x8_t, fa[t & 3]),
                                                                acc[t][j], 0, 0, 0);
    };

    // ---- epilogue: lane (fr, fq) of wave (wm, wn) holds, for M-tile t, pixel (row 4 wm + (t >> 1), col 16 (t & 1) + fr) x channels
    //      n0 + wn * 64 + fq * 16 + j * 4 + reg  (acc[t][j][reg]).  bias + embedding are the accumulators' initial value.
    const float* zf = reinterpret_cast<const float*>(g_zero_page_w);
    auto load_cadd = [&](const TileW& t, float (&cadd)[16]) {
        const int n = t.n0 + wn * 64 + fq * 16;
        const float* bp = p.bias ? p.bias + n : zf;
        const float* ep = p.emb ? p.emb + (int64_t)t.tb * p.emb_stride + n : zf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + q * 4);
            const float4 e4 = *reinterpret_cast<const float4*>(ep + q * 4);
            cadd[q * 4] = b4.x + e4.x; cadd[q * 4 + 1] = b4.y + e4.y; cadd[q * 4 + 2] = b4.z + e4.z; cadd[q * 4 + 3] = b4.w + e4.w;
        }
    };
    auto row16_sum = [&](float x) {
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
        return x;
    };
    auto init_acc = [&](const float (&cadd)[16]) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = f32x4_t{cadd[j * 4], cadd[j * 4 + 1], cadd[j * 4 + 2], cadd[j * 4 + 3]};
    };
    auto epilogue = [&](const TileW& t, const TileW& nx) {
        float cnext[16];
        load_cadd(nx, cnext);                        // in flight while this tile is stored
        const int n = t.n0 + wn * 64 + fq * 16;
        const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
        const float sc = p.out_scale;
        float gsum[2] = {0.f, 0.f}, gsq[2] = {0.f, 0.f};              // this lane's two 8-channel chunks over its 8 pixels
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int64_t m = ((int64_t)t.tb * p.Hout + t.y0 + wm * 4 + (tt >> 1)) * p.Wout + t.x0 + 16 * (tt & 1) + fr;
            T* op = reinterpret_cast<T*>(p.out) + m * p.Cout + n;
            const T* rp = reinterpret_cast<const T*>(p.res) + res_row(p, t.tb, t.y0 + wm * 4 + (tt >> 1), t.x0 + 16 * (tt & 1) + fr) * p.Cout + n;
#pragma unroll
            for (int c = 0; c < 2; ++c) {                                     // 8 channels = N-tiles 2c, 2c+1
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[tt][2 * c + (k >> 2)][k & 3];
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
                if (has_stats) {                     // of the STORED (bf16-rounded) values
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
                const int part = ((t.y0 / PH) * tiles_x + t.x0 / PW) * 4 + wm;
                float* dst = p.stats + (((int64_t)t.tb * p.stats_P + part) * (p.Cout >> 3) + (n >> 3)) * 2;
                *reinterpret_cast<float4*>(dst) = float4{r4[0], r4[1], r4[2], r4[3]};
            }
        }
        init_acc(cnext);
    };

    // ---- prologue (first tile only): halo of k-block 0, weights of steps 0..2
    {
        float c0[16];
        load_cadd(cur, c0);
        init_acc(c0);
    }
    issue_A(cur, 0, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    issue_B(cur.n0, 0, 0);
    issue_B(cur.n0, 1, 1);
    issue_B(cur.n0, 2, 2);
    wwait<NB>();                                     // halo 0 + weights 0, 1 landed (weights 2 may fly)
    __syncthreads();
    load_fb(fb0, 0);
    [&]<int... t>(std::integer_sequence<int, t...>) {
        (load_fa(0, std::integral_constant<int, 0>{}, std::integral_constant<int, t>{}), ...);
    }(std::make_integer_sequence<int, 4>{});

    constexpr int wdist = 3;
    int bcur = 0, hs = 0;                            // weight stage / halo stage of the current k-step (run across tiles)
    for (;;) {
        const bool has_next = tl + gx < chunk_len;
        const TileW nxt = has_next ? decode(tl + gx) : cur;
        int kt = 0;
        for (int cb = 0; cb < ncb; cb += 2) {
            // one k-step: `par` says which fragment set holds this step's operands (steps alternate; two k-blocks = 18 steps per trip)
            auto step = [&](int cbx, auto tap_c, auto par_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int par = decltype(par_c)::value;
                const bool last_cb = cbx + 1 == ncb;
                const bool more = !last_cb || has_next;          // a halo follows this one in the stream
                const int bnext = (bcur + 1) & 3;
                constexpr int ntap = tap == 8 ? 0 : tap + 1;
                const int nast = tap == 8 ? hs ^ 1 : hs;
                // fragments interleaved with this step's 32 MFMAs: the next step's four weight fragments + eight pixel fragments
                // (M-tiles 4-7 of this step, 0-3 of the next): 4 x [4 MFMA, 2 reads] + 4 x [4 MFMA, 1 read]
                if constexpr (par == 0) load_fb(fb1, bnext); else load_fb(fb0, bnext);
                [&]<int... t>(std::integer_sequence<int, t...>) {
                    ((par == 0 ? mma4(fb0, std::integral_constant<int, t>{}) : mma4(fb1, std::integral_constant<int, t>{}),
                      t < 4 ? load_fa(hs, std::integral_constant<int, tap>{}, std::integral_constant<int, (t + 4) & 7>{})
                            : load_fa(nast, std::integral_constant<int, ntap>{}, std::integral_constant<int, (t + 4) & 7>{})), ...);
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
                {   // DMA: weights of step kt + 3, then (taps 0-4) one halo piece of the next k-block
                    const int k3 = kt + wdist;
                    const bool wrap = k3 >= nk;
                    issue_B(wrap ? nxt.n0 : cur.n0, wrap ? (has_next ? k3 - nk : nk - 1) : k3, (bcur + wdist) & 3);
                    if constexpr (tap <= 4) {
                        if (more) issue_A(last_cb ? nxt : cur, last_cb ? 0 : cbx + 1, hs ^ 1, std::integral_constant<int, tap>{}, std::integral_constant<int, 1>{});
                    }
                }
                // retire weights kt + 2; younger instructions may stay in flight: this step's weight piece (NB) and the halo pieces of
                // this step and the previous one (issue order per step: weights, then the halo piece)
                if constexpr (tap == 0 || tap == 5) { if (more) wwait<NB + 1>(); else wwait<NB>(); }
                else if constexpr (tap >= 1 && tap <= 4) { if (more) wwait<NB + 2>(); else wwait<NB>(); }
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

bool wide_eligible(const KParams& p, int dtype) {
    // opt-in only (NLC_CONV_FORCE_WIDE): measured level with conv_halo on the 256^2 maps and 8-10 % slower on the 128^2 / 64^2 ones
    // (profiles/r02_summary.md), so the production dispatch (NLC_CONV_AUTO) never selects it
    if (dtype != NLC_BF16 || p.policy != NLC_CONV_FORCE_WIDE) return false;
    if (!(p.KH == 3 && p.KW == 3 && p.pad_t == 1 && p.pad_l == 1 && p.stride == 1)) return false;
    const int HL = p.ups ? 2 * p.Hin : p.Hin, WL = p.ups ? 2 * p.Win : p.Win;
    if (p.Hout % PH || p.Wout % PW || p.Hout != HL || p.Wout != WL) return false;
    if (p.C0 % 64 || p.C1 % 64 || p.Ctot / KBE > 256) return false;
    if (p.Cout % BN || p.out_mode != NLC_OUT_NHWC || p.act != NLC_ACT_NONE || p.gn_coef) return false;
    if ((int64_t)p.B * p.Hout * p.Wout >= (1ll << 31)) return false;
    if ((int64_t)p.B * p.Hin * p.Win * (p.C0 > p.C1 ? p.C0 : p.C1) * 2 >= (1ll << 31)) return false;         // 32-bit halo offsets, OOB marker 2^31
    if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return false;
    if (p.emb && ((reinterpret_cast<uintptr_t>(p.emb) & 15) || (p.emb_stride & 3))) return false;
    return true;
}

}  // namespace

int nlc_conv_wide_stats_partials(const KParams& p, int dtype) {
    if (!wide_eligible(p, dtype)) return 0;
    return (p.Hout / PH) * (p.Wout / PW) * 4;
}

int nlc_conv_wide_dispatch(const KParams& p, int dtype, hipStream_t stream) {
    if (!wide_eligible(p, dtype)) return NLC_EUNSUPPORTED;
    static DeviceOnce once;
    const int slot = nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    });
    const int ncu = once.ncu[slot];
    const int nblk = p.B * (p.Hout / PH) * (p.Wout / PW) * p.NT;
    const int grid = nblk < ncu ? nblk : ncu;        // one persistent workgroup per CU
    if (p.ups) hipLaunchKernelGGL(conv_wide_kernel<true>, dim3(grid), dim3(WT), WIDE_LDS, stream, p);
    else hipLaunchKernelGGL(conv_wide_kernel<false>, dim3(grid), dim3(WT), WIDE_LDS, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { nlc_set_error("nlc_conv2d(wide): launch failed: %s", hipGetErrorString(e)); return NLC_ELAUNCH; }
    return NLC_OK;
}
