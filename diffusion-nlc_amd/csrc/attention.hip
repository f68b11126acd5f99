// Flash-style multi-head softmax attention on token-major tensors, gfx950.
//
//   qkv : [B][T][3][H][D]      out : [B][T][H][D]      out = softmax(q k^T) v   (scale pre-folded)
//
// One 256-thread workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns a
// 16-row MFMA M-tile.  Keys/values stream through LDS in tiles of KVT tokens; the T x T score
// matrix never reaches HBM (the reference materialises it: src/unet_adm.py:349-353).
//   S = Q K^T : A = Q rows (k = d contiguous, fragments held in registers), B = K rows from LDS.
//   online softmax in f32 on the MFMA C layout (row = 4*(lane>>4)+reg, col = lane&15):
//       row max via 4 xor-shuffles inside each 16-lane group, row sums kept per lane and
//       reduced once at the end.
//   O += P V  : P goes through a per-wave LDS scratch (C layout -> A layout),
//       bf16: V B-fragments by ds_read_b64_tr_b16 (hardware transposed read of the row-major V tile)
//       f32 : ds_read_b32, with the same k permutation (k = 16kk + 4*(lane>>4) + j) on P and V.
// All LDS images are row-major with the 16-byte chunk index XOR-swizzled by row bits.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int NTHREADS = 256;
constexpr int QB = 64;      // query rows per workgroup

// reductions over the 16 lanes of a DPP row (= the 16 score columns a lane group holds), result in every lane: VALU
// only (quad_perm xor 1, xor 2, row_half_mirror, row_mirror).  A __shfl_xor is a ds_bpermute (~100 cycles); the 16 of
// them per key tile cost several times the tile's 16 MFMAs.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x)); x = fmaxf(x, dpp_mov<0x4E>(x));
    x = fmaxf(x, dpp_mov<0x141>(x)); x = fmaxf(x, dpp_mov<0x140>(x));
    return x;
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_mov<0xB1>(x); x += dpp_mov<0x4E>(x); x += dpp_mov<0x141>(x); x += dpp_mov<0x140>(x);
    return x;
}

template <int RB> __device__ __forceinline__ int swz_chunk(int row, int c) {
    if constexpr (RB >= 256) return c ^ (row & 15);
    else if constexpr (RB == 128) return c ^ ((row >> 1) & 7);
    else if constexpr (RB == 64) return c ^ ((4 - ((row >> 2) & 3)) & 3);
    else return c;
}
template <int RB> __device__ __forceinline__ int lds_addr(int row, int c) { return row * RB + (swz_chunk<RB>(row, c) << 4); }

template <typename T> using AttnMma = Mfma16<T>;

template <typename T, int D, bool DMA_OK = true> struct AttnCfg {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int PER = 16 / ES;
    static constexpr int KVT = (D * ES <= 256) ? 64 : 32;
    static constexpr int RBK = D * ES;                 // row bytes of K / V tiles
    static constexpr int RBP = KVT * ES;               // row bytes of the P scratch
    static constexpr int KS = 4 * PER;                 // elements per MFMA k-substep
    static constexpr int NKS = D / KS;                 // substeps over the head dim
    static constexpr int NPS = KVT / KS;               // substeps over the key tile
    static constexpr int NNT = KVT / 16;               // S N-tiles
    static constexpr int NDT = D / 16;                 // O N-tiles
    static constexpr bool QREG = (NKS * 4 <= 64);      // keep Q fragments in registers
    static constexpr int K_BYTES = KVT * RBK;
    static constexpr int P_BYTES = 16 * RBP;
    // K / V tiles by LDS-DMA into two stages (the next tile lands while this one is multiplied) wherever the rows are whole 256-byte
    // multiples (the XOR swizzle of lds_addr is then an involution on the low 4 chunk bits: applied on the source side) and two stages
    // fit.  Taken by launches of at most two workgroups per CU (launch()): with nobody else on the CU to cover a workgroup's load
    // round trip it is 16-36 % faster (T = 256, D = 512, B = 8: 39.5 -> 32.8 us; T = 1024, D = 256, B = 16: 93.8 -> 60.0); a launch
    // that oversubscribes the chip covers the round trips with other workgroups and is 11 % SLOWER with it (T = 256, D = 256, B = 200:
    // 62.8 -> 69.8 us).  Four stages instead of two: level or slower (62.0 / 75.5 us on the last two) - a tile is ~3 800 cycles of one
    // wave per SIMD issuing its VALU, LDS and MFMA work in line, not a load round trip.  (A V-tile swizzle built so that the eight rows
    // a 32-lane half of ds_read_b64_tr_b16 touches fall into eight different 32-byte bank groups - on paper the K swizzle used for V
    // gives them four - measured level to 6 % slower, profiles/r04i_attn_v_swizzle_not_kept.log: not kept.)
    static constexpr bool DMA = DMA_OK && RBK >= 256 && RBK <= 1024 && QREG && (4 * K_BYTES + 4 * P_BYTES) <= 160 * 1024;
    static constexpr int NSTG = DMA ? 2 : 1;
    static constexpr int PIECES = K_BYTES / 1024;      // 1-KiB wave-instructions per K (or V) tile
    static constexpr int RPP = 1024 / (RBK <= 1024 ? RBK : 1024);      // rows per piece
    static constexpr int LDS = NSTG * 2 * K_BYTES + 4 * P_BYTES;
};

__device__ __forceinline__ void attn_glds16(const void* gptr, unsigned lds_base);

template <typename T, int D, bool DMA_OK>
__global__ __launch_bounds__(NTHREADS, 1) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int B, int Tn, int H, int base2) {
    using C = AttnCfg<T, D, DMA_OK>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    char* Ps = smem + C::NSTG * 2 * C::K_BYTES + wave * C::P_BYTES;

    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int64_t tok_stride = (int64_t)3 * H * D;
    const T* qbase = qkv + (int64_t)b * Tn * tok_stride + (int64_t)h * D;
    const T* kbase = qbase + (int64_t)H * D;
    const T* vbase = qbase + (int64_t)2 * H * D;

    int qrow = q0 + wave * 16 + fr;
    if (qrow >= Tn) qrow = Tn - 1;           // clamped rows are computed but never stored
    const T* qptr = qbase + (int64_t)qrow * tok_stride;

    uint4 qf[C::QREG ? C::NKS : 1];
    if constexpr (C::QREG) {
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qptr + (ks * 4 + fq) * C::PER);
    }

    f32x4_t o[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

    constexpr int CH_PER_ROW = D / C::PER;
    constexpr int TILE_CHUNKS = C::KVT * CH_PER_ROW;

    // ---- DMA form: piece p = wave + 4 i of a tile is LDS bytes [1024 p, 1024 p + 1024) = rows p RPP ...; the lane in chunk slot
    //      `slot` of row r fetches global chunk slot ^ (r & 15).  Keys beyond Tn: the last key's rows (masked to p = 0 below).
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    [[maybe_unused]] auto issue_tile = [&](int kv0, int stage) {
        constexpr int LPR = 64 / C::RPP;                      // lanes (= 16-byte chunks) per row
#pragma unroll
        for (int i = 0; i < C::PIECES / 4; ++i) {
            const int pc = __builtin_amdgcn_readfirstlane(wave) + 4 * i;
            const int row = pc * C::RPP + lane / LPR, slot = lane % LPR;
            int key = kv0 + row;
            if (key >= Tn) key = Tn - 1;
            const int64_t off = (int64_t)key * tok_stride + ((slot ^ (row & 15)) * C::PER);
            const unsigned dst = lds0 + stage * 2 * C::K_BYTES + pc * 1024;
            attn_glds16(kbase + off, dst);
            attn_glds16(vbase + off, dst + C::K_BYTES);
        }
    };
    if constexpr (C::DMA) {
        // the Q fragments are consumed here: the compiler's own s_waitcnt vmcnt(0) for them would otherwise sit in front of their first
        // use inside the loop and drain the tile in flight at every iteration
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
        issue_tile(0, 0);
    }

    for (int kv0 = 0, tix = 0; kv0 < Tn; kv0 += C::KVT, ++tix) {
        char* Ks = smem + (C::DMA ? (tix % C::NSTG) * 2 * C::K_BYTES : 0);
        char* Vs = Ks + C::K_BYTES;
        if constexpr (C::DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile tix (nothing else is in flight)
            __syncthreads();                                    // everyone's pieces landed; everyone is done with tile tix - 1
            if (kv0 + C::KVT < Tn) issue_tile(kv0 + C::KVT, (tix + 1) & 1);
        } else {
        __syncthreads();     // previous tile fully consumed
        for (int e = tid; e < TILE_CHUNKS; e += NTHREADS) {
            const int row = e / CH_PER_ROW, c = e - row * CH_PER_ROW;
            const int key = kv0 + row;
            uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (key < Tn) {
                kk = *reinterpret_cast<const uint4*>(kbase + (int64_t)key * tok_stride + c * C::PER);
                vv = *reinterpret_cast<const uint4*>(vbase + (int64_t)key * tok_stride + c * C::PER);
            }
            *reinterpret_cast<uint4*>(Ks + lds_addr<C::RBK>(row, c)) = kk;
            *reinterpret_cast<uint4*>(Vs + lds_addr<C::RBK>(row, c)) = vv;
        }
        __syncthreads();
        }
        // (register double-buffering of the K / V tiles was measured slower, 125 vs 112 us at T = 1024: several
        //  workgroups per CU already hide the load latency and the extra 16 VGPRs cost occupancy)

        // ---- S = Q K^T  (16 x KVT per wave)
        f32x4_t s[C::NNT];
#pragma unroll
        for (int j = 0; j < C::NNT; ++j) s[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
            uint4 a;
            if constexpr (C::QREG) a = qf[ks];
            else a = *reinterpret_cast<const uint4*>(qptr + (ks * 4 + fq) * C::PER);
#pragma unroll
            for (int j = 0; j < C::NNT; ++j) {
                const uint4 bb = *reinterpret_cast<const uint4*>(Ks + lds_addr<C::RBK>(j * 16 + fr, ks * 4 + fq));
                AttnMma<T>::run(a, bb, s[j]);
            }
        }
        // ---- mask + online softmax (rows 4*fq+reg, cols 16*j+fr)
        float tmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < C::NNT; ++j) {
            const bool valid = (kv0 + j * 16 + fr) < Tn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!valid) s[j][r] = -INFINITY;
                tmax[r] = fmaxf(tmax[r], s[j][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax[r] = row16_max(tmax[r]);
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mn = fmaxf(mrow[r], tmax[r]);
            alpha[r] = base2 ? ((sizeof(T) == 4) ? exp2f(mrow[r] - mn) : __builtin_amdgcn_exp2f(mrow[r] - mn))
                             : ((sizeof(T) == 4) ? expf(mrow[r] - mn) : __expf(mrow[r] - mn));
            mrow[r] = mn;
            lrow[r] *= alpha[r];
        }
#pragma unroll
        for (int j = 0; j < C::NNT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = base2 ? ((sizeof(T) == 4) ? exp2f(s[j][r] - mrow[r]) : __builtin_amdgcn_exp2f(s[j][r] - mrow[r]))
                                       : ((sizeof(T) == 4) ? expf(s[j][r] - mrow[r]) : __expf(s[j][r] - mrow[r]));
                lrow[r] += pv;
                // P scratch: row 4*fq+r, col 16*j+fr
                const int prow = fq * 4 + r, pcol = j * 16 + fr;
                T* dst = reinterpret_cast<T*>(Ps + lds_addr<C::RBP>(prow, pcol / C::PER)) + (pcol % C::PER);
                ElemTraits<T>::store(dst, pv);
            }
        }
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha[r];
        // P scratch is private to the wave: a wave-level LDS fence is enough
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();

        // ---- O += P V
#pragma unroll
        for (int ps = 0; ps < C::NPS; ++ps) {
            const uint4 pa = *reinterpret_cast<const uint4*>(Ps + lds_addr<C::RBP>(fr, ps * 4 + fq));
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                if constexpr (sizeof(T) == 2) {
                    // B[k = 8*fq + j][n = fr], j = 0..7 : two 4-row transposed reads
                    const int krow = ps * 32 + fq * 8 + (fr >> 2);
                    const int col = dt * 16 + (fr & 3) * 4;            // element column, 4 per lane
                    const int c16 = col >> 3, sub = (col & 7) * 2;     // 16-byte chunk, byte offset in it
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(Vs + lds_addr<C::RBK>(krow, c16) + sub));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(Vs + lds_addr<C::RBK>(krow + 4, c16) + sub));
                    uint4 vb;
                    vb.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
                    vb.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                    vb.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
                    vb.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
                    AttnMma<T>::run(pa, vb, o[dt]);
                } else {
                    // k = 16*ps + 4*fq + j  for MFMA j (same permutation as the P fragment)
                    const int col = dt * 16 + fr;
                    uint4 vb;
                    const int k0 = ps * 16 + fq * 4;
                    vb.x = *reinterpret_cast<const unsigned*>(Vs + lds_addr<C::RBK>(k0 + 0, col >> 2) + (col & 3) * 4);
                    vb.y = *reinterpret_cast<const unsigned*>(Vs + lds_addr<C::RBK>(k0 + 1, col >> 2) + (col & 3) * 4);
                    vb.z = *reinterpret_cast<const unsigned*>(Vs + lds_addr<C::RBK>(k0 + 2, col >> 2) + (col & 3) * 4);
                    vb.w = *reinterpret_cast<const unsigned*>(Vs + lds_addr<C::RBK>(k0 + 3, col >> 2) + (col & 3) * 4);
                    AttnMma<T>::run(pa, vb, o[dt]);
                }
            }
        }
    }

    // ---- finish: reduce row sums across the 16 lanes of each group, normalise, store
#pragma unroll
    for (int r = 0; r < 4; ++r) lrow[r] = row16_sum(lrow[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + wave * 16 + fq * 4 + r;
        if (row >= Tn) continue;
        const float inv = 1.0f / lrow[r];
        T* optr = out + (((int64_t)b * Tn + row) * H + h) * D;
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) ElemTraits<T>::store(optr + dt * 16 + fr, o[dt][r] * inv);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// bf16, head dim 64, T a multiple of 256: the shape of every attention block of the ADM-256 UNet (T = 1024 / 256, 64
// channels per head).  Everything stays in registers between the two matrix products:
//   * 512 threads = 8 waves, each wave owns 32 query rows of a 256-row query block of one (batch, head); the 64-key K / V
//     tiles are shared by the 8 waves and arrive by LDS-DMA (one 1-KiB piece of K and one of V per wave and tile, source-side
//     swizzle), double buffered, one barrier per tile;
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = the wave's Q rows, loaded once): lane (q = l & 31,
//     h = l >> 5) then holds 32 of the 64 scores of ITS query row, so the row maximum is 16 v_max3 + one half exchange
//     (v_permlane32_swap) and the exponentials / row sums are lane-local - no shuffles, no LDS;
//   * O^T = V^T P^T reuses the score registers directly as the B operand (an accumulator tile is the next MFMA's operand
//     for a product that sums over its ROW index; cdna_hip_programming.md §3): registers 8s..8s+7 of a 32-key block, cast
//     to bf16, are k-step s, in the k order key = 16 s + 8 (j >> 2) + 4 h + (j & 3).  The V^T fragments are read from the
//     row-major V tile in that same key order with two ds_read_b64_tr_b16 each;
//   * O^T has the query on the lane as well, so the online-softmax rescale and the final 1 / l are lane-local too.
// exp(x) is evaluated as exp2(x log2 e) with the multiply folded into one FMA per score.
constexpr int FQ_ROWS = 256, FKV = 64;
constexpr int ATTN_W = 8, ATTN_NST = 2, ATTN_QB = 1;   // the shipped shape: 8 waves x 32 queries (chosen by measurement, see below); T % 256 == 0
constexpr int F_SUB = 1;                               // 64-key tiles per stage (= per barrier); 2 measured slower (55.9 vs 51.7 us at T = 1024)
constexpr int F_TILE = FKV * 128;                      // one K or V tile: 64 rows x 128 B
constexpr int F_STAGE = F_SUB * F_TILE;
// Workgroup shape W (waves = 32-query blocks per workgroup; the K / V tiles are shared by them) and K / V stages NST (NST - 1 tiles
// in flight while one computes) are template parameters: (8, 2) ships; (8, 4), (4, 2), (4, 4) and a software-pipelined form were
// measured against it in round 3 (profiles/r03_summary.md §6: all within +-4 % or slower - neither DMA depth nor barrier lockstep
// nor the VALU count bounds it; the diagnostic dispatch and the pipelined kernel are in the history, commit 489a09b).


__device__ __forceinline__ int fk_swz(int row) { return (row >> 1) & 7; }                              // ds_read_b128 of K rows
__device__ __forceinline__ int fv_swz(int row) { return ((row & 3) << 1) | ((row >> 2) & 1); }          // ds_read_b64_tr_b16 of V

__device__ __forceinline__ void attn_glds16(const void* gptr, unsigned lds_base) {
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

// wait until all but the youngest `tiles` x TP of this wave's DMA pieces have landed (tiles <= 0: all of them)
template <int TP> __device__ __forceinline__ void dma_wait_tiles(int tiles) {
    if (tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TP) : "memory");
    else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TP) : "memory");
    else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Softmax bookkeeping.  Per wave and 64-key tile at 64 channels per head: 20 MFMAs (8 S^T + 8 O^T + 4 row sums) = 640 matrix cycles
// against ~95 VALU instructions - 32 v_exp_f32 at 8 issue cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), 16 v_max3,
// 16 v_cvt_pk and ~30 others at 4 - = ~500 issue cycles: neither port is saturated, and every per-score VALU instruction that was
// removed bought a few per cent at most (round 2: 11.4 VALU per MFMA, matrix pipe 28 % busy).  Per score there is now the exponential
// and half a convert, nothing else:
//   * BASE2 (log2 e folded into the q projection at pack time, nlc_attention(logit_log2 = 1)): p = 2^s needs no multiply;
//   * no per-tile maximum subtraction: softmax is invariant to the offset, and 2^s cannot overflow while the running maximum stays
//     below 2^64 (bf16 probabilities have f32's exponent range; for f16 the window is 2^+-8), so the offset `moff` stays 0
//     (wave-uniformly) until a lane's running maximum leaves [-64, 64] log2 units - only then
//     does that wave take the general path (subtract, rescale O and l) for the rest of its rows; the tile maximum itself is still
//     computed (16 v_max3), it is what detects the excursion;
//   * the row sums l come from the matrix pipe: one more MFMA per k-step with an all-ones A operand and the same P^T fragment
//     (every row of the result is sum_k P[k][q]; only register 0 is kept consistent), instead of 32 adds per tile - and they sum
//     the ROUNDED probabilities the PV product uses.
// Where a launch's time goes (round 5, in-kernel stamps: tools/attn_stamps.py, profiles/r05j_attn_stamps.log; 8 heads, T = 1024, B = 16:
// 512 workgroups = two per CU, all resident from the start): a wave lives 67-69 k cycles at an in-kernel clock of 1.80-1.84 GHz
// (s_memtime / s_memrealtime) - prologue 7 k, sixteen tiles of 2.9 k (S^T + tile maximum 1.37 k, exponentials 0.35 k, converts + O^T
// 0.63 k, DMA wait + barrier 0.55 k - the older half of the workgroup waits 0.8 k there for the younger half, the arbitration loser
// on every segment), stores 4.8 k.  Four waves per SIMD x 640 matrix cycles per tile = 2 560 of every 2 930: the tile loop keeps the
// matrix pipe ~87 % busy; over a median wave's lifetime that is 60 %, and over the whole launch 48-50 % - the two workgroups of a CU
// do not finish together (wave lifetimes 30 ... 45 us: the first-dispatched one wins the arbitration throughout, and the last third
// of the launch runs with one workgroup per CU).  (The 37-38 % of rounds 3-4 divided the same matrix cycles by a cycle count derived
// from GRBM_GUI_ACTIVE, which reads high on launches this short - the guide's DVFS note - not by the in-kernel clock.)  Tried on top
// (round 5, interleaved A/B, profiles/r05j_attn_prio.log): static priority for the second-dispatched workgroup, priority alternating
// between the two every 1 / 2 / 4 tiles - level or slower.
// QB = 32-query blocks per wave.  QB = 2 (round 4; instantiate <T, BASE2, 4, NST, 2> to measure): a wave owns 64 queries; every K
// fragment read feeds the S^T MFMAs of both blocks and every V^T fragment read the O^T MFMAs of both - half the LDS fragment reads per
// MFMA - and the two blocks are independent dependency chains inside one wave.  ~2x the registers (230 VGPRs: two waves per SIMD
// instead of four), so W = 4 waves per workgroup keeps 256 queries per workgroup.  Measured level with QB = 1 (profiles/r04_summary.md:
// 48.3 vs 47.6 us under the counters at T = 1024, 12.9 vs 12.2 us at T = 256; LDS instructions halved, time a wave waits for LDS
// 2.07 M -> 0.44 M cycles): neither the fragment reads nor the lockstep of the phases bound this kernel.  QB = 1 ships.
template <typename T, bool BASE2, int W, int NST, int QB = 1>
__global__ __launch_bounds__(W * 64, QB == 2 ? 2 : (W == 8 ? 2 : 4)) void attn_d64_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int H) {
    constexpr int F_NST = NST;
    constexpr int WG_ROWS = W * 32 * QB;                   // queries per workgroup
    constexpr int PP = 8 / W;                              // DMA pieces (8 rows x 128 B) of K, and of V, per wave and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    // XCD-aware block order: the query blocks of one (batch, head) read the same K / V, so they should run on the same
    // XCD (its L2 then serves all but the first read).  Workgroup ids are dealt round-robin over the 8 XCDs; XCD x
    // (= id & 7) takes a contiguous chunk of the [pair][query block] list (bijective for any block count).
    const int nqb = Tn / WG_ROWS, nblk = gridDim.x;
    int lin;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, cq = nblk >> 3, cr = nblk & 7;
        lin = (xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq) + idx;
    }
    const int pair = lin / nqb, qblk = lin - pair * nqb;
    const int b = pair / H, hd = pair - b * H, q0 = qblk * WG_ROWS + wave * 32 * QB;
    const int64_t tok = (int64_t)3 * H * 64;            // elements between consecutive tokens
    const T* qb = qkv + (int64_t)b * Tn * tok + (int64_t)hd * 64;
    const T* kb = qb + (int64_t)H * 64;
    const T* vb = qb + (int64_t)2 * H * 64;

    // this wave's DMA piece of every tile: rows 8 wave .. 8 wave + 7; lane -> (row, 16-byte slot), source chunk = slot ^ swizzle(row)
    const int drow = wave * 8 + (lane >> 3), dslot = lane & 7;
    const T* ksrc = kb + (int64_t)drow * tok + ((dslot ^ fk_swz(drow)) << 3);
    const T* vsrc = vb + (int64_t)drow * tok + ((dslot ^ fv_swz(drow)) << 3);
    const unsigned dma_off = (unsigned)wave * 1024u;
    auto issue = [&](int tile, int stage) {               // `tile` counts stages of F_SUB x 64 keys
#pragma unroll
        for (int u = 0; u < F_SUB; ++u) {
            const int64_t o = (int64_t)(tile * F_SUB + u) * FKV * tok;
#pragma unroll
            for (int j = 0; j < PP; ++j) {                 // rows drow + 8 W j (same swizzle: it ignores bit 5 and above of the row)
                attn_glds16(ksrc + o + (int64_t)j * 8 * W * tok, lds0 + stage * F_STAGE + u * F_TILE + dma_off + j * W * 1024);
                attn_glds16(vsrc + o + (int64_t)j * 8 * W * tok, lds0 + (F_NST + stage) * F_STAGE + u * F_TILE + dma_off + j * W * 1024);
            }
        }
    };

    // per-lane LDS offsets (within a tile): K row reads and V transposed reads
    int koff[4];                                         // key block kb2 adds 32 rows = 4096 bytes (same swizzle: (row >> 1) & 7 ignores bit 5)
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = q * 128 + (((2 * s + h) ^ fk_swz(q)) << 4);
    // V^T fragment of (16-key step ks, 32-column block db): two transposed reads, rows r0 = 16 ks + 4 h + qq (+ 8), columns
    // db * 32 + 16 cb + 4 pp .. + 3  with  cb = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3
    const int cb = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    auto voff = [&](int ks, int db, int second) {
        const int row = 16 * ks + 4 * h + qq + 8 * second;
        const int colb = (db * 32 + cb * 16 + 4 * pp) * 2;
        return row * 128 + ((((colb >> 4)) ^ fv_swz(row)) << 4) + (colb & 15);
    };
    int vo[4][2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int db = 0; db < 2; ++db) { vo[ks][db][0] = voff(ks, db, 0); vo[ks][db][1] = voff(ks, db, 1); }

    f32x16_t o0[QB], o1[QB], lacc[QB];                   // O^T blocks: d = 32 db + (reg & 3) + 8 (reg >> 2) + 4 h, query on the lane; lacc[0] = row sum
    float mrun[QB], moff[QB];                            // running maximum and current offset of this lane's queries, log2 units
#pragma unroll
    for (int u = 0; u < QB; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[u][r] = 0.f; o1[u][r] = 0.f; lacc[u][r] = 0.f; }
        mrun[u] = -1e30f; moff[u] = 0.f;
    }
    constexpr float L2E = 1.44269504088896340736f;
    const unsigned one2 = std::is_same<T, bf16_raw>::value ? 0x3F803F80u : 0x3C003C00u;      // two 1.0 in T
    const uint4 ones = make_uint4(one2, one2, one2, one2);

    const int ntiles = Tn / (FKV * F_SUB);               // >= 4 (dispatch: T % 256 == 0)
    constexpr int DEPTH = F_NST - 1;                     // tiles in flight beyond the one being computed
    constexpr int TP = 2 * PP;                           // DMA pieces per wave and tile (K and V)
    // The first K / V tiles go out BEFORE the Q loads: one memory round trip for both instead of two in a row (tools/attn_stamps.py:
    // the prologue was 7.3 k of a wave's 69 k cycles at T = 1024, 3.9 k of 19 k at T = 256).
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    // Q^T fragments (B operand): lane (q, h) holds Q[q0 + q][16 s + 8 h + j], j = 0..7
    uint4 qf[QB][4];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const T* qp = qb + (int64_t)(q0 + 32 * u + q) * tok + h * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[u][s] = *reinterpret_cast<const uint4*>(qp + s * 16);
    }
    // The compiler's own bookkeeping must see the Q loads as complete BEFORE the loop: otherwise it waits for them with counted
    // vmcnt(N) inside it, and since it cannot see the asm DMAs those counts drain the prefetch at once.  vmcnt retires in order, so
    // this wait also covers the DMA issued above (all of it: the first tile is needed now anyway; with deeper rings the later tiles
    // land a little earlier than they must).
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int st = t & (F_NST - 1);
        if (t + DEPTH < ntiles) issue(t + DEPTH, (t + DEPTH) & (F_NST - 1));      // that stage was last read during tile t - 1 (barrier passed)
#pragma unroll
      for (int u = 0; u < F_SUB; ++u) {
        const char* Kt = smem + st * F_STAGE + u * F_TILE;
        const char* Vt = smem + (F_NST + st) * F_STAGE + u * F_TILE;
        // ---- S^T = K Q^T : two 32-key blocks
        f32x16_t sA[QB], sB[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[u][r] = 0.f; sB[u][r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 ka = *reinterpret_cast<const uint4*>(Kt + koff[s]);
            const uint4 kc = *reinterpret_cast<const uint4*>(Kt + koff[s] + 4096);
#pragma unroll
            for (int u = 0; u < QB; ++u) {                  // one K fragment read, QB query blocks
                sA[u] = Mfma16<T>::run32(ka, qf[u][s], sA[u]);
                sB[u] = Mfma16<T>::run32(kc, qf[u][s], sB[u]);
            }
        }
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        f32x16_t& s0 = sA[u];
        f32x16_t& s1 = sB[u];
        float& mrun_ = mrun[u];
        float& moff_ = moff[u];
        // ---- tile maximum of the RAW scores (v_max3_f32 from inline asm: fmaxf() on an MFMA output makes hipcc insert a quieting
        // v_max(x, x) per element first).  The first maximum is plain C so that the compiler itself pads the MFMA -> VALU read
        // hazard of both accumulators; every asm statement depends on it through `tm`.
        float tm = fmaxf(s0[0], s1[0]), tm2 = fmaxf(s0[1], s1[1]);        // two chains: a v_max3 waits for the previous one of its chain
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tm) : "v"(s0[r]), "v"(s1[r]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tm2) : "v"(s0[r + 1]), "v"(s1[r + 1]));
        }
        tm = fmaxf(tm, tm2);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tm), __float_as_uint(tm), false, false);
            tm = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));      // lanes l and l ^ 32 hold the two halves of one row
        }
        const float mnew = fmaxf(mrun_, BASE2 ? tm : tm * L2E);
        // an offset is (re)chosen when the running maximum leaves [-64, 64] relative to it; on the first tile also when it lies
        // far BELOW (all logits very negative: 2^s would underflow)
        // (bf16 has f32's exponent range; an f16 probability must stay below 2^16 and well above 2^-24, so there the window is +-8)
        constexpr float WIN = std::is_same<T, bf16_raw>::value ? 64.f : 8.f;
        const float rel = mnew - moff_;
        const bool need = rel > WIN || (mrun_ < -1e29f && rel < -WIN);
        mrun_ = mnew;
        if (__builtin_amdgcn_ballot_w64(need || moff_ != 0.f) != 0) {
            // general path (rare): some query of this wave carries an offset - subtract it IN PLACE, then the common exponentials
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                const float mo = need ? mnew : moff_;
                // nothing is accumulated before the first tile; afterwards need implies mo > moff, so alpha < 1
                const float alpha = (lacc[u][0] == 0.f && o0[u][0] == 0.f) ? 1.f : __builtin_amdgcn_exp2f(moff_ - mo);
                moff_ = mo;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[u][r] *= alpha; o1[u][r] *= alpha; }
                lacc[u][0] *= alpha;
            }
            const float sub = BASE2 ? moff_ : moff_ * (1.0f / L2E);       // in the units of the raw scores
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] -= sub; s1[r] -= sub; }
        }
        // one exponential per score (BASE2: nothing else; natural logits: the log2 e multiply)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(BASE2 ? s0[r] : s0[r] * L2E);
            s1[r] = __builtin_amdgcn_exp2f(BASE2 ? s1[r] : s1[r] * L2E);
        }
      }
        // ---- O^T += V^T P^T and l += 1^T P^T : P^T k-steps come straight from the score registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 pf[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                if constexpr (std::is_same<T, bf16_raw>::value) {
                    bf16x8_t pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (__bf16)((ks < 2 ? sA[u] : sB[u])[(ks & 1) * 8 + j]);
                    pf[u] = __builtin_bit_cast(uint4, pb);
                } else {
                    f16x8_t ph;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ph[j] = (f16_raw)((ks < 2 ? sA[u] : sB[u])[(ks & 1) * 8 + j]);
                    pf[u] = __builtin_bit_cast(uint4, ph);
                }
                lacc[u] = Mfma16<T>::run32(ones, pf[u], lacc[u]);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(Vt + vo[ks][db][0]));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(Vt + vo[ks][db][1]));
                uint4 va;
                va.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
                va.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
                va.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
                va.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
#pragma unroll
                for (int u = 0; u < QB; ++u) {              // one V^T fragment, QB query blocks
                    if (db == 0) o0[u] = Mfma16<T>::run32(va, pf[u], o0[u]);
                    else o1[u] = Mfma16<T>::run32(va, pf[u], o1[u]);
                }
            }
        }
      }
        // own DMA pieces of tile t + 1 landed (the pieces of later tiles may stay in flight) ...
        dma_wait_tiles<TP>(min(DEPTH, ntiles - 1 - t) - 1);
        __syncthreads();                                       // ... and everybody's are published; tile t's stage is free
    }
    // ---- finish: normalise by the row sum (every row of lacc is the same sum), store 4 consecutive channels (8 bytes) per register quad
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const float inv = 1.0f / lacc[u][0];
        T* op = out + ((int64_t)b * Tn + q0 + 32 * u + q) * ((int64_t)H * 64) + (int64_t)hd * 64 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16_t& o = db ? o1[u] : o0[u];
                const float ov[8] = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv, 0.f, 0.f, 0.f, 0.f};
                const uint4 pk = f32_to_chunk<T>(ov);
                *reinterpret_cast<uint2*>(op + db * 32 + 8 * g) = make_uint2(pk.x, pk.y);
            }
    }
}


template <typename T, bool BASE2, int W, int NST, int QB>
int launch_d64v(const void* qkv, void* out, int B, int Tn, int H, hipStream_t st) {
    constexpr int LDS = 2 * NST * F_STAGE;
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_d64_kernel<T, BASE2, W, NST, QB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    hipLaunchKernelGGL((attn_d64_kernel<T, BASE2, W, NST, QB>), dim3((Tn / (W * 32 * QB)) * H * B), dim3(W * 64), LDS, st, (const T*)qkv, (T*)out, Tn, H);
    NLC_CHECK_LAUNCH("nlc_attention(d64)");
    return NLC_OK;
}

template <typename T, bool BASE2>
int launch_d64(const void* qkv, void* out, int B, int Tn, int H, hipStream_t st) {
    return launch_d64v<T, BASE2, ATTN_W, ATTN_NST, ATTN_QB>(qkv, out, B, Tn, H, st);
}

template <typename T, int D, bool DMA_OK>
int launch_v(const void* qkv, void* out, int B, int Tn, int H, int base2, hipStream_t st) {
    using C = AttnCfg<T, D, DMA_OK>;
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<T, D, DMA_OK>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    });
    dim3 grid(cdiv(Tn, QB), H, B);
    hipLaunchKernelGGL((attn_kernel<T, D, DMA_OK>), grid, dim3(NTHREADS), C::LDS, st, (const T*)qkv, (T*)out, B, Tn, H, base2);
    NLC_CHECK_LAUNCH("nlc_attention");
    return NLC_OK;
}
template <typename T, int D>
int launch(const void* qkv, void* out, int B, int Tn, int H, int base2, hipStream_t st) {
    if constexpr (AttnCfg<T, D, true>::DMA) {            // the double-buffered form for launches of at most two workgroups per CU (AttnCfg)
        static DeviceOnce once;
        const int ncu = once.ncu[nlc_device_once(once, [] {})];
        if ((int64_t)cdiv(Tn, QB) * H * B <= 2 * ncu) return launch_v<T, D, true>(qkv, out, B, Tn, H, base2, st);
    }
    return launch_v<T, D, false>(qkv, out, B, Tn, H, base2, st);
}

template <typename T>
int dispatch(const void* qkv, void* out, int B, int Tn, int H, int D, int base2, hipStream_t st) {
    switch (D) {
        case 32: return launch<T, 32>(qkv, out, B, Tn, H, base2, st);
        case 64: return launch<T, 64>(qkv, out, B, Tn, H, base2, st);
        case 128: return launch<T, 128>(qkv, out, B, Tn, H, base2, st);
        case 256: return launch<T, 256>(qkv, out, B, Tn, H, base2, st);
        case 512: return launch<T, 512>(qkv, out, B, Tn, H, base2, st);
        default:
            nlc_set_error("nlc_attention: head dim %d unsupported (32,64,128,256,512)", D);
            return NLC_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" int nlc_attention(const void* qkv, void* out, int B, int T, int H, int D, int dtype, int logit_log2, void* stream) {
    const int base2 = logit_log2 ? 1 : 0;
    NLC_REQUIRE(qkv && out, "nlc_attention: null pointer");
    NLC_REQUIRE(B > 0 && T > 0 && H > 0 && D > 0, "nlc_attention: bad dims");
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_attention: bad dtype %d", dtype);
    NLC_REQUIRE(H <= 65535 && B <= 65535 && (int64_t)B * H * (T / 64 + 1) < (1ll << 31), "nlc_attention: grid too large");
    // the ADM-256 shapes (64 channels per head, T = 1024 / 256): register-resident kernel above
    if (nlc_is16(dtype) && D == 64 && T % FQ_ROWS == 0) {
        if (dtype == NLC_BF16)
            return base2 ? launch_d64<bf16_raw, true>(qkv, out, B, T, H, (hipStream_t)stream) : launch_d64<bf16_raw, false>(qkv, out, B, T, H, (hipStream_t)stream);
        return base2 ? launch_d64<f16_raw, true>(qkv, out, B, T, H, (hipStream_t)stream) : launch_d64<f16_raw, false>(qkv, out, B, T, H, (hipStream_t)stream);
    }
    if (dtype == NLC_BF16) return dispatch<bf16_raw>(qkv, out, B, T, H, D, base2, (hipStream_t)stream);
    if (dtype == NLC_F16) return dispatch<f16_raw>(qkv, out, B, T, H, D, base2, (hipStream_t)stream);
    return dispatch<float>(qkv, out, B, T, H, D, base2, (hipStream_t)stream);
}
