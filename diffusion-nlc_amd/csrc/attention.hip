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

template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_raw> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct AttnMma<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

template <typename T, int D> struct AttnCfg {
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
    static constexpr int LDS = 2 * K_BYTES + 4 * P_BYTES;
};

template <typename T, int D>
__global__ __launch_bounds__(NTHREADS, 1) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int B, int Tn, int H) {
    using C = AttnCfg<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + C::K_BYTES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    char* Ps = smem + 2 * C::K_BYTES + wave * C::P_BYTES;

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

    for (int kv0 = 0; kv0 < Tn; kv0 += C::KVT) {
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
            alpha[r] = (sizeof(T) == 4) ? expf(mrow[r] - mn) : __expf(mrow[r] - mn);
            mrow[r] = mn;
            lrow[r] *= alpha[r];
        }
#pragma unroll
        for (int j = 0; j < C::NNT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (sizeof(T) == 4) ? expf(s[j][r] - mrow[r]) : __expf(s[j][r] - mrow[r]);
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

template <typename T, int D>
int launch(const void* qkv, void* out, int B, int Tn, int H, hipStream_t st) {
    using C = AttnCfg<T, D>;
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    });
    dim3 grid(cdiv(Tn, QB), H, B);
    hipLaunchKernelGGL((attn_kernel<T, D>), grid, dim3(NTHREADS), C::LDS, st, (const T*)qkv, (T*)out, B, Tn, H);
    NLC_CHECK_LAUNCH("nlc_attention");
    return NLC_OK;
}

template <typename T>
int dispatch(const void* qkv, void* out, int B, int Tn, int H, int D, hipStream_t st) {
    switch (D) {
        case 32: return launch<T, 32>(qkv, out, B, Tn, H, st);
        case 64: return launch<T, 64>(qkv, out, B, Tn, H, st);
        case 128: return launch<T, 128>(qkv, out, B, Tn, H, st);
        case 256: return launch<T, 256>(qkv, out, B, Tn, H, st);
        case 512: return launch<T, 512>(qkv, out, B, Tn, H, st);
        default:
            nlc_set_error("nlc_attention: head dim %d unsupported (32,64,128,256,512)", D);
            return NLC_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" int nlc_attention(const void* qkv, void* out, int B, int T, int H, int D, int dtype, void* stream) {
    NLC_REQUIRE(qkv && out, "nlc_attention: null pointer");
    NLC_REQUIRE(B > 0 && T > 0 && H > 0 && D > 0, "nlc_attention: bad dims");
    NLC_REQUIRE(dtype == NLC_F32 || dtype == NLC_BF16, "nlc_attention: bad dtype %d", dtype);
    NLC_REQUIRE(H <= 65535 && B <= 65535, "nlc_attention: grid too large");
    if (dtype == NLC_BF16) return dispatch<bf16_raw>(qkv, out, B, T, H, D, (hipStream_t)stream);
    return dispatch<float>(qkv, out, B, T, H, D, (hipStream_t)stream);
}
