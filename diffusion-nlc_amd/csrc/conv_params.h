// Tile geometry and kernel parameter block shared by the generic (conv_igemm.hip) and the fast
// (conv_fast.hip) implicit-GEMM convolution kernels.
#pragma once
#include "common.h"
#include <type_traits>

constexpr int BM = 128;                              // output pixels per tile
constexpr int BN = 128;                              // output channels per tile
constexpr int KB_BYTES = 128;                        // bytes of K per row per k-step
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM + BN) * KB_BYTES;    // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // 64 KiB -> 2 workgroups per CU

// Division of a 31-bit index by a launch-invariant divisor: q = n / d as one 32 x 32 -> high-32 multiply and a shift
// (Granlund-Montgomery, N = 31: l = ceil(log2 d), mul = ceil(2^(31 + l) / d) < 2^32, q = mulhi(n, mul) >> (l - 1); d = 1: q = n).
// The constants are computed on the host (fill_params); a runtime division costs ~40 VALU each on this chip and the tile prologues
// did eight of them per lane.
struct FastDiv {
    unsigned mul; int shr;      // shr < 0: d == 1
    __host__ static FastDiv make(int d) {
        FastDiv f{0u, -1};
        if (d <= 1) return f;
        int l = 0;
        while ((1ll << l) < d) ++l;
        const unsigned long long num = 1ull << (31 + l);
        f.mul = (unsigned)((num + (unsigned)d - 1) / (unsigned)d);
        f.shr = l - 1;
        return f;
    }
    __device__ __forceinline__ int div(int n) const { return shr < 0 ? n : (int)(__umulhi((unsigned)n, mul) >> shr); }
};

struct KParams {
    const char* x0; const char* x1;
    int C0, C1, Ctot;
    int B, Hin, Win, Hout, Wout, Cout;
    int KH, KW, stride, pad_t, pad_l, ups;
    const char* w; int Cin_pad, Cout_pad;
    const float* bias; const float* emb; int emb_stride;
    const char* res; float out_scale; int act;
    char* out; int out_mode;
    int M, MT, NT;
    int ksplit;         // conv_fast only: >1 = split the channel blocks over blockIdx.y, raw f32 partials to `partial`
    float* partial;     // split-K partial sums + arrival counters (caller workspace); reduced by the last-arriving workgroup of a tile
    long long* stats;   // conv_halo / conv_fast (16-bit, NHWC out, Cout % 128 == 0) or NULL: [B][Cout/gran][4] 64-bit accumulators of the
                        //   (sum, sum of squares) of the STORED (rounded) outputs per chunk (Stat16: hi / lo limbs), added to atomically
    int policy;         // NLC_CONV_* kernel-selection policy of this call (nlc_conv_desc.policy)
    int tuning;         // A/B switches (nlc_conv_desc.tuning)
    const float* gn_coef; int gn_act;   // conv_halo (bf16) only: input = act(a x + b) applied in LDS (nlc_conv_desc.gn_coef)
    char* norm_out;     // conv_pwr only (nlc_conv_desc.norm_out): also write act(a x + b) of the input, [M][C0 + C1]; the conv itself uses x
    int res_ups;        // res is [B][Hout/2][Wout/2][Cout]: output pixel (y, x) adds res pixel (y >> 1, x >> 1) (nlc_conv_desc.res_upsample2x)
    int math;           // NLC_MATH_* (nlc_conv_desc.math): f32 tensors only; F16X3 = weights packed as (hi, lo) f16 halves
    const float* w_scale; // F16X3: [Cout_pad] power-of-two factor of every output channel's conv sum (nlc_conv_desc.w_scale), else NULL
    int stats_gran;     // channels per chunk of `stats` (nlc_conv_desc.stats_granule): 8, or 4 for consumers whose groups are 4 channels wide
    FastDiv div_hwo, div_wo;    // by Hout * Wout and by Wout (output row index -> image, y, x)
};

// ---- ride-along GroupNorm statistics of a lane's 16 consecutive output channels: (sum, sum of squares) of the STORED values per
//      4-channel granule, accumulated over the pixels the lane stores; emitted per 8 channels (granule pairs added) or per 4.
struct Stat16 {
    float s[4], q[4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] = 0.f; q[k] = 0.f; }
    }
    __device__ __forceinline__ void add8(int half, const float (&sv)[8]) {      // channels 8 half .. 8 half + 7
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[2 * half + (k >> 2)] += sv[k]; q[2 * half + (k >> 2)] = fmaf(sv[k], sv[k], q[2 * half + (k >> 2)]); }
    }
    // the same from the packed 16-byte chunk as stored (16-bit types): v_dot2c_f32_{bf16,f16} adds both halves of a dword in one
    // instruction - sum = dot2(d, (1, 1)), sum of squares = dot2(d, d) - 16 VALU per 16 channels instead of 48 (unpack, add, fma)
    template <typename T> __device__ __forceinline__ void add_chunk(int half, const uint4& pk) {
        static_assert(sizeof(T) == 2, "16-bit storage only");
        const unsigned d[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = 2 * half + (k >> 1);
            if constexpr (std::is_same<T, bf16_raw>::value) {
                typedef __attribute__((ext_vector_type(2))) __bf16 v2;
                const v2 x = __builtin_bit_cast(v2, d[k]);
                s[g] = __builtin_amdgcn_fdot2_f32_bf16(x, __builtin_bit_cast(v2, 0x3f803f80u), s[g], false);
                q[g] = __builtin_amdgcn_fdot2_f32_bf16(x, x, q[g], false);
            } else {
                typedef __attribute__((ext_vector_type(2))) _Float16 v2;
                const v2 x = __builtin_bit_cast(v2, d[k]);
                s[g] = __builtin_amdgcn_fdot2(x, __builtin_bit_cast(v2, 0x3c003c00u), s[g], false);
                q[g] = __builtin_amdgcn_fdot2(x, x, q[g], false);
            }
        }
    }
    // ---- totals (include/nlc_hip.h, "GroupNorm statistics ride along"): every (image, chunk) owns four 64-bit accumulators
    //      (sum.hi, sum.lo, sumsq.hi, sumsq.lo) that ALL contributions of a launch are added into with integer atomics.  A
    //      contribution v (an f32 partial sum, exact in f64) is split into hi = floor(v) and lo = floor((v - hi) * 2^44): integer
    //      addition is associative, so the totals do not depend on the order in which workgroups arrive - bit-reproducible without a
    //      partials array, a fixed-order reduction pass or a finalize launch.  Resolution 2^-44 per contribution (5.7e-14), range
    //      +-2^63; the consumer (groupnorm.hip: gn_channel_coefs) reads value = hi + lo * 2^-44 in f64.
    //      A contribution that is not finite or not below 2^45 in magnitude has no limbs (the conversion would be undefined): it sets
    //      bit 62 of the chunk's sumsq.hi word with an atomic OR instead.  Legitimate sumsq.hi totals stay below 2^61 (< 2^16
    //      contributions of < 2^45 each, never negative), so no add can carry into that bit and no add can clear it: the mark is
    //      sticky and order-independent like the sums, and the consumer turns a marked chunk's group into NaN (mean, rstd) - what
    //      F.group_norm gives for a group that holds an inf / NaN.
    static constexpr double LIMB = 17592186044416.0;           // 2^44
    static constexpr float LIMB_DOMAIN = 35184372088832.f;     // 2^45
    static constexpr unsigned long long POISON = 1ull << 62;
    static __device__ __forceinline__ bool in_domain(float v) { return fabsf(v) < LIMB_DOMAIN; }      // false for NaN too
    static __device__ __forceinline__ void limbs(float v, long long& hi, long long& lo) {
        const double d = (double)v, fl = floor(d);
        hi = (long long)fl;
        lo = (long long)((d - fl) * LIMB);
    }
    // `word`: the accumulator a lane was about to add a limb of v to, index within its chunk = (word - base) & 3
    static __device__ __forceinline__ void poison(long long* chunk_sumsq_hi) {
        atomicOr(reinterpret_cast<unsigned long long*>(chunk_sumsq_hi), POISON);
    }
    // After a 16-lane all-reduce (every lane of the row holds the row's sums for its 16 channels from n): lane `fr` of the row adds
    // ONE limb - its index within the slice's 16 (per-4 granules) or 8 (per-8 chunks) accumulators - so that a wave issues a single
    // atomic instruction per tile.
    __device__ __forceinline__ void emit_row(long long* tot, int b, int Cout, int n, int gran, int fr) const {
        const float v8[8] = {s[0] + s[1], q[0] + q[1], s[2] + s[3], q[2] + q[3], 0.f, 0.f, 0.f, 0.f};
        const float v4[8] = {s[0], q[0], s[1], q[1], s[2], q[2], s[3], q[3]};
        const int vi = fr >> 1;
        float v = gran == 4 ? v4[0] : v8[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = (vi == k) ? (gran == 4 ? v4[k] : v8[k]) : v;
        if (gran != 4 && fr >= 8) return;
        long long* dst = tot + (gran == 4 ? ((int64_t)b * (Cout >> 2) + (n >> 2)) : ((int64_t)b * (Cout >> 3) + (n >> 3))) * 4 + fr;
        if (!in_domain(v)) { poison(dst - (fr & 3) + 2); return; }
        long long hi, lo;
        limbs(v, hi, lo);
        atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)((fr & 1) ? lo : hi));
    }
    // The same for a lane that holds 8 channels from n (in s[0..1], q[0..1]: add_chunk(0, .) only): 8 (per-4 granules) or 4 (one per-8
    // chunk) accumulators per row.
    __device__ __forceinline__ void emit_row8(long long* tot, int b, int Cout, int n, int gran, int fr) const {
        const float v8[2] = {s[0] + s[1], q[0] + q[1]};
        const float v4[4] = {s[0], q[0], s[1], q[1]};
        if (fr >= (gran == 4 ? 8 : 4)) return;
        const int vi = fr >> 1;
        float v = gran == 4 ? v4[0] : v8[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) v = (vi == k) ? (gran == 4 ? v4[k] : v8[k & 1]) : v;
        long long* dst = tot + (gran == 4 ? ((int64_t)b * (Cout >> 2) + (n >> 2)) : ((int64_t)b * (Cout >> 3) + (n >> 3))) * 4 + fr;
        if (!in_domain(v)) { poison(dst - (fr & 3) + 2); return; }
        long long hi, lo;
        limbs(v, hi, lo);
        atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)((fr & 1) ? lo : hi));
    }
    // one lane alone adds everything it holds (launches whose 16-pixel rows may straddle two images: never taken by the networks)
    __device__ __forceinline__ void emit_lane(long long* tot, int b, int Cout, int n, int gran) const {
        const float v8[4] = {s[0] + s[1], q[0] + q[1], s[2] + s[3], q[2] + q[3]};
        const float v4[8] = {s[0], q[0], s[1], q[1], s[2], q[2], s[3], q[3]};
        long long* dst = tot + (gran == 4 ? ((int64_t)b * (Cout >> 2) + (n >> 2)) : ((int64_t)b * (Cout >> 3) + (n >> 3))) * 4;
        const int nv = gran == 4 ? 8 : 4;
        for (int k = 0; k < nv; ++k) {
            const float v = gran == 4 ? v4[k] : v8[k];
            if (!in_domain(v)) { poison(dst + (k >> 1) * 4 + 2); continue; }
            long long hi, lo;
            limbs(v, hi, lo);
            atomicAdd(reinterpret_cast<unsigned long long*>(dst + 2 * k), (unsigned long long)hi);
            atomicAdd(reinterpret_cast<unsigned long long*>(dst + 2 * k + 1), (unsigned long long)lo);
        }
    }
};

// (mean, rstd) of group g of image b from the totals that rode along with the producing convolutions (P::tot0 / tot1, granule shifts tsh0 / tsh1, C0, C1, gs, invN (double), eps): the
// group's chunks - over both sources of a concatenated input; a group is a whole number of chunks of either source (host-checked) -
// added in f64 (value = hi + lo * 2^-44: exact integers, so no order dependence upstream), var = E[x^2] - mean^2 in f64.
// Every apply thread evaluates this for the one or two groups its channels lie in: 1 ... 8 32-byte loads - there is no partials
// array, no reduction pass and no finalize launch between a convolution and the normalisation that follows it.
template <typename P>
__device__ __forceinline__ void gn_group_from_totals_t(const P& p, int b, int g, float& mean, float& rstd) {
    double S = 0.0, Q = 0.0;
    long long marks = 0;             // bit 62 of a chunk's sumsq.hi word: a contribution was inf / NaN / out of the limbs' domain (Stat16::poison)
    const int ch = g * p.gs, ch1 = ch + p.gs;
    auto add = [&](const long long* tot, int nq, int first_chunk, int nchunk) {      // the group's chunks of one source are contiguous
        const longlong2* t = reinterpret_cast<const longlong2*>(tot + ((int64_t)b * nq + first_chunk) * 4);
        for (int k = 0; k < nchunk; ++k) {
            const longlong2 ts = t[2 * k], tq = t[2 * k + 1];
            S += (double)ts.x + (double)ts.y * (1.0 / Stat16::LIMB);
            Q += (double)tq.x + (double)tq.y * (1.0 / Stat16::LIMB);
            marks |= tq.x;
        }
    };
    const int e0 = min(ch1, p.C0), s1 = max(ch, p.C0);
    if (ch < e0) add(p.tot0, p.C0 >> p.tsh0, ch >> p.tsh0, (e0 - ch) >> p.tsh0);
    if (s1 < ch1) add(p.tot1, p.C1 >> p.tsh1, (s1 - p.C0) >> p.tsh1, (ch1 - s1) >> p.tsh1);
    const double mean_d = S * p.invN;
    double var = Q * p.invN - mean_d * mean_d;          // f64: E[x^2] - mean^2 keeps ~9 digits after the cancellation
    if (var < 0.0) var = 0.0;
    mean = (float)mean_d;
    rstd = rsqrtf((float)(var + (double)p.eps));         // (the totals describe 16-bit tensors: a 1-ulp f32 reciprocal square root is ample)
    if (marks & (long long)Stat16::POISON) mean = rstd = __builtin_nanf("");
}


// ---- NLC_MATH_F16X3: an f32 operand as two f16 halves, x ~= hi + lo with hi = f16(x) (round to nearest), lo = f16(x - hi); the
//      residual x - hi is exact in f32, so hi + lo carries 22 significand bits of x (lo falls into f16's subnormal range, absolute
//      precision 2^-25, for |x| < 2^-2).  Four f32 values -> (two packed hi words, two packed lo words).
__device__ __forceinline__ void f16x3_split4(const uint4& x, uint2& hi, uint2& lo) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const float f0 = __uint_as_float(x.x), f1 = __uint_as_float(x.y), f2 = __uint_as_float(x.z), f3 = __uint_as_float(x.w);
    const h2 h01 = {(_Float16)f0, (_Float16)f1}, h23 = {(_Float16)f2, (_Float16)f3};
    const h2 l01 = {(_Float16)(f0 - (float)h01[0]), (_Float16)(f1 - (float)h01[1])};
    const h2 l23 = {(_Float16)(f2 - (float)h23[0]), (_Float16)(f3 - (float)h23[1])};
    hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
// The X3 kernels run with MODE.FP16_OVFL = 1 (hwreg(HW_REG_MODE, 23, 1)): an f32 -> f16 conversion that overflows gives +-65504
// instead of +-inf, so an activation outside the mode's domain (|x| >= 65504, include/nlc_hip.h) saturates - hi = +-65504, lo = the
// clamped remainder - and the convolution sum stays finite; true infinities / NaNs of the input still propagate.  One scalar
// instruction at kernel entry; nothing else in those kernels converts to f16.
__device__ __forceinline__ void f16x3_enter() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
__device__ __forceinline__ void mfma_f16(const uint4& a, const uint4& b, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// row (pixel index) of the residual tensor that output pixel (y, x) of image tb / output row m adds
__device__ __forceinline__ int64_t res_row(const KParams& p, int tb, int y, int x) {
    return p.res_ups ? ((int64_t)tb * (p.Hout >> 1) + (y >> 1)) * (p.Wout >> 1) + (x >> 1) : ((int64_t)tb * p.Hout + y) * p.Wout + x;
}
__device__ __forceinline__ int64_t res_row_m(const KParams& p, int64_t m) {
    if (!p.res_ups) return m;
    const int HWo = p.Hout * p.Wout;
    const int b = p.div_hwo.div((int)m), rem = (int)(m - (int64_t)b * HWo);           // (m < 2^31: dispatch)
    const int y = p.div_wo.div(rem), x = rem - y * p.Wout;
    return ((int64_t)b * (p.Hout >> 1) + (y >> 1)) * (p.Wout >> 1) + (x >> 1);
}

// conv_fast.hip: NLC_OK, NLC_ELAUNCH, or NLC_EUNSUPPORTED (shape not handled -> use the generic kernel)
int nlc_conv_fast_dispatch(const KParams& p, int dtype, hipStream_t stream);
// split-K policy for the shapes conv_fast takes (few output tiles, long K): number of splits, 1 = none
int nlc_conv_fast_ksplit(const KParams& p, int dtype);
int64_t nlc_conv_fast_split_bytes(const KParams& p, int ks);     // workspace bytes of a conv_fast launch split ks ways
// conv_halo.hip: split-K factor of the halo kernel for this launch (1 = none): fewer tiles than CUs, long K, bf16, Cout % 128 == 0
int nlc_conv_halo_ksplit(const KParams& p, int dtype);
int nlc_conv_halo_plain_ok(const KParams& p, int dtype);          // the un-split halo kernel would take this launch
// conv_fast.hip: same for the fast path (0 when K would be split or a tile could straddle two images)
int nlc_conv_fast_stats_partials(const KParams& p, int dtype);
// conv_halo.hip: partials per image the halo kernel would emit GroupNorm statistics with for this launch (0: it would not)
int nlc_conv_halo_stats_partials(const KParams& p, int dtype);
// conv_pw.hip: persistent pointwise (1x1) kernel for launches with many tiles; same return convention as the dispatchers below
int nlc_conv_pw_ok(const KParams& p, int dtype);
int nlc_conv_pw_dispatch(const KParams& p, int dtype, hipStream_t stream);
int nlc_conv_pw_norm_ok(const KParams& p, int dtype);    // 1 if the launch can write nlc_conv_desc.norm_out
// conv_narrow.hip: 3x3 with at most 16 output channels (the networks' last layer)
int nlc_conv_narrow_ok(const KParams& p, int dtype);
int nlc_conv_narrow_dispatch(const KParams& p, int dtype, hipStream_t stream);
// conv_halo.hip: 1 if the halo kernel would take this launch AND can apply a GroupNorm prologue (bf16)
int nlc_conv_halo_prologue_ok(const KParams& p, int dtype);
// conv_halo.hip: 3x3 with the input halo resident in LDS; same return convention
int nlc_conv_halo_dispatch(const KParams& p, int dtype, hipStream_t stream);
