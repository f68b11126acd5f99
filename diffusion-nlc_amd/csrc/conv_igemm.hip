// Implicit-GEMM convolution / linear on MFMA for gfx950.
//
//   out[m][n] = act( (sum_{tap,c} in[pixel(m)+tap][c] * w[n][tap][c] + bias[n] + emb[b(m)][n]
//                     + res[m][n]) * out_scale )
//
// M = B*Hout*Wout output pixels (rows), N = Cout, K = KH*KW*Cin.  Activations are NHWC so
// the K dimension (channels of one tap) is contiguous for both operands; no im2col buffer
// is ever formed: the A tile rows are gathered straight from the input image, zero-filled
// outside the image (padding), optionally through a virtual nearest 2x upsample and across
// the channel-concatenation of two tensors.
//
// Tile: 128 (pixels) x 128 (cout) per 256-thread workgroup, 4 waves as 2x2, each wave
// 64x64 = 4x4 MFMA tiles of 16x16.  K advances 128 bytes per row per step (64 bf16 / 32 f32),
// double-buffered in LDS (64 KiB -> 2 workgroups per CU), global->register->LDS staging with
// the next step's loads in flight under the current step's MFMAs.  LDS rows are 128 B with
// the 16-byte chunk index XOR-swizzled by (row>>1)&7 so every ds_read_b128 lane group hits
// 16 distinct 16-byte slots (conflict-free).
//
//   bf16: v_mfma_f32_16x16x32_bf16, f32 accumulate.
//   f32 : v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) - 4 MFMAs per 16-byte fragment pair,
//         with the same k permutation on both operands.
#include "common.h"
#include "conv_params.h"
#include "conv_small.h"
#include <cstring>
#include <stdlib.h>

namespace {

template <typename T> using Mma = Mfma16<T>;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * KB_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// X3 (T = float, NLC_MATH_F16X3 weights): this generic kernel only has to UNDERSTAND the (hi, lo) weight packing - it rebuilds
// w = hi + lo (the 22-bit operand the split-f16 kernels multiply with) and runs the exact f32 MFMAs on it.
template <typename T, bool X3 = false>
__global__ __launch_bounds__(NTHREADS, 2) void conv_igemm_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = ElemTraits<T>::kPerChunk;
    constexpr int KBE = Mma<T>::KBE;
    constexpr int ES = (int)sizeof(T);

    // ---- XCD-aware block -> tile mapping: blocks b and b+8 share an XCD (own L2); give each
    //      XCD a contiguous run of tiles so the input rows its tiles share stay in that L2.
    const int nblk = p.MT * p.NT;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / p.NT, nt = bid - mt * p.NT;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lc = tid & 7;        // 16-byte chunk within the 128-byte k-block row
    const int lr = tid >> 3;       // 0..31, rows lr + 32*i

    const int HWo = p.Hout * p.Wout;
    const int HL = p.ups ? 2 * p.Hin : p.Hin;
    const int WL = p.ups ? 2 * p.Win : p.Win;
    const int ntaps = p.KH * p.KW;
    const int ncb = p.Cin_pad / KBE;
    const int nk = ntaps * ncb;

    // ---- per-thread gather rows (4 A rows, 4 B rows)
    int64_t a_img[4]; int a_iy0[4], a_ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lr + 32 * i;
        if (m < p.M) {
            const int b = m / HWo; const int rem = m - b * HWo;
            const int oy = rem / p.Wout; const int ox = rem - oy * p.Wout;
            a_img[i] = (int64_t)b * p.Hin * p.Win;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
        } else {
            a_img[i] = 0; a_iy0[i] = -(1 << 28); a_ix0[i] = -(1 << 28);
        }
    }
    const int64_t wrow = (int64_t)ntaps * p.Cin_pad * ES;
    const char* b_row0 = p.w + (int64_t)(n0 + lr) * wrow + lc * PER * ES;
    const char* b_row1 = b_row0 + 32 * wrow;
    const char* b_row2 = b_row0 + 64 * wrow;
    const char* b_row3 = b_row0 + 96 * wrow;

    // staging registers are named scalars (not arrays): keeps them out of scratch
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NLC_LOAD_A(i, dst)                                                                              \
    {                                                                                                   \
        int iy = a_iy0[i] + r, ix = a_ix0[i] + s;                                                       \
        const bool ok = cvalid && iy >= 0 && iy < HL && ix >= 0 && ix < WL;                             \
        if (p.ups) { iy >>= 1; ix >>= 1; }                                                              \
        const char* ptr = ok ? src + ((a_img[i] + (int64_t)iy * p.Win + ix) * C + ch) * ES : p.x0;      \
        uint4 v = *reinterpret_cast<const uint4*>(ptr);                                                 \
        v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;             \
        dst = v;                                                                                        \
    }
#define NLC_LOAD_REGS(kt_)                                                                              \
    {                                                                                                   \
        const int tap = (kt_) / ncb, cb = (kt_) - tap * ncb;                                            \
        const int r = tap / p.KW, s = tap - r * p.KW;                                                   \
        const int cch = cb * KBE + lc * PER; /* logical (concatenated) channel */                       \
        const char* src; int C, ch;                                                                     \
        if (cch < p.C0) { src = p.x0; C = p.C0; ch = cch; }                                             \
        else { src = p.x1; C = p.C1; ch = cch - p.C0; }                                                 \
        const bool cvalid = cch < p.Ctot;                                                               \
        if (!cvalid) { src = p.x0; C = p.C0; ch = 0; }                                                  \
        NLC_LOAD_A(0, ra0) NLC_LOAD_A(1, ra1) NLC_LOAD_A(2, ra2) NLC_LOAD_A(3, ra3)                     \
        const int64_t boff = ((int64_t)tap * p.Cin_pad + cb * KBE) * ES;                                \
        rb0 = *reinterpret_cast<const uint4*>(b_row0 + boff);                                           \
        rb1 = *reinterpret_cast<const uint4*>(b_row1 + boff);                                           \
        rb2 = *reinterpret_cast<const uint4*>(b_row2 + boff);                                           \
        rb3 = *reinterpret_cast<const uint4*>(b_row3 + boff);                                           \
    }
#define NLC_STORE_REGS(stage_)                                                                          \
    {                                                                                                   \
        char* As_ = smem + (stage_) * STAGE_BYTES;                                                      \
        char* Bs_ = As_ + BM * KB_BYTES;                                                                \
        *reinterpret_cast<uint4*>(As_ + lds_off(lr, lc)) = ra0;                                         \
        *reinterpret_cast<uint4*>(As_ + lds_off(lr + 32, lc)) = ra1;                                    \
        *reinterpret_cast<uint4*>(As_ + lds_off(lr + 64, lc)) = ra2;                                    \
        *reinterpret_cast<uint4*>(As_ + lds_off(lr + 96, lc)) = ra3;                                    \
        *reinterpret_cast<uint4*>(Bs_ + lds_off(lr, lc)) = rb0;                                         \
        *reinterpret_cast<uint4*>(Bs_ + lds_off(lr + 32, lc)) = rb1;                                    \
        *reinterpret_cast<uint4*>(Bs_ + lds_off(lr + 64, lc)) = rb2;                                    \
        *reinterpret_cast<uint4*>(Bs_ + lds_off(lr + 96, lc)) = rb3;                                    \
    }

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // X3: the packed weights of output row n carry the factor 2^e[n] (nlc_pack_conv_weights_ex); multiplying the rebuilt weight by
    // w_scale[n] = 2^-e[n] is exact, so this kernel's sum is the unscaled one
    float wsj[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (X3 && std::is_same<T, float>::value) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wsj[j] = p.w_scale[n0 + wn * 64 + j * 16 + fr];          // < Cout_pad: in range
    }
    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE_BYTES;
        const char* Bs = As + BM * KB_BYTES;
        if constexpr (X3 && std::is_same<T, float>::value) {
            uint4 fa0[4], fa1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa0[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, fq));
                fa1[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x8_t wh = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, fq)));
                const f16x8_t wl = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, 4 + fq)));
                float wv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) wv[e] = ((float)wh[e] + (float)wl[e]) * wsj[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float av[8] = {__uint_as_float(fa0[i].x), __uint_as_float(fa0[i].y), __uint_as_float(fa0[i].z), __uint_as_float(fa0[i].w),
                                         __uint_as_float(fa1[i].x), __uint_as_float(fa1[i].y), __uint_as_float(fa1[i].z), __uint_as_float(fa1[i].w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], wv[e], acc[i][j], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 fa[4], fb[4];
            const int chunk = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 16 + fr, chunk));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * 64 + j * 16 + fr, chunk));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    NLC_LOAD_REGS(0)
    NLC_STORE_REGS(0)
    __syncthreads();
    // steady state: next step's global loads are in flight under this step's MFMAs; the last
    // step is peeled so the loop body has no conditional around the loads.
    for (int kt = 0; kt < nk - 1; ++kt) {
        NLC_LOAD_REGS(kt + 1)
        compute(kt & 1);
        NLC_STORE_REGS((kt & 1) ^ 1)
        __syncthreads();
    }
    compute((nk - 1) & 1);

    // ---- epilogue: C[row = i*16 + fq*4 + reg][col = j*16 + fr]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wm * 64 + i * 16 + fq * 4 + reg;
            if (m >= p.M) continue;
            const int b = m / HWo;
            const int rem = m - b * HWo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fr;
                if (n >= p.Cout) continue;
                float v = acc[i][j][reg];
                if (p.bias) v += p.bias[n];
                if (p.emb) v += p.emb[(int64_t)b * p.emb_stride + n];
                if (p.res) v += ElemTraits<T>::load(reinterpret_cast<const T*>(p.res) + res_row_m(p, m) * p.Cout + n);
                v *= p.out_scale;
                v = apply_act(v, p.act);
                if (p.out_mode == NLC_OUT_NHWC)
                    ElemTraits<T>::store(reinterpret_cast<T*>(p.out) + (int64_t)m * p.Cout + n, v);
                else
                    reinterpret_cast<float*>(p.out)[((int64_t)b * p.Cout + n) * HWo + rem] = v;
            }
        }
    }
}

template <typename T, bool X3 = false>
int launch(const KParams& p, hipStream_t stream) {
    static DeviceOnce once;
    (void)nlc_device_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, X3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    hipLaunchKernelGGL((conv_igemm_kernel<T, X3>), dim3(p.MT * p.NT), dim3(NTHREADS), LDS_BYTES, stream, p);
    NLC_CHECK_LAUNCH("nlc_conv2d");
    return NLC_OK;
}

}  // namespace

extern "C" int nlc_conv_pack_dims(int dtype, int* cout_mult, int* cin_mult) {
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_conv_pack_dims: bad dtype %d", dtype);
    if (cout_mult) *cout_mult = BN;
    if (cin_mult) *cin_mult = nlc_is16(dtype) ? Mma<bf16_raw>::KBE : Mma<float>::KBE;
    return NLC_OK;
}

// ONE function fills the kernel parameter block from the descriptor, for the three host queries and for nlc_conv2d alike, so that
// a query predicts the dispatch from exactly the state the launch sees (Cout_pad, residual, math mode, ... included).
static void fill_params(const nlc_conv_desc* d, KParams& p) {
    p.x0 = (const char*)d->x0; p.x1 = (const char*)d->x1;
    p.C0 = d->C0; p.C1 = d->C1; p.Ctot = d->C0 + d->C1;
    p.B = d->B; p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.ups = d->upsample2x ? 1 : 0;
    p.w = (const char*)d->w; p.Cin_pad = d->Cin_pad; p.Cout_pad = d->Cout_pad;
    p.bias = d->bias; p.emb = d->emb; p.emb_stride = d->emb_stride;
    p.res = (const char*)d->res; p.out_scale = d->out_scale; p.act = d->act; p.res_ups = d->res_upsample2x ? 1 : 0;
    p.out = (char*)d->out; p.out_mode = d->out_mode;
    const int64_t M64 = (int64_t)d->B * d->Hout * d->Wout;
    p.M = (int)M64; p.MT = cdiv(M64, BM); p.NT = cdiv(d->Cout, BN);
    p.ksplit = 1; p.partial = nullptr;
    p.stats = nullptr;
    p.gn_coef = d->gn_coef; p.gn_act = d->gn_act; p.math = d->math; p.w_scale = d->w_scale; p.norm_out = (char*)d->norm_out;
    p.policy = d->policy; p.tuning = d->tuning;
    p.stats_gran = d->stats_granule == 4 ? 4 : 8;
    p.div_hwo = FastDiv::make(d->Hout * d->Wout); p.div_wo = FastDiv::make(d->Wout);
}

static bool desc_sane(const nlc_conv_desc* d, int dtype) {
    return d && nlc_dtype_ok(dtype) && d->B > 0 && d->Hout > 0 && d->Wout > 0 && d->Cout > 0 && d->policy != NLC_CONV_GENERIC;
}

// The small-map 3x3 kernel (conv_small.hip) takes a launch when the caller asks for it - a GroupNorm of the input to apply on the way
// into LDS (nlc_conv_desc.gn_in) or policy NLC_CONV_FORCE_SMALL - and the geometry fits; tuning bit 21 = never (A/B).
static bool small_wanted(const nlc_conv_desc* d, const KParams& p, int dtype, SmallGeom& g) {
    if (!(d->gn_in || d->policy == NLC_CONV_FORCE_SMALL) || (d->tuning & (1 << 21))) return false;
    return nlc_conv_small_geom(p, dtype, g);
}

extern "C" int nlc_conv2d_gn_in_supported(const nlc_conv_desc* d, int dtype) {
    if (!desc_sane(d, dtype) || (d->tuning & (1 << 21))) return 0;
    KParams p{};
    fill_params(d, p);
    SmallGeom g;
    return nlc_conv_small_geom(p, dtype, g) ? 1 : 0;
}

// nlc_gn_in -> the kernel's view (validated)
static int fill_gn_in(const nlc_gn_in* q, const KParams& p, GnIn& g) {
    const int g0 = q->granule0 == 4 ? 4 : 8, g1 = q->granule1 == 4 ? 4 : 8;
    NLC_REQUIRE((q->granule0 == 0 || q->granule0 == 4 || q->granule0 == 8) && (q->granule1 == 0 || q->granule1 == 4 || q->granule1 == 8),
                "nlc_conv2d: gn_in granules must be 0 (= 8), 4 or 8");
    NLC_REQUIRE(q->stats0 && (p.C1 == 0) == (q->stats1 == nullptr), "nlc_conv2d: gn_in stats0 / stats1 must match x0 / x1");
    NLC_REQUIRE(q->groups > 0 && p.Ctot % q->groups == 0, "nlc_conv2d: gn_in groups=%d must divide C0+C1=%d", q->groups, p.Ctot);
    const int gs = p.Ctot / q->groups;
    NLC_REQUIRE(gs % g0 == 0 && (p.C1 == 0 || gs % g1 == 0), "nlc_conv2d: gn_in group size %d must be a multiple of the statistics granules (%d, %d)", gs, g0, g1);
    NLC_REQUIRE((q->scale == nullptr) == (q->shift == nullptr), "nlc_conv2d: gn_in scale / shift must come together");
    NLC_REQUIRE(!q->scale || q->ss_stride >= p.Ctot, "nlc_conv2d: gn_in ss_stride < C0+C1");
    NLC_REQUIRE(q->act == NLC_ACT_NONE || q->act == NLC_ACT_SILU, "nlc_conv2d: bad gn_in act %d", q->act);
    NLC_REQUIRE(((reinterpret_cast<uintptr_t>(q->stats0) | reinterpret_cast<uintptr_t>(q->stats1)) & 15) == 0, "nlc_conv2d: gn_in statistics must be 16-byte aligned");
    g.tot0 = (const long long*)q->stats0; g.tot1 = (const long long*)q->stats1;
    g.tsh0 = g0 == 4 ? 2 : 3; g.tsh1 = g1 == 4 ? 2 : 3;
    g.C0 = p.C0; g.C1 = p.C1; g.gs = gs;
    g.invN = 1.0 / ((double)p.Hin * p.Win * gs);
    g.eps = q->eps; g.gamma = q->gamma; g.beta = q->beta; g.scale = q->scale; g.shift = q->shift; g.ss_stride = q->ss_stride;
    g.act = q->act; g.div_gs = FastDiv::make(gs);
    return NLC_OK;
}

extern "C" int64_t nlc_conv2d_workspace_bytes(const nlc_conv_desc* d, int dtype) {
    if (!desc_sane(d, dtype)) return 0;
    KParams p{};
    fill_params(d, p);
    { SmallGeom sg; if (small_wanted(d, p, dtype, sg)) return nlc_conv_small_split_bytes(p, sg); }
    if (nlc_conv_narrow_ok(p, dtype)) return 0;
    const int64_t M64 = (int64_t)d->B * d->Hout * d->Wout;
    const int ks = nlc_conv_halo_ksplit(p, dtype);
    if (ks > 1) return (int64_t)ks * M64 * d->Cout * (int64_t)sizeof(float) + 4096;      // arrival counters in front of the halo kernel's partial sums
    if (nlc_conv_halo_plain_ok(p, dtype)) return 0;
    return nlc_conv_fast_split_bytes(p, nlc_conv_fast_ksplit(p, dtype));
}

extern "C" int nlc_conv2d_stats_partials(const nlc_conv_desc* d, int dtype) {
    if (!desc_sane(d, dtype)) return 0;
    KParams p{};
    fill_params(d, p);
    if (nlc_conv_narrow_ok(p, dtype)) return 0;          // the <= 16-channel kernel emits none
    const int P = nlc_conv_halo_stats_partials(p, dtype);
    if (P > 0) return P;
    return nlc_conv_fast_stats_partials(p, dtype);
}

extern "C" int nlc_conv2d_norm_out_supported(const nlc_conv_desc* d, int dtype) {
    if (!desc_sane(d, dtype)) return 0;
    KParams p{};
    fill_params(d, p);
    if (nlc_conv_narrow_ok(p, dtype)) return 0;
    return nlc_conv_pw_norm_ok(p, dtype);
}

extern "C" int nlc_conv2d_prologue_supported(const nlc_conv_desc* d, int dtype) {
    if (!desc_sane(d, dtype)) return 0;
    KParams p{};
    fill_params(d, p);
    p.gn_coef = nullptr; p.norm_out = nullptr;           // "the gn_* fields themselves are not looked at"
    if (nlc_conv_narrow_ok(p, dtype)) return 0;
    return nlc_conv_halo_prologue_ok(p, dtype);
}

// nlc_conv_desc.debug bit 1 (NLC_MATH_F16X3): the operand split needs |x| < 65504 (include/nlc_hip.h).  max |x| over the input
// tensor(s) as an atomicMax on the f32 bit patterns (monotonic for non-negative values; a NaN's pattern is above infinity's), read
// back through a device-global word.  Synchronises the stream: tests and triage only.
__device__ unsigned g_x3_absmax_bits;
__global__ __launch_bounds__(256) void x3_absmax_kernel(const float* __restrict__ x, int64_t n) {
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(fabsf(x[i])));
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&g_x3_absmax_bits, m);
}
static int check_x3_domain(const nlc_conv_desc* d, hipStream_t stream) {
    unsigned bits = 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_x3_absmax_bits), &bits, sizeof(bits), 0, hipMemcpyHostToDevice, stream) != hipSuccess) {
        nlc_set_error("nlc_conv2d: debug bit 1: could not reset the domain-check word");
        return NLC_ELAUNCH;
    }
    const int64_t pix = (int64_t)d->B * d->Hin * d->Win;
    const int64_t n0 = pix * d->C0, n1 = pix * d->C1;
    hipLaunchKernelGGL(x3_absmax_kernel, dim3((unsigned)(cdiv(n0, 256) < 1024 ? cdiv(n0, 256) : 1024)), dim3(256), 0, stream, (const float*)d->x0, n0);
    if (d->x1) hipLaunchKernelGGL(x3_absmax_kernel, dim3((unsigned)(cdiv(n1, 256) < 1024 ? cdiv(n1, 256) : 1024)), dim3(256), 0, stream, (const float*)d->x1, n1);
    NLC_CHECK_LAUNCH("nlc_conv2d(debug: F16X3 domain check)");
    if (hipMemcpyFromSymbolAsync(&bits, HIP_SYMBOL(g_x3_absmax_bits), sizeof(bits), 0, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        nlc_set_error("nlc_conv2d: debug bit 1: could not read back the domain-check word");
        return NLC_ELAUNCH;
    }
    float mx;
    memcpy(&mx, &bits, sizeof(mx));
    NLC_REQUIRE(bits < 0x477fe000u /* 65504.0f */, "nlc_conv2d: NLC_MATH_F16X3 input has max |x| = %g (or a NaN): outside the domain |x| < 65504 of the f16 operand split", (double)mx);
    return NLC_OK;
}

// nlc_conv_desc.debug bit 0: the split-K arrival counters (first 4 KiB of the workspace) must be zero when a launch starts - every
// launch leaves them zero, so a non-zero word means an aborted launch or a foreign writer poisoned the workspace, after which
// tiles would reduce early or never.  Synchronises the stream: tests and triage only.
static int check_counters(const KParams& p, hipStream_t stream) {
    static thread_local int host[1024];
    if (hipMemcpyAsync(host, p.partial, sizeof(host), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        nlc_set_error("nlc_conv2d: could not read back the workspace's arrival counters");
        return NLC_ELAUNCH;
    }
    for (int i = 0; i < 1024; ++i)
        NLC_REQUIRE(host[i] == 0, "nlc_conv2d: split-K arrival counter %d of the workspace is %d, not 0 (poisoned workspace: re-zero its first 4096 bytes)", i, host[i]);
    return NLC_OK;
}

static int validate_conv_desc(const nlc_conv_desc* d, int dtype) {
    NLC_REQUIRE(d != nullptr, "nlc_conv2d: null descriptor");
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_conv2d: bad dtype %d", dtype);
    NLC_REQUIRE(d->math == NLC_MATH_NATIVE || (d->math == NLC_MATH_F16X3 && dtype == NLC_F32), "nlc_conv2d: math %d needs dtype NLC_F32", d->math);
    NLC_REQUIRE((d->math == NLC_MATH_F16X3) == (d->w_scale != nullptr),
                "nlc_conv2d: w_scale (from nlc_pack_conv_weights_ex) is required with NLC_MATH_F16X3 and must be NULL otherwise");
    const int per = nlc_is16(dtype) ? 8 : 4;
    const int kbe = nlc_is16(dtype) ? Mma<bf16_raw>::KBE : Mma<float>::KBE;
    NLC_REQUIRE(d->x0 && d->w && d->out, "nlc_conv2d: null tensor pointer");
    NLC_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->Cout > 0,
                "nlc_conv2d: non-positive dims");
    NLC_REQUIRE(d->C0 > 0 && d->C0 % per == 0, "nlc_conv2d: C0=%d must be a positive multiple of %d", d->C0, per);
    NLC_REQUIRE(d->C1 >= 0 && d->C1 % per == 0 && ((d->C1 == 0) == (d->x1 == nullptr)),
                "nlc_conv2d: C1=%d / x1 mismatch (multiple of %d required)", d->C1, per);
    NLC_REQUIRE(d->Cin_pad % kbe == 0 && d->Cin_pad >= d->C0 + d->C1 && d->Cin_pad - (d->C0 + d->C1) < kbe,
                "nlc_conv2d: Cin_pad=%d must be C0+C1=%d rounded up to %d", d->Cin_pad, d->C0 + d->C1, kbe);
    NLC_REQUIRE(d->Cout_pad % BN == 0 && d->Cout_pad >= d->Cout, "nlc_conv2d: Cout_pad=%d must be a multiple of %d >= Cout=%d",
                d->Cout_pad, BN, d->Cout);
    NLC_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7 && d->stride >= 1, "nlc_conv2d: bad kernel/stride");
    NLC_REQUIRE(d->pad_t >= 0 && d->pad_l >= 0, "nlc_conv2d: negative padding");
    {   // every output pixel's top-left tap must start inside [-(pad), HL): the far side is zero-filled
        const int HL = d->upsample2x ? 2 * d->Hin : d->Hin, WL = d->upsample2x ? 2 * d->Win : d->Win;
        NLC_REQUIRE((d->Hout - 1) * d->stride - d->pad_t < HL && (d->Wout - 1) * d->stride - d->pad_l < WL,
                    "nlc_conv2d: output %dx%d does not fit input %dx%d (stride %d pad %d,%d)", d->Hout, d->Wout, HL, WL,
                    d->stride, d->pad_t, d->pad_l);
    }
    NLC_REQUIRE(d->act >= NLC_ACT_NONE && d->act <= NLC_ACT_GELU, "nlc_conv2d: bad act %d", d->act);
    NLC_REQUIRE(d->out_mode == NLC_OUT_NHWC || d->out_mode == NLC_OUT_NCHW_F32, "nlc_conv2d: bad out_mode");
    NLC_REQUIRE(!(d->emb) || d->emb_stride >= d->Cout, "nlc_conv2d: emb_stride < Cout");
    const int64_t M64 = (int64_t)d->B * d->Hout * d->Wout;
    NLC_REQUIRE(M64 < (1ll << 31) - BM, "nlc_conv2d: too many output pixels");
    NLC_REQUIRE(!d->res_upsample2x || (d->res && d->Hout % 2 == 0 && d->Wout % 2 == 0), "nlc_conv2d: res_upsample2x needs a residual and even Hout, Wout");
    NLC_REQUIRE(!d->gn_coef || d->gn_act == NLC_ACT_NONE || d->gn_act == NLC_ACT_SILU, "nlc_conv2d: bad gn_act %d", d->gn_act);
    NLC_REQUIRE(d->policy >= NLC_CONV_AUTO && d->policy <= NLC_CONV_FORCE_SMALL, "nlc_conv2d: bad policy %d", d->policy);
    NLC_REQUIRE(d->stats_granule == 0 || d->stats_granule == 4 || d->stats_granule == 8, "nlc_conv2d: stats_granule must be 0 (= 8), 4 or 8");
    return NLC_OK;
}

// The small-map kernel's parameter block for this descriptor: NLC_OK (sp filled), NLC_EUNSUPPORTED (not a small-map launch), or an error
static int prepare_small(const nlc_conv_desc* d, const KParams& p, int dtype, hipStream_t stream, SmallParams& sp) {
    if (!small_wanted(d, p, dtype, sp.geo)) return NLC_EUNSUPPORTED;
    sp.k = p;
    if (d->gn_in) { const int gr = fill_gn_in(d->gn_in, p, sp.gn); if (gr != NLC_OK) return gr; }
    if (d->stats_out) {
        NLC_REQUIRE(d->stats_bytes >= (int64_t)p.B * (p.Cout / p.stats_gran) * 4 * (int64_t)sizeof(long long), "nlc_conv2d: stats_out too small");
        NLC_REQUIRE((reinterpret_cast<uintptr_t>(d->stats_out) & 15) == 0, "nlc_conv2d: stats_out must be 16-byte aligned (its consumers read 16-byte pairs)");
        sp.k.stats = (long long*)d->stats_out;
    }
    if (sp.geo.ks > 1) {
        NLC_REQUIRE(d->workspace && d->workspace_bytes >= nlc_conv_small_split_bytes(p, sp.geo),
                    "nlc_conv2d: this small-map launch splits K %d ways: it needs the workspace of nlc_conv2d_workspace_bytes", sp.geo.ks);
        sp.k.partial = (float*)d->workspace;
        if (d->debug & 1) { const int cr = check_counters(sp.k, stream); if (cr != NLC_OK) return cr; }
    }
    return NLC_OK;
}

extern "C" int nlc_resblock_small(const nlc_conv_desc* d1, const nlc_conv_desc* d2, void* barrier, int dtype, void* stream) {
    int rc = validate_conv_desc(d1, dtype);
    if (rc != NLC_OK) return rc;
    rc = validate_conv_desc(d2, dtype);
    if (rc != NLC_OK) return rc;
    NLC_REQUIRE(barrier && (reinterpret_cast<uintptr_t>(barrier) & 7) == 0, "nlc_resblock_small: barrier must be two zeroed, 8-byte aligned ints");
    NLC_REQUIRE(d2->x0 == d1->out && !d2->x1, "nlc_resblock_small: the second convolution's input must be the first one's output");
    NLC_REQUIRE(d1->workspace == d2->workspace, "nlc_resblock_small: both convolutions use ONE split-K workspace (the phases do not overlap)");
    KParams p1, p2;
    fill_params(d1, p1);
    fill_params(d2, p2);
    SmallParams s1{}, s2{};
    rc = prepare_small(d1, p1, dtype, (hipStream_t)stream, s1);
    if (rc != NLC_OK) { if (rc == NLC_EUNSUPPORTED) nlc_set_error("nlc_resblock_small: the first convolution is not a small-map launch"); return rc; }
    rc = prepare_small(d2, p2, dtype, (hipStream_t)stream, s2);
    if (rc != NLC_OK) { if (rc == NLC_EUNSUPPORTED) nlc_set_error("nlc_resblock_small: the second convolution is not a small-map launch"); return rc; }
    s1.out_sc1 = 1;                                  // h crosses the grid barrier to other CUs: write-through
    rc = nlc_resblock_small_dispatch(s1, s2, (int*)barrier, dtype, (hipStream_t)stream);
    if (rc == NLC_EUNSUPPORTED) nlc_set_error("nlc_resblock_small: no one-launch form for this geometry (the two convolutions must tile alike and the grid must be resident at once)");
    return rc;
}

extern "C" int nlc_conv2d(const nlc_conv_desc* d, int dtype, void* stream) {
    { const int vr = validate_conv_desc(d, dtype); if (vr != NLC_OK) return vr; }
    const int per = nlc_is16(dtype) ? 8 : 4;
    const int kbe = nlc_is16(dtype) ? Mma<bf16_raw>::KBE : Mma<float>::KBE;
    (void)per; (void)kbe;

    if ((d->debug & 2) && d->math == NLC_MATH_F16X3) { const int cr = check_x3_domain(d, (hipStream_t)stream); if (cr != NLC_OK) return cr; }
    KParams p;
    fill_params(d, p);
    // stride-1 3x3 / 1x1 "same" convolutions take the LDS-DMA fast path; everything else (strided,
    // odd kernels, cropped outputs) the generic gather kernel.  policy NLC_CONV_GENERIC forces the latter (A/B runs).
    const bool force_generic = d->policy == NLC_CONV_GENERIC;
    {
        SmallParams sp{};
        const int pr = prepare_small(d, p, dtype, (hipStream_t)stream, sp);
        if (pr == NLC_OK) return nlc_conv_small_dispatch(sp, dtype, (hipStream_t)stream);
        if (pr != NLC_EUNSUPPORTED) return pr;
        NLC_REQUIRE(!d->gn_in, "nlc_conv2d: gn_in given but this launch cannot apply it (ask nlc_conv2d_gn_in_supported first)");
    }
    if (!force_generic) {
        int Pfast = 0;
        if (d->stats_out) {
            const int Phalo = nlc_conv_halo_stats_partials(p, dtype);
            Pfast = Phalo > 0 ? 0 : nlc_conv_fast_stats_partials(p, dtype);
            const int P = nlc_conv_narrow_ok(p, dtype) ? 0 : (Phalo > 0 ? Phalo : Pfast);
            NLC_REQUIRE(P > 0, "nlc_conv2d: stats_out given but this launch does not emit statistics (ask nlc_conv2d_stats_partials first)");
            NLC_REQUIRE(d->stats_bytes >= (int64_t)p.B * (p.Cout / p.stats_gran) * 4 * (int64_t)sizeof(long long), "nlc_conv2d: stats_out too small");
            NLC_REQUIRE((reinterpret_cast<uintptr_t>(d->stats_out) & 15) == 0, "nlc_conv2d: stats_out must be 16-byte aligned (its consumers read 16-byte pairs)");
            p.stats = (long long*)d->stats_out;
        }
        if (p.norm_out) {                            // pointwise launch that also writes act(GroupNorm(x)) of its input
            NLC_REQUIRE(p.gn_coef, "nlc_conv2d: norm_out needs gn_coef (nlc_groupnorm_coef)");
            NLC_REQUIRE((reinterpret_cast<uintptr_t>(p.norm_out) & 15) == 0, "nlc_conv2d: norm_out must be 16-byte aligned");
            NLC_REQUIRE(nlc_conv_pw_norm_ok(p, dtype),
                        "nlc_conv2d: norm_out given but this launch cannot write it (ask nlc_conv2d_norm_out_supported first)");
            if (!Pfast) p.stats = nullptr;
            return nlc_conv_pw_dispatch(p, dtype, (hipStream_t)stream);
        }
        NLC_REQUIRE(!p.gn_coef || nlc_conv_halo_prologue_ok(p, dtype),
                    "nlc_conv2d: gn_coef given but this launch has no GroupNorm prologue (ask nlc_conv2d_prologue_supported first)");
        int rc = nlc_conv_narrow_dispatch(p, dtype, (hipStream_t)stream);
        if (rc != NLC_EUNSUPPORTED) return rc;
        const int hks = nlc_conv_halo_ksplit(p, dtype);
        if (hks > 1 && d->workspace && d->workspace_bytes >= (int64_t)hks * p.M * p.Cout * (int64_t)sizeof(float) + 4096) {
            p.ksplit = hks; p.partial = (float*)d->workspace;
            if (d->debug & 1) { const int cr = check_counters(p, (hipStream_t)stream); if (cr != NLC_OK) return cr; }
        } else {
            NLC_REQUIRE(hks <= 1 || !d->stats_out || nlc_conv_halo_plain_ok(p, dtype), "nlc_conv2d: stats_out on a split-K shape needs the workspace of nlc_conv2d_workspace_bytes");
        }
        rc = nlc_conv_halo_dispatch(p, dtype, (hipStream_t)stream);
        if (rc != NLC_EUNSUPPORTED) return rc;
        p.ksplit = 1; p.partial = nullptr;
        if (!Pfast) p.stats = nullptr;
        rc = nlc_conv_pw_dispatch(p, dtype, (hipStream_t)stream);         // 1x1 with >= 1024 tiles: the persistent pointwise kernel
        if (rc != NLC_EUNSUPPORTED) return rc;
        if (d->workspace) {                          // split-K needs [ksplit][M][Cout] f32 of caller workspace
            const int ks = nlc_conv_fast_ksplit(p, dtype);
            if (ks > 1 && d->workspace_bytes >= nlc_conv_fast_split_bytes(p, ks)) {
                p.ksplit = ks; p.partial = (float*)d->workspace;
                if (d->debug & 1) { const int cr = check_counters(p, (hipStream_t)stream); if (cr != NLC_OK) return cr; }
            }
        }
        rc = nlc_conv_fast_dispatch(p, dtype, (hipStream_t)stream);
        if (rc != NLC_EUNSUPPORTED) return rc;
        p.ksplit = 1; p.partial = nullptr;
    }
    NLC_REQUIRE(!p.norm_out, "nlc_conv2d: norm_out given but this launch cannot write it");
    NLC_REQUIRE(!p.gn_coef, "nlc_conv2d: gn_coef given but this launch has no GroupNorm prologue");
    NLC_REQUIRE(!p.stats, "nlc_conv2d: stats_out given but the generic kernel emits no statistics");
    if (dtype == NLC_BF16) return launch<bf16_raw>(p, (hipStream_t)stream);
    if (dtype == NLC_F16) return launch<f16_raw>(p, (hipStream_t)stream);
    if (p.math == NLC_MATH_F16X3) return launch<float, true>(p, (hipStream_t)stream);
    return launch<float>(p, (hipStream_t)stream);
}
