// GroupNorm (+FiLM scale/shift) (+SiLU) on channels-last activations, gfx950.
//
// HBM-bound: 2 reads + 1 write of the activation.  Two launches:
//   1. gn_stats:  grid (NBLK, B).  Each workgroup streams a contiguous pixel range of one
//      image with fully coalesced 16-byte loads (a pixel's channels are contiguous), keeps
//      per-thread f32 shifted sums for a FIXED set of channels, merges them in LDS in a fixed
//      order (deterministic - no atomics) into per-group (count, mean, M2) partials -> workspace.
//   2. gn_apply:  grid (NBLK2, B).  Each workgroup folds the partials (f64) into
//      a[c] = rstd*gamma*(1+scale), b[c] = (beta-mean*rstd*gamma)*(1+scale)+shift  in LDS, then
//      streams  y = act(a[c]*x + b[c])  with 16-byte loads/stores.
// Fast path (one 16-byte chunk per thread per pixel, i.e. C <= 2048 bf16 / 1024 f32 - every GroupNorm of the three
// networks in bf16): both streaming loops are unrolled 4 pixels deep with the loads issued first (>= 4 x 16 B in
// flight per thread; one load per trip leaves HBM latency exposed), the per-channel a/b coefficients live in
// registers, and a tiny gn_finalize kernel folds the partials once per image instead of once per apply block.
// The input may be the channel concatenation of two tensors (UNet skip connections), so the
// torch.cat of the reference (src/unet_adm.py:662) is never materialised.
// y = a*x + b is the same factorisation ATen's CPU group_norm kernel uses.
#include "common.h"
#include "conv_params.h"      // Stat16: the fixed-point format of the ride-along statistics totals
#include <stdlib.h>

namespace {

constexpr int NT = 256;
constexpr int MAX_SLOTS = 4;       // chunk slots per thread per pixel (C <= 256*4*PER)
constexpr int MAX_NBLK = 64;

struct GNParams {
    const char* x0; const char* x1;
    int C0, C1, C, B, HW, G, gs;
    int nch;            // 16-byte chunks per pixel
    int nslot;          // chunk slots per thread
    int tpp;            // threads per pixel (= min(nch, 256))
    int ps;             // pixels processed concurrently per block
    int nblk;           // stats blocks per image
    int pix_per_blk;
    float eps;
    const float* gamma; const float* beta;
    const float* scale; const float* shift; int ss_stride;
    int silu;
    char* out;
    double* ws;         // [B][nblk][G][3] = (count, mean, M2)
    const float* coef;  // [B][C][2] = (a, b) per (image, channel) or NULL: the apply threads compute them
    // statistics that rode along in the producing convolutions' epilogues (nlc_conv_desc.stats_out): per source the totals
    // [B][C_src / tg][4] of 64-bit accumulators (sum.hi, sum.lo, sumsq.hi, sumsq.lo) per tg-channel chunk (conv_params.h: Stat16), or NULL
    const long long* tot0; const long long* tot1; int tg0, tg1;
    int tsh0, tsh1;     // log2 of the granules (2 | 3)
    double invN;        // 1 / (HW * gs): elements per group
    int vec_affine;     // gamma / beta / scale / shift (those present) are 16-byte aligned incl. the row stride: float4 loads
    FastDiv div_gs;     // channel -> group
};

template <typename T>
__device__ __forceinline__ uint4 load_chunk(const GNParams& p, int b, int64_t pix, int chunk) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int c = chunk * PER;
    const char* ptr;
    if (c < p.C0) ptr = p.x0 + (((int64_t)b * p.HW + pix) * p.C0 + c) * (int)sizeof(T);
    else ptr = p.x1 + (((int64_t)b * p.HW + pix) * p.C1 + (c - p.C0)) * (int)sizeof(T);
    return *reinterpret_cast<const uint4*>(ptr);
}

// Statistics are carried as (count, mean, M2) triples and merged with Chan's formula in f64;
// each thread accumulates sums of (x - x_first) so that E[x^2]-E[x]^2 cancellation never
// appears even for two-element groups with |mean| >> std.
template <typename T>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const GNParams p) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [ps][C][3] = (n, mean, M2)

    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int active = p.tpp * p.ps;
    const int tx = tid % p.tpp, pl = tid / p.tpp;
    const int pix0 = blk * p.pix_per_blk;
    const int pix1 = min(pix0 + p.pix_per_blk, p.HW);

    float s[MAX_SLOTS][PER], q[MAX_SLOTS][PER], x0[MAX_SLOTS][PER];
#pragma unroll
    for (int k = 0; k < MAX_SLOTS; ++k)
#pragma unroll
        for (int j = 0; j < PER; ++j) { s[k][j] = 0.f; q[k][j] = 0.f; x0[k][j] = 0.f; }
    int cnt = 0;

    if (tid < active) {
        for (int pix = pix0 + pl; pix < pix1; pix += p.ps, ++cnt) {
#pragma unroll
            for (int k = 0; k < MAX_SLOTS; ++k) {
                const int chunk = tx + k * p.tpp;
                if (k < p.nslot && chunk < p.nch) {
                    float f[PER];
                    chunk_to_f32<T>(load_chunk<T>(p, b, pix, chunk), f);
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        if (cnt == 0) x0[k][j] = f[j];
                        const float d = f[j] - x0[k][j];
                        s[k][j] += d; q[k][j] += d * d;
                    }
                }
            }
        }
        const float n = (float)cnt;
#pragma unroll
        for (int k = 0; k < MAX_SLOTS; ++k) {
            const int chunk = tx + k * p.tpp;
            if (k < p.nslot && chunk < p.nch) {
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int c = chunk * PER + j;
                    float* r = red + ((int64_t)pl * p.C + c) * 3;
                    const float ms = cnt ? s[k][j] / n : 0.f;
                    r[0] = n;
                    r[1] = x0[k][j] + ms;
                    r[2] = cnt ? fmaxf(q[k][j] - s[k][j] * ms, 0.f) : 0.f;
                }
            }
        }
    }
    __syncthreads();
    for (int g = tid; g < p.G; g += NT) {
        double N = 0.0, sm = 0.0;
        for (int l = 0; l < p.ps; ++l)
            for (int c = g * p.gs; c < (g + 1) * p.gs; ++c) {
                const float* r = red + ((int64_t)l * p.C + c) * 3;
                N += (double)r[0]; sm += (double)r[0] * (double)r[1];
            }
        const double mean = N > 0.0 ? sm / N : 0.0;
        double m2 = 0.0;
        for (int l = 0; l < p.ps; ++l)
            for (int c = g * p.gs; c < (g + 1) * p.gs; ++c) {
                const float* r = red + ((int64_t)l * p.C + c) * 3;
                const double dm = (double)r[1] - mean;
                m2 += (double)r[2] + (double)r[0] * dm * dm;
            }
        double* w = p.ws + (((int64_t)b * p.nblk + blk) * p.G + g) * 3;
        w[0] = N; w[1] = mean; w[2] = m2;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const GNParams p) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ca = reinterpret_cast<float*>(smem);   // [C]
    float* cb = ca + p.C;                          // [C]
    float* gm = cb + p.C;                          // [G] mean
    float* gr = gm + p.G;                          // [G] rstd

    const int b = blockIdx.y, tid = threadIdx.x;
    for (int g = tid; g < p.G; g += NT) {
        const double* w = p.ws + ((int64_t)b * p.nblk * p.G + g) * 3;
        double N = 0.0, sm = 0.0;
        for (int k = 0; k < p.nblk; ++k) { const double* r = w + (int64_t)k * p.G * 3; N += r[0]; sm += r[0] * r[1]; }
        const double mean = sm / N;
        double m2 = 0.0;
        for (int k = 0; k < p.nblk; ++k) {
            const double* r = w + (int64_t)k * p.G * 3;
            const double dm = r[1] - mean;
            m2 += r[2] + r[0] * dm * dm;
        }
        const double var = m2 / N;
        gm[g] = (float)mean;
        gr[g] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += NT) {
        const int g = c / p.gs;
        float a = gr[g] * (p.gamma ? p.gamma[c] : 1.f);
        float bb = (p.beta ? p.beta[c] : 0.f) - gm[g] * a;
        if (p.scale) {
            const float sc = 1.f + p.scale[(int64_t)b * p.ss_stride + c];
            const float sh = p.shift[(int64_t)b * p.ss_stride + c];
            a *= sc; bb = bb * sc + sh;
        }
        ca[c] = a; cb[c] = bb;
    }
    __syncthreads();

    const int64_t total = (int64_t)p.HW * p.nch;      // chunks in this image
    T* outb = reinterpret_cast<T*>(p.out) + (int64_t)b * p.HW * p.C;
    for (int64_t e = (int64_t)blockIdx.x * NT + tid; e < total; e += (int64_t)gridDim.x * NT) {
        const int64_t pix = e / p.nch;
        const int chunk = (int)(e - pix * p.nch);
        float f[PER];
        chunk_to_f32<T>(load_chunk<T>(p, b, pix, chunk), f);
        const int c0 = chunk * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            float y = ca[c0 + j] * f[j] + cb[c0 + j];
            if (p.silu) y = (sizeof(T) == 4) ? silu_exact(y) : silu_f(y);
            f[j] = y;
        }
        *reinterpret_cast<uint4*>(outb + pix * p.C + c0) = f32_to_chunk<T>(f);
    }
}

// ---- fast path: nslot == 1 ---------------------------------------------------------------------------------
constexpr int UNR = 8;                       // 16-byte loads in flight per lane of the streaming kernels; 4 / 16 measured within 3 % (round 2)
constexpr int GN_APPLY_MAXBLK = 8192;        // largest apply grid (1024 ... 8192 workgroups measured within 3 %)

template <typename T>
__global__ __launch_bounds__(NT) void gn_stats_fast_kernel(const GNParams p) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [ps][C][3] = (n, mean, M2)
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int active = p.tpp * p.ps;
    const int tx = tid % p.tpp, pl = tid / p.tpp;
    const int pix0 = blk * p.pix_per_blk;
    const int pix1 = min(pix0 + p.pix_per_blk, p.HW);
    if (tid < active) {
        float s[PER], q[PER], x0[PER];
        int cnt = 0;
        int pix = pix0 + pl;
        if (pix < pix1) chunk_to_f32<T>(load_chunk<T>(p, b, pix, tx), x0);
#pragma unroll
        for (int j = 0; j < PER; ++j) { s[j] = 0.f; q[j] = 0.f; if (pix >= pix1) x0[j] = 0.f; }
        for (; pix + (UNR - 1) * p.ps < pix1; pix += UNR * p.ps, cnt += UNR) {
            uint4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = load_chunk<T>(p, b, pix + u * p.ps, tx);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float f[PER];
                chunk_to_f32<T>(v[u], f);
#pragma unroll
                for (int j = 0; j < PER; ++j) { const float d = f[j] - x0[j]; s[j] += d; q[j] += d * d; }
            }
        }
        for (; pix < pix1; pix += p.ps, ++cnt) {
            float f[PER];
            chunk_to_f32<T>(load_chunk<T>(p, b, pix, tx), f);
#pragma unroll
            for (int j = 0; j < PER; ++j) { const float d = f[j] - x0[j]; s[j] += d; q[j] += d * d; }
        }
        const float n = (float)cnt;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            float* r = red + ((int64_t)pl * p.C + tx * PER + j) * 3;
            const float ms = cnt ? s[j] / n : 0.f;
            r[0] = n;
            r[1] = x0[j] + ms;
            r[2] = cnt ? fmaxf(q[j] - s[j] * ms, 0.f) : 0.f;
        }
    }
    __syncthreads();
    for (int g = tid; g < p.G; g += NT) {
        double N = 0.0, sm = 0.0;
        for (int l = 0; l < p.ps; ++l)
            for (int c = g * p.gs; c < (g + 1) * p.gs; ++c) {
                const float* r = red + ((int64_t)l * p.C + c) * 3;
                N += (double)r[0]; sm += (double)r[0] * (double)r[1];
            }
        const double mean = N > 0.0 ? sm / N : 0.0;
        double m2 = 0.0;
        for (int l = 0; l < p.ps; ++l)
            for (int c = g * p.gs; c < (g + 1) * p.gs; ++c) {
                const float* r = red + ((int64_t)l * p.C + c) * 3;
                const double dm = (double)r[1] - mean;
                m2 += (double)r[2] + (double)r[0] * dm * dm;
            }
        double* w = p.ws + (((int64_t)b * p.nblk + blk) * p.G + g) * 3;
        w[0] = N; w[1] = mean; w[2] = m2;
    }
}

// one workgroup per image: Chan-merge the per-block partials (f64, fixed order) -> (mean, rstd) per group,
// written as two floats behind the partials of the image's block 0..nblk-1 region: stat[b][g] = {mean, rstd}
__global__ __launch_bounds__(NT) void gn_finalize_kernel(const GNParams p, float* __restrict__ stat) {
    // 8 lanes per group walk the (up to 64) partials with stride 8 and combine by xor-shuffles in a fixed order
    // (a single thread per group chained 2 x nblk dependent L2 loads: 13 us per launch, 2 ms per NLC step)
    const int b = blockIdx.x;
    const int sub = threadIdx.x & 7;
    for (int g = threadIdx.x >> 3; g < p.G; g += NT / 8) {
        const double* w = p.ws + ((int64_t)b * p.nblk * p.G + g) * 3;
        double N = 0.0, sm = 0.0;
        for (int k = sub; k < p.nblk; k += 8) { const double* r = w + (int64_t)k * p.G * 3; N += r[0]; sm += r[0] * r[1]; }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { N += __shfl_xor(N, o, 64); sm += __shfl_xor(sm, o, 64); }
        const double mean = sm / N;
        double m2 = 0.0;
        for (int k = sub; k < p.nblk; k += 8) {
            const double* r = w + (int64_t)k * p.G * 3;
            const double dm = r[1] - mean;
            m2 += r[2] + r[0] * dm * dm;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        if (sub == 0) {
            const double var = m2 / N;
            stat[((int64_t)b * p.G + g) * 2 + 0] = (float)mean;
            stat[((int64_t)b * p.G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
    }
}

// (mean, rstd) of a group from the ride-along totals: conv_params.h, gn_group_from_totals_t (shared with conv_small.hip, which
// applies the same normalisation on its input's way into LDS)
__device__ __forceinline__ void gn_group_from_totals(const GNParams& p, int b, int g, float& mean, float& rstd) {
    gn_group_from_totals_t(p, b, g, mean, rstd);
}

// nlc_groupnorm_coef: the (a, b) table of GroupNorm (+FiLM) per (image, channel) from the totals, for a convolution that applies the
// normalisation in its LDS prologue (nlc_conv_desc.gn_coef).  One thread per (image, channel).
__global__ __launch_bounds__(NT) void gn_coef_from_totals_kernel(const GNParams p, float* __restrict__ coef) {
    const int b = blockIdx.y, ch = blockIdx.x * NT + threadIdx.x;
    if (ch >= p.C) return;
    float mean, rstd;
    gn_group_from_totals(p, b, p.div_gs.div(ch), mean, rstd);
    float aa = rstd * (p.gamma ? p.gamma[ch] : 1.f);
    float bb = (p.beta ? p.beta[ch] : 0.f) - mean * aa;
    if (p.scale) {
        const float sc = 1.f + p.scale[(int64_t)b * p.ss_stride + ch];
        const float sh = p.shift[(int64_t)b * p.ss_stride + ch];
        aa *= sc; bb = bb * sc + sh;
    }
    *reinterpret_cast<float2*>(coef + ((int64_t)b * p.C + ch) * 2) = float2{aa, bb};
}

// y = ca[j] * x + cb[j] for the PER channels from c0 of image b: GroupNorm (mean, rstd from `stat`, or folded here from the few
// partials of a small map) x gamma / beta x FiLM (1 + scale, shift)
template <int PER>
__device__ __forceinline__ void gn_channel_coefs(const GNParams& p, const float* __restrict__ stat, int b, int c0,
                                                 float (&ca)[PER], float (&cb)[PER]) {
    if (p.coef) {                                    // folded once per (image, channel) by the finalize kernel: 2 PER floats, one round trip
        const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)b * p.C + c0) * 2);
#pragma unroll
        for (int q = 0; q < PER / 2; ++q) {
            const float4 v = cf[q];                  // a0 b0 a1 b1
            ca[2 * q] = v.x; cb[2 * q] = v.y; ca[2 * q + 1] = v.z; cb[2 * q + 1] = v.w;
        }
        return;
    }
    if (p.tot0) {
        // ride-along totals: affine parameters first (independent of the totals, 16-byte loads when aligned), then (mean, rstd) of the
        // one or two groups these PER channels lie in - one integer division, no f64 division or square root
        float ga[PER], be[PER], sc[PER], sh[PER];
        if (p.vec_affine) {
#pragma unroll
            for (int q = 0; q < PER / 4; ++q) {
                const float4 g4 = p.gamma ? reinterpret_cast<const float4*>(p.gamma + c0)[q] : float4{1.f, 1.f, 1.f, 1.f};
                const float4 b4 = p.beta ? reinterpret_cast<const float4*>(p.beta + c0)[q] : float4{0.f, 0.f, 0.f, 0.f};
                ga[4 * q] = g4.x; ga[4 * q + 1] = g4.y; ga[4 * q + 2] = g4.z; ga[4 * q + 3] = g4.w;
                be[4 * q] = b4.x; be[4 * q + 1] = b4.y; be[4 * q + 2] = b4.z; be[4 * q + 3] = b4.w;
                if (p.scale) {
                    const float4 s4 = reinterpret_cast<const float4*>(p.scale + (int64_t)b * p.ss_stride + c0)[q];
                    const float4 h4 = reinterpret_cast<const float4*>(p.shift + (int64_t)b * p.ss_stride + c0)[q];
                    sc[4 * q] = s4.x; sc[4 * q + 1] = s4.y; sc[4 * q + 2] = s4.z; sc[4 * q + 3] = s4.w;
                    sh[4 * q] = h4.x; sh[4 * q + 1] = h4.y; sh[4 * q + 2] = h4.z; sh[4 * q + 3] = h4.w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                ga[j] = p.gamma ? p.gamma[c0 + j] : 1.f;
                be[j] = p.beta ? p.beta[c0 + j] : 0.f;
                if (p.scale) { sc[j] = p.scale[(int64_t)b * p.ss_stride + c0 + j]; sh[j] = p.shift[(int64_t)b * p.ss_stride + c0 + j]; }
            }
        }
        int g = p.div_gs.div(c0);
        int left = (g + 1) * p.gs - c0;                  // channels of group g from c0 on
        float mean, rstd;
        gn_group_from_totals(p, b, g, mean, rstd);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (left == 0) { ++g; left = p.gs; gn_group_from_totals(p, b, g, mean, rstd); }
            --left;
            float a = rstd * ga[j];
            float bb = be[j] - mean * a;
            if (p.scale) { const float f = 1.f + sc[j]; a *= f; bb = bb * f + sh[j]; }
            ca[j] = a; cb[j] = bb;
        }
        return;
    }
    float mean = 0.f, rstd = 0.f;
    int gprev = -1;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = c0 + j;
        const int g = c / p.gs;
        if (stat) { mean = stat[((int64_t)b * p.G + g) * 2]; rstd = stat[((int64_t)b * p.G + g) * 2 + 1]; }
        else if (g != gprev) {
            // few partials (small maps): every thread folds them itself, same f64 Chan merge and order as gn_finalize
            const double* w = p.ws + ((int64_t)b * p.nblk * p.G + g) * 3;
            double N = 0.0, sm = 0.0;
            for (int k = 0; k < p.nblk; ++k) { const double* r = w + (int64_t)k * p.G * 3; N += r[0]; sm += r[0] * r[1]; }
            const double mu = sm / N;
            double m2 = 0.0;
            for (int k = 0; k < p.nblk; ++k) {
                const double* r = w + (int64_t)k * p.G * 3;
                const double dm = r[1] - mu;
                m2 += r[2] + r[0] * dm * dm;
            }
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(m2 / N + (double)p.eps));
            gprev = g;
        }
        float a = rstd * (p.gamma ? p.gamma[c] : 1.f);
        float bb = (p.beta ? p.beta[c] : 0.f) - mean * a;
        if (p.scale) {
            const float sc = 1.f + p.scale[(int64_t)b * p.ss_stride + c];
            const float sh = p.shift[(int64_t)b * p.ss_stride + c];
            a *= sc; bb = bb * sc + sh;
        }
        ca[j] = a; cb[j] = bb;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_fast_kernel(const GNParams p, const float* __restrict__ stat, int pix_per_blk) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int active = p.tpp * p.ps;
    if (tid >= active) return;
    const int tx = tid % p.tpp, pl = tid / p.tpp;
    const int c0 = tx * PER;
    float ca[PER], cb[PER];
    const int pix0 = blockIdx.x * pix_per_blk;
    const int pix1 = min(pix0 + pix_per_blk, p.HW);
    T* outb = reinterpret_cast<T*>(p.out) + (int64_t)b * p.HW * p.C + c0;
    int pix = pix0 + pl;
    auto act1 = [&](const uint4& vv, int at) {
        float f[PER];
        chunk_to_f32<T>(vv, f);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            float y = ca[j] * f[j] + cb[j];
            if (p.silu) y = (sizeof(T) == 4) ? silu_exact(y) : silu_f(y);
            f[j] = y;
        }
        *reinterpret_cast<uint4*>(outb + (int64_t)at * p.C) = f32_to_chunk<T>(f);
    };
    {
        // the first trip's loads are issued BEFORE the coefficient chain (totals -> group statistics -> a, b: a dependent memory
        // round trip plus some f64 arithmetic), so that the two latencies overlap - on the small maps a thread has one trip in all
        const bool full0 = pix + (UNR - 1) * p.ps < pix1;
        uint4 v0[UNR];
        if (full0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) v0[u] = load_chunk<T>(p, b, pix + u * p.ps, tx);
        } else if (pix < pix1) {
            v0[0] = load_chunk<T>(p, b, pix, tx);
        }
        gn_channel_coefs<PER>(p, stat, b, c0, ca, cb);
        if (full0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) act1(v0[u], pix + u * p.ps);
            pix += UNR * p.ps;
        } else if (pix < pix1) {
            act1(v0[0], pix);
            pix += p.ps;
        }
    }
    for (; pix + (UNR - 1) * p.ps < pix1; pix += UNR * p.ps) {
        uint4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = load_chunk<T>(p, b, pix + u * p.ps, tx);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float f[PER];
            chunk_to_f32<T>(v[u], f);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float y = ca[j] * f[j] + cb[j];
                if (p.silu) y = (sizeof(T) == 4) ? silu_exact(y) : silu_f(y);
                f[j] = y;
            }
            *reinterpret_cast<uint4*>(outb + (int64_t)(pix + u * p.ps) * p.C) = f32_to_chunk<T>(f);
        }
    }
    for (; pix < pix1; pix += p.ps) {
        float f[PER];
        chunk_to_f32<T>(load_chunk<T>(p, b, pix, tx), f);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            float y = ca[j] * f[j] + cb[j];
            if (p.silu) y = (sizeof(T) == 4) ? silu_exact(y) : silu_f(y);
            f[j] = y;
        }
        *reinterpret_cast<uint4*>(outb + (int64_t)pix * p.C) = f32_to_chunk<T>(f);
    }
}

// GroupNorm (+FiLM) (+SiLU) followed by AvgPool2d(2), and AvgPool2d(2) of the raw input beside it - the two branches of a
// down-sampling ResBlock (src/unet_adm.py:193-195: h = in_rest(x); h = h_upd(h); x = x_upd(x)) from ONE read of x: the
// full-resolution normalised tensor is never written.  Thread = (output pixel lane, 8/4-channel chunk); the four
// activated values are averaged in f32 in the order avgpool_kernel uses ((a + b) + c + d) * 0.25, so the f32 path is
// bit-identical to the two separate passes (the bf16 one skips an intermediate rounding).
template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_pool_kernel(const GNParams p, const float* __restrict__ stat, int W, int opix_per_blk,
                                                           T* __restrict__ out_x) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int active = p.tpp * p.ps;
    if (tid >= active) return;
    const int tx = tid % p.tpp, pl = tid / p.tpp;
    const int c0 = tx * PER;
    float ca[PER], cb[PER];
    gn_channel_coefs<PER>(p, stat, b, c0, ca, cb);
    const int Wo = W >> 1, OHW = p.HW >> 2;
    const int pix0 = blockIdx.x * opix_per_blk;
    const int pix1 = min(pix0 + opix_per_blk, OHW);
    const T* xb = reinterpret_cast<const T*>(p.x0) + (int64_t)b * p.HW * p.C + c0;
    T* outb = reinterpret_cast<T*>(p.out) + (int64_t)b * OHW * p.C + c0;
    T* outx = out_x + (int64_t)b * OHW * p.C + c0;
    constexpr int U = 2;
    auto one = [&](const uint4 (&v)[4], int opix) {
        float hs[PER], xs[PER];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[PER];
            chunk_to_f32<T>(v[k], f);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float y = ca[j] * f[j] + cb[j];
                if (p.silu) y = (sizeof(T) == 4) ? silu_exact(y) : silu_f(y);
                if (k == 0) { hs[j] = y; xs[j] = f[j]; } else { hs[j] += y; xs[j] += f[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) { hs[j] *= 0.25f; xs[j] *= 0.25f; }
        *reinterpret_cast<uint4*>(outb + (int64_t)opix * p.C) = f32_to_chunk<T>(hs);
        *reinterpret_cast<uint4*>(outx + (int64_t)opix * p.C) = f32_to_chunk<T>(xs);
    };
    auto load4 = [&](uint4 (&v)[4], int opix) {
        const int oy = opix / Wo, ox = opix - oy * Wo;
        const T* p00 = xb + ((int64_t)(2 * oy) * W + 2 * ox) * p.C;
        v[0] = *reinterpret_cast<const uint4*>(p00);
        v[1] = *reinterpret_cast<const uint4*>(p00 + p.C);
        v[2] = *reinterpret_cast<const uint4*>(p00 + (int64_t)W * p.C);
        v[3] = *reinterpret_cast<const uint4*>(p00 + (int64_t)W * p.C + p.C);
    };
    int opix = pix0 + pl;
    for (; opix + (U - 1) * p.ps < pix1; opix += U * p.ps) {
        uint4 v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) load4(v[u], opix + u * p.ps);
#pragma unroll
        for (int u = 0; u < U; ++u) one(v[u], opix + u * p.ps);
    }
    for (; opix < pix1; opix += p.ps) {
        uint4 v[4];
        load4(v, opix);
        one(v, opix);
    }
}

int fill_params(GNParams& p, const void* x0, const void* x1, int C0, int C1, int B, int HW, int groups, int dtype) {
    const int per = nlc_is16(dtype) ? 8 : 4, es = nlc_is16(dtype) ? 2 : 4;
    p.x0 = (const char*)x0; p.x1 = (const char*)x1;
    p.C0 = C0; p.C1 = C1; p.C = C0 + C1; p.B = B; p.HW = HW; p.G = groups; p.gs = p.C / groups;
    p.nch = p.C / per;
    p.coef = nullptr;
    p.tot0 = nullptr; p.tot1 = nullptr; p.tg0 = 8; p.tg1 = 8; p.tsh0 = 3; p.tsh1 = 3;
    p.invN = 1.0 / ((double)HW * p.gs); p.vec_affine = 0; p.div_gs = FastDiv::make(p.gs);
    p.tpp = p.nch < NT ? p.nch : NT;
    p.nslot = cdiv(p.nch, p.tpp);
    p.ps = NT / p.tpp; if (p.ps < 1) p.ps = 1;
    int64_t bytes = (int64_t)HW * p.C * es;
    int nblk = cdiv(bytes, 16384);                   // >= 16 KiB per stats block: small maps are latency-, not bandwidth-bound
    if (nblk < 1) nblk = 1;
    if (nblk > MAX_NBLK) nblk = MAX_NBLK;
    if (nblk > HW) nblk = HW;
    p.pix_per_blk = cdiv(HW, nblk);
    p.nblk = cdiv(HW, p.pix_per_blk);
    return 0;
}

}  // namespace

// run `...` with TG = the storage type of `dtype` (three instantiations of every GroupNorm kernel)
#define GN_SWITCH(dtype, ...)                                            \
    do {                                                                 \
        if ((dtype) == NLC_BF16) { using TG = bf16_raw; __VA_ARGS__; }   \
        else if ((dtype) == NLC_F16) { using TG = f16_raw; __VA_ARGS__; } \
        else { using TG = float; __VA_ARGS__; }                          \
    } while (0)

static int affine_aligned(const float* gamma, const float* beta, const float* scale, const float* shift, int ss_stride) {
    const uintptr_t m = reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(scale) |
                        reinterpret_cast<uintptr_t>(shift);
    return ((m & 15) == 0 && (ss_stride & 3) == 0) ? 1 : 0;
}

// Pixels per apply workgroup: `unit` (= ps x unroll x 4 trips) for the big maps, capped so the grid stays <= GN_APPLY_MAXBLK
// workgroups; halved down to `ps` (one chunk per thread) while the launch would have fewer than 512 workgroups - the 8x8 ... 32x32
// levels otherwise ran 16 ... 128 workgroups of four dependent trips each on a 256-CU chip (12-15 us per launch instead of ~5).
static int apply_pix_per_block(int HW, int B, int ps, int unit) {
    int ppb = unit;
    while ((int64_t)cdiv(HW, ppb) * B > GN_APPLY_MAXBLK) ppb *= 2;
    while (ppb > ps && (int64_t)cdiv(HW, ppb) * B < 512) ppb /= 2;
    return ppb;
}

extern "C" int64_t nlc_groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
    (void)HW;
    return (int64_t)B * MAX_NBLK * groups * 3 * sizeof(double) + (int64_t)B * groups * 2 * sizeof(float) + 16 +
           (int64_t)B * C * 2 * sizeof(float);      // + the per-(image, channel) coefficient table of the ride-along-statistics path
}

extern "C" int nlc_groupnorm(const void* x0, const void* x1, int C0, int C1, int B, int HW, int groups, float eps,
                             const float* gamma, const float* beta, const float* scale, const float* shift,
                             int ss_stride, int silu, void* out, void* workspace, int dtype, void* stream) {
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_groupnorm: bad dtype %d", dtype);
    const int per = nlc_is16(dtype) ? 8 : 4;
    NLC_REQUIRE(x0 && out && workspace, "nlc_groupnorm: null pointer");
    NLC_REQUIRE(B > 0 && HW > 0 && C0 > 0 && C1 >= 0 && groups > 0, "nlc_groupnorm: bad dims");
    NLC_REQUIRE(C0 % per == 0 && C1 % per == 0, "nlc_groupnorm: C0=%d, C1=%d must be multiples of %d", C0, C1, per);
    NLC_REQUIRE((C1 == 0) == (x1 == nullptr), "nlc_groupnorm: x1/C1 mismatch");
    const int C = C0 + C1;
    NLC_REQUIRE(C % groups == 0, "nlc_groupnorm: C=%d not divisible by groups=%d", C, groups);
    NLC_REQUIRE(C / per <= NT * MAX_SLOTS, "nlc_groupnorm: C=%d too large", C);
    NLC_REQUIRE((scale == nullptr) == (shift == nullptr), "nlc_groupnorm: scale/shift must come together");
    NLC_REQUIRE(!scale || ss_stride >= C, "nlc_groupnorm: ss_stride < C");
    GNParams p;
    fill_params(p, x0, x1, C0, C1, B, HW, groups, dtype);
    p.eps = eps; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.ss_stride = ss_stride;
    p.silu = silu; p.out = (char*)out; p.ws = (double*)workspace;
    const size_t lds_stats = (size_t)p.ps * p.C * 3 * sizeof(float);
    const size_t lds_apply = ((size_t)2 * p.C + 2 * p.G) * sizeof(float);
    NLC_REQUIRE(lds_stats <= 64 * 1024 && lds_apply <= 64 * 1024, "nlc_groupnorm: LDS budget exceeded (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    const int es = nlc_is16(dtype) ? 2 : 4;
    int64_t chunks = (int64_t)HW * p.nch;
    int nblk2 = cdiv(chunks, NT * 8);
    if (nblk2 < 1) nblk2 = 1;
    if (nblk2 > 2048 / B + 1) nblk2 = 2048 / B + 1;
    (void)es;
    static const bool fast_ok = !(getenv("NLC_GN_FAST") && getenv("NLC_GN_FAST")[0] == '0');   // diagnostic switch
    if (p.nslot == 1 && fast_ok) {
        float* stat = reinterpret_cast<float*>(p.ws + (int64_t)B * MAX_NBLK * groups * 3);
        // apply blocks: 4 unrolled trips of `ps` pixels each per thread, capped so the grid stays <= ~8k blocks
        const int ppb = apply_pix_per_block(HW, B, p.ps, p.ps * UNR * 4);
        const int nblk_a = cdiv(HW, ppb);
        const bool inline_fin = p.nblk <= 8;         // small maps: no finalize launch, the apply threads fold the partials
        if (inline_fin) stat = nullptr;
        GN_SWITCH(dtype,
            hipLaunchKernelGGL(gn_stats_fast_kernel<TG>, dim3(p.nblk, B), dim3(NT), lds_stats, st, p);
            if (!inline_fin) hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(NT), 0, st, p, stat);
            hipLaunchKernelGGL(gn_apply_fast_kernel<TG>, dim3(nblk_a, B), dim3(NT), 0, st, p, stat, ppb));
    } else {
        GN_SWITCH(dtype,
            hipLaunchKernelGGL(gn_stats_kernel<TG>, dim3(p.nblk, B), dim3(NT), lds_stats, st, p);
            hipLaunchKernelGGL(gn_apply_kernel<TG>, dim3(nblk2, B), dim3(NT), lds_apply, st, p));
    }
    NLC_CHECK_LAUNCH("nlc_groupnorm");
    return NLC_OK;
}

extern "C" int nlc_groupnorm_prestats(const void* x0, const void* x1, int C0, int C1, int B, int HW, int groups, float eps,
                                      const float* gamma, const float* beta, const float* scale, const float* shift,
                                      int ss_stride, int silu, void* out, int dtype,
                                      const void* stats0, int gran0, const void* stats1, int gran1, void* stream) {
    const int g0 = gran0 == 4 ? 4 : 8, g1 = gran1 == 4 ? 4 : 8;
    NLC_REQUIRE((gran0 == 0 || gran0 == 4 || gran0 == 8) && (gran1 == 0 || gran1 == 4 || gran1 == 8), "nlc_groupnorm_prestats: granules must be 0 (= 8), 4 or 8");
    NLC_REQUIRE(nlc_is16(dtype), "nlc_groupnorm_prestats: bf16 / f16 only (the f32 path computes its statistics itself)");
    NLC_REQUIRE(x0 && out && stats0, "nlc_groupnorm_prestats: null pointer");
    NLC_REQUIRE(B > 0 && HW > 0 && C0 > 0 && C1 >= 0 && groups > 0, "nlc_groupnorm_prestats: bad dims");
    NLC_REQUIRE((C1 == 0) == (x1 == nullptr) && (C1 == 0) == (stats1 == nullptr), "nlc_groupnorm_prestats: x1 / stats1 / C1 mismatch");
    const int C = C0 + C1;
    NLC_REQUIRE(C % groups == 0 && (C / groups) % g0 == 0 && (C1 == 0 || (C / groups) % g1 == 0) && C0 % 8 == 0 && C1 % 8 == 0,
                "nlc_groupnorm_prestats: group size %d must be a multiple of the statistics granules (%d, %d) and C0=%d, C1=%d of 8", C / groups, g0, g1, C0, C1);
    NLC_REQUIRE(C / 8 <= NT, "nlc_groupnorm_prestats: C=%d too large for the one-chunk-per-thread apply kernel", C);
    NLC_REQUIRE((scale == nullptr) == (shift == nullptr), "nlc_groupnorm_prestats: scale/shift must come together");
    NLC_REQUIRE(!scale || ss_stride >= C, "nlc_groupnorm_prestats: ss_stride < C");
    NLC_REQUIRE(((reinterpret_cast<uintptr_t>(stats0) | reinterpret_cast<uintptr_t>(stats1)) & 15) == 0, "nlc_groupnorm_prestats: statistics must be 16-byte aligned");
    GNParams p;
    fill_params(p, x0, x1, C0, C1, B, HW, groups, dtype);
    p.eps = eps; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.ss_stride = ss_stride;
    p.silu = silu; p.out = (char*)out; p.ws = nullptr;
    p.tot0 = (const long long*)stats0; p.tot1 = (const long long*)stats1; p.tg0 = g0; p.tg1 = g1;
    p.tsh0 = g0 == 4 ? 2 : 3; p.tsh1 = g1 == 4 ? 2 : 3;
    p.vec_affine = affine_aligned(gamma, beta, scale, shift, ss_stride);
    const int ppb = apply_pix_per_block(HW, B, p.ps, p.ps * UNR * 4);
    // ONE launch: every apply thread derives its channels' (a, b) from the totals (gn_group_from_totals)
    NLC_SWITCH_16(dtype, hipLaunchKernelGGL(gn_apply_fast_kernel<T16>, dim3(cdiv(HW, ppb), B), dim3(NT), 0, (hipStream_t)stream, p, (const float*)nullptr, ppb));
    NLC_CHECK_LAUNCH("nlc_groupnorm_prestats");
    return NLC_OK;
}

extern "C" int nlc_groupnorm_pool2x2(const void* x, int C, int B, int H, int W, int groups, float eps, const float* gamma,
                                     const float* beta, const float* scale, const float* shift, int ss_stride, int silu,
                                     void* out_norm, void* out_x, void* workspace, int dtype, const void* stats0, int gran0,
                                     void* stream) {
    const int g0 = gran0 == 4 ? 4 : 8;
    NLC_REQUIRE(gran0 == 0 || gran0 == 4 || gran0 == 8, "nlc_groupnorm_pool2x2: granule must be 0 (= 8), 4 or 8");
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_groupnorm_pool2x2: bad dtype %d", dtype);
    const int per = nlc_is16(dtype) ? 8 : 4;
    NLC_REQUIRE(x && out_norm && out_x && (workspace || stats0), "nlc_groupnorm_pool2x2: null pointer");
    NLC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && groups > 0 && H % 2 == 0 && W % 2 == 0, "nlc_groupnorm_pool2x2: bad dims (H, W even)");
    NLC_REQUIRE(C % per == 0 && C % groups == 0, "nlc_groupnorm_pool2x2: C=%d must be a multiple of %d and of groups=%d", C, per, groups);
    NLC_REQUIRE(C / per <= NT, "nlc_groupnorm_pool2x2: C=%d too large for the one-chunk-per-thread kernels", C);
    NLC_REQUIRE((scale == nullptr) == (shift == nullptr), "nlc_groupnorm_pool2x2: scale/shift must come together");
    NLC_REQUIRE(!scale || ss_stride >= C, "nlc_groupnorm_pool2x2: ss_stride < C");
    NLC_REQUIRE(!stats0 || (nlc_is16(dtype) && (C / groups) % g0 == 0 && (reinterpret_cast<uintptr_t>(stats0) & 15) == 0),
                "nlc_groupnorm_pool2x2: ride-along statistics are bf16 / f16 only, group size a multiple of their granule");
    const int HW = H * W;
    GNParams p;
    fill_params(p, x, nullptr, C, 0, B, HW, groups, dtype);
    p.eps = eps; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.ss_stride = ss_stride;
    p.silu = silu; p.out = (char*)out_norm; p.ws = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    float* stat = nullptr;
    if (stats0) {
        p.tot0 = (const long long*)stats0; p.tg0 = g0; p.tsh0 = g0 == 4 ? 2 : 3;   // the apply threads derive (a, b) from the totals: no other launch
        p.vec_affine = affine_aligned(gamma, beta, scale, shift, ss_stride);
    } else {
        stat = reinterpret_cast<float*>(p.ws + (int64_t)B * MAX_NBLK * groups * 3);
        const size_t lds_stats = (size_t)p.ps * p.C * 3 * sizeof(float);
        NLC_REQUIRE(lds_stats <= 64 * 1024, "nlc_groupnorm_pool2x2: LDS budget exceeded (C=%d)", C);
        GN_SWITCH(dtype, hipLaunchKernelGGL(gn_stats_fast_kernel<TG>, dim3(p.nblk, B), dim3(NT), lds_stats, st, p));
        if (p.nblk <= 8) stat = nullptr;             // small maps: the apply threads fold the partials themselves
        else hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(NT), 0, st, p, stat);
    }
    const int OHW = HW / 4;
    const int oppb = apply_pix_per_block(OHW, B, p.ps, p.ps * 2 * 4);      // 4 unrolled trips of 2 x ps output pixels (= 8 x ps input pixels)
    GN_SWITCH(dtype, hipLaunchKernelGGL(gn_apply_pool_kernel<TG>, dim3(cdiv(OHW, oppb), B), dim3(NT), 0, st, p, stat, W, oppb, (TG*)out_x));
    NLC_CHECK_LAUNCH("nlc_groupnorm_pool2x2");
    return NLC_OK;
}

extern "C" int nlc_groupnorm_coef(int C0, int C1, int B, int HW, int groups, float eps, const float* gamma, const float* beta,
                                  const float* scale, const float* shift, int ss_stride, const void* stats0, int gran0,
                                  const void* stats1, int gran1, float* coef, void* stream) {
    const int g0 = gran0 == 4 ? 4 : 8, g1 = gran1 == 4 ? 4 : 8;
    NLC_REQUIRE(coef && stats0, "nlc_groupnorm_coef: null pointer");
    NLC_REQUIRE(B > 0 && HW > 0 && C0 > 0 && C1 >= 0 && groups > 0, "nlc_groupnorm_coef: bad dims");
    NLC_REQUIRE((C1 == 0) == (stats1 == nullptr), "nlc_groupnorm_coef: stats1 / C1 mismatch");
    const int C = C0 + C1;
    NLC_REQUIRE(C % groups == 0 && (C / groups) % g0 == 0 && (C1 == 0 || (C / groups) % g1 == 0) && C0 % 8 == 0 && C1 % 8 == 0,
                "nlc_groupnorm_coef: group size %d must be a multiple of the statistics granules and C0=%d, C1=%d of 8", C / groups, C0, C1);
    NLC_REQUIRE((scale == nullptr) == (shift == nullptr), "nlc_groupnorm_coef: scale/shift must come together");
    NLC_REQUIRE(!scale || ss_stride >= C, "nlc_groupnorm_coef: ss_stride < C");
    NLC_REQUIRE(((reinterpret_cast<uintptr_t>(stats0) | reinterpret_cast<uintptr_t>(stats1)) & 15) == 0, "nlc_groupnorm_coef: statistics must be 16-byte aligned");
    GNParams p{};
    p.C0 = C0; p.C1 = C1; p.C = C; p.B = B; p.HW = HW; p.G = groups; p.gs = C / groups; p.eps = eps;
    p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.ss_stride = ss_stride;
    p.tot0 = (const long long*)stats0; p.tot1 = (const long long*)stats1; p.tg0 = g0; p.tg1 = g1;
    p.tsh0 = g0 == 4 ? 2 : 3; p.tsh1 = g1 == 4 ? 2 : 3;
    p.invN = 1.0 / ((double)HW * p.gs); p.div_gs = FastDiv::make(p.gs);
    hipLaunchKernelGGL(gn_coef_from_totals_kernel, dim3(cdiv(C, NT), B), dim3(NT), 0, (hipStream_t)stream, p, coef);
    NLC_CHECK_LAUNCH("nlc_groupnorm_coef");
    return NLC_OK;
}
