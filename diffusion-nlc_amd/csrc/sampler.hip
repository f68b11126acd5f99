// Sampler-state kernels: everything the DDIM/EDM + NLC loop does to the f32 NCHW state
// between two network evaluations.  All HBM-bound and per-sample; the per-sample scalars
// (sigma_t, sigma_prev, t, c_in) live in small device arrays so the loop has no host syncs.
// Formulas restate src/experiments.py:186-207,273-293,360-370,401-431,457-459 and
// src/schedulers.py:185-190,367-390,407-449 (+ the variants at :465-627), operation by operation
// in f32 so results track the CPU reference to rounding.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int RT = 1024;   // threads for one-block-per-sample reductions

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) { for (int i = 0; i < nw; ++i) r += sh[i]; sh[0] = r; }
    __syncthreads();
    r = sh[0];
    __syncthreads();
    return r;
}

// sumsq[b] = sum_{d<D} x[b*row_stride + d]^2, one workgroup per sample, fixed summation order.
__global__ __launch_bounds__(RT) void row_sumsq_kernel(const float* __restrict__ x, float* __restrict__ sumsq,
                                                      int64_t row_stride, int64_t D) {
    __shared__ float sh[RT / 64];
    const float* row = x + (int64_t)blockIdx.x * row_stride;
    float acc = 0.f;
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    int64_t i0 = 0;
    if (vec) {
        const int64_t n4 = D >> 2;
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (int64_t i = threadIdx.x; i < n4; i += RT) {
            const float4 v = r4[i];
            acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
        }
        i0 = n4 << 2;
    }
    for (int64_t i = i0 + threadIdx.x; i < D; i += RT) acc += row[i] * row[i];
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) sumsq[blockIdx.x] = tot;
}

// first index i with table[i] >= v  (torch.searchsorted, right=False), n if none
__device__ __forceinline__ int lower_bound(const float* __restrict__ table, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (table[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// sigma -> t.  Discrete schedules: left searchsorted (src/schedulers.py:185-190).  Continuous-t schedules
// (slopes != null): Interp1d of (sigmas, arange) as sigma_to_t_interp does (src/schedulers.py:210-220,
// src/torchinterp1d.py:10-154): ind = clamp(searchsorted-1, 0, n-2); t = ind + slope[ind]*(v - sigmas[ind]),
// slope[i] = 1/(eps + sigmas[i+1]-sigmas[i]) precomputed on the host in f32 exactly as the reference does.
__device__ __forceinline__ float t_lookup(const float* __restrict__ sigmas, const float* __restrict__ slopes, int n, float v) {
    const int lb = lower_bound(sigmas, n, v);
    if (slopes == nullptr) return (float)lb;
    const int ind = min(max(lb - 1, 0), n - 2);
    return (float)ind + slopes[ind] * (v - sigmas[ind]);
}

__global__ __launch_bounds__(RT) void refine_sigma_kernel(nlc_sigma_desc d) {
    __shared__ float sh[RT / 64];
    __shared__ float s_min;
    const int B = d.B;
    // pass 1: sigma and t per sample, batch minimum of t (src/experiments.py:411)
    float tmin = 3.0e38f;
    for (int b = threadIdx.x; b < B; b += RT) {
        const float raw = d.sigma_in ? d.sigma_in[b] : d.sigma_sched;
        float sg = raw;
        float tf = d.t_in ? d.t_in[b] : d.t_sched;
        if (d.refine) {
            const float norm_x = sqrtf(d.sumsq[b]) / d.sqrt_dim;
            const float min_dist = fmaxf(norm_x - d.norm_max, 0.f);
            const float max_dist = norm_x + d.norm_min;
            sg = fminf(fmaxf(raw, min_dist), max_dist);
            tf = t_lookup(d.sigmas, d.t_slopes, d.n_sigmas, sg);
            tmin = fminf(tmin, tf);
        }
        d.sigma_t[b] = sg;
        d.sigma_prev[b] = d.prev_is_ratio ? raw * d.sigma_prev_sched : d.sigma_prev_sched;
        d.t[b] = tf;   // provisional
    }
    // block minimum (wave shuffle + one LDS round)
    for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = tmin;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = sh[0];
        for (int i = 1; i < RT / 64; ++i) m = fminf(m, sh[i]);
        s_min = m;
    }
    __syncthreads();
    const float shift = (d.refine && s_min > 0.f) ? d.time_shift : 0.f;
    for (int b = threadIdx.x; b < B; b += RT) {
        float tf = d.t[b] - shift;
        tf = fminf(fmaxf(tf, 0.f), 1000.f);
        d.t[b] = tf;
        const float sg = d.sigma_t[b];
        d.c_in[b] = sqrtf(1.0f / (sg * sg + 1.0f));
    }
}

__global__ void sigma_correct_kernel(const float* __restrict__ r, int partial, const float* __restrict__ sigmas,
                                     const float* __restrict__ slopes, int n_sigmas, float* __restrict__ sigma_t,
                                     float* __restrict__ sigma_prev, float* __restrict__ t, float* __restrict__ c_in, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float st = sigma_t[b], sp = sigma_prev[b];
    const float dist_hat = st * (1.0f + r[b]);
    const float dist_prev_hat = dist_hat * (sp / st);
    float tf = t_lookup(sigmas, slopes, n_sigmas, dist_hat);
    tf = fminf(fmaxf(tf, 0.f), 1000.f);
    sigma_t[b] = dist_hat;
    if (!partial) sigma_prev[b] = dist_prev_hat;
    t[b] = tf;
    c_in[b] = sqrtf(1.0f / (dist_hat * dist_hat + 1.0f));
}

// projection_loop's sigma re-estimation (image_sample.py:485-497), one thread per sample:
//   cur_norm = ||x_prev|| / sqrt(D) ; cur_dist = sqrt(cur_norm^2 + norm_max^2 - 2 cur_norm norm_max cos + 1e-8)
//   sigma <- term0 + r1*sigma_prev + r2*(sigma_t * cur_norm/last_norm) + r3*cur_dist ; t <- lookup(sigma)
__global__ void proj_sigma_kernel(const float* __restrict__ sumsq, float sqrt_dim, float norm_max, float norm_max_sq,
                                  float costheta, float term0, float r1, float r2, float r3,
                                  const float* __restrict__ sigmas, const float* __restrict__ slopes, int n_sigmas,
                                  float* __restrict__ last_norm, float* __restrict__ sigma_t,
                                  const float* __restrict__ sigma_prev, float* __restrict__ t, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float cur_norm = sqrtf(sumsq[b]) / sqrt_dim;
    const float cur_dist = sqrtf(((cur_norm * cur_norm + norm_max_sq) - ((2.0f * cur_norm) * norm_max) * costheta) + 1e-8f);
    const float norm_ratio = cur_norm / last_norm[b];
    const float s1 = sigma_prev[b];
    const float s2 = sigma_t[b] * norm_ratio;
    const float sg = ((term0 + r1 * s1) + r2 * s2) + r3 * cur_dist;
    sigma_t[b] = sg;
    t[b] = t_lookup(sigmas, slopes, n_sigmas, sg);
    last_norm[b] = cur_norm;
}

// ---- exact per-sample quantile of |x| (torch.quantile, linear interpolation) by radix select.
//      One workgroup per sample; 4 passes over 8-bit digits of the (non-negative) float bit pattern.
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__global__ __launch_bounds__(RT) void quantile_kernel(const float* __restrict__ x, float q, float max_value,
                                                     float* __restrict__ s_out, int64_t D) {
    // one histogram per wave: the digits of a pass cluster in a handful of bins (pass 0 sees the exponents of one image), and
    // 1024 threads hammering the same LDS words serialised the whole pass (257 us per launch with a single histogram)
    constexpr int NW = RT / 64;
    __shared__ unsigned histw[NW][256];
    unsigned* hist = histw[0];
    __shared__ unsigned s_prefix, s_k, s_cnt_le, s_next;
    const int wv = threadIdx.x >> 6;
    const float* row = x + (int64_t)blockIdx.x * D;
    // rank arithmetic in f32 exactly as ATen does: rank = q * (n-1)
    const float rank = q * (float)(D - 1);
    const float rank_lo = floorf(rank);
    const float w = rank - rank_lo;
    unsigned k = (unsigned)rank_lo;            // 0-based order statistic
    unsigned prefix = 0, mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < NW * 256; i += RT) (&histw[0][0])[i] = 0;
        __syncthreads();
        // eight loads in flight per thread (one per iteration made every pass a chain of 192 dependent L2 round trips)
        int64_t i = threadIdx.x;
        for (; i + 7 * RT < D; i += 8 * RT) {
            unsigned u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = abs_bits(row[i + e * RT]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if ((u[e] & mask) == prefix) atomicAdd(&histw[wv][(u[e] >> shift) & 255u], 1u);
        }
        for (; i < D; i += RT) {
            const unsigned u = abs_bits(row[i]);
            if ((u & mask) == prefix) atomicAdd(&histw[wv][(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 256) {                     // fold the per-wave histograms (integer counts: order-free)
            unsigned a = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += histw[w][threadIdx.x];
            hist[threadIdx.x] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned acc = 0; int bin = 0;
            for (; bin < 256; ++bin) {
                if (acc + hist[bin] > k) break;
                acc += hist[bin];
            }
            s_k = k - acc;
            s_prefix = prefix | ((unsigned)bin << shift);
        }
        __syncthreads();
        k = s_k; prefix = s_prefix; mask |= (255u << shift);
        __syncthreads();
    }
    // prefix is now the bit pattern of sorted[lo]; find sorted[lo+1]
    if (threadIdx.x == 0) { s_cnt_le = 0; s_next = 0x7fffffffu; }
    __syncthreads();
    unsigned cnt = 0, nxt = 0x7fffffffu;
    {
        int64_t i = threadIdx.x;
        for (; i + 7 * RT < D; i += 8 * RT) {
            unsigned u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = abs_bits(row[i + e * RT]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { if (u[e] <= prefix) ++cnt; else nxt = min(nxt, u[e]); }
        }
        for (; i < D; i += RT) {
            const unsigned u = abs_bits(row[i]);
            if (u <= prefix) ++cnt; else nxt = min(nxt, u);
        }
    }
    atomicAdd(&s_cnt_le, cnt);
    atomicMin(&s_next, nxt);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v_lo = __uint_as_float(prefix);
        const unsigned lo_idx = (unsigned)rank_lo;
        float v_hi = v_lo;
        if (ceilf(rank) > rank_lo && s_cnt_le <= lo_idx + 1) v_hi = __uint_as_float(s_next);
        // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
        const float diff = v_hi - v_lo;
        float s = (w < 0.5f) ? (v_lo + w * diff) : (v_hi - diff * (1.0f - w));
        s = fminf(fmaxf(s, 1.0f), max_value);
        s_out[blockIdx.x] = s;
    }
}

// ---- the same quantile with G workgroups per sample (nlc_dynamic_threshold_ws): the single-workgroup form above reads a sample five
//      times through ONE CU (16 CUs busy at B = 16: 175 us for 12.6 MB).  Here every digit pass is two launches - slices of the sample
//      histogrammed by G workgroups into a global per-sample histogram (integer atomics: order-free, exact), then one workgroup per
//      sample scans it, picks the bin and clears it - and the kernel boundary is the only synchronisation: no workgroup ever waits for
//      another.  Workspace per sample (u32): [4][256] histograms + {prefix, k, count <= value, 0x7fffffff - next larger value, ...};
//      all zero on entry, all zero again on exit.  Same arithmetic as above, so the result is bit-identical to it and to torch.quantile.
constexpr int QW_WORDS = 4 * 256 + 8;
constexpr int QT = 256;

__device__ __forceinline__ unsigned q_rank0(float q, int64_t D) { return (unsigned)floorf(q * (float)(D - 1)); }

__global__ __launch_bounds__(QT) void qhist_kernel(const float* __restrict__ x, int64_t D, int pass, unsigned* __restrict__ ws) {
    constexpr int NW = QT / 64;
    __shared__ unsigned histw[NW][256];
    const int g = blockIdx.x, G = gridDim.x, b = blockIdx.y, wv = threadIdx.x >> 6;
    unsigned* w = ws + (int64_t)b * QW_WORDS;
    const unsigned prefix = pass ? w[1024] : 0u;
    const unsigned mask = pass ? (0xffffffffu << (32 - 8 * pass)) : 0u;
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < NW * 256; i += QT) (&histw[0][0])[i] = 0;
    __syncthreads();
    const float* row = x + (int64_t)b * D;
    const int64_t i1 = D * (g + 1) / G;
    int64_t i = D * g / G + threadIdx.x;
    for (; i + 7 * QT < i1; i += 8 * QT) {
        unsigned u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = abs_bits(row[i + e * QT]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if ((u[e] & mask) == prefix) atomicAdd(&histw[wv][(u[e] >> shift) & 255u], 1u);
    }
    for (; i < i1; i += QT) {
        const unsigned u = abs_bits(row[i]);
        if ((u & mask) == prefix) atomicAdd(&histw[wv][(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    unsigned a = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) a += histw[k][threadIdx.x];
    if (a) atomicAdd(&w[pass * 256 + threadIdx.x], a);
}

__global__ __launch_bounds__(QT) void qscan_kernel(float q, int64_t D, int pass, unsigned* __restrict__ ws) {
    __shared__ unsigned sc[QT];
    unsigned* w = ws + (int64_t)blockIdx.x * QW_WORDS;
    const int tid = threadIdx.x, shift = 24 - 8 * pass;
    const unsigned prefix = pass ? w[1024] : 0u;
    const unsigned k = pass ? w[1025] : q_rank0(q, D);
    const unsigned c = w[pass * 256 + tid];
    w[pass * 256 + tid] = 0;                                   // the histogram is zero again for the next call
    sc[tid] = c;
    for (int off = 1; off < QT; off <<= 1) {                   // inclusive prefix sum over the 256 bins
        __syncthreads();
        const unsigned v = tid >= off ? sc[tid - off] : 0u;
        __syncthreads();
        sc[tid] += v;
    }
    const unsigned incl = sc[tid], excl = incl - c;
    if (excl <= k && k < incl) {                               // exactly one bin holds order statistic k (everyone has read prefix / k)
        w[1024] = prefix | ((unsigned)tid << shift);
        w[1025] = k - excl;
    }
}

__global__ __launch_bounds__(QT) void qcount_kernel(const float* __restrict__ x, int64_t D, unsigned* __restrict__ ws) {
    __shared__ unsigned s_cnt, s_nxt;
    const int g = blockIdx.x, G = gridDim.x, b = blockIdx.y;
    unsigned* w = ws + (int64_t)b * QW_WORDS;
    const unsigned prefix = w[1024];                           // bit pattern of sorted[lo]
    if (threadIdx.x == 0) { s_cnt = 0; s_nxt = 0x7fffffffu; }
    __syncthreads();
    const float* row = x + (int64_t)b * D;
    const int64_t i1 = D * (g + 1) / G;
    unsigned cnt = 0, nxt = 0x7fffffffu;
    int64_t i = D * g / G + threadIdx.x;
    for (; i + 7 * QT < i1; i += 8 * QT) {
        unsigned u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = abs_bits(row[i + e * QT]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (u[e] <= prefix) ++cnt; else nxt = min(nxt, u[e]); }
    }
    for (; i < i1; i += QT) {
        const unsigned u = abs_bits(row[i]);
        if (u <= prefix) ++cnt; else nxt = min(nxt, u);
    }
    atomicAdd(&s_cnt, cnt);
    atomicMin(&s_nxt, nxt);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cnt) atomicAdd(&w[1026], s_cnt);
        atomicMax(&w[1027], 0x7fffffffu - s_nxt);              // zero-initialised workspace: the maximum of (0x7fffffff - value)
    }
}

__global__ void qfinish_kernel(float q, float max_value, int64_t D, float* __restrict__ s_out, unsigned* __restrict__ ws, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned* w = ws + (int64_t)b * QW_WORDS;
    const unsigned prefix = w[1024], cnt_le = w[1026], nxt = 0x7fffffffu - w[1027];
    w[1024] = 0; w[1025] = 0; w[1026] = 0; w[1027] = 0;
    const float rank = q * (float)(D - 1);
    const float rank_lo = floorf(rank);
    const float wq = rank - rank_lo;
    const float v_lo = __uint_as_float(prefix);
    const unsigned lo_idx = (unsigned)rank_lo;
    float v_hi = v_lo;
    if (ceilf(rank) > rank_lo && cnt_le <= lo_idx + 1) v_hi = __uint_as_float(nxt);
    const float diff = v_hi - v_lo;                            // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    float s = (wq < 0.5f) ? (v_lo + wq * diff) : (v_hi - diff * (1.0f - wq));
    s_out[b] = fminf(fmaxf(s, 1.0f), max_value);
}

// ---- fused scheduler update ---------------------------------------------------------------
struct SchedScalars {
    float st, sp, eps_mul, eps_div, min_lv, max_lv, abp;
    bool norm;
};
__device__ __forceinline__ SchedScalars sched_scalars(const nlc_sched_desc& d, int b) {
    SchedScalars s;
    s.st = d.sigma_t[b]; s.sp = d.sigma_prev[b];
    s.norm = d.eps_norm_sumsq != nullptr;
    s.eps_mul = sqrtf((float)((int64_t)d.C * d.HW));
    s.eps_div = s.norm ? fmaxf(sqrtf(d.eps_norm_sumsq[b]), 1e-12f) : 1.f;
    // get_eps_logvar, src/schedulers.py:367-380
    const float st2 = s.st * s.st, sp2 = s.sp * s.sp;
    float beta_t = fabsf((st2 - sp2) / (st2 + 1.0f));
    beta_t = fmaxf(beta_t, 1e-20f);
    const float alpha_t = 1.0f / (st2 + 1.0f);
    const float alpha_prev = 1.0f / (sp2 + 1.0f);
    float coef = (1.0f - alpha_prev) / (1.0f - alpha_t);
    coef = fminf(fmaxf(coef, 0.f), 1.f);
    const float post_var = beta_t * coef;
    s.max_lv = logf(beta_t);
    s.min_lv = logf(fmaxf(post_var, d.min_var_coef));
    s.abp = alpha_prev;
    return s;
}
__device__ __forceinline__ float sched_eps(const nlc_sched_desc& d, const SchedScalars& s, int b, int c, int p) {
    float e = d.eps_out[((int64_t)b * d.Cnet + c) * d.HW + p];
    if (s.norm) e = (s.eps_mul * e) / s.eps_div;      // utils.normalize: sqrt(D)*x/denom
    return e;
}

__global__ void sched_x0_kernel(const nlc_sched_desc d) {
    const int b = blockIdx.y;
    const SchedScalars s = sched_scalars(d, b);
    const int n = d.C * d.HW;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        const int c = i / d.HW, p = i - c * d.HW;
        const float e = sched_eps(d, s, b, c, p);
        const int64_t o = (int64_t)b * n + i;
        d.x0[o] = d.xt[o] - s.st * e;                 // pred_xstart, src/schedulers.py:407-409
        if (d.eps_used) d.eps_used[o] = e;
    }
}

__global__ void sched_step_kernel(const nlc_sched_desc d, int* nan_flag) {
    const int b = blockIdx.y;
    const SchedScalars s = sched_scalars(d, b);
    const int n = d.C * d.HW;
    const float sp2 = s.sp * s.sp;
    const float dyn = d.dyn_s ? d.dyn_s[b] : 1.f;
    const float simple_sig = sqrtf(1.0f - d.eta * d.eta);   // math.sqrt(1-eta**2), python double -> f32 scalar
    bool saw_nan = false;
    const bool do_clip = d.phases == 0 || (d.phases & 1), do_step = d.phases == 0 || (d.phases & 2);
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        const int c = i / d.HW, p = i - c * d.HW;
        const int64_t o = (int64_t)b * n + i;
        float e = sched_eps(d, s, b, c, p);
        const float xt = d.xt[o];
        float x0 = d.x0[o];                                  // pre-clip x0_hat from phase 0
        if (do_clip) {
            if (d.clip == NLC_CLIP_CLAMP) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            else if (d.clip == NLC_CLIP_DYNAMIC) x0 = fminf(fmaxf(x0, -dyn), dyn) / dyn;
            if (d.mask) { if (d.mask[(int64_t)c * d.HW + p] != 0.f) x0 = d.known[o]; }
            d.x0[o] = x0;
        }
        if (!do_step) continue;
        float lv = 0.f;
        if (d.logvar_ext) lv = d.logvar_ext[o];
        else if (d.var_mode == NLC_VAR_LEARNED) {
            const float v = d.eps_out[((int64_t)b * d.Cnet + d.C + c) * d.HW + p];
            const float frac = (v + 1.0f) / 2.0f;
            lv = frac * s.max_lv + (1.0f - frac) * s.min_lv;
        } else if (d.var_mode == NLC_VAR_FIXEDSMALL) lv = s.min_lv;
        else if (d.var_mode == NLC_VAR_FIXEDLARGE) lv = s.max_lv;
        float z = d.noise ? d.noise[o] : 0.f;
        float xp;
        switch (d.variant) {
            case NLC_SCHED_DDIM: {
                float nsig = 0.f;
                if (d.eta > 0.f) { nsig = d.eta * expf(0.5f * lv) / sqrtf(s.abp); if (!(s.sp > 0.f)) z = 0.f; }
                else z = 0.f;
                const float sig = sqrtf(fmaxf(sp2 - nsig * nsig, 0.f));
                const float nsig2 = sqrtf(sp2 - sig * sig);
                xp = x0 + sig * e + nsig2 * z;
            } break;
            case NLC_SCHED_DDIM_SIMPLE_ORIG:
            case NLC_SCHED_DDIM_SIMPLE: {
                if (d.variant == NLC_SCHED_DDIM_SIMPLE_ORIG) e = (xt - x0) / s.st;
                const float sig = simple_sig * s.sp;
                xp = x0 + sig * e;
                if (d.eta > 0.f) xp = xp + (d.eta * s.sp) * z;
            } break;
            case NLC_SCHED_DDIM_SIMPLE_DRAG: {
                e = (xt - x0) / s.st;
                xp = x0 + s.sp * e;
                if (d.eta > 0.f) xp = xp + (d.eta * s.sp) * z;
            } break;
            case NLC_SCHED_DDPM: {
                const float nsig = expf(0.5f * lv) / sqrtf(s.abp);
                const float sig = sqrtf(fmaxf(sp2 - nsig * nsig, 0.f));
                xp = x0 + sig * e;
                if (!(s.sp > 0.f)) z = 0.f;
                xp = xp + nsig * z;
            } break;
            case NLC_SCHED_DDIM_ORIG: {
                e = (xt - x0) / s.st;
                float nsig = 0.f;
                if (d.eta > 0.f) { nsig = d.eta * expf(0.5f * lv) / sqrtf(s.abp); if (!(s.sp > 0.f)) z = 0.f; }
                else z = 0.f;
                const float sig = sqrtf(fmaxf(sp2 - nsig * nsig, 0.f));
                xp = x0 + sig * e + nsig * z;
            } break;
            default: {   // NLC_SCHED_DDPM_ORIG
                const float ab = 1.0f / (s.st * s.st + 1.0f);
                const float abp = s.abp;
                const float a_t = ab / abp;
                const float beta_t = 1.0f - a_t;
                const float zt = xt * sqrtf(ab);
                const float c1 = beta_t * sqrtf(abp) / (1.0f - ab);
                const float c2 = (1.0f - abp) * sqrtf(a_t) / (1.0f - ab);
                const float mean = c1 * x0 + c2 * zt;
                const float m = (s.sp > 0.f) ? 1.f : 0.f;
                const float zp = mean + m * expf(0.5f * lv) * z;
                xp = zp / sqrtf(abp);
            } break;
        }
        d.x_prev[o] = xp;
        if (d.eps_used) d.eps_used[o] = e;
        saw_nan |= (xp != xp);
    }
    if (nan_flag && saw_nan) atomicOr(nan_flag, 1);
}

__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ scale, float scalar,
                                  float* __restrict__ out, int64_t D) {
    const int b = blockIdx.y;
    const float sc = scale ? scale[b] * scalar : scalar;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < D; i += (int64_t)gridDim.x * NT)
        out[(int64_t)b * D + i] = x[(int64_t)b * D + i] * sc;
}

// out[b,:] = ca[b]*x[b,:] + cb[b]*y[b,:], two roundings of the products and one of the sum (this file is compiled with
// -ffp-contract=off): the forward process x_n = x_0 sqrt(abar_t) + noise sqrt(1 - abar_t), src/schedulers.py:323-329
__global__ void lincomb_rows_kernel(const float* __restrict__ x, const float* __restrict__ ca, const float* __restrict__ y,
                                    const float* __restrict__ cb, float* __restrict__ out, int64_t D) {
    const int b = blockIdx.y;
    const float a = ca[b], c = cb ? cb[b] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < D; i += (int64_t)gridDim.x * NT) {
        const int64_t o = (int64_t)b * D + i;
        out[o] = y ? x[o] * a + y[o] * c : x[o] * a;
    }
}

int check_sched(const nlc_sched_desc* d, const char* name) {
    if (!d) { nlc_set_error("%s: null descriptor", name); return NLC_EINVAL; }
    if (!d->xt || !d->eps_out || !d->sigma_t || !d->sigma_prev || !d->x0) { nlc_set_error("%s: null pointer", name); return NLC_EINVAL; }
    if (d->B <= 0 || d->C <= 0 || d->HW <= 0 || d->Cnet < d->C) { nlc_set_error("%s: bad dims", name); return NLC_EINVAL; }
    if (d->B > 65535) { nlc_set_error("%s: B too large", name); return NLC_EINVAL; }
    if ((int64_t)d->C * d->HW >= (1ll << 31)) { nlc_set_error("%s: sample too large", name); return NLC_EINVAL; }
    if (d->variant < NLC_SCHED_DDIM || d->variant > NLC_SCHED_DDIM_ORIG) { nlc_set_error("%s: bad variant %d", name, d->variant); return NLC_EINVAL; }
    // (a caller-supplied log-variance overrides var_mode)
    if (d->var_mode == NLC_VAR_LEARNED && !d->logvar_ext && d->Cnet < 2 * d->C) { nlc_set_error("%s: learned variance needs Cnet >= 2C", name); return NLC_EINVAL; }
    if ((d->mask == nullptr) != (d->known == nullptr)) { nlc_set_error("%s: mask/known must come together", name); return NLC_EINVAL; }
    return NLC_OK;
}

}  // namespace

extern "C" int nlc_row_sumsq(const float* x, float* sumsq, int B, int64_t row_stride, int64_t D, void* stream) {
    NLC_REQUIRE(x && sumsq && B > 0 && D > 0 && row_stride >= D, "nlc_row_sumsq: bad arguments");
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(B), dim3(RT), 0, (hipStream_t)stream, x, sumsq, row_stride, D);
    NLC_CHECK_LAUNCH("nlc_row_sumsq");
    return NLC_OK;
}

extern "C" int nlc_refine_sigma_ex(const nlc_sigma_desc* d, void* stream) {
    NLC_REQUIRE(d, "nlc_refine_sigma_ex: null descriptor");
    NLC_REQUIRE(d->sigma_t && d->sigma_prev && d->t && d->c_in && d->B > 0, "nlc_refine_sigma_ex: null output / bad B");
    NLC_REQUIRE(!d->refine || (d->sumsq && d->sigmas && d->n_sigmas > 1), "nlc_refine_sigma_ex: refine needs sumsq and the sigma table");
    NLC_REQUIRE(!d->prev_is_ratio || d->sigma_prev_sched >= 0.f, "nlc_refine_sigma_ex: negative sigma_prev ratio");
    hipLaunchKernelGGL(refine_sigma_kernel, dim3(1), dim3(RT), 0, (hipStream_t)stream, *d);
    NLC_CHECK_LAUNCH("nlc_refine_sigma_ex");
    return NLC_OK;
}

extern "C" int nlc_refine_sigma(const float* sumsq, float sqrt_dim, float norm_max, float norm_min, float sigma_sched,
                                float sigma_prev_sched, int refine, const float* sigmas, int n_sigmas, int t_sched,
                                int time_shift, float* sigma_t, float* sigma_prev, float* t, float* c_in, int B,
                                void* stream) {
    nlc_sigma_desc d{};
    d.sumsq = sumsq; d.sqrt_dim = sqrt_dim; d.norm_max = norm_max; d.norm_min = norm_min; d.sigma_sched = sigma_sched;
    d.sigma_prev_sched = sigma_prev_sched; d.refine = refine; d.sigmas = sigmas; d.n_sigmas = n_sigmas;
    d.t_sched = (float)t_sched; d.time_shift = (float)time_shift; d.sigma_t = sigma_t; d.sigma_prev = sigma_prev; d.t = t;
    d.c_in = c_in; d.B = B;
    return nlc_refine_sigma_ex(&d, stream);
}

extern "C" int nlc_sigma_correct(const float* r, int partial, const float* sigmas, const float* t_slopes, int n_sigmas,
                                 float* sigma_t, float* sigma_prev, float* t, float* c_in, int B, void* stream) {
    NLC_REQUIRE(r && sigmas && sigma_t && sigma_prev && t && c_in && B > 0 && n_sigmas > 1, "nlc_sigma_correct: bad arguments");
    hipLaunchKernelGGL(sigma_correct_kernel, dim3(cdiv(B, NT)), dim3(NT), 0, (hipStream_t)stream, r, partial, sigmas, t_slopes,
                       n_sigmas, sigma_t, sigma_prev, t, c_in, B);
    NLC_CHECK_LAUNCH("nlc_sigma_correct");
    return NLC_OK;
}

extern "C" int nlc_proj_sigma(const float* sumsq, float sqrt_dim, float norm_max, float norm_max_sq, float costheta,
                              float term0, float r1, float r2, float r3, const float* sigmas, const float* t_slopes,
                              int n_sigmas, float* last_norm, float* sigma_t, const float* sigma_prev, float* t, int B,
                              void* stream) {
    NLC_REQUIRE(sumsq && sigmas && last_norm && sigma_t && sigma_prev && t && B > 0 && n_sigmas > 1, "nlc_proj_sigma: bad arguments");
    hipLaunchKernelGGL(proj_sigma_kernel, dim3(cdiv(B, NT)), dim3(NT), 0, (hipStream_t)stream, sumsq, sqrt_dim, norm_max,
                       norm_max_sq, costheta, term0, r1, r2, r3, sigmas, t_slopes, n_sigmas, last_norm, sigma_t, sigma_prev, t, B);
    NLC_CHECK_LAUNCH("nlc_proj_sigma");
    return NLC_OK;
}

extern "C" int nlc_dynamic_threshold(const float* x0_hat, float q, float max_value, float* s_out, int B, int64_t D,
                                     void* stream) {
    NLC_REQUIRE(x0_hat && s_out && B > 0 && D > 1, "nlc_dynamic_threshold: bad arguments");
    NLC_REQUIRE(q >= 0.f && q <= 1.f, "nlc_dynamic_threshold: q out of range");
    NLC_REQUIRE(D < (1ll << 24), "nlc_dynamic_threshold: D too large for exact f32 rank arithmetic");
    hipLaunchKernelGGL(quantile_kernel, dim3(B), dim3(RT), 0, (hipStream_t)stream, x0_hat, q, max_value, s_out, D);
    NLC_CHECK_LAUNCH("nlc_dynamic_threshold");
    return NLC_OK;
}

extern "C" int64_t nlc_dynamic_threshold_ws_bytes(int B) { return B > 0 ? (int64_t)B * QW_WORDS * 4 : 0; }

extern "C" int nlc_dynamic_threshold_ws(const float* x0_hat, float q, float max_value, float* s_out, int B, int64_t D,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    NLC_REQUIRE(x0_hat && s_out && B > 0 && D > 1, "nlc_dynamic_threshold_ws: bad arguments");
    NLC_REQUIRE(q >= 0.f && q <= 1.f, "nlc_dynamic_threshold_ws: q out of range");
    NLC_REQUIRE(D < (1ll << 24), "nlc_dynamic_threshold_ws: D too large for exact f32 rank arithmetic");
    NLC_REQUIRE(B <= 65535, "nlc_dynamic_threshold_ws: B too large");
    NLC_REQUIRE(workspace && workspace_bytes >= nlc_dynamic_threshold_ws_bytes(B) && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0,
                "nlc_dynamic_threshold_ws: workspace of nlc_dynamic_threshold_ws_bytes(B) bytes (zeroed once by the caller) required");
    unsigned* ws = reinterpret_cast<unsigned*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    int G = 512 / B;                                           // ~two workgroups per CU over the whole batch
    const int gmax = (int)((D + 8 * QT - 1) / (8 * QT));       // ... each with at least one full trip of its load loop
    if (G > gmax) G = gmax;
    if (G < 1) G = 1;
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(qhist_kernel, dim3(G, B), dim3(QT), 0, st, x0_hat, D, pass, ws);
        hipLaunchKernelGGL(qscan_kernel, dim3(B), dim3(QT), 0, st, q, D, pass, ws);
    }
    hipLaunchKernelGGL(qcount_kernel, dim3(G, B), dim3(QT), 0, st, x0_hat, D, ws);
    hipLaunchKernelGGL(qfinish_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, q, max_value, D, s_out, ws, B);
    NLC_CHECK_LAUNCH("nlc_dynamic_threshold_ws");
    return NLC_OK;
}

extern "C" int nlc_sched_x0(const nlc_sched_desc* d, void* stream) {
    int rc = check_sched(d, "nlc_sched_x0"); if (rc) return rc;
    int gx = cdiv((int64_t)d->C * d->HW, NT * 4); if (gx < 1) gx = 1; if (gx > 256) gx = 256;
    hipLaunchKernelGGL(sched_x0_kernel, dim3(gx, d->B), dim3(NT), 0, (hipStream_t)stream, *d);
    NLC_CHECK_LAUNCH("nlc_sched_x0");
    return NLC_OK;
}

extern "C" int nlc_sched_step(const nlc_sched_desc* d, int* nan_flag, void* stream) {
    int rc = check_sched(d, "nlc_sched_step"); if (rc) return rc;
    NLC_REQUIRE(d->x_prev || d->phases == 1, "nlc_sched_step: null x_prev");
    NLC_REQUIRE(d->phases >= 0 && d->phases <= 3, "nlc_sched_step: bad phases %d", d->phases);
    NLC_REQUIRE(d->clip >= NLC_CLIP_NONE && d->clip <= NLC_CLIP_DYNAMIC, "nlc_sched_step: bad clip %d", d->clip);
    NLC_REQUIRE(d->clip != NLC_CLIP_DYNAMIC || d->dyn_s, "nlc_sched_step: dynamic clip needs dyn_s");
    NLC_REQUIRE(d->var_mode >= NLC_VAR_NONE && d->var_mode <= NLC_VAR_LEARNED, "nlc_sched_step: bad var_mode %d", d->var_mode);
    const bool needs_noise = (d->variant == NLC_SCHED_DDPM || d->variant == NLC_SCHED_DDPM_ORIG || d->eta > 0.f);
    const bool stepping = d->phases == 0 || (d->phases & 2);
    NLC_REQUIRE(!stepping || !needs_noise || d->noise, "nlc_sched_step: this variant/eta needs a noise tensor");
    NLC_REQUIRE(!stepping || !needs_noise || d->var_mode != NLC_VAR_NONE || d->logvar_ext || d->variant == NLC_SCHED_DDIM_SIMPLE ||
                    d->variant == NLC_SCHED_DDIM_SIMPLE_ORIG || d->variant == NLC_SCHED_DDIM_SIMPLE_DRAG,
                "nlc_sched_step: stochastic variant with sampler_var 'none' (the reference raises here too)");
    int gx = cdiv((int64_t)d->C * d->HW, NT * 4); if (gx < 1) gx = 1; if (gx > 256) gx = 256;
    hipLaunchKernelGGL(sched_step_kernel, dim3(gx, d->B), dim3(NT), 0, (hipStream_t)stream, *d, nan_flag);
    NLC_CHECK_LAUNCH("nlc_sched_step");
    return NLC_OK;
}

extern "C" int nlc_lincomb_rows(const float* x, const float* ca, const float* y, const float* cb, float* out, int B, int64_t D,
                                void* stream) {
    NLC_REQUIRE(x && ca && out && B > 0 && D > 0 && B <= 65535, "nlc_lincomb_rows: bad arguments");
    NLC_REQUIRE((y == nullptr) == (cb == nullptr), "nlc_lincomb_rows: y / cb must come together");
    int gx = cdiv(D, NT * 4); if (gx < 1) gx = 1; if (gx > 256) gx = 256;
    hipLaunchKernelGGL(lincomb_rows_kernel, dim3(gx, B), dim3(NT), 0, (hipStream_t)stream, x, ca, y, cb, out, D);
    NLC_CHECK_LAUNCH("nlc_lincomb_rows");
    return NLC_OK;
}

extern "C" int nlc_scale_rows(const float* x, const float* scale, float scalar, float* out, int B, int64_t D, void* stream) {
    NLC_REQUIRE(x && out && B > 0 && D > 0 && B <= 65535, "nlc_scale_rows: bad arguments");
    int gx = cdiv(D, NT * 4); if (gx < 1) gx = 1; if (gx > 256) gx = 256;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(gx, B), dim3(NT), 0, (hipStream_t)stream, x, scale, scalar, out, D);
    NLC_CHECK_LAUNCH("nlc_scale_rows");
    return NLC_OK;
}
