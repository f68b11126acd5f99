// EDM / Heun + NLC sampler state kernels (src/experiments.py:777-918).  The reference keeps the state
// and eps in FLOAT64 and runs the network in float32 (:860,872,789-802); these kernels do the same:
// f64 elementwise updates with per-sample f64 coefficients, f32 preconditioning scalars.
// HBM-bound, one pass each.  Compiled with -ffp-contract=off (op-by-op restatement).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int RT = 1024;

__global__ void cast_f64_f32_kernel(const double* __restrict__ x, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) out[i] = (float)x[i];
}

__global__ __launch_bounds__(RT) void row_sumsq_f64_kernel(const double* __restrict__ x, double* __restrict__ sumsq, int64_t D) {
    __shared__ double sh[RT / 64];
    const double* row = x + (int64_t)blockIdx.x * D;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < D; i += RT) acc += row[i] * row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < RT / 64; ++i) t += sh[i];
        sumsq[blockIdx.x] = t;
    }
}

// torch.nn.CosineSimilarity(dim=1, eps) of two [B, D] f64 tensors (src/experiments.py:870,912-915; ATen
// cosine_similarity: each operand is divided by max(||.||, eps) and the quotients are dotted; one workgroup per row,
// two passes: norms, then the dot of the normalised rows).
__global__ __launch_bounds__(RT) void row_cosine_f64_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                           double eps, double* __restrict__ out, int64_t D) {
    __shared__ double sh[3][RT / 64];
    __shared__ double s_na, s_nb;
    const double* ra = a + (int64_t)blockIdx.x * D;
    const double* rb = b + (int64_t)blockIdx.x * D;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    double aa = 0.0, bb = 0.0;
    for (int64_t i = threadIdx.x; i < D; i += RT) { aa += ra[i] * ra[i]; bb += rb[i] * rb[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { aa += __shfl_xor(aa, o, 64); bb += __shfl_xor(bb, o, 64); }
    if (l == 0) { sh[0][w] = aa; sh[1][w] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < RT / 64; ++i) { ta += sh[0][i]; tb += sh[1][i]; }
        s_na = fmax(sqrt(ta), eps); s_nb = fmax(sqrt(tb), eps);
    }
    __syncthreads();
    const double na = s_na, nb = s_nb;
    double ab = 0.0;
    for (int64_t i = threadIdx.x; i < D; i += RT) ab += (ra[i] / na) * (rb[i] / nb);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ab += __shfl_xor(ab, o, 64);
    if (l == 0) sh[2][w] = ab;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < RT / 64; ++i) t += sh[2][i];
        out[blockIdx.x] = t;
    }
}

// encode_edm / pred_edm scalars (src/experiments.py:779-783,790-797): sigma is cast to f32 first.
__global__ void edm_scalars_kernel(const double* __restrict__ sigma, float sigma_data, float* __restrict__ c_in,
                                   float* __restrict__ c_noise, float* __restrict__ c_skip, float* __restrict__ c_out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = (float)sigma[b];
    const float sd2 = sigma_data * sigma_data;
    const float s2 = s * s;
    c_skip[b] = sd2 / (s2 + sd2);
    c_out[b] = s * sigma_data / sqrtf(s2 + sd2);
    c_in[b] = 1.0f / sqrtf(sd2 + s2);
    c_noise[b] = logf(s) / 4.0f;
}

// D_x = c_skip*x32 + c_out*F (f32, :801) ; denoised = (double)D_x ; eps = (x - denoised)/sigma_div (:836-840)
__global__ void edm_eps_kernel(const double* __restrict__ x, const float* __restrict__ x32, const float* __restrict__ F,
                               const float* __restrict__ c_skip, const float* __restrict__ c_out,
                               const double* __restrict__ sigma_div, double* __restrict__ eps, double* __restrict__ denoised,
                               int64_t D) {
    const int b = blockIdx.y;
    const float cs = c_skip[b], co = c_out[b];
    const double sd = sigma_div[b];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < D; i += (int64_t)gridDim.x * NT) {
        const int64_t o = (int64_t)b * D + i;
        const float dx = cs * x32[o] + co * F[o];
        const double dn = (double)dx;
        if (denoised) denoised[o] = dn;
        eps[o] = (x[o] - dn) / sd;
    }
}

// out = ca[b]*x (+ cb[b]*y)
__global__ void f64_lincomb_kernel(const double* __restrict__ x, const double* __restrict__ ca, const double* __restrict__ y,
                                   const double* __restrict__ cb, double* __restrict__ out, int64_t D) {
    const int b = blockIdx.y;
    const double a = ca[b];
    const double c = y ? cb[b] : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < D; i += (int64_t)gridDim.x * NT) {
        const int64_t o = (int64_t)b * D + i;
        double v = a * x[o];
        if (y) v = v + c * y[o];
        out[o] = v;
    }
}

inline int gridx(int64_t D) { int g = cdiv(D, NT * 4); if (g < 1) g = 1; if (g > 256) g = 256; return g; }

}  // namespace

extern "C" int nlc_cast_f64_f32(const double* x, float* out, int64_t n, void* stream) {
    NLC_REQUIRE(x && out && n > 0, "nlc_cast_f64_f32: bad arguments");
    int g = cdiv(n, NT * 4); if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, x, out, n);
    NLC_CHECK_LAUNCH("nlc_cast_f64_f32");
    return NLC_OK;
}

extern "C" int nlc_row_sumsq_f64(const double* x, double* sumsq, int B, int64_t D, void* stream) {
    NLC_REQUIRE(x && sumsq && B > 0 && D > 0, "nlc_row_sumsq_f64: bad arguments");
    hipLaunchKernelGGL(row_sumsq_f64_kernel, dim3(B), dim3(RT), 0, (hipStream_t)stream, x, sumsq, D);
    NLC_CHECK_LAUNCH("nlc_row_sumsq_f64");
    return NLC_OK;
}

extern "C" int nlc_row_cosine_f64(const double* a, const double* b, double eps, double* out, int B, int64_t D, void* stream) {
    NLC_REQUIRE(a && b && out && B > 0 && D > 0 && eps >= 0.0, "nlc_row_cosine_f64: bad arguments");
    hipLaunchKernelGGL(row_cosine_f64_kernel, dim3(B), dim3(RT), 0, (hipStream_t)stream, a, b, eps, out, D);
    NLC_CHECK_LAUNCH("nlc_row_cosine_f64");
    return NLC_OK;
}

extern "C" int nlc_edm_scalars(const double* sigma, float sigma_data, float* c_in, float* c_noise, float* c_skip,
                               float* c_out, int B, void* stream) {
    NLC_REQUIRE(sigma && c_in && c_noise && c_skip && c_out && B > 0, "nlc_edm_scalars: bad arguments");
    hipLaunchKernelGGL(edm_scalars_kernel, dim3(cdiv(B, NT)), dim3(NT), 0, (hipStream_t)stream, sigma, sigma_data, c_in, c_noise,
                       c_skip, c_out, B);
    NLC_CHECK_LAUNCH("nlc_edm_scalars");
    return NLC_OK;
}

extern "C" int nlc_edm_eps(const double* x, const float* x32, const float* F, const float* c_skip, const float* c_out,
                           const double* sigma_div, double* eps, double* denoised, int B, int64_t D, void* stream) {
    NLC_REQUIRE(x && x32 && F && c_skip && c_out && sigma_div && eps && B > 0 && D > 0 && B <= 65535, "nlc_edm_eps: bad arguments");
    hipLaunchKernelGGL(edm_eps_kernel, dim3(gridx(D), B), dim3(NT), 0, (hipStream_t)stream, x, x32, F, c_skip, c_out, sigma_div, eps,
                       denoised, D);
    NLC_CHECK_LAUNCH("nlc_edm_eps");
    return NLC_OK;
}

extern "C" int nlc_f64_lincomb(const double* x, const double* ca, const double* y, const double* cb, double* out, int B,
                               int64_t D, void* stream) {
    NLC_REQUIRE(x && ca && out && B > 0 && D > 0 && B <= 65535, "nlc_f64_lincomb: bad arguments");
    NLC_REQUIRE((y == nullptr) == (cb == nullptr), "nlc_f64_lincomb: y and cb must come together");
    hipLaunchKernelGGL(f64_lincomb_kernel, dim3(gridx(D), B), dim3(NT), 0, (hipStream_t)stream, x, ca, y, cb, out, D);
    NLC_CHECK_LAUNCH("nlc_f64_lincomb");
    return NLC_OK;
}
