// Inpainting measurement operator on the NCHW f32 sampler state.
// functions/svd_operators.py:324-359 expresses A and A^+ as permute / index_select / scatter chains over
// a pixel-interleaved (H*W, C) flattening; both are pure gathers:
//   A(x)[b][j]          = x[b][c][p]            with kept[j] = p*C + c
//   A_pinv(y)[b][c][p]  = inv[p*C + c] >= 0 ? y[b][inv[p*C + c]] : 0
// The per-step projection  x0 <- x0 - A^+(A x0 - y)  ("copy the known pixels") is fused into
// nlc_sched_step (mask / known), so these two kernels run once per batch.
#include "common.h"

namespace {
constexpr int NT = 256;

__global__ void inpaint_A_kernel(const float* __restrict__ x, const int64_t* __restrict__ kept, float* __restrict__ out,
                                 int64_t nk, int C, int64_t HW) {
    const int b = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * NT + threadIdx.x; j < nk; j += (int64_t)gridDim.x * NT) {
        const int64_t idx = kept[j];
        const int64_t p = idx / C; const int c = (int)(idx - p * C);
        out[(int64_t)b * nk + j] = x[((int64_t)b * C + c) * HW + p];
    }
}

__global__ void inpaint_Apinv_kernel(const float* __restrict__ y, const int32_t* __restrict__ inv, float* __restrict__ out,
                                     int64_t nk, int C, int64_t HW) {
    const int b = blockIdx.y;
    const int64_t n = (int64_t)C * HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i / HW); const int64_t p = i - (int64_t)c * HW;
        const int32_t j = inv[p * C + c];
        out[(int64_t)b * n + i] = j >= 0 ? y[(int64_t)b * nk + j] : 0.f;
    }
}
}  // namespace

extern "C" int nlc_inpaint_A(const float* x, const int64_t* kept, float* out, int B, int C, int64_t HW, int64_t nk, void* stream) {
    NLC_REQUIRE(x && kept && out && B > 0 && B <= 65535 && C > 0 && HW > 0 && nk > 0, "nlc_inpaint_A: bad arguments");
    int g = cdiv(nk, NT); if (g > 1024) g = 1024;
    hipLaunchKernelGGL(inpaint_A_kernel, dim3(g, B), dim3(NT), 0, (hipStream_t)stream, x, kept, out, nk, C, HW);
    NLC_CHECK_LAUNCH("nlc_inpaint_A");
    return NLC_OK;
}

extern "C" int nlc_inpaint_Apinv(const float* y, const int32_t* inv, float* out, int B, int C, int64_t HW, int64_t nk, void* stream) {
    NLC_REQUIRE(y && inv && out && B > 0 && B <= 65535 && C > 0 && HW > 0 && nk > 0, "nlc_inpaint_Apinv: bad arguments");
    int g = cdiv((int64_t)C * HW, NT); if (g > 1024) g = 1024;
    hipLaunchKernelGGL(inpaint_Apinv_kernel, dim3(g, B), dim3(NT), 0, (hipStream_t)stream, y, inv, out, nk, C, HW);
    NLC_CHECK_LAUNCH("nlc_inpaint_Apinv");
    return NLC_OK;
}
