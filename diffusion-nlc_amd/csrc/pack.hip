// Load-time weight packing for nlc_conv2d (include/nlc_hip.h: nlc_pack_conv_weights): the reference's
// [Cout][Cin][KH*KW] f32 tensors -> [Cout_pad][KH*KW][Cin_pad] in the compute dtype, with the output-row permutation,
// per-row scale (attention scale / BatchNorm fold) and input-column permutation applied on the way.
// HBM-bound, runs once per tensor: one thread per packed element, writes coalesced (the padded layout is the
// contiguous one), reads strided by KH*KW - irrelevant at load time.
#include "common.h"
#include "conv_params.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int Cout_pad, int Cin_pad,
                                                   const int32_t* __restrict__ row_perm, const double* __restrict__ row_scale,
                                                   const int32_t* __restrict__ col_perm, T* __restrict__ packed) {
    const int64_t n = (int64_t)Cout_pad * taps * Cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cin_pad);
        const int64_t rt = i / Cin_pad;
        const int tap = (int)(rt % taps);
        const int r = (int)(rt / taps);
        float v = 0.f;
        if (r < Cout && c < Cin) {
            const int sr = row_perm ? row_perm[r] : r, sc = col_perm ? col_perm[c] : c;
            const double x = (double)w[((int64_t)sr * Cin + sc) * taps + tap];
            v = (float)(row_scale ? x * row_scale[r] : x);
        }
        ElemTraits<T>::store(packed + i, v);
    }
}

__global__ void pack_bias_kernel(const float* __restrict__ bias, int Cout, const int32_t* __restrict__ row_perm,
                                 const double* __restrict__ row_scale, const double* __restrict__ bias_add,
                                 float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= Cout) return;
    double b = bias ? (double)bias[row_perm ? row_perm[r] : r] : 0.0;
    if (row_scale) b *= row_scale[r];
    if (bias_add) b += bias_add[r];
    out[r] = (float)b;
}

}  // namespace

extern "C" int nlc_pack_conv_weights(const float* w, const float* bias, int Cout, int Cin, int KH, int KW,
                                     const int32_t* row_perm, const double* row_scale, const double* bias_add,
                                     const int32_t* col_perm, int dtype, void* packed, float* bias_out, void* stream) {
    NLC_REQUIRE(w && packed, "nlc_pack_conv_weights: null pointer");
    NLC_REQUIRE(dtype == NLC_F32 || dtype == NLC_BF16, "nlc_pack_conv_weights: bad dtype %d", dtype);
    NLC_REQUIRE(Cout > 0 && Cin > 0 && KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7, "nlc_pack_conv_weights: bad dims");
    NLC_REQUIRE(bias_out || (!bias && !bias_add), "nlc_pack_conv_weights: bias / bias_add given without bias_out");
    int cm = 0, km = 0;
    (void)nlc_conv_pack_dims(dtype, &cm, &km);
    const int Cout_pad = (Cout + cm - 1) / cm * cm, Cin_pad = (Cin + km - 1) / km * km;
    const int taps = KH * KW;
    const int64_t n = (int64_t)Cout_pad * taps * Cin_pad;
    int grid = cdiv(n, 256);
    if (grid > 4096) grid = 4096;
    if (dtype == NLC_BF16)
        hipLaunchKernelGGL(pack_kernel<bf16_raw>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, Cout_pad, Cin_pad,
                           row_perm, row_scale, col_perm, (bf16_raw*)packed);
    else
        hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, Cout_pad, Cin_pad,
                           row_perm, row_scale, col_perm, (float*)packed);
    NLC_CHECK_LAUNCH("nlc_pack_conv_weights");
    if (bias_out && (bias || bias_add)) {
        hipLaunchKernelGGL(pack_bias_kernel, dim3(cdiv(Cout, 256)), dim3(256), 0, (hipStream_t)stream, bias, Cout, row_perm, row_scale,
                           bias_add, bias_out);
        NLC_CHECK_LAUNCH("nlc_pack_conv_weights(bias)");
    }
    return NLC_OK;
}
