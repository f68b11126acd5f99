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

// Per packed output row r: the power of two 2^e that brings the largest |value| of the row into [2^14, 2^15), written as its INVERSE
// (what the convolution multiplies its sum by).  One workgroup per row; max is order-independent, so no fixed order is needed.
__global__ __launch_bounds__(256) void pack_x3_rowscale_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int Cout_pad,
                                                               const int32_t* __restrict__ row_perm, const double* __restrict__ row_scale,
                                                               float* __restrict__ w_scale_out) {
    const int r = blockIdx.x;
    __shared__ float red[256];
    float m = 0.f;
    if (r < Cout) {
        const int sr = row_perm ? row_perm[r] : r;
        const float* src = w + (int64_t)sr * Cin * taps;
        const double rs = row_scale ? row_scale[r] : 1.0;
        for (int i = threadIdx.x; i < Cin * taps; i += 256) {
            const float v = fabsf((float)((double)src[i] * rs));
            m = (v > m || v != v) ? v : m;                       // a NaN sticks (and ends in scale 1 below)
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { const float o = red[threadIdx.x + s]; if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m = red[0];
        int e = 0;
        if (m > 0.f && m < __builtin_huge_valf()) {              // finite, non-zero: ilogb(m) = k, m in [2^k, 2^(k+1)) -> e = 14 - k
            e = 14 - ilogbf(m);
            e = e > 100 ? 100 : (e < -100 ? -100 : e);           // 2^-e stays a normal f32
        }
        w_scale_out[r] = ldexpf(1.0f, -e);
    }
}

// NLC_MATH_F16X3 packing of an f32 weight tensor: one thread per f16 slot of the packed tensor (two slots per weight).  Within a
// 32-channel k-block (64 slots = 8 chunks of 8): chunk c < 4 holds the hi halves of channels 4c..4c+3 and 16+4c..16+4c+3, chunk 4 + c
// the lo halves of the same eight channels (include/nlc_hip.h: nlc_pack_conv_weights_ex; conv_halo.hip reads it).  Values are
// pre-multiplied by the row's power of two 2^e[r] = 1 / w_scale[r] (kernel above).
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int Cout_pad, int Cin_pad,
                                                      const int32_t* __restrict__ row_perm, const double* __restrict__ row_scale,
                                                      const int32_t* __restrict__ col_perm, const float* __restrict__ w_scale,
                                                      f16_raw* __restrict__ packed) {
    const int64_t n = (int64_t)Cout_pad * taps * Cin_pad * 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int slot = (int)(i & 63);
        const int64_t blk = i >> 6;                          // (row, tap, 32-channel block)
        const int ncb = Cin_pad / 32;
        const int cb = (int)(blk % ncb);
        const int64_t rt = blk / ncb;
        const int tap = (int)(rt % taps);
        const int r = (int)(rt / taps);
        const int chunk = slot >> 3, e = slot & 7;
        const int c = cb * 32 + (e < 4 ? 4 * (chunk & 3) + e : 16 + 4 * (chunk & 3) + (e - 4));
        float v = 0.f;
        if (r < Cout && c < Cin) {
            const int sr = row_perm ? row_perm[r] : r, sc = col_perm ? col_perm[c] : c;
            const double x = (double)w[((int64_t)sr * Cin + sc) * taps + tap];
            v = (float)(row_scale ? x * row_scale[r] : x);
            v *= 1.0f / w_scale[r];                              // * 2^e[r]: exact (both factors are powers of two / normal)
        }
        const f16_raw hi = (f16_raw)v;
        packed[i] = chunk < 4 ? hi : (f16_raw)(v - (float)hi);
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
    return nlc_pack_conv_weights_ex(w, bias, Cout, Cin, KH, KW, row_perm, row_scale, bias_add, col_perm, dtype, NLC_MATH_NATIVE, packed,
                                    bias_out, nullptr, stream);
}

extern "C" int nlc_pack_conv_weights_ex(const float* w, const float* bias, int Cout, int Cin, int KH, int KW,
                                        const int32_t* row_perm, const double* row_scale, const double* bias_add,
                                        const int32_t* col_perm, int dtype, int math, void* packed, float* bias_out, float* w_scale_out,
                                        void* stream) {
    NLC_REQUIRE(w && packed, "nlc_pack_conv_weights: null pointer");
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_pack_conv_weights: bad dtype %d", dtype);
    NLC_REQUIRE(math == NLC_MATH_NATIVE || (math == NLC_MATH_F16X3 && dtype == NLC_F32), "nlc_pack_conv_weights: math %d needs dtype NLC_F32", math);
    NLC_REQUIRE(Cout > 0 && Cin > 0 && KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7, "nlc_pack_conv_weights: bad dims");
    NLC_REQUIRE(bias_out || (!bias && !bias_add), "nlc_pack_conv_weights: bias / bias_add given without bias_out");
    NLC_REQUIRE(math != NLC_MATH_F16X3 || w_scale_out, "nlc_pack_conv_weights: NLC_MATH_F16X3 needs w_scale_out (f32 [Cout_pad])");
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
    else if (dtype == NLC_F16)
        hipLaunchKernelGGL(pack_kernel<f16_raw>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, Cout_pad, Cin_pad,
                           row_perm, row_scale, col_perm, (f16_raw*)packed);
    else if (math == NLC_MATH_F16X3) {
        hipLaunchKernelGGL(pack_x3_rowscale_kernel, dim3(Cout_pad), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, Cout_pad,
                           row_perm, row_scale, w_scale_out);
        NLC_CHECK_LAUNCH("nlc_pack_conv_weights(row scale)");
        hipLaunchKernelGGL(pack_x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, taps, Cout_pad, Cin_pad,
                           row_perm, row_scale, col_perm, (const float*)w_scale_out, (f16_raw*)packed);
    }
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
