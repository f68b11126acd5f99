// HBM-bound helpers on channels-last activations: 2x resampling, pad, layout changes,
// sinusoidal embeddings and the Cin<=4 first-layer convolution.  All use 16-byte accesses on
// the contiguous channel axis and grid-stride loops (<= 2048 workgroups).
#include "common.h"
#include "conv_params.h"

namespace {

constexpr int NT = 256;
static inline int grid_for(int64_t work) {
    int g = cdiv(work, NT);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return g;
}

// ---- 2x2 average pool (src/unet_adm.py:136, AvgPool2d(2); edm_networks.py:91-93 with f=[1,1])
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int Ho = H / 2, Wo = W / 2, nch = C / PER;
    const int64_t total = (int64_t)B * Ho * Wo * nch;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int ch = (int)(e % nch); int64_t r = e / nch;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
        const T* p00 = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + ch * PER;
        float a[PER], t[PER];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(p00), a);
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(p00 + C), t);
#pragma unroll
        for (int j = 0; j < PER; ++j) a[j] += t[j];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(p00 + (int64_t)W * C), t);
#pragma unroll
        for (int j = 0; j < PER; ++j) a[j] += t[j];
        chunk_to_f32<T>(*reinterpret_cast<const uint4*>(p00 + (int64_t)W * C + C), t);
#pragma unroll
        for (int j = 0; j < PER; ++j) a[j] = (a[j] + t[j]) * 0.25f;
        *reinterpret_cast<uint4*>(out + (((int64_t)b * Ho + oy) * Wo + ox) * C + ch * PER) = f32_to_chunk<T>(a);
    }
}

// ---- nearest 2x upsample (F.interpolate nearest, src/unet_adm.py:107)
template <typename T>
__global__ void upsample_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int Ho = H * 2, Wo = W * 2, nch = C / PER;
    const int64_t total = (int64_t)B * Ho * Wo * nch;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int ch = (int)(e % nch); int64_t r = e / nch;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + ch * PER);
        *reinterpret_cast<uint4*>(out + (((int64_t)b * Ho + oy) * Wo + ox) * C + ch * PER) = v;
    }
}

// ---- zero pad one pixel right/bottom (F.pad (0,1,0,1), src/unet_simple.py:69-70)
template <typename T>
__global__ void pad_rb_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C) {
    constexpr int PER = ElemTraits<T>::kPerChunk;
    const int Ho = H + 1, Wo = W + 1, nch = C / PER;
    const int64_t total = (int64_t)B * Ho * Wo * nch;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int ch = (int)(e % nch); int64_t r = e / nch;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (oy < H && ox < W) v = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + oy) * W + ox) * C + ch * PER);
        *reinterpret_cast<uint4*>(out + (((int64_t)b * Ho + oy) * Wo + ox) * C + ch * PER) = v;
    }
}

// ---- layout changes through a 32x32 LDS transpose tile: [B][HW][C] <-> [B][C][HW]
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? ElemTraits<T>::load(x + ((int64_t)b * HW + p) * C + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) out[((int64_t)b * C + c) * HW + p] = tile[tx][i];
    }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int HW, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (p < HW && c < C) ? x[((int64_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) ElemTraits<T>::store(out + ((int64_t)b * HW + p) * C + c, tile[tx][i]);
    }
}

// ---- sinusoidal embeddings.  The frequency table is computed by the host exactly as the
//      reference does (so t*freq is bit-identical); only sin/cos run here.
__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                            int B, int dim, int sin_first) {
    const int half = dim / 2;
    const int64_t total = (int64_t)B * half;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int i = (int)(e % half); const int b = (int)(e / half);
        const float a = t[b] * freqs[i];
        const float cs = cosf(a), sn = sinf(a);
        float* o = out + (int64_t)b * dim;
        if (sin_first) { o[i] = sn; o[half + i] = cs; }
        else { o[i] = cs; o[half + i] = sn; }
        if ((dim & 1) && i == 0) o[dim - 1] = 0.f;
    }
}

// ---- first-layer convolution, Cin <= 4.  One workgroup = 32 consecutive output pixels of one
//      image row x all Cout; the KHxKWxCin input patch per pixel sits in LDS, each thread keeps
//      its channel's weights in registers.  Output stores are channel-contiguous (coalesced).
constexpr int FPIX = 32;
constexpr int FMAXK = 49 * 4;
template <typename T>
__global__ __launch_bounds__(NT) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, int B, int Cin, int H, int W, int Cout,
                                                        int KH, int KW) {
    __shared__ float patch[FPIX][FMAXK + 1];
    const int K = KH * KW * Cin;
    const int tiles_x = (W + FPIX - 1) / FPIX;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * FPIX; bid /= tiles_x;
    const int oy = bid % H; const int b = bid / H;
    const float sc = in_scale ? in_scale[b] : 1.0f;
    const int ph = KH / 2, pw = KW / 2;
    for (int e = threadIdx.x; e < FPIX * K; e += NT) {
        const int p = e / K, k = e - p * K;
        const int tap = k / Cin, c = k - tap * Cin;
        const int r = tap / KW, s = tap - r * KW;
        const int iy = oy + r - ph, ix = tx0 + p + s - pw;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * Cin + c) * H + iy) * W + ix] * sc;
        patch[p][k] = v;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < Cout; n += NT) {
        const float* wn = w + (int64_t)n * K;
        const float bv = bias ? bias[n] : 0.f;
        float acc[FPIX];
#pragma unroll
        for (int p = 0; p < FPIX; ++p) acc[p] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float wk = wn[k];
#pragma unroll
            for (int p = 0; p < FPIX; ++p) acc[p] = fmaf(patch[p][k], wk, acc[p]);
        }
#pragma unroll
        for (int p = 0; p < FPIX; ++p) {
            const int ox = tx0 + p;
            if (ox < W) ElemTraits<T>::store(out + (((int64_t)b * H + oy) * W + ox) * Cout + n, acc[p] + bv);
        }
    }
}

// ---- bf16 first-layer conv on the matrix cores.  K = KH*KW*Cin <= 32 is ONE v_mfma_f32_16x16x32_bf16 k-step, so a
//      wave keeps its 64 output channels' weights in registers for the whole launch and the kernel is a pure stream:
//      per 64-pixel tile the workgroup builds the im2col patch [64][32] (bf16, input scale fused) in LDS from the NCHW
//      f32 state, each of the 4 waves multiplies it with its 64 channels (operands swapped: a lane ends up with 16
//      consecutive channels of one pixel), adds the bias and stores 16-byte chunks.  HBM-bound on the output write
//      (the VALU version ran at 32 TFLOP/s of f32 FMAs: 0.45 ms for the 537 MB of ADM-256's first layer).
//      GroupNorm statistics of the output ride along (per chunk (sum, sumsq) totals, added atomically once per workgroup).
constexpr int F1_PIX = 64;
constexpr int F1_LD = 80;          // LDS bytes per patch row: 64 B of k + 16 B pad (conflict-free 16-byte fragment reads)
template <typename T>
__global__ __launch_bounds__(NT) void conv_first_mfma_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             T* __restrict__ out, int Cin, int H, int W, int Cout,
                                                             int KH, int KW, int tiles_per_blk, int blks_per_img,
                                                             long long* __restrict__ stats, int stats_gran) {
    __shared__ __attribute__((aligned(16))) char patch[2][F1_PIX * F1_LD];
    const int K = KH * KW * Cin;
    const int HW = H * W;
    const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const float sc = in_scale ? in_scale[b] : 1.0f;
    const int ph = KH / 2, pw = KW / 2;
    // this wave's weights: MFMA tile j row fr <-> channel wave*64 + (fr>>2)*16 + j*4 + (fr&3); k = fq*8 .. fq*8+7
    uint4 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = wave * 64 + (fr >> 2) * 16 + j * 4 + (fr & 3);
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int kk = fq * 8 + k; f[k] = (ch < Cout && kk < K) ? w[(int64_t)ch * K + kk] : 0.f; }
        wf[j] = f32_to_chunk<T>(f);
    }
    const int n = wave * 64 + fq * 16;                       // this lane's 16 output channels
    float cb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cb[k] = (bias && n + k < Cout) ? bias[n + k] : 0.f;
    Stat16 st16;
    st16.zero();
    const int tile0 = blk * tiles_per_blk;
    const int ntile = (HW + F1_PIX - 1) / F1_PIX;
    const int pp = tid & 63, kq = tid >> 6;                  // patch builder: pixel pp, k = kq, kq+4, ...
    // (Requesting the patch values of tile t + 1 before tile t is multiplied and stored - eight registers across the MFMAs - measured
    //  SLOWER: 198 -> 215 us at B = 16, 49.6 -> 55.7 us on EDM-32; the two to four workgroups per CU already cover the round trip.)
    for (int ti = 0; ti < tiles_per_blk; ++ti) {
        const int tile = tile0 + ti;
        if (tile >= ntile) break;                            // workgroup-uniform
        char* pt = patch[ti & 1];
        {
            const int m = tile * F1_PIX + pp;
            const int oy = m / W, ox = m - oy * W;
            for (int k = kq; k < 32; k += 4) {
                float v = 0.f;
                if (k < K && m < HW) {
                    const int tap = k / Cin, c = k - tap * Cin;
                    const int r = tap / KW, s_ = tap - r * KW;
                    const int iy = oy + r - ph, ix = ox + s_ - pw;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * Cin + c) * H + iy) * W + ix] * sc;
                }
                ElemTraits<T>::store(reinterpret_cast<T*>(pt + pp * F1_LD + k * 2), v);
            }
        }
        __syncthreads();                                     // double-buffered patch: one barrier per tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 af = *reinterpret_cast<const uint4*>(pt + (i * 16 + fr) * F1_LD + fq * 16);
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4_t acc = f32x4_t{cb[j * 4], cb[j * 4 + 1], cb[j * 4 + 2], cb[j * 4 + 3]};
                Mfma16<T>::run(wf[j], af, acc);
                v[j * 4] = acc[0]; v[j * 4 + 1] = acc[1]; v[j * 4 + 2] = acc[2]; v[j * 4 + 3] = acc[3];
            }
            const int m = tile * F1_PIX + i * 16 + fr;
            if (m < HW && n < Cout) {
                T* op = out + ((int64_t)b * HW + m) * Cout + n;
                if (n + 16 <= Cout && (Cout & 7) == 0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const uint4 pk = f32_to_chunk<T>(v + c * 8);
                        *reinterpret_cast<uint4*>(op + c * 8) = pk;
                        if (stats) st16.add_chunk<T>(c, pk);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) if (n + k < Cout) ElemTraits<T>::store(op + k, v[k]);
                }
            }
        }
    }
    if (stats) {             // host guarantees Cout % 16 == 0 and HW % 64 == 0 here: every lane contributed whole chunks
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) { st16.s[q] += __shfl_xor(st16.s[q], o, 64); st16.q[q] += __shfl_xor(st16.q[q], o, 64); }
        if (n < Cout) st16.emit_row(stats, b, Cout, n, stats_gran, fr);          // totals: one atomic instruction per wave
    }
}

// workgroups per image of the MFMA first-layer kernel (also the number of statistics partials), 0 = not eligible
static int conv_first_mfma_blocks(int Cin, int H, int W, int Cout, int KH, int KW, int dtype, int B = 16) {
    if (!nlc_is16(dtype) || KH * KW * Cin > 32 || Cout > 256) return 0;
    const int ntile = ((int64_t)H * W + F1_PIX - 1) / F1_PIX;
    int want = B > 0 ? (1024 + B - 1) / B : 64;              // ~four workgroups per CU over the batch, at least 64 per image (B = 8: 86 -> 63 us; twice as many: 75)
    if (want < 64) want = 64;
    int nb = ntile < want ? ntile : want;
    return nb < 1 ? 1 : nb;
}

}  // namespace

// launch KERNEL<T> for the storage type of `dtype`; X / OUT name the (void*) tensors that take the element type
#define LAUNCH_T(dtype, KERNEL, grid, st, XEXPR, OEXPR, ...)                                                                              \
    do {                                                                                                                                  \
        if ((dtype) == NLC_BF16) hipLaunchKernelGGL(KERNEL<bf16_raw>, grid, dim3(NT), 0, st, (const bf16_raw*)(XEXPR), (bf16_raw*)(OEXPR), __VA_ARGS__); \
        else if ((dtype) == NLC_F16) hipLaunchKernelGGL(KERNEL<f16_raw>, grid, dim3(NT), 0, st, (const f16_raw*)(XEXPR), (f16_raw*)(OEXPR), __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<float>, grid, dim3(NT), 0, st, (const float*)(XEXPR), (float*)(OEXPR), __VA_ARGS__);                \
    } while (0)

static int check_nhwc(const char* name, const void* x, const void* out, int B, int H, int W, int C, int dtype) {
    if (!nlc_dtype_ok(dtype)) { nlc_set_error("%s: bad dtype %d", name, dtype); return NLC_EINVAL; }
    if (!x || !out) { nlc_set_error("%s: null pointer", name); return NLC_EINVAL; }
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) { nlc_set_error("%s: bad dims", name); return NLC_EINVAL; }
    return NLC_OK;
}

extern "C" int nlc_avgpool2x2(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream) {
    int rc = check_nhwc("nlc_avgpool2x2", x, out, B, H, W, C, dtype); if (rc) return rc;
    const int per = nlc_is16(dtype) ? 8 : 4;
    NLC_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % per == 0, "nlc_avgpool2x2: H,W must be even and C a multiple of %d", per);
    const int64_t work = (int64_t)B * (H / 2) * (W / 2) * (C / per);
    LAUNCH_T(dtype, avgpool_kernel, dim3(grid_for(work)), (hipStream_t)stream, x, out, B, H, W, C);
    NLC_CHECK_LAUNCH("nlc_avgpool2x2");
    return NLC_OK;
}

extern "C" int nlc_upsample2x(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream) {
    int rc = check_nhwc("nlc_upsample2x", x, out, B, H, W, C, dtype); if (rc) return rc;
    const int per = nlc_is16(dtype) ? 8 : 4;
    NLC_REQUIRE(C % per == 0, "nlc_upsample2x: C must be a multiple of %d", per);
    const int64_t work = (int64_t)B * H * 2 * W * 2 * (C / per);
    LAUNCH_T(dtype, upsample_kernel, dim3(grid_for(work)), (hipStream_t)stream, x, out, B, H, W, C);
    NLC_CHECK_LAUNCH("nlc_upsample2x");
    return NLC_OK;
}

extern "C" int nlc_pad_rb(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream) {
    int rc = check_nhwc("nlc_pad_rb", x, out, B, H, W, C, dtype); if (rc) return rc;
    const int per = nlc_is16(dtype) ? 8 : 4;
    NLC_REQUIRE(C % per == 0, "nlc_pad_rb: C must be a multiple of %d", per);
    const int64_t work = (int64_t)B * (H + 1) * (W + 1) * (C / per);
    LAUNCH_T(dtype, pad_rb_kernel, dim3(grid_for(work)), (hipStream_t)stream, x, out, B, H, W, C);
    NLC_CHECK_LAUNCH("nlc_pad_rb");
    return NLC_OK;
}

extern "C" int nlc_nhwc_to_nchw_f32(const void* x, float* out, int B, int H, int W, int C, int dtype, void* stream) {
    int rc = check_nhwc("nlc_nhwc_to_nchw_f32", x, out, B, H, W, C, dtype); if (rc) return rc;
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
    NLC_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "nlc_nhwc_to_nchw_f32: grid too large");
    if (dtype == NLC_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_raw>, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_raw*)x, out, HW, C);
    else if (dtype == NLC_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_raw>, grid, dim3(NT), 0, (hipStream_t)stream, (const f16_raw*)x, out, HW, C);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, out, HW, C);
    NLC_CHECK_LAUNCH("nlc_nhwc_to_nchw_f32");
    return NLC_OK;
}

extern "C" int nlc_nchw_f32_to_nhwc(const float* x, void* out, int B, int H, int W, int C, int dtype, void* stream) {
    int rc = check_nhwc("nlc_nchw_f32_to_nhwc", x, out, B, H, W, C, dtype); if (rc) return rc;
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
    NLC_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "nlc_nchw_f32_to_nhwc: grid too large");
    if (dtype == NLC_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_raw>, grid, dim3(NT), 0, (hipStream_t)stream, x, (bf16_raw*)out, HW, C);
    else if (dtype == NLC_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_raw>, grid, dim3(NT), 0, (hipStream_t)stream, x, (f16_raw*)out, HW, C);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, x, (float*)out, HW, C);
    NLC_CHECK_LAUNCH("nlc_nchw_f32_to_nhwc");
    return NLC_OK;
}

extern "C" int nlc_timestep_embedding(const float* t, const float* freqs, float* out, int B, int dim, int sin_first, void* stream) {
    NLC_REQUIRE(t && freqs && out && B > 0 && dim >= 2, "nlc_timestep_embedding: bad arguments");
    hipLaunchKernelGGL(temb_kernel, dim3(grid_for((int64_t)B * (dim / 2))), dim3(NT), 0, (hipStream_t)stream, t, freqs, out, B, dim, sin_first);
    NLC_CHECK_LAUNCH("nlc_timestep_embedding");
    return NLC_OK;
}

extern "C" int nlc_conv_first_stats_partials(int Cin, int H, int W, int Cout, int KH, int KW, int dtype) {
    if (Cin < 1 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 16) || ((int64_t)H * W) % F1_PIX) return 0;
    const int nb = conv_first_mfma_blocks(Cin, H, W, Cout, KH, KW, dtype);
    if (nb <= 0) return 0;
    const int ntile = (int)(((int64_t)H * W) / F1_PIX);
    const int tpb = (ntile + nb - 1) / nb;
    return (ntile + tpb - 1) / tpb == nb ? nb : 0;
}

extern "C" int nlc_conv_first(const float* x_nchw, const float* in_scale, const float* w, const float* bias, void* out_nhwc,
                              int B, int Cin, int H, int W, int Cout, int KH, int KW, int dtype,
                              void* stats_out, int64_t stats_bytes, int stats_granule, void* stream) {
    const int gran = stats_granule == 4 ? 4 : 8;
    NLC_REQUIRE(stats_granule == 0 || stats_granule == 4 || stats_granule == 8, "nlc_conv_first: stats_granule must be 0, 4 or 8");
    NLC_REQUIRE(nlc_dtype_ok(dtype), "nlc_conv_first: bad dtype %d", dtype);
    NLC_REQUIRE(x_nchw && w && out_nhwc, "nlc_conv_first: null pointer");
    NLC_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "nlc_conv_first: bad dims");
    NLC_REQUIRE(Cin >= 1 && Cin <= 4, "nlc_conv_first: Cin=%d must be in 1..4", Cin);
    NLC_REQUIRE(KH >= 1 && KH <= 7 && KW >= 1 && KW <= 7 && (KH & 1) && (KW & 1), "nlc_conv_first: odd kernel 1..7 required");
    NLC_REQUIRE(B <= 65535 && (int64_t)H * W < (1ll << 30), "nlc_conv_first: image too large");
    {
        const int nb = conv_first_mfma_blocks(Cin, H, W, Cout, KH, KW, dtype, B);
        const bool want_stats = stats_out != nullptr;
        if (want_stats) {
            NLC_REQUIRE(nb > 0 && (Cout % 16) == 0 && ((int64_t)H * W) % F1_PIX == 0,
                        "nlc_conv_first: stats_out given but this launch does not emit statistics (ask nlc_conv_first_stats_partials)");
            NLC_REQUIRE(stats_bytes >= (int64_t)B * (Cout / gran) * 4 * (int64_t)sizeof(long long), "nlc_conv_first: stats_out too small");
            NLC_REQUIRE((reinterpret_cast<uintptr_t>(stats_out) & 15) == 0, "nlc_conv_first: stats_out must be 16-byte aligned (its consumers read 16-byte pairs)");
        }
        if (nb > 0) {
            const int ntile = (int)(((int64_t)H * W + F1_PIX - 1) / F1_PIX);
            const int tpb = (ntile + nb - 1) / nb;
            const int nblk = (ntile + tpb - 1) / tpb;
            NLC_SWITCH_16(dtype, hipLaunchKernelGGL(conv_first_mfma_kernel<T16>, dim3(nblk, B), dim3(NT), 0, (hipStream_t)stream, x_nchw, in_scale, w,
                                                    bias, (T16*)out_nhwc, Cin, H, W, Cout, KH, KW, tpb, nb, (long long*)stats_out, gran));
            NLC_CHECK_LAUNCH("nlc_conv_first");
            return NLC_OK;
        }
    }
    const int tiles_x = (W + FPIX - 1) / FPIX;
    const int64_t nblk = (int64_t)B * H * tiles_x;
    NLC_REQUIRE(nblk < (1ll << 31), "nlc_conv_first: grid too large");
    if (dtype == NLC_BF16) hipLaunchKernelGGL(conv_first_kernel<bf16_raw>, dim3((unsigned)nblk), dim3(NT), 0, (hipStream_t)stream, x_nchw, in_scale, w, bias, (bf16_raw*)out_nhwc, B, Cin, H, W, Cout, KH, KW);
    else if (dtype == NLC_F16) hipLaunchKernelGGL(conv_first_kernel<f16_raw>, dim3((unsigned)nblk), dim3(NT), 0, (hipStream_t)stream, x_nchw, in_scale, w, bias, (f16_raw*)out_nhwc, B, Cin, H, W, Cout, KH, KW);
    else hipLaunchKernelGGL(conv_first_kernel<float>, dim3((unsigned)nblk), dim3(NT), 0, (hipStream_t)stream, x_nchw, in_scale, w, bias, (float*)out_nhwc, B, Cin, H, W, Cout, KH, KW);
    NLC_CHECK_LAUNCH("nlc_conv_first");
    return NLC_OK;
}
