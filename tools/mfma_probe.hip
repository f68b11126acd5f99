// Probe (run once on the GPU box): does v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 inputs?  NLC_MATH_F16X3 relies on it for the
// `lo` halves of small operands (conv_params.h: f16x3_split4).  Prints the products of a subnormal / smallest-normal A with B = 2^10.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ void probe(const _Float16* a_vals, float* out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < 4; ++t) {
        f16x8_t a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)0.f; b[k] = (_Float16)0.f; }
        if ((lane >> 4) == 0) { a[0] = a_vals[t]; b[0] = (_Float16)1024.f; }      // k = 0 only: D[i][j] = a * 1024 for every (i, j)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        if (lane == 0) out[t] = acc[0];
    }
}
int main() {
    _Float16 h[4];
    const float vals[4] = {5.9604645e-8f /* 2^-24: smallest subnormal */, 3.0517578e-5f /* 2^-15 */, 6.1035156e-5f /* 2^-14: smallest normal */, 1.0f};
    for (int i = 0; i < 4; ++i) h[i] = (_Float16)vals[i];
    _Float16* d; float* o; float r[4];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 4; ++i) {
        const float expect = vals[i] * 1024.f;
        printf("a = %.9g  a * 1024 via MFMA = %.9g  expected %.9g  %s\n", vals[i], r[i], expect, r[i] == expect ? "kept" : "FLUSHED / wrong");
        if (r[i] != expect) ok = 0;
    }
    printf("%s\n", ok ? "MFMA_F16_SUBNORMALS_KEPT" : "MFMA_F16_SUBNORMALS_FLUSHED");
    return 0;
}
