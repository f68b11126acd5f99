// Probe (GPU box): what does the ACCESS SHAPE of an LDS-DMA activation stream cost?  The pointwise-convolution kernels fetch a k-step as
// 128-byte column slices of 128 pixel rows (rows K*2 bytes apart) and sit at 3.3-4.5 TB/s; the streaming kernels read whole rows and
// reach 5.8.  This reads the same [M][K] bf16 tensor through the same ring (16-KiB stages, 256-thread workgroups, two per CU, counted
// vmcnt waits, one barrier per stage) in both shapes, with 2..8 stages in flight, optionally with the output write stream beside it.
//   hipcc --offload-arch=gfx950 -O2 tools/dma_shape_probe.hip -o gpurun_out/dma_shape_probe && gpurun_out/dma_shape_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* gptr, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void waitv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: column slices (tile = 128 rows, k-step = 128 bytes of each); MODE 1: whole rows (a stage = 16 KiB of consecutive bytes)
template <int MODE, int D>
__global__ __launch_bounds__(256, 2) void stream_kernel(const char* x, int64_t M, int RB, char* out, int out_rb, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nk = RB / 128;                                   // k-steps per 128-row tile
    const int64_t steps_all = (M / 128) * nk;                  // 16-KiB stages in the tensor
    const int G = gridDim.x;
    const int64_t s0 = steps_all * blockIdx.x / G, s1 = steps_all * (blockIdx.x + 1) / G;   // (MODE 0: ranges are whole tiles when steps_all/G % nk == 0)
    const int total = (int)(s1 - s0);
    auto issue = [&](int64_t s, int stage) {
        const unsigned base = lds0 + stage * 16384 + wave * 1024;
        if (MODE == 0) {
            const int64_t tile = s / nk; const int kb = (int)(s - tile * nk);
            const int lr = tid >> 3, gc = tid & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(x + ((tile * 128 + lr + 32 * i) * RB + kb * 128 + gc * 16), base + i * 4096);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(x + (s * 16384 + (i * 4 + wave) * 1024 + lane * 16), base + i * 4096);
        }
    };
    unsigned acc = 0;
    for (int d = 0; d < D - 1 && d < total; ++d) issue(s0 + d, d);
    int stage = 0;
    for (int g = 0; g < total; ++g) {
        if (g + D - 1 < total) { int st = stage + D - 1; if (st >= D) st -= D; issue(s0 + g + D - 1, st); waitv<4 * (D - 1)>(); }
        else waitv<0>();
        __syncthreads();
        acc ^= *reinterpret_cast<const unsigned*>(smem + stage * 16384 + tid * 64);
        if (out && ((g + 1) % nk) == 0) {                      // the output tile of a 128-pixel tile: 128 x out_rb bytes, whole rows
            const int64_t tile = (s0 + g) / nk;
            char* op = out + tile * 128 * (int64_t)out_rb;
            const uint4 v = {acc, acc, acc, acc};
            for (int o = tid * 16; o < 128 * out_rb; o += 256 * 16) *reinterpret_cast<uint4*>(op + o) = v;
        }
        __syncthreads();
        if (++stage == D) stage = 0;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int D>
float run(const char* x, int64_t M, int RB, char* out, int out_rb, unsigned* sink, int grid, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<MODE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, D * 16384);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<MODE, D>), dim3(grid), dim3(256), D * 16384, 0, x, M, RB, out, out_rb, sink);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, D>), dim3(grid), dim3(256), D * 16384, 0, x, M, RB, out, out_rb, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int64_t M = 1 << 20;
    unsigned* sink; hipMalloc(&sink, 4);
    for (int RB : {1024, 512, 1536}) {
        char* x; char* out;
        if (hipMalloc(&x, M * RB) != hipSuccess || hipMalloc(&out, M * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(x, 1, M * RB); hipMemset(out, 0, M * 512);
        const double gb = (double)M * RB / 1e9, gbw = (double)M * 512 / 1e9;
        for (int wr = 0; wr < 2; ++wr) {
            char* o = wr ? out : nullptr;
            const double tot = gb + (wr ? gbw : 0.0);
#define ROW(MODE, D, G) { float ms = run<MODE, D>(x, M, RB, o, 512, sink, G, 5); \
            printf("row %4d B  %s  %s  stages in flight %d  grid %4d : %8.1f us  %6.2f TB/s\n", RB, MODE ? "whole rows   " : "column slices", \
                   wr ? "+write 512B/px" : "read only     ", D - 1, G, ms * 1e3, tot / ms); }
            ROW(0, 3, 512) ROW(1, 3, 512) ROW(0, 5, 512) ROW(1, 5, 512) ROW(0, 5, 256) ROW(1, 5, 256) ROW(0, 9, 256) ROW(1, 9, 256)
        }
        hipFree(x); hipFree(out);
    }
    return 0;
}
