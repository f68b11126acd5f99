// Shared device/host helpers for libnlc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "nlc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef unsigned short bf16_raw;   // storage type for bf16 in memory
typedef _Float16 f16_raw;          // storage type for IEEE half (NLC_F16): a distinct type, so every kernel template has both 16-bit forms
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define NLC_WAVE 64

void nlc_set_error(const char* fmt, ...);

#define NLC_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            nlc_set_error(__VA_ARGS__);                          \
            return NLC_EINVAL;                                   \
        }                                                        \
    } while (0)

#define NLC_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            nlc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return NLC_ELAUNCH;                                                   \
        }                                                                         \
    } while (0)

// ---- scalar conversions -------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw v) {
    return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_raw, b);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kPerChunk = 4;            // elements per 16-byte chunk
    static constexpr int kDtype = NLC_F32;
    __device__ static float load(const float* p) { return *p; }
    __device__ static void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_raw> {
    static constexpr int kPerChunk = 8;
    static constexpr int kDtype = NLC_BF16;
    __device__ static float load(const bf16_raw* p) { return bf16_to_f32(*p); }
    __device__ static void store(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct ElemTraits<f16_raw> {
    static constexpr int kPerChunk = 8;
    static constexpr int kDtype = NLC_F16;
    __device__ static float load(const f16_raw* p) { return (float)*p; }
    __device__ static void store(f16_raw* p, float v) { *p = (f16_raw)v; }      // round-to-nearest-even (v_cvt_f16_f32)
};

static inline bool nlc_is16(int dtype) { return dtype == NLC_BF16 || dtype == NLC_F16; }
static inline bool nlc_dtype_ok(int dtype) { return dtype == NLC_F32 || nlc_is16(dtype); }
// run `...` with T16 = the 16-bit storage type of `dtype` (NLC_BF16 or NLC_F16)
#define NLC_SWITCH_16(dtype, ...)                                       \
    do {                                                                \
        if ((dtype) == NLC_BF16) { using T16 = bf16_raw; __VA_ARGS__; } \
        else { using T16 = f16_raw; __VA_ARGS__; }                      \
    } while (0)

// unpack a 16-byte chunk to floats / pack floats into a chunk
template <typename T> __device__ __forceinline__ void chunk_to_f32(const uint4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
    f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_raw>(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
    f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
    f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
    f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void chunk_to_f32<f16_raw>(const uint4& c, float* f) {
    const f16x8_t h = __builtin_bit_cast(f16x8_t, c);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (float)h[k];
}
template <typename T> __device__ __forceinline__ uint4 f32_to_chunk(const float* f);
template <> __device__ __forceinline__ uint4 f32_to_chunk<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 f32_to_chunk<bf16_raw>(const float* f) {
    uint4 c;
    c.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
    c.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
    c.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
    c.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
    return c;
}

template <> __device__ __forceinline__ uint4 f32_to_chunk<f16_raw>(const float* f) {
    f16x8_t h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (f16_raw)f[k];
    return __builtin_bit_cast(uint4, h);
}

// ---- matrix cores: one 16-byte A fragment x one 16-byte B fragment into a 16x16 f32 tile.  16-bit types: ONE
//      v_mfma_f32_16x16x32_{bf16,f16} (K = 32; the two run at the same rate on gfx950); f32: four exact
//      v_mfma_f32_16x16x4_f32 with the same k permutation on both operands.  KBE = elements per 128-byte k-block.
template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_raw> {
    static constexpr int KBE = 64;
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16_t run32(const uint4& a, const uint4& b, const f32x16_t& acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_raw> {
    static constexpr int KBE = 64;
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16_t run32(const uint4& a, const uint4& b, const f32x16_t& acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mfma16<float> {
    static constexpr int KBE = 32;
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// 16-bit paths: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of expf + IEEE division (~25 instructions per element)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == NLC_ACT_SILU) return silu_exact(v);
    if (act == NLC_ACT_GELU) return gelu_erf(v);
    return v;
}

// wave-wide reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Per-DEVICE one-time launch setup.  hipFuncSetAttribute (the dynamic-LDS opt-in) binds to the current device's copy of
// the code object, so a process that drives several GPUs has to do it once per device, not once per process; the
// CU count is a per-device fact too.  The flags are only ever set (idempotent work behind them), so two host threads
// racing on the first launch at worst both do the setup.
constexpr int NLC_MAX_DEVICES = 64;
struct DeviceOnce {
    bool done[NLC_MAX_DEVICES] = {};
    int ncu[NLC_MAX_DEVICES] = {};
};
// returns the current device id (clamped into the table) and runs `setup()` the first time this device is seen
template <typename F>
static inline int nlc_device_once(DeviceOnce& st, F&& setup) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const int slot = dev < NLC_MAX_DEVICES ? dev : NLC_MAX_DEVICES - 1;
    if (!st.done[slot] || dev >= NLC_MAX_DEVICES) {
        setup();
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        st.ncu[slot] = n;
        st.done[slot] = true;
    }
    return slot;
}
