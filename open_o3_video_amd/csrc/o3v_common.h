// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the Open-o3-Video generate path.
// bf16 values are carried as raw 16-bit patterns (uint16_t); arithmetic is fp32 with the rounding
// points of the HF reference (every tensor the reference materialises in bf16 is rounded here too).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define O3V_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// low / high half of a packed pair
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// value after a bf16 materialisation
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// F.gelu(approximate="tanh") in fp32 (ATen's GeluKernel: kBeta = sqrt(2/pi), kKappa = 0.044715)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float inner = 0.79788456080286535588f * (x + 0.044715f * (x * x * x));
    return 0.5f * x * (1.0f + tanhf(inner));
}

// error codes of the C ABI (include/o3v.h)
#define O3V_OK 0
#define O3V_ERR_ARG (-1)
#define O3V_ERR_SHAPE (-2)
#define O3V_ERR_LAUNCH (-3)
#define O3V_ERR_WORKSPACE (-4)

// hipGetLastError() reports the last error of ANY earlier runtime call of this host thread (torch's own included),
// so clear it right before our launch and read it right after.
#define O3V_KLAUNCH(...)               \
    do {                               \
        (void)hipGetLastError();       \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define O3V_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return O3V_ERR_LAUNCH;       \
    } while (0)
