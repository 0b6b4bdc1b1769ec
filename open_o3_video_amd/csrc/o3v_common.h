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

// ---- RMSNorm of ONE row by ONE wave (TF:65-79): out = w * bf16(x * rsqrt(mean(x^2) + eps)); the row is held in VGPRs.  Shared by
// rmsnorm_kernel (o3v_elem.hip) and by the finishing waves of the residual linears that normalise for the next linear
// (o3v_gemm.hip, TailNorm), so both round identically: chunks lane + 64 i ascending in one fmaf chain, then the wave butterfly.
// ld_chunk(c) returns the c-th 16-byte chunk of the row (a plain or an L1-bypassing load).
template <int MAXCH, class LoadChunk>
__device__ __forceinline__ void rmsnorm_row_wave(LoadChunk ld_chunk, const bf16_t* __restrict__ w, bf16_t* __restrict__ orow_, int cols,
                                                 float eps) {
    const int lane = threadIdx.x & 63;
    const int nch = cols >> 3;  // 16-byte chunks per row
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    uint4 v[MAXCH], wv[MAXCH];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {  // the weight chunks ride along with the row: one memory latency instead of two
        int c = lane + i * 64;
        if (c < nch) wv[i] = wr[c];
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        int c = lane + i * 64;
        if (c < nch) {
            v[i] = ld_chunk(c);
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = bf_lo(p[j]), b = bf_hi(p[j]);
                ss = fmaf(a, a, ss);
                ss = fmaf(b, b, ss);
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)cols + eps);
    uint4* orow = reinterpret_cast<uint4*>(orow_);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        int c = lane + i * 64;
        if (c < nch) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(&wv[i]);
            uint4 o;
            uint32_t* po = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = rbf(bf_lo(p[j]) * rstd), b = rbf(bf_hi(p[j]) * rstd);
                po[j] = pack_bf2(bf_lo(q[j]) * a, bf_hi(q[j]) * b);
            }
            orow[c] = o;
        }
    }
}

// ---- per-head q/k RMSNorm + rotation (Qwen3-VL, TF3:480-484), shared by o3v_qkv_norm_rope_cache and the one-launch decode block
// so that both round identically.  A "lane share" of a head is an 8-wide chunk of each rotary half (lo: dims 8c.., hi: D/2 + 8c..).
__device__ __forceinline__ float qkn_chain(const u32x4& lo, const u32x4& hi) {  // sum of squares of a lane share, fixed order
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ss = fmaf(bf_lo(lo[j]), bf_lo(lo[j]), ss);
        ss = fmaf(bf_hi(lo[j]), bf_hi(lo[j]), ss);
        ss = fmaf(bf_lo(hi[j]), bf_lo(hi[j]), ss);
        ss = fmaf(bf_hi(hi[j]), bf_hi(hi[j]), ss);
    }
    return ss;
}
__device__ __forceinline__ float qkn_rstd(float ss, int D, float eps) { return 1.0f / sqrtf(ss / (float)D + eps); }
__device__ __forceinline__ u32x4 qkn_scale(const u32x4& x, const u32x4& w, float rstd) {  // w * bf16(x * rstd), rounded to bf16
    u32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = pack_bf2(bf_lo(w[j]) * rbf(bf_lo(x[j]) * rstd), bf_hi(w[j]) * rbf(bf_hi(x[j]) * rstd));
    return y;
}
// rotate a lane share: out_lo = bf16(bf16(lo*cos) + bf16(-hi*sin)), out_hi = bf16(bf16(hi*cos) + bf16(lo*sin))   (TF:598-599 in bf16)
__device__ __forceinline__ void rope_share(const u32x4& lo, const u32x4& hi, const u32x4& cv, const u32x4& sv, u32x4& ol, u32x4& oh) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = bf_lo(lo[j]), a1 = bf_hi(lo[j]), b0 = bf_lo(hi[j]), b1 = bf_hi(hi[j]);
        const float c0 = bf_lo(cv[j]), c1 = bf_hi(cv[j]), s0 = bf_lo(sv[j]), s1 = bf_hi(sv[j]);
        const float l0 = rbf(__fadd_rn(rbf(__fmul_rn(a0, c0)), rbf(__fmul_rn(-b0, s0))));
        const float l1 = rbf(__fadd_rn(rbf(__fmul_rn(a1, c1)), rbf(__fmul_rn(-b1, s1))));
        const float h0 = rbf(__fadd_rn(rbf(__fmul_rn(b0, c0)), rbf(__fmul_rn(a0, s0))));
        const float h1 = rbf(__fadd_rn(rbf(__fmul_rn(b1, c1)), rbf(__fmul_rn(a1, s1))));
        ol[j] = pack_bf2(l0, l1);
        oh[j] = pack_bf2(h0, h1);
    }
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
// o3v_tl_launches: kernel launches enqueued by this host thread; o3v_llm_decode reports the
// launches per decode layer from it (o3v_decode_state.host_stats)
inline thread_local long long o3v_tl_launches = 0;  // (C++17 inline variable: one instance per thread for the whole library)
#define O3V_KLAUNCH(...)               \
    do {                               \
        (void)hipGetLastError();       \
        ++o3v_tl_launches;             \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define O3V_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return O3V_ERR_LAUNCH;       \
    } while (0)
