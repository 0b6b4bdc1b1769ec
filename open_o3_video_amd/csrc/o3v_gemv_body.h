// Weight-streaming GEMV body of the decode linears (M <= 8 rows), shared by the stand-alone kernels of o3v_gemm.hip and
// the role-fused decode launch of o3v_fused.hip: both instantiate THIS code, so their results are bit-identical.
#pragma once
#include "o3v_common.h"
#include "o3v_handoff.h"

#define EPI_NONE 0
#define EPI_RESIDUAL 1
#define EPI_GELU 2
#define EPI_SWIGLU 3
#define EPI_GELU_TANH 6  // Qwen3-VL vision MLP: gelu_pytorch_tanh
#define EPI_PARTIAL 5  // split-K pass of the MFMA GEMM: raw fp32 tile to the workspace, epilogue in the reduce kernel
#define EPI_QKVROPE 4  // decode only: bias, M-RoPE, write q / append k,v to the cache (TF:557-599, :652-664)

namespace {

// ------------------------------------------------------------------------------------------------
// Small-M weight-streaming GEMV (decode).  Block = 4 waves arranged as (4/KS) row groups x KS K-slices:
// a wave owns R weight rows over its K-slice, walks it with 16-byte non-temporal loads (64 lanes x 8 bf16
// = 512 k per step, U steps in flight), fp32 FMA, one wave reduction, K-slices summed through LDS.
//   * x[M,K] is tiny and shared by every wave: read through L1/L2, or -- NORM variant -- normalised once per
//     block into LDS: the RMSNorm that precedes every q/k/v and gate/up projection (TF:65-79, :733-748) is
//     fused here with its two bf16 rounding points, which removes one launch per projection.
//   * the weights stream from HBM exactly once.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// 8 bf16 x 8 bf16 -> fp32 accumulate with v_dot2c_f32_bf16 (2 MACs per instruction on the packed pairs, no unpacking):
// 4 VALU instructions per 16-byte chunk pair instead of 24, which keeps the M = 8 (group rollout) GEMV HBM-bound.
__device__ __forceinline__ void fma8(const u32x4& w, const u32x4& x, float& acc) {
    // (element-wise locals: bit_cast applied directly to `w[j]` of a vector reference is mis-lowered by hipcc 7.2 --
    // all four j read element 0)
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    const uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w0), __builtin_bit_cast(bf16x2_t, x0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w1), __builtin_bit_cast(bf16x2_t, x1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w2), __builtin_bit_cast(bf16x2_t, x2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w3), __builtin_bit_cast(bf16x2_t, x3), acc, false);
}

// 16 fp8 (OCP e4m3fn) weights x 16 bf16 of x -> fp32 accumulate: v_cvt_scalef32_pk_bf16_fp8 widens two fp8 to two bf16
// EXACTLY (3 mantissa bits fit in 7; scale 1.0), then the same v_dot2c_f32_bf16 as the bf16 path.  8 cvt + 8 dot2 per 16
// weight bytes; the per-row dequantisation scale multiplies the finished sum.
__device__ __forceinline__ void fma16_fp8(const u32x4& w, const u32x4& xa, const u32x4& xb, float& acc) {
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    const uint32_t x0 = xa[0], x1 = xa[1], x2 = xa[2], x3 = xa[3], x4 = xb[0], x5 = xb[1], x6 = xb[2], x7 = xb[3];
#define O3V_F8(WW, XL, XH)                                                                                                   \
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(WW, 1.0f, false),                        \
                                          __builtin_bit_cast(bf16x2_t, XL), acc, false);                                    \
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(WW, 1.0f, true),                         \
                                          __builtin_bit_cast(bf16x2_t, XH), acc, false);
    O3V_F8(w0, x0, x1)
    O3V_F8(w1, x2, x3)
    O3V_F8(w2, x4, x5)
    O3V_F8(w3, x6, x7)
#undef O3V_F8
}

struct RopeArgs {  // EPI_QKVROPE destinations (one token per row m, cache slot `slot`, table row m*cs_stride+cs_off)
    const bf16_t *cosT, *sinT;
    bf16_t *qout, *kc, *vc;
    int slot, Hq, Hkv, D, Tmax, cs_stride, cs_off;
    int raw = 0;  // 1: no rotation (cos = 1, sin = 0; cosT / sinT unused) -- the q/k norm of Qwen3-VL comes first, elsewhere
};

// PUB: the outputs (q and the new K/V row; a SwiGLU / plain output row) are handed to other workgroups of the SAME launch
// (o3v_fused.hip): they are stored write-through (sc1) so that a drained store is visible beyond this XCD's L2.
template <bool PUB>
__device__ __forceinline__ void gemv_store_bf16(bf16_t* p, bf16_t v) {
    if (PUB)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_short ... sc1
    else
        *p = v;
}

// `bid` = index of this block among the blocks of the linear, `smem` = the block's dynamic LDS:
// [NORM: M*K bf16] [KS>1: 4*R*M f32] [NORM: 4*M f32]
// UU > 0: 512-wide k steps requested per trip (default: 2 for R >= 4, else 4); chunk order, hence every sum, is the same
// for any UU.
// NW = waves per workgroup (default 4).  A CU moves ~24 GB/s however many workgroups it hosts, so a projection streams at
// the chip's rate only when every CU gets the same number of weight rows: the launcher picks NW so that the grid is a whole
// number of workgroups per CU (e.g. o_proj at 7B: 1792 waves = 7 per CU -> NW 7, one workgroup per CU).  The rows of a
// wave and their sums do not depend on NW; only the fused RMSNorm's sum of squares is split over NW waves.
// WB = bytes per weight element: 2 = bf16 rows; 1 = fp8 (OCP e4m3fn) rows with one fp32 dequantisation scale per output row
// (`wscale`): a 16-byte weight chunk then spans 16 k (two 16-byte chunks of x) and the row's sum is multiplied by its scale
// before the bias.  K % 16 == 0 for WB == 1.
template <int M, int R, int KS, int EPI, bool NORM, bool PUB = false, int UU = 0, int NW = 4, int WB = 2>
__device__ __forceinline__ void gemv_body(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                          const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                          bf16_t* __restrict__ out, const bf16_t* __restrict__ norm_w, float eps, int N,
                                          int K, int ldx, int ldw, int ldo, int ldr, const RopeArgs& ra, const int bid,
                                          char* smem, const float* __restrict__ wscale = nullptr) {
    static_assert(NW % KS == 0, "K slices must divide the waves of a workgroup");
    static_assert(WB == 1 || WB == 2, "bf16 or fp8 weights");
    constexpr int RG = NW / KS;                                  // row groups per block
    constexpr int NT = NW * 64;                                  // threads per block
    constexpr int XPC = 2 / WB;                                  // 16-byte chunks of x per 16-byte chunk of weights
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = wave / KS, ks = wave % KS;
    const int nch = K >> (WB == 1 ? 4 : 3);                      // weight chunks per row
    const int nxc = K >> 3;                                      // x chunks per row

    // ---- rows of this wave.  SWIGLU: R/2 output columns = R/2 (gate,up) row pairs of the 16-row-interleaved weight
    const int grp = bid * RG + rg;
    int rows[R];
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int no = grp * (R / 2) + r;
            const int g = (no >> 4) * 32 + (no & 15);
            rows[2 * r] = g;
            rows[2 * r + 1] = g + 16;
        }
    } else if (EPI == EPI_QKVROPE) {
        // a wave owns the rotary pair (j, j + D/2) of one head, so the rotation needs no second wave
        const int half = ra.D >> 1;
        rows[0] = (grp / half) * ra.D + (grp % half);
        rows[R - 1] = rows[0] + half;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = grp * R + r;
    }
    const u32x4* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = rows[r] < N ? rows[r] : N - 1;  // tail rows re-read a valid row, never stored
        wp[r] = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(W) + (size_t)rr * ldw * WB);
    }
    // Epilogue operands (bias, residual, rotary cos/sin) are requested here, a whole kernel ahead of their use: these
    // kernels live for 7-25 us, and a dependent L2 round trip at the tail is 5-10 % of that.
    float e_scale[R];
#pragma unroll
    for (int r = 0; r < R; ++r) e_scale[r] = (WB == 1) ? wscale[rows[r] < N ? rows[r] : N - 1] : 1.0f;
    float e_bias[R], e_res[EPI == EPI_RESIDUAL ? R : 1][EPI == EPI_RESIDUAL ? M : 1], e_cos[EPI == EPI_QKVROPE ? M : 1],
        e_sin[EPI == EPI_QKVROPE ? M : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = rows[r] < N ? rows[r] : N - 1;
        e_bias[r] = bias ? bf2f(bias[rr]) : 0.f;
        if (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int m = 0; m < M; ++m) e_res[r][m] = bf2f(res[(size_t)m * ldr + rr]);
        }
    }
    if (EPI == EPI_QKVROPE) {
        const int jj = (rows[0] < N ? rows[0] : 0) % ra.D;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const size_t cs = ((size_t)m * ra.cs_stride + ra.cs_off) * ra.D + jj;
            e_cos[m] = ra.raw ? 1.0f : bf2f(ra.cosT[cs]);  // raw: x * 1 + (-y) * 0 == x exactly
            e_sin[m] = ra.raw ? 0.0f : bf2f(ra.sinT[cs]);
        }
    }
    // K-slice in whole 64-chunk steps
    const int steps = (nch + 63) >> 6;
    const int sps = (steps + KS - 1) / KS;
    const int c_begin = ks * sps * 64;
    int c_end = c_begin + sps * 64;
    c_end = c_end < nch ? c_end : nch;
    constexpr int U = UU > 0 ? UU : ((R >= 4) ? 2 : 4);

    // weight loads of one trip (U steps x R rows, 16 B per lane each): issued as early as possible
    u32x4 wv[U][R];
    u32x4 xg[NORM ? 1 : U][NORM ? 1 : M][XPC];  // un-normalised x comes from global memory (L2): fetched one trip ahead, with the weights
    auto load_x = [&](int c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64 + lane;
            const int cc = c < c_end ? c : c_begin;
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int h = 0; h < XPC; ++h)
                    xg[NORM ? 0 : u][NORM ? 0 : m][h] =
                        *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + ((size_t)cc * XPC + h) * 8);
        }
    };
    auto load_w = [&](int c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64 + lane;
            const bool in = c < c_end;
            const int cc = in ? c : c_begin;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                wv[u][r] = __builtin_nontemporal_load(wp[r] + cc);
                if (!in) wv[u][r] = (u32x4){0, 0, 0, 0};
            }
        }
        if (!NORM) load_x(c0);
    };

    if (NORM) {
        // ---- block-wide RMSNorm of x into LDS: xs[m][k] = bf16(w[k] * bf16(x[m][k] * rstd[m])).
        // x (a few KiB, L2-resident) is requested first, then the first trip of weight loads, so that the
        // HBM latency of the weights runs under the norm instead of after it.
        // The sum of squares is added in ONE order whatever the workgroup size: 256 virtual threads t (chunks t, t + 256, ...
        // ascending, one fmaf chain each), 4 virtual waves of 64 (the wave_sum butterfly), their sums added 0..3.  Real wave w
        // plays the virtual waves w, w + NW, ... -- with 4 waves exactly its own, with fewer several, waves past the fourth
        // none.  So rstd, and with it every normalised element, does not depend on the decomposition the launcher picked
        // (or on which kernel -- stand-alone, one-launch block, norm_finish of the layer block -- computes it).
        constexpr int VW = (4 + NW - 1) / NW;        // virtual waves per real wave (at most)
        constexpr bool ALLV = (4 % NW) == 0;         // 1, 2, 4 waves: every (wave, v) is one of the four virtual waves
        constexpr int XC = 2;                        // chunks per virtual thread kept in registers (K <= 4096)
        const bool small = (nxc <= XC * 256) && (M <= 4);
        u32x4 xr[VW][XC][M <= 4 ? M : 1], wnr[VW][XC];  // x chunks and the norm-weight chunks that go with them: one latency, not two
        float ss[VW][M];
#pragma unroll
        for (int v = 0; v < VW; ++v)
#pragma unroll
            for (int m = 0; m < M; ++m) ss[v][m] = 0.f;
        if (small) {
#pragma unroll
            for (int v = 0; v < VW; ++v)
#pragma unroll
                for (int i = 0; i < XC; ++i) {
                    const int vw = wave + v * NW;
                    const int c = vw * 64 + lane + i * 256;
                    const bool in = (ALLV || vw < 4) && c < nxc;
#pragma unroll
                    for (int m = 0; m < (M <= 4 ? M : 1); ++m)
                        xr[v][i][m] = in ? *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8) : (u32x4){0, 0, 0, 0};
                    wnr[v][i] = in ? *reinterpret_cast<const u32x4*>(norm_w + (size_t)c * 8) : (u32x4){0, 0, 0, 0};
                }
        }
        if (c_begin < c_end) load_w(c_begin);
        if (small) {
#pragma unroll
            for (int v = 0; v < VW; ++v)
#pragma unroll
                for (int i = 0; i < XC; ++i)
#pragma unroll
                    for (int m = 0; m < (M <= 4 ? M : 1); ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ss[v][m] = fmaf(bf_lo(xr[v][i][m][j]), bf_lo(xr[v][i][m][j]), ss[v][m]);
                            ss[v][m] = fmaf(bf_hi(xr[v][i][m][j]), bf_hi(xr[v][i][m][j]), ss[v][m]);
                        }
        } else {
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const int vw = wave + v * NW;
                for (int c = vw * 64 + lane; (ALLV || vw < 4) && c < nxc; c += 256) {
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const u32x4 xv4 = *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            ss[v][m] = fmaf(bf_lo(xv4[j]), bf_lo(xv4[j]), ss[v][m]);
                            ss[v][m] = fmaf(bf_hi(xv4[j]), bf_hi(xv4[j]), ss[v][m]);
                        }
                    }
                }
            }
        }
        float* red = reinterpret_cast<float*>(smem + (size_t)M * K * 2 + NW * R * M * 4);  // [4 virtual waves][M]
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            const int vw = wave + v * NW;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                ss[v][m] = wave_sum(ss[v][m]);
                if (lane == 0 && (ALLV || vw < 4)) red[vw * M + m] = ss[v][m];
            }
        }
        __syncthreads();
        float rstd[M];
#pragma unroll
        for (int m = 0; m < M; ++m)
        {
            float t = red[m];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) t += red[w2 * M + m];
            rstd[m] = 1.0f / sqrtf(t / (float)K + eps);
        }
        auto norm_store = [&](int c, int m, const u32x4& v, const u32x4& wn) {
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf2(bf_lo(wn[j]) * rbf(bf_lo(v[j]) * rstd[m]), bf_hi(wn[j]) * rbf(bf_hi(v[j]) * rstd[m]));
            *reinterpret_cast<u32x4*>(smem + ((size_t)m * K + (size_t)c * 8) * 2) = o;
        };
        if (small) {
#pragma unroll
            for (int v = 0; v < VW; ++v)
#pragma unroll
                for (int i = 0; i < XC; ++i) {
                    const int vw = wave + v * NW;
                    const int c = vw * 64 + lane + i * 256;
                    if ((ALLV || vw < 4) && c < nxc) {
#pragma unroll
                        for (int m = 0; m < (M <= 4 ? M : 1); ++m) norm_store(c, m, xr[v][i][m], wnr[v][i]);
                    }
                }
        } else {
            for (int c = threadIdx.x; c < nxc; c += NT) {   // element-wise: any assignment
                const u32x4 wn = *reinterpret_cast<const u32x4*>(norm_w + (size_t)c * 8);
#pragma unroll
                for (int m = 0; m < M; ++m)
                    norm_store(c, m, *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8), wn);
            }
        }
        __syncthreads();
    } else {
        if (c_begin < c_end) load_w(c_begin);
    }

    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

    for (int c0 = c_begin; c0 < c_end; c0 += 64 * U) {
        u32x4 xv[U][M][XPC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64 + lane;
            const int cc = c < c_end ? c : c_begin;  // out-of-range lanes meet zeroed weights
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int h = 0; h < XPC; ++h)
                    xv[u][m][h] = NORM ? *reinterpret_cast<const u32x4*>(smem + ((size_t)m * K + ((size_t)cc * XPC + h) * 8) * 2)
                                       : xg[NORM ? 0 : u][NORM ? 0 : m][h];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    if constexpr (WB == 1)
                        fma16_fp8(wv[u][r], xv[u][m][0], xv[u][m][XPC - 1], acc[r][m]);
                    else
                        fma8(wv[u][r], xv[u][m][0], acc[r][m]);
                }
        if (c0 + 64 * U < c_end) load_w(c0 + 64 * U);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);

    if (KS > 1) {
        float* part = reinterpret_cast<float*>(smem + (NORM ? (size_t)M * K * 2 : 0));
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) part[(wave * R + r) * M + m] = acc[r][m];
        }
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float t = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < KS; ++k2) t += part[((rg * KS + k2) * R + r) * M + m];
                acc[r][m] = t;
            }
    }
    if (lane != 0) return;
    if (WB == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] *= e_scale[r];
    }
    if (EPI == EPI_QKVROPE) {
        if (rows[0] >= N) return;
        const int half = ra.D >> 1, head = rows[0] / ra.D, j = rows[0] % ra.D;
        const float b0 = e_bias[0], b1 = e_bias[R - 1];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float v0 = rbf(acc[0][m] + b0), v1 = rbf(acc[R - 1][m] + b1);
            if (head >= ra.Hq + ra.Hkv) {  // v: no rotation
                bf16_t* dst = ra.vc + (((size_t)m * ra.Hkv + (head - ra.Hq - ra.Hkv)) * ra.Tmax + ra.slot) * ra.D;
                gemv_store_bf16<PUB>(dst + j, f2bf(v0));
                gemv_store_bf16<PUB>(dst + j + half, f2bf(v1));
                continue;
            }
            const float c = e_cos[EPI == EPI_QKVROPE ? m : 0], sn = e_sin[EPI == EPI_QKVROPE ? m : 0];
            // TF:598-599 in bf16: bf16(bf16(x*cos) + bf16(rotate_half(x)*sin))
            const float o0 = __fadd_rn(rbf(__fmul_rn(v0, c)), rbf(__fmul_rn(-v1, sn)));
            const float o1 = __fadd_rn(rbf(__fmul_rn(v1, c)), rbf(__fmul_rn(v0, sn)));
            bf16_t* dst = head < ra.Hq ? ra.qout + ((size_t)m * ra.Hq + head) * ra.D
                                       : ra.kc + (((size_t)m * ra.Hkv + (head - ra.Hq)) * ra.Tmax + ra.slot) * ra.D;
            gemv_store_bf16<PUB>(dst + j, f2bf(o0));
            gemv_store_bf16<PUB>(dst + j + half, f2bf(o1));
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int no = grp * (R / 2) + r;
            if (no >= (N >> 1)) continue;
            const float bg = e_bias[2 * r];
            const float bu = e_bias[2 * r + 1];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float g = rbf(acc[2 * r][m] + bg), u = rbf(acc[2 * r + 1][m] + bu);
                gemv_store_bf16<PUB>(out + (size_t)m * ldo + no, f2bf(rbf(silu_f(g)) * u));
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = rows[r];
            if (n >= N) continue;
            const float bv = e_bias[r];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v = acc[r][m] + bv;
                if (EPI == EPI_RESIDUAL) v = rbf(v) + e_res[EPI == EPI_RESIDUAL ? r : 0][EPI == EPI_RESIDUAL ? m : 0];
                if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
            if (EPI == EPI_GELU_TANH) v = gelu_tanh_f(rbf(v));
                gemv_store_bf16<PUB>(out + (size_t)m * ldo + n, f2bf(v));
            }
        }
    }
}


}  // namespace
