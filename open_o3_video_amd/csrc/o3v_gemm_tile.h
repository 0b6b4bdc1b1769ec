// Shared pieces of the LDS-staged MFMA GEMMs (o3v_gemm.hip: bf16; o3v_fp8.hip: fp8 x fp8): the 128-byte-row tile swizzle,
// the global -> LDS staging of a 128-row tile and the fused epilogue of one wave's 64 x 64 fp32 sub-tile.
#pragma once
#include "o3v_common.h"
#include "o3v_gemv_body.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void lds_void;

// byte offset inside a [128][64] bf16 tile of logical (row, 16-byte chunk c in 0..7)
__device__ __forceinline__ int swz_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// issue the global->LDS copies of one 128x64 tile: 16 wave-instructions of 1 KiB, 4 per wave.
// LDS position p = instr*1024 + lane*16 holds logical chunk (row = p/128, c = ((p%128)/16) ^ ((row>>1)&7)).
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int rows_valid, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int instr = wave * 4 + i;
        const int row = instr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < rows_valid ? grow : rows_valid - 1;  // clamp: tail rows re-read a valid row, never stored
        const bf16_t* src = g + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + instr * 1024), 16, 0, 0);
    }
}

// Epilogue of one wave's 64x64 fp32 sub-tile whose top-left output element is (mrow0, ncol0); `et` is the wave's own
// 64 x 68 float staging area in LDS.  C/D map of 16x16x32: col = lane&15, row = (lane>>4)*4 + reg.
template <int EPI>
__device__ __forceinline__ void wave_epilogue(const f32x4 (&acc)[4][4], float* et, int lane, int mrow0, int ncol0, int M, int N,
                                              const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                              bf16_t* __restrict__ out, int ldo, int ldr) {
    const int fr = lane & 15, fg = lane >> 4;
    // Fast path: the wave's 64x64 fp32 sub-tile goes through LDS (row stride 68 floats: the two 32-lane halves of a
    // ds_write_b32 land on disjoint banks) and comes back row-wise, 8 consecutive columns per lane, so bias / residual are
    // 16-byte loads and every output row segment is a 16-byte store of a full 128-byte line per 8 lanes.
    const bool vec_ok = ((N & 7) == 0) && ((ldo & 7) == 0) && (EPI != EPI_RESIDUAL || (ldr & 7) == 0);
    if (vec_ok) {
        constexpr int ES = 68;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) et[(i * 16 + fg * 4 + r) * ES + j * 16 + fr] = acc[i][j][r];
        // same wave writes and reads its own region; LDS operations of a wave complete in order
        if (EPI == EPI_SWIGLU) {
            const int no0 = (ncol0) >> 1;  // first output column of this wave (32 per wave)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 16 + (lane >> 2), oc = (lane & 3) * 8;
                const int m = mrow0 + row, no = no0 + oc;
                if (m >= M || no >= (N >> 1)) continue;
                const int gcol = (oc >> 4) * 32 + (oc & 15);  // gate columns in the tile; up = +16
                const float* gp = et + row * ES + gcol;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(gp + 16), u1 = *reinterpret_cast<const f32x4*>(gp + 20);
                float gb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (bias) {
                    const int ng = ncol0 + gcol;
                    const u32x4 bg = *reinterpret_cast<const u32x4*>(bias + ng), bu = *reinterpret_cast<const u32x4*>(bias + ng + 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        gb[2 * q] = bf_lo(bg[q]);
                        gb[2 * q + 1] = bf_hi(bg[q]);
                        ub[2 * q] = bf_lo(bu[q]);
                        ub[2 * q + 1] = bf_hi(bu[q]);
                    }
                }
                float o8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float gv = rbf((q < 4 ? g0[q] : g1[q - 4]) + gb[q]);
                    const float uv = rbf((q < 4 ? u0[q] : u1[q - 4]) + ub[q]);
                    o8[q] = rbf(silu_f(gv)) * uv;
                }
                u32x4 pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf2(o8[2 * q], o8[2 * q + 1]);
                *reinterpret_cast<u32x4*>(out + (size_t)m * ldo + no) = pk;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                const int m = mrow0 + row, n = ncol0 + c8;
                if (m >= M || n >= N) continue;
                const float* ep = et + row * ES + c8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep), v1 = *reinterpret_cast<const f32x4*>(ep + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (bias) {
                    const u32x4 bb = *reinterpret_cast<const u32x4*>(bias + n);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] += bf_lo(bb[q]);
                        v[2 * q + 1] += bf_hi(bb[q]);
                    }
                }
                if (EPI == EPI_RESIDUAL) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(res + (size_t)m * ldr + n);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[2 * q] = rbf(v[2 * q]) + bf_lo(rr[q]);
                        v[2 * q + 1] = rbf(v[2 * q + 1]) + bf_hi(rr[q]);
                    }
                }
                if (EPI == EPI_GELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = gelu_erf_f(rbf(v[q]));
                }
                if (EPI == EPI_GELU_TANH) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = gelu_tanh_f(rbf(v[q]));
                }
                u32x4 pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf2(v[2 * q], v[2 * q + 1]);
                *reinterpret_cast<u32x4*>(out + (size_t)m * ldo + n) = pk;
            }
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
        // W rows interleaved in 16-row groups: even groups = gate rows, odd groups = up rows of the same
        // 16 output columns (host packs them), so acc[i][2jj] / acc[i][2jj+1] meet in one lane.
        const int No = N >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int ng = ncol0 + jj * 32 + fr;  // gate row index in the packed weight
                const int no = ((ncol0) >> 1) + jj * 16 + fr;
                if (no >= No) continue;
                const float bg = bias ? bf2f(bias[ng]) : 0.f;
                const float bu = bias ? bf2f(bias[ng + 16]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + i * 16 + fg * 4 + r;
                    if (m >= M) continue;
                    const float g = rbf(acc[i][2 * jj][r] + bg);
                    const float u = rbf(acc[i][2 * jj + 1][r] + bu);
                    out[(size_t)m * ldo + no] = f2bf(rbf(silu_f(g)) * u);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol0 + j * 16 + fr;
                if (n >= N) continue;
                const float bv = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + i * 16 + fg * 4 + r;
                    if (m >= M) continue;
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldr + n]);
                    if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
            if (EPI == EPI_GELU_TANH) v = gelu_tanh_f(rbf(v));
                    out[(size_t)m * ldo + n] = f2bf(v);
                }
            }
        }
    }
}

}  // namespace
