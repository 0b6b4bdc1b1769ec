// bf16 GEMMs of the Qwen2.5-VL generate path on gfx950 (MI355X).
//
//   o3v_gemm_bf16 : out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)   MFMA-bound  (ViT, merger, LLM prefill)
//   o3v_gemv_bf16 : the same contract for M <= 8 rows               HBM-bound   (decode: weights streamed once)
//
// Both operands are K-contiguous ("NT" form: nn.Linear stores W as [out,in]), so an MFMA fragment is
// one 16-byte load for A and for B.  fp32 accumulation; the epilogue applies the reference's rounding
// points (TF:modeling_qwen2_5_vl.py:85-96, :541-554 MLPs, :137-150 merger, :692-757 residual adds).
//
// GEMM tile: 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Staging: global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave-instruction, LDS image lane-linear) with
// the XOR swizzle applied to the per-lane SOURCE address and to the ds_read_b128 (both sides), which
// makes every ds_read_b128 lane-group conflict-free on the 128-byte tile rows.  Two LDS buffers, one
// barrier per K-tile (the stage of tile t+1 is issued before the MFMAs of tile t).
#include "o3v_common.h"

#define EPI_NONE 0
#define EPI_RESIDUAL 1
#define EPI_GELU 2
#define EPI_SWIGLU 3

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

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ out, int M, int N, int K, int lda, int ldw,
                                                        int ldo, int ldr, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16K | B 16K]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
    // tiles that walk M fastest, so neighbours reuse the same W panel out of L2.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage_tile(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile(W, ldw, n0, N, 0, smem + TILE_BYTES, wave, lane);
    __syncthreads();  // emits vmcnt(0) for the pending LDS-DMA, then the barrier

    const int fr = lane & 15, fg = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * 2 * TILE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
        if (t + 1 < nk) {
            stage_tile(A, lda, m0, M, (t + 1) * BK, nxt, wave, lane);
            stage_tile(W, ldw, n0, N, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(cur + swz_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(cur + TILE_BYTES + swz_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D map of 16x16x32: col = lane&15, row = (lane>>4)*4 + reg.
    if (EPI == EPI_SWIGLU) {
        // W rows interleaved in 16-row groups: even groups = gate rows, odd groups = up rows of the same
        // 16 output columns (host packs them), so acc[i][2jj] / acc[i][2jj+1] meet in one lane.
        const int No = N >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int ng = n0 + wn * 64 + jj * 32 + fr;  // gate row index in the packed weight
                const int no = ((n0 + wn * 64) >> 1) + jj * 16 + fr;
                if (no >= No) continue;
                const float bg = bias ? bf2f(bias[ng]) : 0.f;
                const float bu = bias ? bf2f(bias[ng + 16]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
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
                const int n = n0 + wn * 64 + j * 16 + fr;
                if (n >= N) continue;
                const float bv = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
                    if (m >= M) continue;
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldr + n]);
                    if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
                    out[(size_t)m * ldo + n] = f2bf(v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Small-M weight-streaming GEMV: each wave owns R weight rows, walks K with 16-byte loads (64 lanes x
// 8 bf16 = 512 k per step), keeps M x-chunks and R w-chunks in flight, reduces across the wave once at
// the end.  x[M,K] is tiny and shared by every wave: it is read through L1/L2, the weights stream
// from HBM exactly once.  M is a template parameter so the accumulators stay in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fma8(const u32x4& w, const u32x4& x, float& acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc = fmaf(bf_lo(w[j]), bf_lo(x[j]), acc);
        acc = fmaf(bf_hi(w[j]), bf_hi(x[j]), acc);
    }
}

template <int M, int R, int EPI>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ out, int N, int K, int ldx, int ldw, int ldo,
                                                        int ldr) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    // SWIGLU: a wave owns R/2 output columns = R/2 (gate,up) row pairs of the 16-row-interleaved weight
    int rows[R];
    const int first = wave_g * R;
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int no = wave_g * (R / 2) + r;  // output column
            const int g = (no >> 4) * 32 + (no & 15);
            rows[2 * r] = g;
            rows[2 * r + 1] = g + 16;
        }
        if (wave_g * (R / 2) >= (N >> 1)) return;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = first + r;
        if (first >= N) return;
    }
    const u32x4* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = rows[r] < N ? rows[r] : N - 1;
        wp[r] = reinterpret_cast<const u32x4*>(W + (size_t)rr * ldw);
    }
    float acc[R][M];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = 0.f;

    const int nch = K >> 3;
    for (int c = lane; c < nch; c += 64) {
        u32x4 wv[R], xv[M];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = __builtin_nontemporal_load(wp[r] + c);
#pragma unroll
        for (int m = 0; m < M; ++m) xv[m] = *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) fma8(wv[r], xv[m], acc[r][m]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);

    if (lane != 0) return;
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < R / 2; ++r) {
            const int no = wave_g * (R / 2) + r;
            if (no >= (N >> 1)) continue;
            const float bg = bias ? bf2f(bias[rows[2 * r]]) : 0.f;
            const float bu = bias ? bf2f(bias[rows[2 * r + 1]]) : 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float g = rbf(acc[2 * r][m] + bg), u = rbf(acc[2 * r + 1][m] + bu);
                out[(size_t)m * ldo + no] = f2bf(rbf(silu_f(g)) * u);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = rows[r];
            if (n >= N) continue;
            const float bv = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v = acc[r][m] + bv;
                if (EPI == EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldr + n]);
                if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
                out[(size_t)m * ldo + n] = f2bf(v);
            }
        }
    }
}

template <int M, int R>
int launch_gemv(const bf16_t* X, const bf16_t* W, const bf16_t* bias, const bf16_t* res, bf16_t* out, int N, int K, int ldx,
                int ldw, int ldo, int ldr, int epi, hipStream_t s) {
    const int per_wave = (epi == EPI_SWIGLU) ? R / 2 : R;
    const int outs = (epi == EPI_SWIGLU) ? N / 2 : N;
    const int waves = (outs + per_wave - 1) / per_wave;
    dim3 grid((waves + 3) / 4), block(256);
#define O3V_GV(E) \
    O3V_KLAUNCH((gemv_bf16_kernel<M, R, E>), grid, block, 0, s, X, W, bias, res, out, N, K, ldx, ldw, ldo, ldr)
    switch (epi) {
        case EPI_NONE: O3V_GV(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_GV(EPI_RESIDUAL); break;
        case EPI_GELU: O3V_GV(EPI_GELU); break;
        case EPI_SWIGLU: O3V_GV(EPI_SWIGLU); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GV
    return O3V_OK;
}

template <int M>
int launch_gemv_m(const bf16_t* X, const bf16_t* W, const bf16_t* bias, const bf16_t* res, bf16_t* out, int N, int K,
                  int ldx, int ldw, int ldo, int ldr, int epi, hipStream_t s) {
    // rows per wave: 4 when that still gives >= 2 waves per SIMD chip-wide, else 2 (small N such as o_proj)
    const int outs = (epi == EPI_SWIGLU) ? N / 2 : N;
    if (M <= 2 && outs >= 4 * 2048 * (epi == EPI_SWIGLU ? 2 : 1))
        return launch_gemv<M, 4>(X, W, bias, res, out, N, K, ldx, ldw, ldo, ldr, epi, s);
    return launch_gemv<M, 2>(X, W, bias, res, out, N, K, ldx, ldw, ldo, ldr, epi, s);
}

}  // namespace

extern "C" int o3v_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                             int lda, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K % BK) || (lda & 7) || (ldw & 7)) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && (N % 32)) return O3V_ERR_SHAPE;
    if (M == 0) return O3V_OK;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(256);
    const size_t shmem = 4 * TILE_BYTES;
#define O3V_GM(E)                                                                                                       \
    O3V_KLAUNCH((gemm_bf16_kernel<E>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)W,            \
                       (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, tiles_m, tiles_n)
    switch (epilogue) {
        case EPI_NONE: O3V_GM(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_GM(EPI_RESIDUAL); break;
        case EPI_GELU: O3V_GM(EPI_GELU); break;
        case EPI_SWIGLU: O3V_GM(EPI_SWIGLU); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GM
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_gemv_bf16(const void* X, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                             int ldx, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    if (!X || !W || !out || M < 0 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K & 7) || (ldx & 7) || (ldw & 7) || M > 8) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && (N % 32)) return O3V_ERR_SHAPE;
    if (M == 0) return O3V_OK;
    const bf16_t *x = (const bf16_t*)X, *w = (const bf16_t*)W, *b = (const bf16_t*)bias, *r = (const bf16_t*)res;
    bf16_t* o = (bf16_t*)out;
    int rc;
    switch (M) {
        case 1: rc = launch_gemv_m<1>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 2: rc = launch_gemv_m<2>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 3: rc = launch_gemv_m<3>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 4: rc = launch_gemv_m<4>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 5: rc = launch_gemv_m<5>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 6: rc = launch_gemv_m<6>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        case 7: rc = launch_gemv_m<7>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
        default: rc = launch_gemv_m<8>(x, w, b, r, o, N, K, ldx, ldw, ldo, ldr, epilogue, stream); break;
    }
    if (rc != O3V_OK) return rc;
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
