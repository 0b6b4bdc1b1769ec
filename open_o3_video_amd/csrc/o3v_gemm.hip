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
#include <cstring>
#include "../../include/o3v.h"
#include "o3v_common.h"
#include "o3v_gemv_body.h"

#include "o3v_gemm_tile.h"

namespace {

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

    // split-K (EPI_PARTIAL): blockIdx.y owns k-steps [t0, t1)
    const int nk_all = K / BK;
    const int per_split = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t0 = (EPI == EPI_PARTIAL) ? (int)blockIdx.y * per_split : 0;
    const int nk = (EPI == EPI_PARTIAL) ? (t0 + per_split < nk_all ? t0 + per_split : nk_all) : nk_all;
    const int fr = lane & 15, fg = lane >> 4;
    if (t0 < nk) {
        stage_tile(A, lda, m0, M, t0 * BK, smem, wave, lane);
        stage_tile(W, ldw, n0, N, t0 * BK, smem + TILE_BYTES, wave, lane);
    }
    __syncthreads();  // emits vmcnt(0) for the pending LDS-DMA, then the barrier

    for (int t = t0; t < nk; ++t) {
        char* cur = smem + ((t - t0) & 1) * 2 * TILE_BYTES;
        char* nxt = smem + ((t - t0 + 1) & 1) * 2 * TILE_BYTES;
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
    if (EPI == EPI_PARTIAL) {
        float* part = reinterpret_cast<float*>(out) + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
                    if (m < M && n < N) part[(size_t)m * N + n] = acc[i][j][r];
                }
            }
        return;
    }
    // main loop ended with a barrier: the tiles are dead, every wave stages its sub-tile in its own 64 x 68 float region
    wave_epilogue<EPI>(acc, reinterpret_cast<float*>(smem) + wave * 64 * 68, lane, m0 + wm * 64, n0 + wn * 64, M, N, bias, res, out,
                       ldo, ldr);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4), each wave a 128 x 64 sub-tile (8 x 4 MFMA blocks).  The 128-tile kernel above is
// LDS-bound: per 32-deep k-slice a wave reads 8 KiB of fragments for 16 MFMAs and the block stages 32 KiB per k-step.
// Here a wave reads 12 KiB for 32 MFMAs and the block stages 64 KiB for four times the work: ~2/3 of the LDS bytes per
// FLOP at the same two waves per SIMD.  One block per CU (128 KiB of tiles); the launcher uses it for the big prefill
// GEMMs whose 256-tile grid still fills the chip evenly.
// ------------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int TILE2_BYTES = BM2 * BK * 2;  // 32 KiB per operand tile

__device__ __forceinline__ void stage_tile256(const bf16_t* __restrict__ g, int ld, int row0, int rows_valid, int k0,
                                              char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int instr = wave * 4 + i;  // 32 wave-instructions of 1 KiB per tile, 4 per wave
        const int row = instr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < rows_valid ? grow : rows_valid - 1;
        const bf16_t* src = g + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + instr * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                           const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                           bf16_t* __restrict__ out, int M, int N, int K, int lda, int ldw,
                                                           int ldo, int ldr, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32K | B 32K]; epilogue: 8 x 64 x 68 floats
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM2, n0 = tn * BM2;

    f32x4 acc[2][4][4];  // [row half][i][j]: rows wm*128 + half*64 + i*16, cols wn*64 + j*16
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    const int fr = lane & 15, fg = lane >> 4;
    stage_tile256(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile256(W, ldw, n0, N, 0, smem + TILE2_BYTES, wave, lane);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * 2 * TILE2_BYTES;
        char* nxt = smem + ((t + 1) & 1) * 2 * TILE2_BYTES;
        if (t + 1 < nk) {
            stage_tile256(A, lda, m0, M, (t + 1) * BK, nxt, wave, lane);
            stage_tile256(W, ldw, n0, N, (t + 1) * BK, nxt + TILE2_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(cur + TILE2_BYTES + swz_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wm * 128 + h * 64 + i * 16 + fr;
                    af[i] = *reinterpret_cast<const bf16x8*>(cur + swz_off(row, ks * 4 + fg));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[h][i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* et = reinterpret_cast<float*>(smem) + wave * 64 * 68;
#pragma unroll
    for (int h = 0; h < 2; ++h)
        wave_epilogue<EPI>(acc[h], et, lane, m0 + wm * 128 + h * 64, n0 + wn * 64, M, N, bias, res, out, ldo, ldr);
}

// ------------------------------------------------------------------------------------------------
// Small-M weight-streaming GEMV (decode): body in o3v_gemv_body.h (shared with the fused decode launch).
// ------------------------------------------------------------------------------------------------
template <int M, int R, int KS, int EPI, bool NORM, int NW = 4, int WB = 2, int UU = 0>
__global__ __launch_bounds__(NW * 64) void gemv_bf16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ out, const bf16_t* __restrict__ norm_w,
                                                            float eps, int N, int K, int ldx, int ldw, int ldo, int ldr,
                                                            RopeArgs ra, const float* __restrict__ wscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NORM: M*K bf16] [KS>1: NW*R*M f32] [NORM: NW*M f32]
    gemv_body<M, R, KS, EPI, NORM, false, UU, NW, WB>(X, W, bias, res, out, norm_w, eps, N, K, ldx, ldw, ldo, ldr, ra, blockIdx.x, smem,
                                                      wscale);
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM on the matrix cores for 2 <= M <= 32 rows (group rollout: G completions decode together).  The
// scalar GEMV above is VALU/LDS-bound beyond M ~ 2 (M*R dot products per weight chunk); here one
// v_mfma_f32_16x16x32_bf16 multiplies 16 weight rows x 32 k against all M rows of x at once:
//   A = W[16 rows][32 k]  (lane (row = lane&15, g = lane>>4) loads 16 B at W[row][k0 + 8g], straight from global)
//   B = x^T[32 k][16 cols] (lane (m = lane&15, g) reads x[m][k0 + 8g..]; rows m >= M are zero)
//   C[row][m]: lane (m = lane&15) holds rows 4g..4g+3 -> bias / residual / SwiGLU / RoPE epilogues are lane-local.
// A wave owns RB row blocks (2 for SwiGLU: the gate and up blocks of the same 16 columns; 2 for the rotary pair
// blocks j and j + D/2), the 4 waves of a block either take 4 different row groups (KS = 1) or split K (KS = 4).
// NORM: RMSNorm of x fused, normalised rows kept in LDS with a 16-byte row skew (conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------------
// CB: 16-row column blocks of x (M <= 16 * CB): every weight fragment feeds CB MFMAs, so 17..32 rows still stream the weights once.
//
// TailNorm (EPI_RESIDUAL, 8..32 rows): the linear also produces h = RMSNorm(out; w) for the NEXT linear, so the stand-alone norm
// launch between them (5.9 us at 8 rows x 3584, nearly all of it launch boundary) disappears.  Protocol of o3v_handoff.h: every
// storing wave writes its part of `out` through (sc1), drains its stores and draws a ticket; the waves that draw the LAST M tickets
// of the launch each take one row -- they wait until the very last ticket holder has published the epoch, read their row past L1
// (sc1) and normalise it with rmsnorm_row_wave, i.e. the code and the rounding of rmsnorm_kernel: h is bit-identical to
// o3v_rmsnorm(out).  The grid (N / 16 row groups, <= 256 workgroups) is resident at once, the waits are bounded and sticky.
struct TailNorm {
    uint32_t* sync = nullptr;   // nullptr: off.  Ticket line, "all stored" line and the time-out word (O3V_SYNC_TAIL_*, O3V_SYNC_TMO_WORD)
    uint32_t epoch = 0, total = 0;  // 1, 2, ...: index of this launch among those sharing the lines; storing waves per launch
    const bf16_t* w = nullptr;  // the next RMSNorm's weight [N]
    bf16_t* h = nullptr;        // [M, ldh] normalised rows
    int ldh = 0;
    float eps = 0.f;
};

// RX: 16-row weight blocks per wave for the single-block epilogues (NONE / RESIDUAL / GELU).  At 17..32 rows of x a k-step reads 2 KiB
// of x fragments (L2) per KiB of weights; two adjacent weight blocks per wave share them (the paired epilogues always did).
template <int EPI, bool NORM, int KS, bool PACKED, int UT = 8, int CB = 1, int RX = 1>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ out, const bf16_t* __restrict__ norm_w,
                                                        float eps, int M, int N, int K, int ldx, int ldw, int ldo, int ldr,
                                                        RopeArgs ra, TailNorm tn) {
    constexpr int RB = (EPI == EPI_SWIGLU || EPI == EPI_QKVROPE) ? 2 : RX;
    constexpr int RG = 4 / KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NORM: M x (K*2+16)] [KS>1: 4 x RB x 64 x 4 f32] [red]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int rg = wave / KS, ks = wave % KS;
    const int xstride = K * 2 + 16;
    const size_t norm_bytes = NORM ? (size_t)M * xstride : 0;  // only the M live rows are staged

    // ---- row blocks of this wave
    const int grp = blockIdx.x * RG + rg;
    int rb0[RB];
    if (EPI == EPI_SWIGLU) {
        rb0[0] = grp * 32;            // gate rows of output columns 16*grp .. 16*grp+15
        rb0[RB - 1] = grp * 32 + 16;  // their up rows
    } else if (EPI == EPI_QKVROPE) {
        const int bph = ra.D / 32;    // 16-row blocks per half head
        const int head = grp / bph, jb = grp % bph;
        rb0[0] = head * ra.D + jb * 16;
        rb0[RB - 1] = rb0[0] + ra.D / 2;
    } else {
#pragma unroll
        for (int b = 0; b < RB; ++b) rb0[b] = (grp * RB + b) * 16;
    }
    const int nks = K >> 5;                      // 32-wide k-steps
    // PACKED: W is the fragment-major image [N/16][K/32][64 lanes][8] (weights.py pack_mfma_fragments): the A fragment
    // of (row block, k-step) is one contiguous KiB, so the wave streams exactly like the scalar GEMV does.
    const bf16_t* wrow[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        if (PACKED) {
            const int blk = (rb0[b] < N ? rb0[b] : 0) >> 4;
            wrow[b] = W + ((size_t)blk * nks * 64 + lane) * 8;
        } else {
            int r = rb0[b] + fr;
            r = r < N ? r : N - 1;
            wrow[b] = W + (size_t)r * ldw + fg * 8;
        }
    }
    const int per = (nks + KS - 1) / KS;
    const int s_begin = ks * per;
    int s_end = s_begin + per;
    s_end = s_end < nks ? s_end : nks;

    // ---- epilogue operands of this lane (C[n = rb0 + 4*fg + r][m = fr]): requested now, a kernel's length ahead of their use
    float e_bias[RB][4], e_res[EPI == EPI_RESIDUAL ? RB : 1][CB][4], e_cos[CB][4], e_sin[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int m = fr + 16 * cb < M ? fr + 16 * cb : 0;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = rb0[b] + fg * 4 + r;
                n = n < N ? n : N - 1;
                e_bias[b][r] = bias ? bf2f(bias[n]) : 0.f;
                if (EPI == EPI_RESIDUAL) e_res[b][cb][r] = bf2f(res[(size_t)m * ldr + n]);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (EPI != EPI_RESIDUAL) e_res[0][cb][r] = 0.f;
            if (EPI == EPI_QKVROPE) {
                const size_t cs = ((size_t)m * ra.cs_stride + ra.cs_off) * ra.D + (rb0[0] < N ? rb0[0] : 0) % ra.D + fg * 4 + r;
                e_cos[cb][r] = bf2f(ra.cosT[cs]);
                e_sin[cb][r] = bf2f(ra.sinT[cs]);
            } else {
                e_cos[cb][r] = e_sin[cb][r] = 0.f;
            }
        }
    }
    // ---- weight stream, double buffered: two trips of U k-steps (UT KiB each) are in flight per wave, and the first one
    // is issued BEFORE the RMSNorm prologue so the HBM latency of the first weights hides the norm
    f32x4 acc[RB][CB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[b][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = UT / RB;
    bf16x8 wf0[U][RB], wf1[U][RB], xf0[U][CB], xf1[U][CB];
    auto load_w = [&](bf16x8 (&wf)[U][RB], bf16x8 (&xf)[U][CB], int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sidx = s0 + u;
            if (sidx < s_end) {  // wave-uniform
                const int kk = sidx * 32;
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    wf[u][b] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow[b] + (PACKED ? (size_t)sidx * 512 : (size_t)kk)));
                if (!NORM) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        xf[u][cb] = fr + 16 * cb < M ? *reinterpret_cast<const bf16x8*>(X + (size_t)(fr + 16 * cb) * ldx + kk + fg * 8)
                                                     : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            } else {
#pragma unroll
                for (int b = 0; b < RB; ++b) wf[u][b] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (!NORM) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) xf[u][cb] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        }
    };
    auto compute = [&](bf16x8 (&wf)[U][RB], bf16x8 (&xf)[U][CB], int s0) {
        if (NORM) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sidx = s0 + u;
                const int kk = (sidx < s_end ? sidx : s_begin) * 32;  // past the end the weights are zero: any finite x will do
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    xf[u][cb] = fr + 16 * cb < M
                                    ? *reinterpret_cast<const bf16x8*>(smem + (size_t)(fr + 16 * cb) * xstride + (size_t)(kk + fg * 8) * 2)
                                    : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][b], xf[u][cb], acc[b][cb], 0, 0, 0);
    };
    load_w(wf0, xf0, s_begin);

    if (NORM) {
        // RMSNorm of the M rows into LDS, one wave per row (rows wave, wave+4, ...): wave-level reduction only, a single
        // block barrier at the end.  Row chunks stay in registers between the two passes when K <= 8 * 512.
        const int nch = K >> 3;
        for (int m = wave; m < M; m += 4) {
            u32x4 xr[8];
            float ss = 0.f;
            const bool in_regs = nch <= 8 * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + i * 64;
                xr[i] = (in_regs && c < nch) ? *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8) : (u32x4){0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ss = fmaf(bf_lo(xr[i][j]), bf_lo(xr[i][j]), ss);
                    ss = fmaf(bf_hi(xr[i][j]), bf_hi(xr[i][j]), ss);
                }
            }
            if (!in_regs) {
                for (int c = lane; c < nch; c += 64) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ss = fmaf(bf_lo(v[j]), bf_lo(v[j]), ss);
                        ss = fmaf(bf_hi(v[j]), bf_hi(v[j]), ss);
                    }
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)K + eps);
            auto put = [&](int c, const u32x4& v) {
                const u32x4 wn = *reinterpret_cast<const u32x4*>(norm_w + (size_t)c * 8);
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf2(bf_lo(wn[j]) * rbf(bf_lo(v[j]) * rstd), bf_hi(wn[j]) * rbf(bf_hi(v[j]) * rstd));
                *reinterpret_cast<u32x4*>(smem + (size_t)m * xstride + (size_t)c * 16) = o;
            };
            if (in_regs) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = lane + i * 64;
                    if (c < nch) put(c, xr[i]);
                }
            } else {
                for (int c = lane; c < nch; c += 64) put(c, *reinterpret_cast<const u32x4*>(X + (size_t)m * ldx + (size_t)c * 8));
            }
        }
        __syncthreads();
    }

    for (int s0 = s_begin; s0 < s_end; s0 += 2 * U) {
        load_w(wf1, xf1, s0 + U);
        compute(wf0, xf0, s0);
        load_w(wf0, xf0, s0 + 2 * U);
        compute(wf1, xf1, s0 + U);
    }
    if (KS > 1) {
        f32x4* part = reinterpret_cast<f32x4*>(smem + norm_bytes);
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) part[((wave * RB + b) * CB + cb) * 64 + lane] = acc[b][cb];
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k2 = 0; k2 < KS; ++k2) t += part[(((rg * KS + k2) * RB + b) * CB + cb) * 64 + lane];
                acc[b][cb] = t;
            }
    }
    // ---- epilogue: this lane holds C[n = rb0 + 4*fg + r][m = fr + 16 cb]
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
    const int m = fr + 16 * cb;
    if (m >= M) continue;
    if (EPI == EPI_SWIGLU) {
        const int no0 = grp * 16 + fg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int no = no0 + r;
            if (no >= (N >> 1)) continue;
            const float bg = e_bias[0][r], bu = e_bias[RB - 1][r];
            const float g = rbf(acc[0][cb][r] + bg), u = rbf(acc[RB - 1][cb][r] + bu);
            out[(size_t)m * ldo + no] = f2bf(rbf(silu_f(g)) * u);
        }
    } else if (EPI == EPI_QKVROPE) {
        const int half = ra.D >> 1, head = rb0[0] / ra.D, j0 = rb0[0] % ra.D + fg * 4;
        if (rb0[0] >= N) return;  // block-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + r;
            const float b0 = e_bias[0][r], b1 = e_bias[RB - 1][r];
            const float v0 = rbf(acc[0][cb][r] + b0), v1 = rbf(acc[RB - 1][cb][r] + b1);
            if (head >= ra.Hq + ra.Hkv) {
                bf16_t* dst = ra.vc + (((size_t)m * ra.Hkv + (head - ra.Hq - ra.Hkv)) * ra.Tmax + ra.slot) * ra.D;
                dst[j] = f2bf(v0);
                dst[j + half] = f2bf(v1);
                continue;
            }
            const float c = e_cos[cb][r], sn = e_sin[cb][r];
            const float o0 = __fadd_rn(rbf(__fmul_rn(v0, c)), rbf(__fmul_rn(-v1, sn)));
            const float o1 = __fadd_rn(rbf(__fmul_rn(v1, c)), rbf(__fmul_rn(v0, sn)));
            bf16_t* dst = head < ra.Hq ? ra.qout + ((size_t)m * ra.Hq + head) * ra.D
                                       : ra.kc + (((size_t)m * ra.Hkv + (head - ra.Hq)) * ra.Tmax + ra.slot) * ra.D;
            dst[j] = f2bf(o0);
            dst[j + half] = f2bf(o1);
        }
    } else {
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rb0[b] + fg * 4 + r;
                if (n >= N) continue;
                float v = acc[b][cb][r] + e_bias[b][r];
                if (EPI == EPI_RESIDUAL) v = rbf(v) + e_res[EPI == EPI_RESIDUAL ? b : 0][cb][r];
                if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
                if (EPI == EPI_GELU_TANH) v = gelu_tanh_f(rbf(v));
                if (EPI == EPI_RESIDUAL && tn.sync)
                    gemv_store_bf16<true>(out + (size_t)m * ldo + n, f2bf(v));  // written through: another XCD's wave normalises the row
                else
                    out[(size_t)m * ldo + n] = f2bf(v);
            }
    }
    }
    if constexpr (EPI == EPI_RESIDUAL) {
        if (!tn.sync) return;  // kernel-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of `out` is in memory
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(tn.sync + O3V_SYNC_TAIL_TICKET, 1u, O3V_RLX_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        const uint32_t idx = old - (tn.epoch - 1u) * tn.total;  // 0 .. total-1 within this launch
        if (idx + (uint32_t)M < tn.total) return;               // not one of the last M storing waves
        const int row = (int)(idx + (uint32_t)M - tn.total);
        // the last ticket holder knows every row is complete; the few others poll the ticket line itself (at most 31 pollers, and
        // only while the last storing waves arrive: one hop less than a separate "done" word)
        if (idx != tn.total - 1u)
            spin_until<1>(tn.sync + O3V_SYNC_TAIL_TICKET, 1, tn.epoch * tn.total, tn.sync + O3V_SYNC_TMO_WORD, 0x500u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)row * ldo), 0, N * 2, 0x00020000);
        rmsnorm_row_wave<8>([&](int c) { return __builtin_bit_cast(uint4, load16_sc1(rs, (uint32_t)c * 16)); }, tn.w,
                            tn.h + (size_t)row * tn.ldh, N, tn.eps);
    }
}

// ------------------------------------------------------------------------------------------------
// The same skinny GEMM on fp8 (OCP e4m3fn) weight rows with one fp32 scale per output row (BASELINE config #5: N = 16
// self-consistency chains on fp8 weights), 4 <= M <= 32 rows of already normalised x.  The weights travel as fp8 -- half the
// bytes of the stream that bounds the kernel -- and are widened EXACTLY to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8: 3
// mantissa bits fit in 7), so the arithmetic is the bf16 MFMA of the kernel above on the dequantised values and the row scale
// multiplies the finished fp32 sum; activations, accumulation and epilogues are unchanged (an fp8 x fp8 MFMA would have to round
// x to fp8 as well).  W8p: fragment-major image [N/16][K/64][64 lanes][16 B] (weights.py pack_mfma_fragments_fp8): lane
// (row = lane & 15, g = lane >> 4) of (row block, 64-wide double step t) holds W[row][64 t + 16 g .. + 16), i.e. the A fragments
// of two MFMAs, whose B fragments are x[m][64 t + 16 g + {0..7}, {8..15}] -- the k order differs from the bf16 kernel's, the set
// does not.  One contiguous KiB per wave-instruction.
// ------------------------------------------------------------------------------------------------
template <int EPI, int KS, int CB>
__global__ __launch_bounds__(256) void gemv_mfma_fp8_kernel(const bf16_t* __restrict__ X, const uint8_t* __restrict__ W8p,
                                                            const float* __restrict__ wscale, const bf16_t* __restrict__ bias,
                                                            const bf16_t* __restrict__ res, bf16_t* __restrict__ out, int M, int N,
                                                            int K, int ldx, int ldo, int ldr, RopeArgs ra) {
    constexpr int RB = (EPI == EPI_SWIGLU || EPI == EPI_QKVROPE) ? 2 : 1;
    constexpr int RG = 4 / KS;
#ifndef O3V_F8_U1  // (overridable for A/B builds: tools/probes/rollout_probe.py)
#define O3V_F8_U1 8
#define O3V_F8_U2 2
#endif
    // double steps per trip: 8 KiB of weights in flight per buffer and wave for the single row block, 4 KiB for the paired ones
    // (7B rollouts on fp8 rows, G = 8 / 16: (4,2) 2.877 / 3.344, (8,4) 2.888 / 3.363, (2,1) 3.007 / 3.352, (8,2) 2.826 / 3.310 ms/step)
    constexpr int U = RB == 1 ? O3V_F8_U1 : O3V_F8_U2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [KS > 1: 4 x RB x CB x 64 x 4 f32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int rg = wave / KS, ks = wave % KS;
    const int grp = blockIdx.x * RG + rg;
    int rb0[RB];
    if (EPI == EPI_SWIGLU) {
        rb0[0] = grp * 32;
        rb0[RB - 1] = grp * 32 + 16;
    } else if (EPI == EPI_QKVROPE) {  // the rotary pair blocks j and j + D/2 of one head (as gemv_mfma_kernel)
        const int bph = ra.D / 32;
        const int head = grp / bph, jb = grp % bph;
        rb0[0] = head * ra.D + jb * 16;
        rb0[RB - 1] = rb0[0] + ra.D / 2;
    } else {
        rb0[0] = grp * 16;
    }
    const int nt = K >> 6;  // 64-wide double steps
    const u32x4* wrow[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int blk = (rb0[b] < N ? rb0[b] : 0) >> 4;
        wrow[b] = reinterpret_cast<const u32x4*>(W8p) + (size_t)blk * nt * 64 + lane;
    }
    const int per = (nt + KS - 1) / KS;
    const int t_begin = ks * per;
    int t_end = t_begin + per;
    t_end = t_end < nt ? t_end : nt;

    float e_scale[RB][4], e_bias[RB][4], e_res[CB][4], e_cos[CB][4], e_sin[CB][4];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int n = rb0[b] + fg * 4 + r;
            n = n < N ? n : N - 1;
            e_scale[b][r] = wscale[n];
            e_bias[b][r] = bias ? bf2f(bias[n]) : 0.f;
        }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int m = fr + 16 * cb < M ? fr + 16 * cb : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int n = rb0[0] + fg * 4 + r;
            n = n < N ? n : N - 1;
            e_res[cb][r] = (EPI == EPI_RESIDUAL) ? bf2f(res[(size_t)m * ldr + n]) : 0.f;
            if (EPI == EPI_QKVROPE) {
                const size_t cs = ((size_t)m * ra.cs_stride + ra.cs_off) * ra.D + (rb0[0] < N ? rb0[0] : 0) % ra.D + fg * 4 + r;
                e_cos[cb][r] = bf2f(ra.cosT[cs]);
                e_sin[cb][r] = bf2f(ra.sinT[cs]);
            } else {
                e_cos[cb][r] = e_sin[cb][r] = 0.f;
            }
        }
    }
    f32x4 acc[RB][CB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[b][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 wf0[U][RB], wf1[U][RB];
    bf16x8 xf0[U][CB][2], xf1[U][CB][2];
    auto load_w = [&](u32x4 (&wf)[U][RB], bf16x8 (&xf)[U][CB][2], int t0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < t_end) {  // wave-uniform
#pragma unroll
                for (int b = 0; b < RB; ++b) wf[u][b] = __builtin_nontemporal_load(wrow[b] + (size_t)t * 64);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const bool has = fr + 16 * cb < M;
                    const bf16_t* xp = X + (size_t)(fr + 16 * cb) * ldx + t * 64 + fg * 16;
                    xf[u][cb][0] = has ? *reinterpret_cast<const bf16x8*>(xp) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    xf[u][cb][1] = has ? *reinterpret_cast<const bf16x8*>(xp + 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
            } else {
#pragma unroll
                for (int b = 0; b < RB; ++b) wf[u][b] = (u32x4){0u, 0u, 0u, 0u};  // fp8 zero: widens to 0.0
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) xf[u][cb][0] = xf[u][cb][1] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };
    auto widen = [](uint32_t lo, uint32_t hi) -> bf16x8 {  // 8 fp8 -> 8 bf16, k order preserved
        const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
        const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
        const u32x4 p = {__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c),
                         __builtin_bit_cast(uint32_t, d)};
        return __builtin_bit_cast(bf16x8, p);
    };
    auto compute = [&](u32x4 (&wf)[U][RB], bf16x8 (&xf)[U][CB][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const bf16x8 a0 = widen(wf[u][b][0], wf[u][b][1]), a1 = widen(wf[u][b][2], wf[u][b][3]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, xf[u][cb][0], acc[b][cb], 0, 0, 0);
                    acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, xf[u][cb][1], acc[b][cb], 0, 0, 0);
                }
            }
    };
    load_w(wf0, xf0, t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * U) {
        load_w(wf1, xf1, t0 + U);
        compute(wf0, xf0);
        load_w(wf0, xf0, t0 + 2 * U);
        compute(wf1, xf1);
    }
    if (KS > 1) {
        f32x4* part = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) part[((wave * RB + b) * CB + cb) * 64 + lane] = acc[b][cb];
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k2 = 0; k2 < KS; ++k2) t += part[(((rg * KS + k2) * RB + b) * CB + cb) * 64 + lane];
                acc[b][cb] = t;
            }
    }
    // ---- epilogue: this lane holds C[n = rb0 + 4*fg + r][m = fr + 16 cb]; the row scale multiplies the fp32 sum first
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int m = fr + 16 * cb;
        if (m >= M) continue;
        if (EPI == EPI_SWIGLU) {
            const int no0 = grp * 16 + fg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int no = no0 + r;
                if (no >= (N >> 1)) continue;
                const float g = rbf(acc[0][cb][r] * e_scale[0][r] + e_bias[0][r]);
                const float u = rbf(acc[RB - 1][cb][r] * e_scale[RB - 1][r] + e_bias[RB - 1][r]);
                out[(size_t)m * ldo + no] = f2bf(rbf(silu_f(g)) * u);
            }
        } else if (EPI == EPI_QKVROPE) {  // bias, M-RoPE, q out / K,V appended to the cache (TF:557-599, :652-664), as gemv_mfma_kernel
            const int half = ra.D >> 1, head = rb0[0] / ra.D, j0 = rb0[0] % ra.D + fg * 4;
            if (rb0[0] >= N) return;  // block-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + r;
                const float v0 = rbf(acc[0][cb][r] * e_scale[0][r] + e_bias[0][r]);
                const float v1 = rbf(acc[RB - 1][cb][r] * e_scale[RB - 1][r] + e_bias[RB - 1][r]);
                if (head >= ra.Hq + ra.Hkv) {
                    bf16_t* dst = ra.vc + (((size_t)m * ra.Hkv + (head - ra.Hq - ra.Hkv)) * ra.Tmax + ra.slot) * ra.D;
                    dst[j] = f2bf(v0);
                    dst[j + half] = f2bf(v1);
                    continue;
                }
                const float c = e_cos[cb][r], sn = e_sin[cb][r];
                const float o0 = __fadd_rn(rbf(__fmul_rn(v0, c)), rbf(__fmul_rn(-v1, sn)));
                const float o1 = __fadd_rn(rbf(__fmul_rn(v1, c)), rbf(__fmul_rn(v0, sn)));
                bf16_t* dst = head < ra.Hq ? ra.qout + ((size_t)m * ra.Hq + head) * ra.D
                                           : ra.kc + (((size_t)m * ra.Hkv + (head - ra.Hq)) * ra.Tmax + ra.slot) * ra.D;
                dst[j] = f2bf(o0);
                dst[j + half] = f2bf(o1);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rb0[0] + fg * 4 + r;
                if (n >= N) continue;
                float v = acc[0][cb][r] * e_scale[0][r] + e_bias[0][r];
                if (EPI == EPI_RESIDUAL) v = rbf(v) + e_res[cb][r];
                out[(size_t)m * ldo + n] = f2bf(v);
            }
        }
    }
}

struct GemvArgs {
    const bf16_t *X, *W, *bias, *res, *norm_w;
    bf16_t* out;
    float eps;
    int N, K, ldx, ldw, ldo, ldr, epi;
    hipStream_t s;
    RopeArgs ra;
    bool packed;  // W points at the MFMA-fragment-major image (M >= 2 path only)
    const float* wscale = nullptr;  // fp8 weights (M <= 3): one dequantisation scale per output row; W then points at bytes
    TailNorm tn{};                  // matrix-core path, EPI_RESIDUAL: normalise the result for the next linear
};

template <int M, int R, int KS, bool NORM, int NW = 4, int WB = 2, int UU = 0>
int launch_gemv(const GemvArgs& a) {
    const int per_wave = (a.epi == EPI_SWIGLU) ? R / 2 : (a.epi == EPI_QKVROPE ? 1 : R);
    const int outs = (a.epi == EPI_SWIGLU || a.epi == EPI_QKVROPE) ? a.N / 2 : a.N;
    const int per_block = per_wave * (NW / KS);
    dim3 grid((outs + per_block - 1) / per_block), block(NW * 64);
    const size_t shmem = (NORM ? (size_t)M * a.K * 2 : 0) + (size_t)NW * R * M * 4 + (NORM ? 4 * M * 4 : 0);  // x, split-K partials, 4 virtual-wave sums
    if (shmem > 160 * 1024) return O3V_ERR_SHAPE;
#define O3V_GV(E)                                                                                                             \
    O3V_KLAUNCH((gemv_bf16_kernel<M, R, KS, E, NORM, NW, WB, UU>), grid, block, shmem, a.s, a.X, a.W, a.bias, a.res, a.out, a.norm_w, \
                a.eps, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, a.wscale)
    switch (a.epi) {
        case EPI_QKVROPE:
            if (R != 2 || !NORM) return O3V_ERR_ARG;  // the rotary pair (j, j+D/2) is the wave's two rows
            if constexpr (R == 2 && NORM) O3V_GV(EPI_QKVROPE);
            break;
        case EPI_NONE: O3V_GV(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_GV(EPI_RESIDUAL); break;
        case EPI_GELU: O3V_GV(EPI_GELU); break;
        case EPI_SWIGLU: O3V_GV(EPI_SWIGLU); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GV
    return O3V_OK;
}

#ifdef O3V_TUNE
static int g_tune_R = 0, g_tune_KS = 0;  // tuning build only (tools/tune_gemv.py): force one decomposition
extern "C" void o3v_gemv_tune(int R, int KS) {
    g_tune_R = R;
    g_tune_KS = KS;
}
#endif

// R = 2 rows per wave, KS K-slices: pick the waves per workgroup so that the grid is a whole number of workgroups per CU
// (MI355X: 256 CUs).  Measured at 7B dims (profiles/r02_gemv_balance.txt): down_proj 896 x 4 waves (3.5 per CU) -> 256 x 14,
// o_proj 448 x 4 (1.75 per CU) -> 256 x 7, q/k/v 576 x 4 (2.25 per CU) -> 768 x 3.
template <int M, int KS, bool NORM, int WB = 2>
int launch_gemv_balanced(const GemvArgs& a) {
    if constexpr (M == 1) {
        constexpr int CUS = 256;
        const int groups = (a.epi == EPI_QKVROPE) ? a.N / 2 : (a.N + 1) / 2;  // waves per K slice
        const int waves = groups * KS;
        if (waves % CUS == 0) {
            const int per_cu = waves / CUS;
            if constexpr (KS == 1) {
                switch (per_cu) {
                    case 3: case 6: case 9: case 12: return launch_gemv<M, 2, 1, NORM, 3, WB>(a);
                    case 5: case 10: case 15: return launch_gemv<M, 2, 1, NORM, 5, WB>(a);
                    case 7: case 14: return launch_gemv<M, 2, 1, NORM, 7, WB>(a);
                    default: break;
                }
            } else {
                switch (per_cu) {
                    case 6: case 12: return launch_gemv<M, 2, 2, NORM, 6, WB>(a);
                    case 10: return launch_gemv<M, 2, 2, NORM, 10, WB>(a);
                    case 14: return launch_gemv<M, 2, 2, NORM, 14, WB>(a);
                    default: break;
                }
            }
        }
    }
    return launch_gemv<M, 2, KS, NORM, 4, WB>(a);
}

template <int M, bool NORM, int WB = 2>
int launch_gemv_m(const GemvArgs& a) {
#ifdef O3V_TUNE
    if (g_tune_R) {
#define O3V_T(RR, KK) if (g_tune_R == RR && g_tune_KS == KK) return launch_gemv<M, RR, KK, NORM, 4, WB>(a)
        O3V_T(2, 1); O3V_T(2, 2); O3V_T(2, 4); O3V_T(4, 1); O3V_T(4, 2); O3V_T(4, 4); O3V_T(8, 1); O3V_T(8, 2);
#undef O3V_T
        // R = 2, KS = 1 with other workgroup sizes (g_tune_R = 100 + NW) and trip depths (g_tune_KS = k-steps per trip)
#define O3V_T2(NN, UV) if (g_tune_R == 100 + NN && g_tune_KS == UV) return launch_gemv<M, 2, 1, NORM, NN, WB, UV>(a)
        O3V_T2(4, 2); O3V_T2(4, 4); O3V_T2(4, 8); O3V_T2(8, 4); O3V_T2(8, 2); O3V_T2(16, 4); O3V_T2(2, 4); O3V_T2(8, 8);
        O3V_T2(1, 4); O3V_T2(1, 2); O3V_T2(2, 2); O3V_T2(3, 4); O3V_T2(2, 6); O3V_T2(1, 6);
#define O3V_T3(NN, UV) if (g_tune_R == 300 + NN && g_tune_KS == UV) return launch_gemv<M, 2, 2, NORM, NN, WB, UV>(a)
        O3V_T3(14, 4); O3V_T3(14, 2); O3V_T3(14, 6); O3V_T3(8, 4); O3V_T3(6, 4); O3V_T3(10, 4); O3V_T3(16, 4); O3V_T3(12, 4); O3V_T3(4, 4);
#undef O3V_T3
#define O3V_T4(NN, UV) if (g_tune_R == 200 + NN && g_tune_KS == UV) return launch_gemv<M, 4, 1, NORM, NN, WB, UV>(a)
        O3V_T4(2, 2); O3V_T4(3, 2); O3V_T4(2, 4); O3V_T4(4, 4); O3V_T4(8, 2); O3V_T4(2, 1);
#undef O3V_T4
#undef O3V_T2
        return O3V_ERR_ARG;
    }
#endif
    // Decomposition: enough waves to keep >= 32 KiB of weight loads in flight per CU, whole 512-k steps per wave.
    const int outs = (a.epi == EPI_SWIGLU) ? a.N / 2 : a.N;
    const int steps = (a.K / (WB == 1 ? 16 : 8) + 63) / 64;
    // decompositions chosen by interleaved A/B on MI355X (tools/tune_gemv.py, profiles/r01_gemv_tune.txt)
    if (a.epi == EPI_QKVROPE) return launch_gemv_balanced<M, 1, NORM, WB>(a);
    if (a.epi == EPI_SWIGLU) {
        // (fp8 rows, half as long: 4 pairs per wave measured slower, 1.958 vs 1.864 ms per decode step at 7B dims)
        // one row of bf16 weights: one (gate, up) pair per wave -- re-measured in round 2 (profiles/r02_gemv_tune_bf16.txt:
        // 42.4 vs 43.5 us at 7B dims, decode step 2.72 -> 2.685 ms); two pairs per wave stay for fp8 rows and for two rows of x
        // (and two waves per workgroup: 41.7 us; 4 waves 42.5, 3 waves 42.1, 1 wave 45.4 -- profiles/r02_gemv_tune_wg.txt)
        if (M == 1 && WB == 2 && outs >= 8192) return launch_gemv<M, 2, 1, NORM, 2, WB>(a);
        if (M <= 2 && outs >= 8192) return launch_gemv<M, 4, 1, NORM, 4, WB>(a);  // 2 (gate,up) pairs per wave
        return launch_gemv<M, 2, 1, NORM, 4, WB>(a);
    }
    if (M == 1 && WB == 2 && outs >= 32768) return launch_gemv<M, 2, 1, NORM, 3, WB>(a);  // lm_head, one bf16 row: 152.1 us (R = 4: 156)
    if (M == 1 && WB == 1 && outs >= 32768) return launch_gemv<M, 4, 1, NORM, 3, WB>(a);  // lm_head, one row of fp8 weights: 80.2 vs 82.0 us
    if (M <= 2 && outs >= 32768) return launch_gemv<M, 4, 1, NORM, 4, WB>(a);     // lm_head
    // long K, few rows (3B down_proj, 2048 x 11008: 1024 wave groups): four K slices per row pair -> 16 waves per CU; 10.3 vs 11.5 us
    // (profiles/r03_tune_3b.txt)
    if (M == 1 && WB == 2 && steps >= 16 && outs <= 2048) return launch_gemv<M, 2, 4, NORM, 4, WB>(a);
    if (steps >= 16) return launch_gemv_balanced<M, 2, NORM, WB>(a);           // long K (down_proj): split K over wave pairs
    return launch_gemv_balanced<M, 1, NORM, WB>(a);                            // o_proj / qkv
}

}  // namespace

namespace {
// Sum of the split-K partials in split order (deterministic) + the bias / residual / GELU epilogue of gemm_bf16_kernel.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ out, int ldo, int ldr) {
    const size_t total = (size_t)M * N, stride = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += part[(size_t)s * stride + i];
        if (bias) v += bf2f(bias[n]);
        if (EPI == EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldr + n]);
        if (EPI == EPI_GELU) v = gelu_erf_f(rbf(v));
            if (EPI == EPI_GELU_TANH) v = gelu_tanh_f(rbf(v));
        out[(size_t)m * ldo + n] = f2bf(v);
    }
}
}  // namespace

// The MFMA GEMM for row counts too small to fill the chip with output tiles (a prompt suffix behind a cached prefix:
// 9..128 rows, N/128 = 28..36 tiles on 256 CUs): K is split across `splits` blocks per tile, fp32 partials go through
// `workspace` (splits * M * N floats) and a second kernel reduces them in a fixed order and applies the epilogue.
extern "C" int o3v_gemm_bf16_splitk(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N,
                                    int K, int lda, int ldw, int ldo, int ldr, int epilogue, int splits, float* workspace,
                                    size_t ws_bytes, hipStream_t stream) {
    if (!A || !W || !out || !workspace || M <= 0 || N <= 0 || K <= 0 || splits < 1) return O3V_ERR_ARG;
    if ((K % BK) || (lda & 7) || (ldw & 7)) return O3V_ERR_SHAPE;
    if (epilogue != EPI_NONE && epilogue != EPI_RESIDUAL && epilogue != EPI_GELU) return O3V_ERR_ARG;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (splits > K / BK) splits = K / BK;
    if (ws_bytes < (size_t)splits * M * N * sizeof(float)) return O3V_ERR_WORKSPACE;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    O3V_KLAUNCH((gemm_bf16_kernel<EPI_PARTIAL>), dim3(tiles_m * tiles_n, splits), dim3(256), 4 * 64 * 68 * 4, stream,
                (const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)workspace, M, N, K, lda,
                ldw, N, 0, tiles_m, tiles_n);
    const int rb = (int)(((size_t)M * N + 255) / 256);
    const dim3 rg(rb < 2048 ? rb : 2048);
#define O3V_RK(E)                                                                                                        \
    O3V_KLAUNCH((splitk_reduce_kernel<E>), rg, dim3(256), 0, stream, (const float*)workspace, splits, M, N, (const bf16_t*)bias, \
                (const bf16_t*)res, (bf16_t*)out, ldo, ldr)
    if (epilogue == EPI_NONE)
        O3V_RK(EPI_NONE);
    else if (epilogue == EPI_RESIDUAL)
        O3V_RK(EPI_RESIDUAL);
    else
        O3V_RK(EPI_GELU);
#undef O3V_RK
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}


// fraction of the chip's block slots a grid of `tiles` blocks keeps busy over its ceil(tiles/slots) rounds
static inline float fill_eff(int tiles, int slots) { return (float)tiles / (float)(((tiles + slots - 1) / slots) * slots); }

// tile: 0 = choose per shape, 128 / 256 / 257 = force the 128-tile kernel, the 256-tile kernel with one __syncthreads() per K-step,
// the phased 256-tile kernel of o3v_gemm8p.hip (tests, A/B measurements; all give bit-identical results)
extern "C" int o3v_gemm_bf16_tile(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                                  int lda, int ldw, int ldo, int ldr, int epilogue, int tile, hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0 || (tile != 0 && tile != 128 && tile != 256 && tile != 257)) return O3V_ERR_ARG;
    if ((K % BK) || (lda & 7) || (ldw & 7)) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && (N % 32)) return O3V_ERR_SHAPE;
    if (M == 0) return O3V_OK;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int t2m = (M + BM2 - 1) / BM2, t2n = (N + BM2 - 1) / BM2;
    // 256-tiles: one block per CU (256 slots), more padding at ragged edges; 128-tiles: two blocks per CU (512 slots)
    // full-grid speed ratio of the 256-tile kernels over the 128-tile one: 1.3 for the kernel with one __syncthreads() per K-step
    // (profiles/r01_gemm_tile_ab.txt), 1.55 for the phased schedule of o3v_gemm8p.hip, which needs an even number of K-tiles
    // (profiles/r03_gemm_phased.txt)
    const bool can_phase = (K % (2 * BK)) == 0 && K >= 4 * BK && tile != 256;
    const float gain256 = can_phase ? 1.55f : 1.3f;
    const float use128 = fill_eff(tiles_m * tiles_n, 512) * ((float)M * N / ((float)tiles_m * BM * (float)tiles_n * BN));
    const float use256 = gain256 * fill_eff(t2m * t2n, 256) * ((float)M * N / ((float)t2m * BM2 * (float)t2n * BM2));
    const bool big = tile ? tile >= 256 : (M >= 1024 && N >= 1024 && use256 > use128);
    if (big && can_phase) return o3v_gemm_bf16_phased(A, W, bias, res, out, M, N, K, lda, ldw, ldo, ldr, epilogue, stream);
    if (tile == 257) return O3V_ERR_SHAPE;  // the phased kernel was asked for and does not take this K
    if (big) {
        dim3 grid(t2m * t2n), block(512);
        const size_t shmem = 8 * 64 * 68 * 4;  // max(2 stages x (A + B) = 128 KiB, epilogue staging 8 waves x 64 x 68 f32)
#define O3V_GM2(E)                                                                                                     \
    O3V_KLAUNCH((gemm256_bf16_kernel<E>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)W,               \
                (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, t2m, t2n)
        switch (epilogue) {
            case EPI_NONE: O3V_GM2(EPI_NONE); break;
            case EPI_RESIDUAL: O3V_GM2(EPI_RESIDUAL); break;
            case EPI_GELU: O3V_GM2(EPI_GELU); break;
            case EPI_GELU_TANH: O3V_GM2(EPI_GELU_TANH); break;
            case EPI_SWIGLU: O3V_GM2(EPI_SWIGLU); break;
            default: return O3V_ERR_ARG;
        }
#undef O3V_GM2
        O3V_CHECK_LAUNCH();
        return O3V_OK;
    }
    dim3 grid(tiles_m * tiles_n), block(256);
    const size_t shmem = 4 * 64 * 68 * 4;  // max(2 stages x (A+B) = 64 KiB, epilogue staging 4 waves x 64 x 68 f32)
#define O3V_GM(E)                                                                                                       \
    O3V_KLAUNCH((gemm_bf16_kernel<E>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)W,            \
                       (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, tiles_m, tiles_n)
    switch (epilogue) {
        case EPI_NONE: O3V_GM(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_GM(EPI_RESIDUAL); break;
        case EPI_GELU: O3V_GM(EPI_GELU); break;
        case EPI_GELU_TANH: O3V_GM(EPI_GELU_TANH); break;
        case EPI_SWIGLU: O3V_GM(EPI_SWIGLU); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GM
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                             int lda, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    return o3v_gemm_bf16_tile(A, W, bias, res, out, M, N, K, lda, ldw, ldo, ldr, epilogue, 0, stream);
}

template <int EPI, bool NORM, int KS, bool PACKED>
static int launch_gemv_mfma_p(const GemvArgs& a, int M);

template <int EPI, bool NORM, int KS>
static int launch_gemv_mfma_t(const GemvArgs& a, int M) {
    if (a.packed) return launch_gemv_mfma_p<EPI, NORM, KS, true>(a, M);
    return launch_gemv_mfma_p<EPI, NORM, KS, false>(a, M);
}

#ifdef O3V_TUNE
static int g_mt_ks = 0, g_mt_ut = 0;
extern "C" void o3v_gemv_mfma_tune(int ks, int ut) {
    g_mt_ks = ks;
    g_mt_ut = ut;
}
#endif

template <int EPI, bool NORM, int KS, bool PACKED>
static int launch_gemv_mfma_p(const GemvArgs& a, int M) {
    constexpr int RB = (EPI == EPI_SWIGLU || EPI == EPI_QKVROPE) ? 2 : 1;
    const int groups = (EPI == EPI_SWIGLU || EPI == EPI_QKVROPE) ? a.N / 32 : (a.N + 15) / 16;
    constexpr int RG = 4 / KS;
    dim3 grid((groups + RG - 1) / RG), block(256);
    const size_t shmem = (NORM ? (size_t)M * (a.K * 2 + 16) : 0) + (KS > 1 ? (size_t)4 * RB * 64 * 16 : 0) + (NORM ? 16 * 4 * 4 : 0);
    const size_t shmem2 = KS > 1 ? (size_t)4 * RB * 2 * 64 * 16 : 0;  // two column blocks of split-K partials
    TailNorm tn = a.tn;
    tn.total = grid.x * RG;  // storing waves: one per row group (with a K split the slice-0 wave of each workgroup)
    if (tn.sync && (EPI != EPI_RESIDUAL || tn.total < (uint32_t)M)) return O3V_ERR_SHAPE;
    // 17..32 rows, single-block epilogues: two adjacent weight blocks per wave (RX = 2) share the x fragments
    const int groups2 = (a.N + 31) / 32;
    const dim3 grid2((groups2 + RG - 1) / RG);
    const size_t shmem2x = KS > 1 ? (size_t)4 * 2 * 2 * 64 * 16 : 0;
    TailNorm tn2 = a.tn;
    tn2.total = grid2.x * RG;
    (void)shmem2x;
#ifdef O3V_TUNE
#define O3V_TUNE_UT(U)                                                                                                         \
    if (g_mt_ut == U) {                                                                                                        \
        if (M > 16) {                                                                                                          \
            if constexpr (!NORM) {                                                                                             \
                O3V_KLAUNCH((gemv_mfma_kernel<EPI, false, KS, PACKED, U, 2>), grid, block, shmem2, a.s, a.X, a.W, a.bias,      \
                            a.res, a.out, a.norm_w, a.eps, M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn);                 \
                return O3V_OK;                                                                                                 \
            }                                                                                                                  \
            return O3V_ERR_SHAPE;                                                                                              \
        }                                                                                                                      \
        O3V_KLAUNCH((gemv_mfma_kernel<EPI, NORM, KS, PACKED, U>), grid, block, shmem, a.s, a.X, a.W, a.bias, a.res, a.out,     \
                    a.norm_w, a.eps, M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn);                                       \
        return O3V_OK;                                                                                                         \
    }
    O3V_TUNE_UT(16)
    O3V_TUNE_UT(4)
    O3V_TUNE_UT(8)
#undef O3V_TUNE_UT
    // 100 + U: two weight blocks per wave (RX = 2) at 17..32 rows, U KiB in flight per wave
#define O3V_TUNE_RX(U)                                                                                                          \
    if (g_mt_ut == 100 + U) {                                                                                                   \
        if constexpr (!NORM && RB == 1) {                                                                                       \
            if (M > 16) {                                                                                                       \
                O3V_KLAUNCH((gemv_mfma_kernel<EPI, false, KS, PACKED, U, 2, 2>), grid2, block, shmem2x, a.s, a.X, a.W, a.bias,  \
                            a.res, a.out, a.norm_w, a.eps, M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn2);                 \
                return O3V_OK;                                                                                                  \
            }                                                                                                                   \
        }                                                                                                                       \
        return O3V_ERR_SHAPE;                                                                                                   \
    }
    O3V_TUNE_RX(16)
    O3V_TUNE_RX(8)
    O3V_TUNE_RX(4)
#undef O3V_TUNE_RX
#endif
    // 16 KiB of weight loads in flight per wave for the single-block epilogues, 8 KiB for the paired (gate/up, q/k/v) ones:
    // A/B on the 7B shapes in profiles/r01_m8_linear.txt
    constexpr int UT = RB == 1 ? 16 : 8;
    if (M > 16) {
        // 17..32 rows: two column blocks per weight fragment (x fragments from L2: no fused norm), half the weight bytes in flight
        // per wave so that the doubled x fragments and accumulators fit the register file
        if constexpr (!NORM && RB == 1 && KS == 1) {
            // many row groups and no K split (lm_head): two weight blocks per wave share the x fragments, 4 KiB in flight per wave --
            // 336.7 -> 237.9 us at 32 rows x 152064 x 3584; with a K split (down_proj, o_proj) the same form is slower, 59 vs 51 us
            // (profiles/r03_tune_m32.txt)
            O3V_KLAUNCH((gemv_mfma_kernel<EPI, false, KS, PACKED, 4, 2, 2>), grid2, block, shmem2x, a.s, a.X, a.W, a.bias, a.res, a.out,
                        a.norm_w, a.eps, M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn2);
            return O3V_OK;
        }
        if constexpr (!NORM) {
            O3V_KLAUNCH((gemv_mfma_kernel<EPI, false, KS, PACKED, 8, 2>), grid, block, shmem2, a.s, a.X, a.W, a.bias, a.res, a.out,
                        a.norm_w, a.eps, M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn);
            return O3V_OK;
        }
        return O3V_ERR_SHAPE;
    }
    O3V_KLAUNCH((gemv_mfma_kernel<EPI, NORM, KS, PACKED, UT>), grid, block, shmem, a.s, a.X, a.W, a.bias, a.res, a.out, a.norm_w, a.eps,
                M, a.N, a.K, a.ldx, a.ldw, a.ldo, a.ldr, a.ra, tn);
    return O3V_OK;
}

static int launch_gemv_mfma(const GemvArgs& a, int M) {
    const int groups = (a.epi == EPI_SWIGLU || a.epi == EPI_QKVROPE) ? a.N / 32 : (a.N + 15) / 16;
    // wave tasks = row groups x K slices: aim at >= 2048 waves (8 per CU) so enough weight loads are in flight
    // (A/B in profiles/r01_m8_linear.txt: the LDS-free form -- x already normalised, fragments from L2 -- keeps 4-way
    // K splits profitable up to 2048 groups; with x staged in LDS two blocks per CU are the limit and 2-way is better)
    int ks = groups >= 2048 ? 1 : ((groups >= 1024 && a.norm_w) ? 2 : 4);
#ifdef O3V_TUNE
    if (g_mt_ks) ks = g_mt_ks;
#endif
#define O3V_MK(E, NRM)                                                       \
    do {                                                                     \
        if (ks == 1) return launch_gemv_mfma_t<E, NRM, 1>(a, M);             \
        if (ks == 2) return launch_gemv_mfma_t<E, NRM, 2>(a, M);             \
        return launch_gemv_mfma_t<E, NRM, 4>(a, M);                          \
    } while (0)
#define O3V_MM(E)                     \
    do {                              \
        if (a.norm_w) O3V_MK(E, true); \
        O3V_MK(E, false);             \
    } while (0)
    switch (a.epi) {
        case EPI_NONE: O3V_MM(EPI_NONE);
        case EPI_RESIDUAL: O3V_MM(EPI_RESIDUAL);
        case EPI_GELU: O3V_MM(EPI_GELU);
        case EPI_SWIGLU: O3V_MM(EPI_SWIGLU);
        case EPI_QKVROPE: O3V_MM(EPI_QKVROPE);
        default: return O3V_ERR_ARG;
    }
#undef O3V_MM
#undef O3V_MK
}

static int gemv_dispatch(const void* X, const void* W, const void* bias, const void* res, void* out, const void* norm_w,
                         float eps, int M, int N, int K, int ldx, int ldw, int ldo, int ldr, int epilogue,
                         hipStream_t stream, const RopeArgs* ra = nullptr, const void* Wp = nullptr,
                         const float* wscale = nullptr, const TailNorm* tn = nullptr) {
    if (!X || !W || (!out && !ra) || M < 0 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K & 7) || (ldx & 7) || (ldw & 7) || M > 32 || (M > 16 && norm_w)) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && (N % 32)) return O3V_ERR_SHAPE;
    if (M == 0) return O3V_OK;
    GemvArgs a{(const bf16_t*)X, (const bf16_t*)W, (const bf16_t*)bias, (const bf16_t*)res, (const bf16_t*)norm_w,
               (bf16_t*)out, eps, N, K, ldx, ldw, ldo, ldr, epilogue, stream, ra ? *ra : RopeArgs{}, false};
    if (wscale) {
        // fp8 (e4m3fn) weight rows + per-row scales: the scalar weight-streaming path, up to 3 rows of x
        if (M > 3 || (K & 15) || (ldw & 15)) return O3V_ERR_SHAPE;
        a.wscale = wscale;
        int rc8;
        switch (M) {
            case 1: rc8 = norm_w ? launch_gemv_m<1, true, 1>(a) : launch_gemv_m<1, false, 1>(a); break;
            case 2: rc8 = norm_w ? launch_gemv_m<2, true, 1>(a) : launch_gemv_m<2, false, 1>(a); break;
            default: rc8 = norm_w ? launch_gemv_m<3, true, 1>(a) : launch_gemv_m<3, false, 1>(a); break;
        }
        if (rc8 != O3V_OK) return rc8;
        O3V_CHECK_LAUNCH();
        return O3V_OK;
    }
    // with the fragment-major weight image every wave-instruction of the matrix-core path reads one contiguous KiB;
    // on row-major weights its 64-byte row segments cost ~25 % of the bandwidth
    // (M = 2, 3: the scalar dot2 GEMV still streams at ~5.8 TB/s, measured in profiles/r01_m8_linear.txt)
    if (Wp && M >= 4) {
        a.W = (const bf16_t*)Wp;
        a.packed = true;
    }
    if (M >= 4 && (K % 32) == 0 && (N % 16) == 0 && (epilogue != EPI_QKVROPE || (a.ra.D % 32) == 0) &&
        (epilogue != EPI_SWIGLU || (N % 32) == 0) && (!a.packed || (N % 16 == 0)) && (!norm_w || (size_t)M * (K * 2 + 16) <= 144 * 1024)) {
        if (tn) a.tn = *tn;
        int rcm = launch_gemv_mfma(a, M);
        if (rcm != O3V_OK) return rcm;
        O3V_CHECK_LAUNCH();
        return O3V_OK;
    }
    if (M > 8 || tn) return O3V_ERR_SHAPE;  // 9..32 rows (and the TailNorm) exist only on the matrix-core path (one or two MFMA column blocks)
    a.W = (const bf16_t*)W;  // scalar path reads the row-major image
    a.packed = false;
    int rc;
#define O3V_M(MM) rc = norm_w ? launch_gemv_m<MM, true>(a) : launch_gemv_m<MM, false>(a)
    switch (M) {
        case 1: O3V_M(1); break;
#ifndef O3V_TUNE
        case 2: O3V_M(2); break;
        case 3: O3V_M(3); break;
        case 4: O3V_M(4); break;
        case 5: O3V_M(5); break;
        case 6: O3V_M(6); break;
        case 7: O3V_M(7); break;
        default: O3V_M(8); break;
#else
        default: return O3V_ERR_SHAPE;
#endif
    }
#undef O3V_M
    if (rc != O3V_OK) return rc;
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_gemv_bf16(const void* X, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                             int ldx, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    return gemv_dispatch(X, W, bias, res, out, nullptr, 0.f, M, N, K, ldx, ldw, ldo, ldr, epilogue, stream);
}

// RMSNorm fused into the projection: out = epi(rmsnorm(X; norm_w, eps) . W^T + bias)   (TF:65-79 + nn.Linear)
extern "C" int o3v_gemv_norm_bf16(const void* X, const void* norm_w, float eps, const void* W, const void* bias,
                                  const void* res, void* out, int M, int N, int K, int ldx, int ldw, int ldo, int ldr,
                                  int epilogue, hipStream_t stream) {
    if (!norm_w) return O3V_ERR_ARG;
    return gemv_dispatch(X, W, bias, res, out, norm_w, eps, M, N, K, ldx, ldw, ldo, ldr, epilogue, stream);
}

// Decode linear with both weight images: `W` row-major [N,K] (M = 1: scalar GEMV at the HBM copy rate) and `Wp` its
// MFMA-fragment-major copy [N/16][K/32][64][8] (2 <= M <= 8: matrix-core skinny GEMM streaming 1 KiB per
// wave-instruction).  norm_w may be NULL (no fused RMSNorm); Wp may be NULL (row-major only).
extern "C" int o3v_linear_decode(const void* X, const void* norm_w, float eps, const void* W, const void* Wp,
                                 const void* bias, const void* res, void* out, int M, int N, int K, int ldx, int ldo,
                                 int ldr, int epilogue, hipStream_t stream) {
    return gemv_dispatch(X, W, bias, res, out, norm_w, eps, M, N, K, ldx, K, ldo, ldr, epilogue, stream, nullptr, Wp);
}

// out = X . W^T + res AND h = RMSNorm(out; next_norm_w) for the linear that follows, 8..32 rows of already normalised X on the
// fragment-major image Wp: one launch instead of the residual linear + o3v_rmsnorm (TailNorm above; h is bit-identical to
// o3v_rmsnorm(out)).  `sync`: the zeroed buffer of o3v_decode_sync_bytes(); `epoch` = 1, 2, ... counts the calls that share it.
// O3V_ERR_SHAPE when the form does not apply (the caller then issues the two launches).
extern "C" int o3v_linear_decode_norm_next(const void* X, const void* Wp, const void* res, void* out, int M, int N, int K, int ldx,
                                           int ldo, int ldr, const void* next_norm_w, float eps, void* h, int ldh, uint32_t* sync,
                                           uint32_t epoch, hipStream_t stream) {
    if (!X || !Wp || !res || !out || !next_norm_w || !h || !sync || epoch == 0) return O3V_ERR_ARG;
    if (M < 8 || M > 32 || N > 4096 || (N & 15) || (ldh & 7) || (ldo & 7)) return O3V_ERR_SHAPE;
    static const bool gfx950 = [] {  // the hand-off rests on measured gfx950 behaviour (o3v_handoff.h)
        int dev = 0;
        hipDeviceProp_t prop;
        return hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
               strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }();
    if (!gfx950) return O3V_ERR_SHAPE;
    TailNorm tn;
    tn.sync = sync;
    tn.epoch = epoch;
    tn.w = (const bf16_t*)next_norm_w;
    tn.h = (bf16_t*)h;
    tn.ldh = ldh;
    tn.eps = eps;
    return gemv_dispatch(X, Wp, nullptr, res, out, nullptr, 0.f, M, N, K, ldx, K, ldo, ldr, EPI_RESIDUAL, stream, nullptr, Wp, nullptr, &tn);
}

// Decode q/k/v projection with everything around it fused: RMSNorm prologue, bias, M-RoPE, q written to qout[M,Hq,D],
// k and v appended to the cache at `slot` (TF:733-736 norm, :636-664 projection + rope + cache update).
extern "C" int o3v_gemv_norm_qkv_rope(const void* X, const void* norm_w, float eps, const void* W, const void* Wp,
                                      const void* bias, int M, int K, int ldx, const void* cosT, const void* sinT,
                                      void* qout, void* kcache, void* vcache, int slot, int Hq, int Hkv, int D, int Tmax,
                                      int cs_stride_row, int cs_off, hipStream_t stream) {
    if (!cosT || !sinT || !qout || !kcache || !vcache || slot < 0 || slot >= Tmax || Hq <= 0 || Hkv <= 0 || (D & 1))
        return O3V_ERR_ARG;
    if (!norm_w && M < 4) return O3V_ERR_ARG;  // un-fused norm (X already normalised) exists on the matrix-core path only
    RopeArgs ra{(const bf16_t*)cosT, (const bf16_t*)sinT, (bf16_t*)qout, (bf16_t*)kcache, (bf16_t*)vcache,
                slot, Hq, Hkv, D, Tmax, cs_stride_row, cs_off};
    return gemv_dispatch(X, W, bias, nullptr, nullptr, norm_w, eps, M, (Hq + 2 * Hkv) * D, K, ldx, K, 0, 0, EPI_QKVROPE, stream,
                         &ra, Wp);
}

// ---- fp8 (OCP e4m3fn) weights with one fp32 scale per output row, M <= 3 rows: the decode linears at half the bytes.
// W8: uint8 [N, K] (K % 16 == 0), scale: f32 [N]; out = epi(scale[n] * (rmsnorm(X) . fp8(W8[n])) + bias).
extern "C" int o3v_linear_decode_fp8(const void* X, const void* norm_w, float eps, const void* W8, const float* scale,
                                     const void* bias, const void* res, void* out, int M, int N, int K, int ldx, int ldo,
                                     int ldr, int epilogue, hipStream_t stream) {
    if (!scale) return O3V_ERR_ARG;
    return gemv_dispatch(X, W8, bias, res, out, norm_w, eps, M, N, K, ldx, K, ldo, ldr, epilogue, stream, nullptr, nullptr, scale);
}

// fp8 rows at 4..32 rows of x (already normalised): the matrix-core kernel above on the fragment-major fp8 image W8p.
// epilogue NONE / RESIDUAL / SWIGLU (W8p then packs the interleaved gate/up matrix) / QKVROPE, K % 64 == 0, N % 16 == 0.
static int launch_fp8_rows(const void* X, const void* W8p, const float* scale, const void* bias, const void* res, void* out, int M, int N,
                           int K, int ldx, int ldo, int ldr, int epilogue, const RopeArgs& ra, hipStream_t stream) {
    if (!X || !W8p || !scale || M < 4 || M > 32 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K & 63) || (N & 15) || (ldx & 7) || ((epilogue == EPI_SWIGLU || epilogue == EPI_QKVROPE) && (N & 31))) return O3V_ERR_SHAPE;
    if (epilogue == EPI_QKVROPE && (ra.D % 32)) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    const int RB = (epilogue == EPI_SWIGLU || epilogue == EPI_QKVROPE) ? 2 : 1;
    const int groups = RB == 2 ? N / 32 : N / 16;
    const int ks = groups >= 2048 ? 1 : 4;  // as the bf16 kernel: >= 8 waves per CU
    const int cbn = M > 16 ? 2 : 1;
#define O3V_F8K(E, KK, CC)                                                                                                       \
    do {                                                                                                                         \
        constexpr int RG_ = 4 / KK;                                                                                              \
        const dim3 grid((groups + RG_ - 1) / RG_), block(256);                                                                   \
        const size_t sh = KK > 1 ? (size_t)4 * RB * CC * 64 * 16 : 0;                                                            \
        O3V_KLAUNCH((gemv_mfma_fp8_kernel<E, KK, CC>), grid, block, sh, stream, (const bf16_t*)X, (const uint8_t*)W8p, scale,    \
                    (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, ldx, ldo, ldr, ra);                          \
    } while (0)
#define O3V_F8E(E)                       \
    do {                                 \
        if (ks == 1) {                   \
            if (cbn == 2)                \
                O3V_F8K(E, 1, 2);        \
            else                         \
                O3V_F8K(E, 1, 1);        \
        } else {                         \
            if (cbn == 2)                \
                O3V_F8K(E, 4, 2);        \
            else                         \
                O3V_F8K(E, 4, 1);        \
        }                                \
    } while (0)
    switch (epilogue) {
        case EPI_NONE: O3V_F8E(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_F8E(EPI_RESIDUAL); break;
        case EPI_SWIGLU: O3V_F8E(EPI_SWIGLU); break;
        case EPI_QKVROPE: O3V_F8E(EPI_QKVROPE); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_F8E
#undef O3V_F8K
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_linear_decode_fp8_rows(const void* X, const void* W8p, const float* scale, const void* bias, const void* res,
                                          void* out, int M, int N, int K, int ldx, int ldo, int ldr, int epilogue, hipStream_t stream) {
    if (!out || epilogue == EPI_QKVROPE) return O3V_ERR_ARG;
    return launch_fp8_rows(X, W8p, scale, bias, res, out, M, N, K, ldx, ldo, ldr, epilogue, RopeArgs{}, stream);
}

// q/k/v on fp8 rows for 4..32 rows of already normalised x, with bias, M-RoPE and the cache append in the epilogue (the
// arguments of o3v_gemv_norm_qkv_rope without the fused norm)
extern "C" int o3v_qkv_rope_fp8_rows(const void* X, const void* W8p, const float* scale, const void* bias, int M, int K, int ldx,
                                     const void* cosT, const void* sinT, void* qout, void* kcache, void* vcache, int slot, int Hq,
                                     int Hkv, int D, int Tmax, int cs_stride_row, int cs_off, hipStream_t stream) {
    if (!cosT || !sinT || !qout || !kcache || !vcache || slot < 0 || slot >= Tmax || Hq <= 0 || Hkv <= 0 || (D & 1)) return O3V_ERR_ARG;
    const RopeArgs ra{(const bf16_t*)cosT, (const bf16_t*)sinT, (bf16_t*)qout, (bf16_t*)kcache, (bf16_t*)vcache,
                      slot, Hq, Hkv, D, Tmax, cs_stride_row, cs_off};
    return launch_fp8_rows(X, W8p, scale, bias, nullptr, nullptr, M, (Hq + 2 * Hkv) * D, K, ldx, 0, 0, EPI_QKVROPE, ra, stream);
}

extern "C" int o3v_gemv_norm_qkv_rope_fp8(const void* X, const void* norm_w, float eps, const void* W8, const float* scale,
                                          const void* bias, int M, int K, int ldx, const void* cosT, const void* sinT,
                                          void* qout, void* kcache, void* vcache, int slot, int Hq, int Hkv, int D, int Tmax,
                                          int cs_stride_row, int cs_off, hipStream_t stream) {
    if (!scale || !norm_w || !cosT || !sinT || !qout || !kcache || !vcache || slot < 0 || slot >= Tmax || Hq <= 0 || Hkv <= 0 ||
        (D & 1))
        return O3V_ERR_ARG;
    RopeArgs ra{(const bf16_t*)cosT, (const bf16_t*)sinT, (bf16_t*)qout, (bf16_t*)kcache, (bf16_t*)vcache,
                slot, Hq, Hkv, D, Tmax, cs_stride_row, cs_off};
    return gemv_dispatch(X, W8, bias, nullptr, nullptr, norm_w, eps, M, (Hq + 2 * Hkv) * D, K, ldx, K, 0, 0, EPI_QKVROPE, stream,
                         &ra, nullptr, scale);
}
