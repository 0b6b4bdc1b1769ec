// 256 x 256 x 64 bf16 GEMM, phased schedule: out = epi(A . W^T + bias) on the matrix cores for the big prefill / log-prob linears
// (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:581-586 generate -> TF:692-757 decoder layer linears; :371-384 log-prob pass).
// The LDS-staged kernels with one __syncthreads() per K-step are in o3v_gemm.hip; this file holds the schedule that keeps the
// staging in flight across barriers.
#include <hip/hip_runtime.h>
#include "../../include/o3v.h"
#include "o3v_common.h"
#include "o3v_gemm_tile.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Same tile, same waves (2 x 4, each a 128 x 64 sub-tile), same MFMA order per accumulator as gemm256_bf16_kernel -- bit-identical
// results -- but the K loop is cut into PHASES of 16 MFMAs (one 64 x 32 quadrant of the wave's sub-tile x the 64-deep K-tile):
//
//   phase:  ds_read the operand pieces the quadrant needs and that are not in registers yet   (12 / 4 / 8 / 0 reads of 16 B)
//           issue the global->LDS copy of ONE half-tile (16 KiB: 2 wave-instructions per wave), 6 half-tiles ahead of its use
//           s_waitcnt vmcnt(6)      all but the 3 youngest half-tiles have landed (loads return in order)
//           s_barrier
//           16 MFMAs (priority raised)
//           s_barrier
//
// The two wave rows run ONE BARRIER APART (wave row 1 passes an extra barrier first, wave row 0 one at the end): a SIMD hosts one wave
// of each row, so while one multiplies the other reads LDS and issues copies -- the matrix pipe and the LDS pipe are both busy all the
// time instead of in turns, and no wave ever waits for vmcnt(0) inside the loop.
//
// A K-tile is staged as four half-tiles, each read from LDS in exactly ONE phase of its K-tile:
//   AH0 = rows  0..63  of both wave rows (block rows 0..63, 128..191)        read in phase 1   (A fragments stay in registers: 1, 2)
//   BH0 = columns 0..31 of the four wave columns (block columns 64 w + 0..31) read in phase 1   (B fragments stay in registers: 1..4)
//   BH1 = columns 32..63 of the four wave columns                             read in phase 2   (registers: 2, 3)
//   AH1 = rows 64..127 of both wave rows                                      read in phase 3   (registers: 3, 4; reuses AH0's)
// quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0).  LDS = a ring of 8 half-tile slots (128 KiB); half-tile g = 4 t + q lives in slot
// g % 8, is issued in phase g - 6 and read in phase 4 t + max(q - 1, 0).  Write-after-read: slot g % 8 held half-tile g - 8, last
// read >= 2 phases (4 barriers) before the copy of g is issued, also across the one-barrier skew of the wave rows.  Read-after-write:
// the wait that retires a half-tile sits before the FIRST barrier of a phase at least two phases before its read (one more than the
// in-step rule asks, for the skew).  The last 6 phases issue nothing and wait for vmcnt(0).
// Needs K / 64 even and >= 4 (the launcher checks; other shapes take the kernels of o3v_gemm.hip).
// ------------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int HT_BYTES = 128 * BK * 2;  // 16 KiB half-tile: 128 rows x 64 k
constexpr int LEAD = 6;                 // half-tiles issued ahead of the phase that reads them

// global->LDS copy of this wave's share (2 KiB) of half-tile q of K-tile t.  LDS row r of an A half-tile is block row
// (r < 64 ? r : r + 64) + 64 * (q == 3); LDS row r of a B half-tile is block column (r / 32) * 64 + r % 32 + 32 * (q == 2).
__device__ __forceinline__ void stage_half(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, int lda, int ldw, int m0, int n0,
                                           int M, int N, int t, int q, char* slot, int wave, int lane) {
    const bool isA = (q == 0 || q == 3);
    const bf16_t* g = isA ? A : W;
    const int ld = isA ? lda : ldw, base = isA ? m0 : n0, lim = isA ? M : N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int instr = wave * 2 + i;  // 16 wave-instructions of 1 KiB per half-tile
        const int r = instr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = isA ? ((r < 64 ? r : r + 64) + (q == 3 ? 64 : 0)) : ((r >> 5) * 64 + (r & 31) + (q == 2 ? 32 : 0));
        row += base;
        row = row < lim ? row : lim - 1;  // clamp: tail rows re-read a valid row, never stored
        __builtin_amdgcn_global_load_lds(g + (size_t)row * ld + (size_t)t * BK + c * 8, (lds_void*)(slot + instr * 1024), 16, 0, 0);
    }
}

#define O3V_READ_A(SLOT)                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) af[ks][i] =     \
        *reinterpret_cast<const bf16x8*>((SLOT) + swz_off(wm * 64 + i * 16 + fr, ks * 4 + fg))
#define O3V_READ_B(DST, SLOT)                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j) DST[ks][j] =    \
        *reinterpret_cast<const bf16x8*>((SLOT) + swz_off(wn * 32 + j * 16 + fr, ks * 4 + fg))
// 16 MFMAs: quadrant (row half H, column half JB) over the K-tile, k ascending
// (the empty asm statements pin the MFMAs -- pure functions of registers -- to their phase: they are ordered against the barriers,
// and hipcc was seen to sink all MFMAs of the fp8 twin of this loop, o3v_fp8.hip, to the end of the loop body without them)
#define O3V_MMA(H, JB, BF)                                                                                                          \
    asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), \
                 "+v"(af[1][3]), "+v"(BF[0][0]), "+v"(BF[0][1]), "+v"(BF[1][0]), "+v"(BF[1][1]));                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        acc[H][i][(JB) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], BF[ks][j], acc[H][i][(JB) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                                  \
    asm volatile("" : "+v"(acc[H][0][(JB) * 2]), "+v"(acc[H][0][(JB) * 2 + 1]), "+v"(acc[H][1][(JB) * 2]),                          \
                 "+v"(acc[H][1][(JB) * 2 + 1]), "+v"(acc[H][2][(JB) * 2]), "+v"(acc[H][2][(JB) * 2 + 1]),                           \
                 "+v"(acc[H][3][(JB) * 2]), "+v"(acc[H][3][(JB) * 2 + 1]))
// issue half-tile (phase + LEAD), wait for all but the 3 youngest, first barrier of the phase
#define O3V_STAGE_WAIT(P)                                                                                                         \
    {                                                                                                                             \
        const int g = p0 + (P) + LEAD;                                                                                            \
        if (g < 4 * nk) {                                                                                                         \
            stage_half(A, W, lda, ldw, m0, n0, M, N, g >> 2, ((P) + LEAD) & 3, smem + (((P) + LEAD) & 7) * HT_BYTES, wave, lane); \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                      \
        } else {                                                                                                                  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                      \
        }                                                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    }
#define O3V_PHASE_END()                 \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    __builtin_amdgcn_sched_barrier(0)

template <int EPI>
__global__ __launch_bounds__(512) void gemm256ph_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                             const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                             bf16_t* __restrict__ out, int M, int N, int K, int lda, int ldw, int ldo,
                                                             int ldr, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 8 half-tile slots of 16 KiB; epilogue: 8 x 64 x 68 floats
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // tile order: groups of GROUP_M row tiles, inside a group row-fastest -- the 256 tiles in flight cover ~16 x 16 tiles, i.e. 16 A
    // panels + 16 W panels are live, not 256 A panels + 1 W panel as with plain row-fastest order on a tall matrix (32 videos prefill
    // together: 562 row tiles), whose A re-reads (N / 256 times the whole activation) then come from HBM instead of L2 / MALL
    constexpr int GROUP_M = 16;
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_grp = bid - grp * per_group;
    const int tm = first_m + in_grp % gm, tn = in_grp / gm;
    const int m0 = tm * BM2, n0 = tn * BM2;
    const int nk = K / BK;
    const int fr = lane & 15, fg = lane >> 4;

    f32x4 acc[2][4][4];  // [row half][i][j]: rows wm*128 + half*64 + i*16, cols wn*64 + j*16
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4], b0[2][2], b1[2][2];

    // prologue: half-tiles 0..5; the first two (AH0, BH0 of K-tile 0) must have landed before phase 0 reads them
#pragma unroll
    for (int g = 0; g < LEAD; ++g) stage_half(A, W, lda, ldw, m0, n0, M, N, g >> 2, g & 3, smem + g * HT_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);

    for (int p0 = 0; p0 < 4 * nk; p0 += 8) {  // 8 phases = 2 K-tiles: slots 0..3 then 4..7
        // ---- K-tile 2i: AH0 slot 0, BH0 slot 1, BH1 slot 2, AH1 slot 3
        O3V_READ_B(b0, smem + 1 * HT_BYTES);
        O3V_READ_A(smem + 0 * HT_BYTES);
        O3V_STAGE_WAIT(0)
        O3V_MMA(0, 0, b0);
        O3V_PHASE_END();
        O3V_READ_B(b1, smem + 2 * HT_BYTES);
        O3V_STAGE_WAIT(1)
        O3V_MMA(0, 1, b1);
        O3V_PHASE_END();
        O3V_READ_A(smem + 3 * HT_BYTES);
        O3V_STAGE_WAIT(2)
        O3V_MMA(1, 1, b1);
        O3V_PHASE_END();
        O3V_STAGE_WAIT(3)
        O3V_MMA(1, 0, b0);
        O3V_PHASE_END();
        // ---- K-tile 2i + 1: slots 4..7
        O3V_READ_B(b0, smem + 5 * HT_BYTES);
        O3V_READ_A(smem + 4 * HT_BYTES);
        O3V_STAGE_WAIT(4)
        O3V_MMA(0, 0, b0);
        O3V_PHASE_END();
        O3V_READ_B(b1, smem + 6 * HT_BYTES);
        O3V_STAGE_WAIT(5)
        O3V_MMA(0, 1, b1);
        O3V_PHASE_END();
        O3V_READ_A(smem + 7 * HT_BYTES);
        O3V_STAGE_WAIT(6)
        O3V_MMA(1, 1, b1);
        O3V_PHASE_END();
        O3V_STAGE_WAIT(7)
        O3V_MMA(1, 0, b0);
        O3V_PHASE_END();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // the barrier wave row 1 took first
    __syncthreads();                            // the tiles are dead: the epilogue stages through the same LDS
    float* et = reinterpret_cast<float*>(smem) + wave * 64 * 68;
#pragma unroll
    for (int h = 0; h < 2; ++h)
        wave_epilogue<EPI>(acc[h], et, lane, m0 + wm * 128 + h * 64, n0 + wn * 64, M, N, bias, res, out, ldo, ldr);
}
#undef O3V_READ_A
#undef O3V_READ_B
#undef O3V_MMA
#undef O3V_STAGE_WAIT
#undef O3V_PHASE_END

}  // namespace

// out = epi(A . W^T + bias) on the phased 256-tile kernel; W row-major [N, K] as o3v_gemm_bf16.  K % 128 == 0 and K >= 256
// (an even number of K-tiles); bit-identical to o3v_gemm_bf16 / o3v_gemm_bf16_tile.  O3V_ERR_SHAPE otherwise.
extern "C" int o3v_gemm_bf16_phased(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                                    int lda, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K % (2 * BK)) || K < 4 * BK || (lda & 7) || (ldw & 7)) return O3V_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && (N % 32)) return O3V_ERR_SHAPE;
    if (M == 0) return O3V_OK;
    const int t2m = (M + BM2 - 1) / BM2, t2n = (N + BM2 - 1) / BM2;
    dim3 grid(t2m * t2n), block(512);
    const size_t shmem = 8 * 64 * 68 * 4;  // max(8 half-tile slots = 128 KiB, epilogue staging 8 waves x 64 x 68 f32)
#define O3V_GP(E)                                                                                                   \
    O3V_KLAUNCH((gemm256ph_bf16_kernel<E>), grid, block, shmem, stream, (const bf16_t*)A, (const bf16_t*)W,          \
                (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, t2m, t2n)
    switch (epilogue) {
        case EPI_NONE: O3V_GP(EPI_NONE); break;
        case EPI_RESIDUAL: O3V_GP(EPI_RESIDUAL); break;
        case EPI_GELU: O3V_GP(EPI_GELU); break;
        case EPI_GELU_TANH: O3V_GP(EPI_GELU_TANH); break;
        case EPI_SWIGLU: O3V_GP(EPI_SWIGLU); break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GP
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
