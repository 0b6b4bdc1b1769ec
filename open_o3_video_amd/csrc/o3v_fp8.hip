// fp8 x fp8 on the matrix cores (BASELINE config #5: "fp8 weights on CDNA4 fp8 MFMA"): the compute-bound linears of the LLM
// prefill and of the log-prob pass (TF:modeling_qwen2_5_vl.py:541-554, :602-689; TF3 the same) as W8A8 --
//
//   weights      the fp8 (OCP e4m3fn) rows + one power-of-two fp32 scale per output row that the decode already streams
//                (weights.py quantize_rows_fp8; row-major [N][K])
//   activations  quantised per token on the fly: one power-of-two scale per row (the smallest that brings the row into +-448),
//                fp8 e4m3fn, round to nearest even (o3v_quantize_rows_fp8 / o3v_rmsnorm_quantize_fp8)
//   product      v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 rate per clock at 4x the K per
//                instruction), fp32 accumulation; out = epilogue(acc * sa[m] * sw[n] + bias) with the bf16 kernels' epilogues
//
// Kernel = the 256 x 256 tile of o3v_gemm.hip with 128-BYTE k-tiles of fp8 instead of 64 bf16: the same LDS image (128-byte
// rows, XOR swizzle on the 16-byte chunks), the same LDS-DMA staging, half as many k-tiles for the same K.  Per k-tile a wave
// reads the same 24 fragments of 16 B... x2 (a 16x16x128 operand is 32 B per lane) and issues 32 MFMAs of twice the length:
// the instruction mix per tile is the bf16 kernel's, the tile covers twice the K.
// Opt-in (engine.fp8_prefill): the bf16 path and its goldens are untouched.  Oracle: oracle/quant_ref.py (the same quantisation
// applied in fp32 torch).
#include "o3v_common.h"
#include "o3v_gemm_tile.h"

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));

constexpr int FBM = 256, FBK = 128;            // rows per operand tile, BYTES (= fp8 elements) per tile row
constexpr int FTILE_BYTES = FBM * FBK;         // 32 KiB per operand tile

// global -> LDS of a 256 x 128 B tile: 32 wave-instructions of 1 KiB, 4 per wave (8 waves); LDS position p = instr*1024 +
// lane*16 holds logical chunk (row = p / 128, c = ((p % 128) / 16) ^ ((row >> 1) & 7)) -- swz_off of o3v_gemm_tile.h
__device__ __forceinline__ void stage_tile_fp8(const uint8_t* __restrict__ g, int ld, int row0, int rows_valid, int k0, char* lds_tile,
                                               int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int instr = wave * 4 + i;
        const int row = instr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < rows_valid ? grow : rows_valid - 1;  // tail rows re-read a valid row, never stored
        const uint8_t* src = g + (size_t)grow * ld + k0 + c * 16;
        __builtin_amdgcn_global_load_lds(src, (lds_void*)(lds_tile + instr * 1024), 16, 0, 0);
    }
}

// A / B operand of the 16x16x128 fp8 MFMA: lane (r = lane & 15, g = lane >> 4) holds k = 32 g .. 32 g + 31 of row r: the two
// 16-byte chunks 2g, 2g + 1 of the 128-byte tile row
__device__ __forceinline__ v8i_t frag_fp8(const char* tile, int row, int fg) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const v4i_t lo = *reinterpret_cast<const v4i_t*>(tile + swz_off(row, 2 * fg));
    const v4i_t hi = *reinterpret_cast<const v4i_t*>(tile + swz_off(row, 2 * fg + 1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);  // a concatenation: the two loads land in the operand's registers
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_fp8_kernel(const uint8_t* __restrict__ A, const float* __restrict__ sa,
                                                          const uint8_t* __restrict__ W, const float* __restrict__ sw,
                                                          const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ out, int M, int N, int K, int lda, int ldw, int ldo,
                                                          int ldr, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32K | B 32K]; epilogue: 8 x 64 x 68 floats
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {  // XCD-aware tile order (as the bf16 kernels): each XCD walks a contiguous run of tiles, M fastest
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * FBM, n0 = tn * FBM;

    f32x4 acc[2][4][4];  // [row half][i][j]: rows wm*128 + half*64 + i*16, cols wn*64 + j*16
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / FBK;
    const int fr = lane & 15, fg = lane >> 4;
    stage_tile_fp8(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile_fp8(W, ldw, n0, N, 0, smem + FTILE_BYTES, wave, lane);
    __syncthreads();

    constexpr int ONE = 0x7f7f7f7f;  // E8M0 block scales of 1.0 in every byte
    for (int t = 0; t < nk; ++t) {
        char* cur = smem + (t & 1) * 2 * FTILE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * 2 * FTILE_BYTES;
        if (t + 1 < nk) {
            stage_tile_fp8(A, lda, m0, M, (t + 1) * FBK, nxt, wave, lane);
            stage_tile_fp8(W, ldw, n0, N, (t + 1) * FBK, nxt + FTILE_BYTES, wave, lane);
        }
        v8i_t bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_fp8(cur + FTILE_BYTES, wn * 64 + j * 16 + fr, fg);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v8i_t af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag_fp8(cur, wm * 128 + h * 64 + i * 16 + fr, fg);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[h][i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bfr[j], acc[h][i][j], 0, 0, 0, ONE, 0, ONE);
        }
        __syncthreads();
    }
    // dequantise: acc[m][n] *= sa[m] * sw[n] (both powers of two: exact), then the bf16 kernels' epilogue.
    // C/D map of the 16x16 shapes: col = lane & 15, row = (lane >> 4) * 4 + reg.
    float* et = reinterpret_cast<float*>(smem) + wave * 64 * 68;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int mrow0 = m0 + wm * 128 + h * 64, ncol0 = n0 + wn * 64;
        float swv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol0 + j * 16 + fr;
            swv[j] = sw[n < N ? n : N - 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + i * 16 + fg * 4 + r;
                const float sm = sa[m < M ? m : M - 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j][r] *= sm * swv[j];
            }
        wave_epilogue<EPI>(acc[h], et, lane, mrow0, ncol0, M, N, bias, res, out, ldo, ldr);
    }
}

// ------------------------------------------------------------------------------------------------
// The same tile on the PHASED schedule of o3v_gemm8p.hip (read its header for the choreography and the wait / re-stage rules): phases
// of 8 MFMAs (a 64 x 32 quadrant of the wave's sub-tile x one 128-byte K-tile, as long as 16 bf16 MFMAs) between raw barriers, the
// LDS copies of half-tiles six deep in flight with counted vmcnt, the two wave rows one barrier apart, grouped tile order.  Same
// MFMA sequence per accumulator as gemm256_fp8_kernel: bit-identical.  K / 128 even and >= 4.
// ------------------------------------------------------------------------------------------------
constexpr int FHT_BYTES = 128 * FBK;  // 16 KiB half-tile: 128 rows x 128 B
constexpr int FLEAD = 6;

__device__ __forceinline__ void stage_half_fp8(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W, int lda, int ldw, int m0, int n0,
                                               int M, int N, int t, int q, char* slot, int wave, int lane) {
    const bool isA = (q == 0 || q == 3);
    const uint8_t* g = isA ? A : W;
    const int ld = isA ? lda : ldw, base = isA ? m0 : n0, lim = isA ? M : N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int instr = wave * 2 + i;
        const int r = instr * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = isA ? ((r < 64 ? r : r + 64) + (q == 3 ? 64 : 0)) : ((r >> 5) * 64 + (r & 31) + (q == 2 ? 32 : 0));
        row += base;
        row = row < lim ? row : lim - 1;
        __builtin_amdgcn_global_load_lds(g + (size_t)row * ld + (size_t)t * FBK + c * 16, (lds_void*)(slot + instr * 1024), 16, 0, 0);
    }
}

#define O3V_FREAD_A(SLOT) _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = frag_fp8((SLOT), wm * 64 + i * 16 + fr, fg)
#define O3V_FREAD_B(DST, SLOT) _Pragma("unroll") for (int j = 0; j < 2; ++j) DST[j] = frag_fp8((SLOT), wn * 32 + j * 16 + fr, fg)
// (hipcc sinks these MFMAs -- pure functions of registers -- out of their phase, all 64 to the end of the loop body, unless their
// operands and results are pinned to the phase: the empty asm statements are ordered against the barriers)
#define O3V_FMMA(H, JB, BF)                                                                                                  \
    asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(BF[0]), "+v"(BF[1]));                         \
    __builtin_amdgcn_s_setprio(1);                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[H][i][(JB) * 2 + j] =    \
        __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], BF[j], acc[H][i][(JB) * 2 + j], 0, 0, 0, ONE, 0, ONE);       \
    __builtin_amdgcn_s_setprio(0);                                                                                           \
    asm volatile("" : "+v"(acc[H][0][(JB) * 2]), "+v"(acc[H][0][(JB) * 2 + 1]), "+v"(acc[H][1][(JB) * 2]),                   \
                 "+v"(acc[H][1][(JB) * 2 + 1]), "+v"(acc[H][2][(JB) * 2]), "+v"(acc[H][2][(JB) * 2 + 1]),                    \
                 "+v"(acc[H][3][(JB) * 2]), "+v"(acc[H][3][(JB) * 2 + 1]))
#define O3V_FSTAGE_WAIT(P)                                                                                                          \
    {                                                                                                                               \
        const int g = p0 + (P) + FLEAD;                                                                                             \
        if (g < 4 * nk) {                                                                                                           \
            stage_half_fp8(A, W, lda, ldw, m0, n0, M, N, g >> 2, ((P) + FLEAD) & 3, smem + (((P) + FLEAD) & 7) * FHT_BYTES, wave, lane); \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                        \
        } else {                                                                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                        \
        }                                                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
#define O3V_FPHASE_END()                \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    __builtin_amdgcn_sched_barrier(0)

template <int EPI>
__global__ __launch_bounds__(512) void gemm256ph_fp8_kernel(const uint8_t* __restrict__ A, const float* __restrict__ sa,
                                                            const uint8_t* __restrict__ W, const float* __restrict__ sw,
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
    constexpr int GROUP_M = 16;  // grouped tile order (o3v_gemm8p.hip)
    const int per_group = GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_grp = bid - grp * per_group;
    const int tm = first_m + in_grp % gm, tn = in_grp / gm;
    const int m0 = tm * FBM, n0 = tn * FBM;
    const int nk = K / FBK;
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int ONE = 0x7f7f7f7f;  // E8M0 block scales of 1.0 in every byte

    f32x4 acc[2][4][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    v8i_t af[4], b0[2], b1[2];

#pragma unroll
    for (int g = 0; g < FLEAD; ++g) stage_half_fp8(A, W, lda, ldw, m0, n0, M, N, g >> 2, g & 3, smem + g * FHT_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);

    for (int p0 = 0; p0 < 4 * nk; p0 += 8) {
        O3V_FREAD_B(b0, smem + 1 * FHT_BYTES);
        O3V_FREAD_A(smem + 0 * FHT_BYTES);
        O3V_FSTAGE_WAIT(0)
        O3V_FMMA(0, 0, b0);
        O3V_FPHASE_END();
        O3V_FREAD_B(b1, smem + 2 * FHT_BYTES);
        O3V_FSTAGE_WAIT(1)
        O3V_FMMA(0, 1, b1);
        O3V_FPHASE_END();
        O3V_FREAD_A(smem + 3 * FHT_BYTES);
        O3V_FSTAGE_WAIT(2)
        O3V_FMMA(1, 1, b1);
        O3V_FPHASE_END();
        O3V_FSTAGE_WAIT(3)
        O3V_FMMA(1, 0, b0);
        O3V_FPHASE_END();
        O3V_FREAD_B(b0, smem + 5 * FHT_BYTES);
        O3V_FREAD_A(smem + 4 * FHT_BYTES);
        O3V_FSTAGE_WAIT(4)
        O3V_FMMA(0, 0, b0);
        O3V_FPHASE_END();
        O3V_FREAD_B(b1, smem + 6 * FHT_BYTES);
        O3V_FSTAGE_WAIT(5)
        O3V_FMMA(0, 1, b1);
        O3V_FPHASE_END();
        O3V_FREAD_A(smem + 7 * FHT_BYTES);
        O3V_FSTAGE_WAIT(6)
        O3V_FMMA(1, 1, b1);
        O3V_FPHASE_END();
        O3V_FSTAGE_WAIT(7)
        O3V_FMMA(1, 0, b0);
        O3V_FPHASE_END();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();
    float* et = reinterpret_cast<float*>(smem) + wave * 64 * 68;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int mrow0 = m0 + wm * 128 + h * 64, ncol0 = n0 + wn * 64;
        float swv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol0 + j * 16 + fr;
            swv[j] = sw[n < N ? n : N - 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + i * 16 + fg * 4 + r;
                const float sm = sa[m < M ? m : M - 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j][r] *= sm * swv[j];
            }
        wave_epilogue<EPI>(acc[h], et, lane, mrow0, ncol0, M, N, bias, res, out, ldo, ldr);
    }
}
#undef O3V_FREAD_A
#undef O3V_FREAD_B
#undef O3V_FMMA
#undef O3V_FSTAGE_WAIT
#undef O3V_FPHASE_END

// ------------------------------------------------------------------------------------------------
// Per-row quantisation bf16 -> fp8 e4m3fn with a power-of-two scale (and optionally the RMSNorm in front of it, with the
// arithmetic of rmsnorm_kernel: y = w * bf16(x * rstd) rounded to bf16 -- the values the bf16 path feeds its linears).
// One wave per row; rows up to 16 chunks per lane (8192 columns) stay in registers, longer rows are read twice.
// scale = 2^e, e = the smallest integer with amax / 2^e <= 448 (amax = 0: scale 1); q = fp8(y / scale) (exact division, RNE).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_scale_for(float amax) {
    if (!(amax > 0.f)) return 1.0f;
    int e;
    const float m = frexpf(amax, &e);  // amax = m * 2^e, m in [0.5, 1); 448 = 0.875 * 2^9: amax / 2^E <= 448 <=> m * 2^(e-E) <= 0.875 * 2^9
    return ldexpf(1.0f, m <= 0.875f ? e - 9 : e - 8);
}

__device__ __forceinline__ uint2 pack8_fp8(const float (&y)[8], float inv) {
    uint32_t lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0] * inv, y[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2] * inv, y[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4] * inv, y[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6] * inv, y[7] * inv, hi, true);
    return make_uint2(lo, hi);
}

template <int MAXCH, bool NORM>
__global__ __launch_bounds__(256) void rows_quant_fp8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             uint8_t* __restrict__ q, float* __restrict__ scale, int rows, int cols,
                                                             int ld_in, int ld_q, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * ld_in);
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    uint2* qr = reinterpret_cast<uint2*>(q + (size_t)row * ld_q);
    auto normed8 = [&](const uint4& v, const uint4& wn, float rstd, float (&y)[8]) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&v);
        const uint32_t* g = reinterpret_cast<const uint32_t*>(&wn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NORM) {
                y[2 * j] = rbf(bf_lo(g[j]) * rbf(bf_lo(p[j]) * rstd));
                y[2 * j + 1] = rbf(bf_hi(g[j]) * rbf(bf_hi(p[j]) * rstd));
            } else {
                y[2 * j] = bf_lo(p[j]);
                y[2 * j + 1] = bf_hi(p[j]);
            }
        }
    };
    float rstd = 1.0f;
    if constexpr (MAXCH > 0) {  // the row lives in registers
        uint4 v[MAXCH > 0 ? MAXCH : 1], wv[MAXCH > 0 ? MAXCH : 1];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                v[i] = xr[c];
                if (NORM) wv[i] = wr[c];
            }
        }
        if (NORM) {
#pragma unroll
            for (int i = 0; i < MAXCH; ++i) {
                const int c = lane + i * 64;
                if (c < nch) {
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = bf_lo(p[j]), b = bf_hi(p[j]);
                        ss = fmaf(a, a, ss);
                        ss = fmaf(b, b, ss);
                    }
                }
            }
            ss = wave_sum(ss);
            rstd = 1.0f / sqrtf(ss / (float)cols + eps);
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                float y[8];
                normed8(v[i], wv[i], rstd, y);
#pragma unroll
                for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(y[j]));
            }
        }
        amax = wave_max(amax);
        const float sc = pow2_scale_for(amax), inv = 1.0f / sc;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                float y[8];
                normed8(v[i], wv[i], rstd, y);
                qr[c] = pack8_fp8(y, inv);
            }
        }
        if (lane == 0) scale[row] = sc;
        return;
    }
    // long rows: three passes over a row that stays in L2
    if (NORM) {
        float ss = 0.f;
        for (int c = lane; c < nch; c += 64) {
            const uint4 v = xr[c];
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(p[j]), b = bf_hi(p[j]);
                ss = fmaf(a, a, ss);
                ss = fmaf(b, b, ss);
            }
        }
        ss = wave_sum(ss);
        rstd = 1.0f / sqrtf(ss / (float)cols + eps);
    }
    float amax = 0.f;
    for (int c = lane; c < nch; c += 64) {
        float y[8];
        normed8(xr[c], NORM ? wr[c] : xr[c], rstd, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(y[j]));
    }
    amax = wave_max(amax);
    const float sc = pow2_scale_for(amax), inv = 1.0f / sc;
    for (int c = lane; c < nch; c += 64) {
        float y[8];
        normed8(xr[c], NORM ? wr[c] : xr[c], rstd, y);
        qr[c] = pack8_fp8(y, inv);
    }
    if (lane == 0) scale[row] = sc;
}

template <bool NORM>
int launch_rows_quant(const void* x, const void* w, void* q, float* scale, int rows, int cols, int ld_in, int ld_q, float eps,
                      hipStream_t stream) {
    if (!x || !q || !scale || (NORM && !w) || rows < 0 || cols <= 0 || (cols & 7) || (ld_in & 7) || (ld_q & 7) || ld_in < cols || ld_q < cols)
        return O3V_ERR_ARG;
    if (rows == 0) return O3V_OK;
    const dim3 grid((rows + 3) / 4), block(256);
#define O3V_RQ(MC)                                                                                                              \
    O3V_KLAUNCH((rows_quant_fp8_kernel<MC, NORM>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (uint8_t*)q, scale, rows, \
                cols, ld_in, ld_q, eps)
    if (cols <= 4 * 512)
        O3V_RQ(4);
    else if (cols <= 8 * 512)
        O3V_RQ(8);
    else
        O3V_RQ(0);
#undef O3V_RQ
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

}  // namespace

extern "C" int o3v_quantize_rows_fp8(const void* x, void* q, float* scale, int rows, int cols, int ld_in, int ld_q, hipStream_t stream) {
    return launch_rows_quant<false>(x, nullptr, q, scale, rows, cols, ld_in, ld_q, 0.f, stream);
}

extern "C" int o3v_rmsnorm_quantize_fp8(const void* x, const void* w, void* q, float* scale, int rows, int cols, int ld_in, int ld_q,
                                        float eps, hipStream_t stream) {
    return launch_rows_quant<true>(x, w, q, scale, rows, cols, ld_in, ld_q, eps, stream);
}

// schedule: 0 = choose (the phased kernel wherever K has an even number >= 4 of 128-byte K-tiles), 1 = the kernel with one
// __syncthreads() per K-tile, 2 = the phased kernel (O3V_ERR_SHAPE where it does not apply); both give the same bits
extern "C" int o3v_gemm_fp8_sched(const void* A8, const float* sa, const void* W8, const float* sw, const void* bias, const void* res,
                                  void* out, int M, int N, int K, int lda, int ldw, int ldo, int ldr, int epilogue, int schedule,
                                  hipStream_t stream);
extern "C" int o3v_gemm_fp8(const void* A8, const float* sa, const void* W8, const float* sw, const void* bias, const void* res, void* out,
                            int M, int N, int K, int lda, int ldw, int ldo, int ldr, int epilogue, hipStream_t stream) {
    return o3v_gemm_fp8_sched(A8, sa, W8, sw, bias, res, out, M, N, K, lda, ldw, ldo, ldr, epilogue, 0, stream);
}

extern "C" int o3v_gemm_fp8_sched(const void* A8, const float* sa, const void* W8, const float* sw, const void* bias, const void* res,
                                  void* out, int M, int N, int K, int lda, int ldw, int ldo, int ldr, int epilogue, int schedule,
                                  hipStream_t stream) {
    if (schedule < 0 || schedule > 2) return O3V_ERR_ARG;
    if (!A8 || !sa || !W8 || !sw || !out || M <= 0 || N <= 0 || K <= 0) return O3V_ERR_ARG;
    if ((K % FBK) || (lda & 15) || (ldw & 15) || lda < K || ldw < K) return O3V_ERR_SHAPE;  // whole 128-byte k-tiles, 16-byte chunks
    if (epilogue == EPI_RESIDUAL && !res) return O3V_ERR_ARG;
    if (epilogue == EPI_SWIGLU && ((N & 31) != 0)) return O3V_ERR_SHAPE;
    const int tiles_m = (M + FBM - 1) / FBM, tiles_n = (N + FBM - 1) / FBM;
    const dim3 grid(tiles_m * tiles_n), block(512);
    const size_t shmem = 8 * 64 * 68 * 4;  // max(2 stages x (A + B) = 128 KiB, epilogue staging 8 waves x 64 x 68 f32), as gemm256_bf16
    const bool can_phase = (K % (2 * FBK)) == 0 && K >= 4 * FBK;
    if (schedule == 2 && !can_phase) return O3V_ERR_SHAPE;
    const bool phased = schedule != 1 && can_phase;
#define O3V_GF(E)                                                                                                            \
    if (phased)                                                                                                              \
        O3V_KLAUNCH((gemm256ph_fp8_kernel<E>), grid, block, shmem, stream, (const uint8_t*)A8, sa, (const uint8_t*)W8, sw,    \
                    (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, tiles_m, tiles_n);    \
    else                                                                                                                     \
        O3V_KLAUNCH((gemm256_fp8_kernel<E>), grid, block, shmem, stream, (const uint8_t*)A8, sa, (const uint8_t*)W8, sw,      \
                    (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)out, M, N, K, lda, ldw, ldo, ldr, tiles_m, tiles_n);
    switch (epilogue) {
        case EPI_NONE: O3V_GF(EPI_NONE) break;
        case EPI_RESIDUAL: O3V_GF(EPI_RESIDUAL) break;
        case EPI_SWIGLU: O3V_GF(EPI_SWIGLU) break;
        default: return O3V_ERR_ARG;
    }
#undef O3V_GF
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
