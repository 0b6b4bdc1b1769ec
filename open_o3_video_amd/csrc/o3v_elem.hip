// HBM-bound elementwise / normalisation / layout kernels of the Qwen2.5-VL generate path (gfx950).
// All of them move 16 B per lane per access (8 bf16) and keep every intermediate in fp32 with the
// reference's bf16 rounding points.  Reference arithmetic: transformers 5.15.0
// models/qwen2_5_vl/modeling_qwen2_5_vl.py ("TF:" below) as called by the Open-o3-Video trainer
// (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:581-586) -- see include/o3v.h for the boundary.
#include "o3v_common.h"

// ------------------------------------------------------------------------------------------------
// RMSNorm (TF:65-79): out = w * bf16(x * rsqrt(mean(x^2)+eps)); one wave per row, row held in VGPRs.
// ------------------------------------------------------------------------------------------------
template <int MAXCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int rows, int cols, int ld_in,
                                                      int ld_out, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * ld_in);
    rmsnorm_row_wave<MAXCH>([&](int c) { return xr[c]; }, w, out + (size_t)row * ld_out, cols, eps);
}

extern "C" int o3v_rmsnorm(const void* x, const void* w, void* out, int rows, int cols, int ld_in, int ld_out,
                           float eps, hipStream_t stream) {
    if (!x || !w || !out || rows < 0 || cols <= 0 || (cols & 7) || (ld_in & 7) || (ld_out & 7)) return O3V_ERR_ARG;
    if (rows == 0) return O3V_OK;
    if (cols > 16 * 512) return O3V_ERR_SHAPE;
    dim3 grid((rows + 3) / 4), block(256);
    if (cols <= 4 * 512)
        O3V_KLAUNCH(rmsnorm_kernel<4>, grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out,
                           rows, cols, ld_in, ld_out, eps);
    else if (cols <= 8 * 512)
        O3V_KLAUNCH(rmsnorm_kernel<8>, grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out,
                           rows, cols, ld_in, ld_out, eps);
    else
        O3V_KLAUNCH(rmsnorm_kernel<16>, grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out,
                           rows, cols, ld_in, ld_out, eps);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm with bias (Qwen3-VL vision blocks and mergers, TF3:122-135, :268-284): fp32 statistics over the row held in
// VGPRs (mean, then the centred sum of squares), y = (x - mean) * rstd * w + b, one rounding to bf16.
// ------------------------------------------------------------------------------------------------
template <int MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int rows,
                                                        int cols, int ld_in, int ld_out, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = cols >> 3;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * ld_in);
    uint4 v[MAXCH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            v[i] = xr[c];
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += bf_lo(p[j]) + bf_hi(p[j]);
        }
    }
    const float mean = wave_sum(sum) / (float)cols;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(p[j]) - mean, d = bf_hi(p[j]) - mean;
                ss = fmaf(a, a, ss);
                ss = fmaf(d, d, ss);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)cols + eps);
    const uint4* wr = reinterpret_cast<const uint4*>(w);
    const uint4* br = reinterpret_cast<const uint4*>(b);
    uint4* orow = reinterpret_cast<uint4*>(out + (size_t)row * ld_out);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const uint4 wv = wr[c], bv = br[c];
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&v[i]);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(&wv);
            const uint32_t* z = reinterpret_cast<const uint32_t*>(&bv);
            uint4 o;
            uint32_t* po = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                po[j] = pack_bf2((bf_lo(p[j]) - mean) * rstd * bf_lo(q[j]) + bf_lo(z[j]),
                                 (bf_hi(p[j]) - mean) * rstd * bf_hi(q[j]) + bf_hi(z[j]));
            orow[c] = o;
        }
    }
}

extern "C" int o3v_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int cols, int ld_in, int ld_out,
                             float eps, hipStream_t stream) {
    if (!x || !w || !b || !out || rows < 0 || cols <= 0 || (cols & 7) || (ld_in & 7) || (ld_out & 7)) return O3V_ERR_ARG;
    if (rows == 0) return O3V_OK;
    if (cols > 16 * 512) return O3V_ERR_SHAPE;
    dim3 grid((rows + 3) / 4), block(256);
#define O3V_LN(CH)                                                                                                        \
    O3V_KLAUNCH(layernorm_kernel<CH>, grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, \
                rows, cols, ld_in, ld_out, eps)
    if (cols <= 4 * 512)
        O3V_LN(4);
    else if (cols <= 8 * 512)
        O3V_LN(8);
    else
        O3V_LN(16);
#undef O3V_LN
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// ViT 2-D RoPE in place on the q and k thirds of qkv [P, 3*H*D] (TF:160-171): fp32,
// q*cos + rotate_half(q)*sin with separately rounded products, one cast back to bf16.
// cos/sin: fp32 [P, D/2] (the table is cat(rot, rot), so column d and d+D/2 share an entry).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cosT,
                                                       const float* __restrict__ sinT, int P, int H, int D) {
    const int half = D >> 1, cpr = half >> 3;  // 8-wide chunks per half head
    const long total = (long)P * 2 * H * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % cpr);
        long r = i / cpr;
        int h = (int)(r % H);
        r /= H;
        int which = (int)(r & 1);  // 0 = q, 1 = k
        int p = (int)(r >> 1);
        bf16_t* base = qkv + (size_t)p * 3 * H * D + (size_t)which * H * D + (size_t)h * D + c * 8;
        uint4 lo = *reinterpret_cast<const uint4*>(base);
        uint4 hi = *reinterpret_cast<const uint4*>(base + half);
        const float* cr = cosT + (size_t)p * half + c * 8;
        const float* sr = sinT + (size_t)p * half + c * 8;
        const uint32_t* pl = reinterpret_cast<const uint32_t*>(&lo);
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(&hi);
        uint4 ol, oh;
        uint32_t* pol = reinterpret_cast<uint32_t*>(&ol);
        uint32_t* poh = reinterpret_cast<uint32_t*>(&oh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = bf_lo(pl[j]), a1 = bf_hi(pl[j]);  // x1 (first half)
            float b0 = bf_lo(ph[j]), b1 = bf_hi(ph[j]);  // x2 (second half)
            float c0 = cr[2 * j], c1 = cr[2 * j + 1], s0 = sr[2 * j], s1 = sr[2 * j + 1];
            // first half: x1*cos + (-x2)*sin ; second half: x2*cos + x1*sin
            float l0 = __fadd_rn(__fmul_rn(a0, c0), __fmul_rn(-b0, s0));
            float l1 = __fadd_rn(__fmul_rn(a1, c1), __fmul_rn(-b1, s1));
            float h0 = __fadd_rn(__fmul_rn(b0, c0), __fmul_rn(a0, s0));
            float h1 = __fadd_rn(__fmul_rn(b1, c1), __fmul_rn(a1, s1));
            pol[j] = pack_bf2(l0, l1);
            poh[j] = pack_bf2(h0, h1);
        }
        *reinterpret_cast<uint4*>(base) = ol;
        *reinterpret_cast<uint4*>(base + half) = oh;
    }
}

extern "C" int o3v_vit_rope(void* qkv, const float* cosT, const float* sinT, int P, int H, int D, hipStream_t stream) {
    if (!qkv || !cosT || !sinT || P < 0 || H <= 0 || D <= 0 || (D & 15)) return O3V_ERR_ARG;
    if (P == 0) return O3V_OK;
    long total = (long)P * 2 * H * (D >> 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(vit_rope_kernel, dim3(blocks), dim3(256), 0, stream, (bf16_t*)qkv, cosT, sinT, P, H, D);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// M-RoPE cos/sin table (TF:525-538 + section select TF:590-596): pos int32 [3, T], inv_freq fp32
// [D/2] (computed on the host exactly as torch does), axis_of[D/2] in {0,1,2} -> cos,sin bf16 [T, D].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mrope_table_kernel(const int* __restrict__ pos, const float* __restrict__ inv_freq,
                                                          const int* __restrict__ axis_of, bf16_t* __restrict__ cosT,
                                                          bf16_t* __restrict__ sinT, int T, int D) {
    const int half = D >> 1;
    long total = (long)T * half;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int d = (int)(i % half);
        int t = (int)(i / half);
        float f = __fmul_rn(inv_freq[d], (float)pos[(size_t)axis_of[d] * T + t]);
        bf16_t c = f2bf(cosf(f)), s = f2bf(sinf(f));
        cosT[(size_t)t * D + d] = c;
        cosT[(size_t)t * D + d + half] = c;
        sinT[(size_t)t * D + d] = s;
        sinT[(size_t)t * D + d + half] = s;
    }
}

extern "C" int o3v_mrope_table(const int* pos, const float* inv_freq, const int* axis_of, void* cosT, void* sinT, int T,
                               int D, hipStream_t stream) {
    if (!pos || !inv_freq || !axis_of || !cosT || !sinT || T < 0 || D <= 0 || (D & 1)) return O3V_ERR_ARG;
    if (T == 0) return O3V_OK;
    long total = (long)T * (D >> 1);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    O3V_KLAUNCH(mrope_table_kernel, dim3(blocks), dim3(256), 0, stream, pos, inv_freq, axis_of, (bf16_t*)cosT,
                       (bf16_t*)sinT, T, D);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// LLM: split qkv [T, (Hq+2Hkv)*D], apply M-RoPE in bf16 (TF:598-599: bf16(bf16(q*cos)+bf16(rh*sin))),
// write q [T,Hq,D] and append k,v to the cache [B,Hkv,Tmax,D] at slot_base + (t % tokens_per_row).
// token t belongs to batch row b = t / tokens_per_row; its cos/sin row is
// b*cs_stride_row + cs_off + (t % tokens_per_row)  (prefill: the [B*S,D] table; decode: step `cs_off`
// of the per-sequence [B,Tnew,D] table).
// ------------------------------------------------------------------------------------------------
// QKNORM (Qwen3-VL, TF3:480-481): q and k heads go through RMSNorm over head_dim (weights q_norm / k_norm) before the
// rotation; the D/16 lanes that hold one head are neighbours, so the sum of squares is a few lane exchanges.
template <bool QKNORM>
__global__ __launch_bounds__(256) void qkv_rope_cache_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ cosT,
                                                             const bf16_t* __restrict__ sinT, bf16_t* __restrict__ qout,
                                                             bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                             int slot_base, int T, int tokens_per_row,
                                                             int Hq, int Hkv, int D, int Tmax, int cs_stride_row,
                                                             int cs_off, const bf16_t* __restrict__ q_norm,
                                                             const bf16_t* __restrict__ k_norm, float eps) {
    const int half = D >> 1, cpr = half >> 3;
    const int HT = Hq + 2 * Hkv;
    const long total = (long)T * HT * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % cpr);
        long r = i / cpr;
        int h = (int)(r % HT);
        int t = (int)(r / HT);
        int b = t / tokens_per_row;
        const int tin = t - b * tokens_per_row;
        const int slot = slot_base + tin;
        const bf16_t* src = qkv + (size_t)t * HT * D + (size_t)h * D + c * 8;
        u32x4 lo = *reinterpret_cast<const u32x4*>(src);
        u32x4 hi = *reinterpret_cast<const u32x4*>(src + half);
        if (QKNORM) {
            // every lane of the head's group takes part (v heads too: their sum is simply not used).  Lane shares are summed
            // pairwise over lane distance 1, 2, 4: ((S0+S1)+(S2+S3)) + ((S4+S5)+(S6+S7)) at D = 128, in every lane
            float ss = qkn_chain(lo, hi);
            for (int m = 1; m < cpr; m <<= 1) ss += __shfl_xor(ss, m, 64);
            if (h < Hq + Hkv) {
                const float rstd = qkn_rstd(ss, D, eps);
                const bf16_t* nw = (h < Hq ? q_norm : k_norm) + c * 8;
                lo = qkn_scale(lo, *reinterpret_cast<const u32x4*>(nw), rstd);
                hi = qkn_scale(hi, *reinterpret_cast<const u32x4*>(nw + half), rstd);
            }
        }
        bf16_t* dst;
        if (h < Hq) {
            dst = qout + ((size_t)t * Hq + h) * D + c * 8;
        } else if (h < Hq + Hkv) {
            dst = kc + (((size_t)b * Hkv + (h - Hq)) * Tmax + slot) * D + c * 8;
        } else {
            dst = vc + (((size_t)b * Hkv + (h - Hq - Hkv)) * Tmax + slot) * D + c * 8;
            *reinterpret_cast<u32x4*>(dst) = lo;
            *reinterpret_cast<u32x4*>(dst + half) = hi;
            continue;
        }
        // cos/sin row: row-major [rows, D]; both halves hold the same value so read the first half only
        size_t csr = (size_t)(b * cs_stride_row + cs_off + tin) * D + c * 8;
        u32x4 ol, oh;
        rope_share(lo, hi, *reinterpret_cast<const u32x4*>(cosT + csr), *reinterpret_cast<const u32x4*>(sinT + csr), ol, oh);
        *reinterpret_cast<u32x4*>(dst) = ol;
        *reinterpret_cast<u32x4*>(dst + half) = oh;
    }
}

extern "C" int o3v_qkv_rope_cache(const void* qkv, const void* cosT, const void* sinT, void* qout, void* kcache,
                                  void* vcache, int slot_base, int T, int tokens_per_row, int Hq, int Hkv, int D,
                                  int Tmax, int cs_stride_row, int cs_off, hipStream_t stream) {
    if (!qkv || !cosT || !sinT || !qout || !kcache || !vcache || slot_base < 0 || T < 0 || tokens_per_row <= 0 || (D & 15) ||
        slot_base + tokens_per_row > Tmax)
        return O3V_ERR_ARG;
    if (T == 0) return O3V_OK;
    long total = (long)T * (Hq + 2 * Hkv) * (D >> 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(qkv_rope_cache_kernel<false>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, (const bf16_t*)cosT,
                       (const bf16_t*)sinT, (bf16_t*)qout, (bf16_t*)kcache, (bf16_t*)vcache, slot_base, T, tokens_per_row, Hq,
                       Hkv, D, Tmax, cs_stride_row, cs_off, (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0.f);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_qkv_norm_rope_cache(const void* qkv, const void* q_norm, const void* k_norm, float eps, const void* cosT,
                                       const void* sinT, void* qout, void* kcache, void* vcache, int slot_base, int T,
                                       int tokens_per_row, int Hq, int Hkv, int D, int Tmax, int cs_stride_row, int cs_off,
                                       hipStream_t stream) {
    if (!qkv || !q_norm || !k_norm || !cosT || !sinT || !qout || !kcache || !vcache || T < 0 || tokens_per_row <= 0 || Hq <= 0 ||
        Hkv <= 0 || D <= 0 || (D & 15))
        return O3V_ERR_ARG;
    const int cpr = D >> 4;
    if (cpr & (cpr - 1) || cpr > 64) return O3V_ERR_SHAPE;  // the head's lanes exchange by xor: a power of two of them
    if (T == 0) return O3V_OK;
    long total = (long)T * (Hq + 2 * Hkv) * cpr;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(qkv_rope_cache_kernel<true>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, (const bf16_t*)cosT,
                       (const bf16_t*)sinT, (bf16_t*)qout, (bf16_t*)kcache, (bf16_t*)vcache, slot_base, T, tokens_per_row, Hq,
                       Hkv, D, Tmax, cs_stride_row, cs_off, (const bf16_t*)q_norm, (const bf16_t*)k_norm, eps);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// DeepStack (Qwen3-VL, TF3:839-862): x[rows[i], :] = bf16(x[rows[i], :] + feat[src[i], :]) after the first decoder layers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ x, const int* __restrict__ rows,
                                                       const int* __restrict__ src, const bf16_t* __restrict__ feat, int n,
                                                       int chunks) {
    const long total = (long)n * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks), r = (int)(i / chunks);
        uint4* xp = reinterpret_cast<uint4*>(x + ((size_t)rows[r] * chunks + c) * 8);
        const uint4 a = *xp, b = *reinterpret_cast<const uint4*>(feat + ((size_t)src[r] * chunks + c) * 8);
        const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
        uint4 o;
        uint32_t* po = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j) po[j] = pack_bf2(bf_lo(pa[j]) + bf_lo(pb[j]), bf_hi(pa[j]) + bf_hi(pb[j]));
        *xp = o;
    }
}

extern "C" int o3v_add_rows(void* x, const int* rows, const int* src, const void* feat, int n, int hidden, hipStream_t stream) {
    if (!x || !rows || !src || !feat || n < 0 || hidden <= 0 || (hidden & 7)) return O3V_ERR_ARG;
    if (n == 0) return O3V_OK;
    long total = (long)n * (hidden >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(add_rows_kernel, dim3(blocks), dim3(256), 0, stream, (bf16_t*)x, rows, src, (const bf16_t*)feat, n, hidden >> 3);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Row gather: dst[i,:] = src[idx[i],:], rows of row_bytes (multiple of 16).  Used for the ViT window
// permutation at 4-token granularity and its inverse (TF:436-439, :464-466).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ idx,
                                                          uint4* __restrict__ dst, int rows, int chunks) {
    long total = (long)rows * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % chunks);
        int r = (int)(i / chunks);
        dst[(size_t)r * chunks + c] = src[(size_t)idx[r] * chunks + c];
    }
}

extern "C" int o3v_gather_rows(const void* src, const int* idx, void* dst, int rows, int row_bytes, hipStream_t stream) {
    if (!src || !idx || !dst || rows < 0 || row_bytes <= 0 || (row_bytes & 15)) return O3V_ERR_ARG;
    if (rows == 0) return O3V_OK;
    long total = (long)rows * (row_bytes >> 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(gather_rows_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, idx, (uint4*)dst, rows,
                       row_bytes >> 4);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Token embedding gather + masked scatter of the merged visual tokens (TF:1206-1215):
// src_row[t] >= 0 -> embed_table[src_row[t]] ; src_row[t] < 0 -> vis[-src_row[t]-1].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_scatter_kernel(const uint4* __restrict__ table, const uint4* __restrict__ vis,
                                                            const int* __restrict__ src_row, uint4* __restrict__ out,
                                                            int T, int chunks) {
    long total = (long)T * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % chunks);
        int t = (int)(i / chunks);
        int r = src_row[t];
        out[(size_t)t * chunks + c] = (r >= 0) ? table[(size_t)r * chunks + c] : vis[(size_t)(-r - 1) * chunks + c];
    }
}

extern "C" int o3v_embed_scatter(const void* table, const void* vis, const int* src_row, void* out, int T, int hidden,
                                 hipStream_t stream) {
    if (!table || !src_row || !out || T < 0 || hidden <= 0 || (hidden & 7)) return O3V_ERR_ARG;
    if (T == 0) return O3V_OK;
    long total = (long)T * (hidden >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(embed_scatter_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)table, (const uint4*)vis,
                       src_row, (uint4*)out, T, hidden >> 3);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// pixel_values f32 [P, K0] (already normalised by the HF processor) -> bf16 [P, Kp] zero padded
// (TF:1090 `pixel_values.type(self.visual.dtype)`; the pad columns meet zero weight columns).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int P,
                                                       int K0, int Kp) {
    const int cpr = Kp >> 3;
    long total = (long)P * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % cpr);
        int p = (int)(i / cpr);
        const float* s = src + (size_t)p * K0 + c * 8;
        uint4 o;
        uint32_t* po = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int k = c * 8 + 2 * j;
            float a = (k < K0) ? s[2 * j] : 0.f;
            float b = (k + 1 < K0) ? s[2 * j + 1] : 0.f;
            po[j] = pack_bf2(a, b);
        }
        *reinterpret_cast<uint4*>(dst + (size_t)p * Kp + c * 8) = o;
    }
}

extern "C" int o3v_cast_pad_f32_bf16(const float* src, void* dst, int P, int K0, int Kp, hipStream_t stream) {
    if (!src || !dst || P < 0 || K0 <= 0 || Kp < K0 || (Kp & 7)) return O3V_ERR_ARG;
    if (P == 0) return O3V_OK;
    long total = (long)P * (Kp >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    O3V_KLAUNCH(cast_pad_kernel, dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, P, K0, Kp);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused frame pipeline: frames [T,3,H,W] (u8 or f32, values 0..255) -> rescale 1/255 (in double, as
// TF:image_transforms.py:118 does), CLIP normalise (fp32), patchify to merge-block-major rows with the
// temporal slice duplicated (TF:image_processing_pil_qwen2_vl.py:152-187), cast bf16, zero pad to Kp.
// ------------------------------------------------------------------------------------------------
// PAIRS (native video input, TF:models/qwen2_vl/video_processing_qwen2_vl.py:236-274): temporal patch f holds frames 2f and
// 2f+1 in its two temporal slices (an odd frame count repeats the last frame); T then counts temporal patches and NF frames.
template <typename TIN, bool PAIRS>
__global__ __launch_bounds__(256) void patchify_kernel(const TIN* __restrict__ frames, bf16_t* __restrict__ dst, int T,
                                                       int H, int W, int Kp, int PS, float m0, float m1, float m2, float s0,
                                                       float s1, float s2, int NF) {
    const int gh = H / PS, gw = W / PS, ppf = gh * gw, gwm = gw >> 1;
    const int cpr = Kp >> 3;
    const long total = (long)T * ppf * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % cpr);
        long pr = i / cpr;
        int pi = (int)(pr % ppf);
        int f = (int)(pr / ppf);
        // merge-block-major patch index -> (row, col) of the patch grid
        int iw = pi & 1, ih = (pi >> 1) & 1, blk = pi >> 2;
        int bw = blk % gwm, bh = blk / gwm;
        int prow = bh * 2 + ih, pcol = bw * 2 + iw;
        uint4 o;
        uint32_t* po = reinterpret_cast<uint32_t*>(&o);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = c * 8 + j;
            float val = 0.f;
            if (k < 3 * 2 * PS * PS) {
                int ch = k / (2 * PS * PS);
                int rem = k % (PS * PS);  // images: temporal slice index dropped, both slices are the same frame
                int py = rem / PS, px = rem % PS;
                int fr = f;
                if (PAIRS) {
                    fr = 2 * f + (k / (PS * PS)) % 2;
                    if (fr > NF - 1) fr = NF - 1;
                }
                double raw = (double)frames[(((size_t)fr * 3 + ch) * H + (prow * PS + py)) * W + (pcol * PS + px)];
                float r = (float)(raw * 0.00392156862745098);
                float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
                float sd = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
                val = __fdiv_rn(__fsub_rn(r, mean), sd);
            }
            v[j] = val;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) po[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
        *reinterpret_cast<uint4*>(dst + ((size_t)f * ppf + pi) * Kp + c * 8) = o;
    }
}

extern "C" int o3v_patchify_ps(const void* frames, int is_u8, void* dst, int T, int H, int W, int Kp, int patch,
                               const float* mean3, const float* std3, hipStream_t stream) {
    if (!frames || !dst || !mean3 || !std3 || T < 0 || H <= 0 || W <= 0 || patch <= 0 || (H % (2 * patch)) || (W % (2 * patch)) ||
        Kp < 6 * patch * patch || (Kp & 7))
        return O3V_ERR_ARG;
    if (T == 0) return O3V_OK;
    long total = (long)T * (H / patch) * (W / patch) * (Kp >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (is_u8)
        O3V_KLAUNCH((patchify_kernel<uint8_t, false>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)frames, (bf16_t*)dst,
                           T, H, W, Kp, patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], T);
    else
        O3V_KLAUNCH((patchify_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, (const float*)frames, (bf16_t*)dst, T,
                           H, W, Kp, patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], T);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_patchify_video(const void* frames, int is_u8, void* dst, int n_frames, int H, int W, int Kp, int patch,
                                  const float* mean3, const float* std3, hipStream_t stream) {
    if (!frames || !dst || !mean3 || !std3 || n_frames < 0 || H <= 0 || W <= 0 || patch <= 0 || (H % (2 * patch)) ||
        (W % (2 * patch)) || Kp < 6 * patch * patch || (Kp & 7))
        return O3V_ERR_ARG;
    if (n_frames == 0) return O3V_OK;
    const int T = (n_frames + 1) / 2;
    long total = (long)T * (H / patch) * (W / patch) * (Kp >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (is_u8)
        O3V_KLAUNCH((patchify_kernel<uint8_t, true>), dim3(blocks), dim3(256), 0, stream, (const uint8_t*)frames, (bf16_t*)dst,
                           T, H, W, Kp, patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], n_frames);
    else
        O3V_KLAUNCH((patchify_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, (const float*)frames, (bf16_t*)dst, T,
                           H, W, Kp, patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], n_frames);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_patchify(const void* frames, int is_u8, void* dst, int T, int H, int W, int Kp, const float* mean3,
                            const float* std3, hipStream_t stream) {
    return o3v_patchify_ps(frames, is_u8, dst, T, H, W, Kp, 14, mean3, std3, stream);
}

// ------------------------------------------------------------------------------------------------
// Frame resize of fetch_video (R:src/r1-v/src/open_r1/vision_process.py:310-315: torchvision resize, BICUBIC,
// antialias=True on the [T,3,H,W] frame tensor == ATen _upsample_bicubic2d_aa): separable, width pass then height
// pass, each output sample a dot product of <= kmax taps whose normalised weights (cubic a = -0.5, support widened by
// the scale when shrinking) are tabulated once per (in, out) size on the host in float32, exactly as ATen computes them.
// The taps are accumulated in tap order with separate multiply and add, fp32 throughout; uint8 sources are rounded half
// to even and clamped to 0..255 at the end (torchvision's uint8 path), the result stays float32 for the patch kernel.
// ------------------------------------------------------------------------------------------------
template <typename TIN>
__global__ __launch_bounds__(256) void resize_aa_w_kernel(const TIN* __restrict__ src, float* __restrict__ dst, long rows,
                                                          int W_in, int W_out, const int* __restrict__ xmin,
                                                          const int* __restrict__ xsize, const float* __restrict__ w, int kmax) {
    const long total = rows * W_out;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const long row = idx / W_out;
        const int i = (int)(idx - row * W_out);
        const TIN* sp = src + row * W_in + xmin[i];
        const float* wp = w + (size_t)i * kmax;
        const int n = xsize[i];
        float t = __fmul_rn((float)sp[0], wp[0]);
        for (int j = 1; j < n; ++j) t = __fadd_rn(t, __fmul_rn((float)sp[j], wp[j]));
        dst[idx] = t;
    }
}

__global__ __launch_bounds__(256) void resize_aa_h_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                          int H_in, int H_out, int W, const int* __restrict__ ymin,
                                                          const int* __restrict__ ysize, const float* __restrict__ w, int kmax,
                                                          int round_u8) {
    const long total = (long)planes * H_out * W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int x = (int)(idx % W);
        const long r = idx / W;
        const int i = (int)(r % H_out);
        const long pl = r / H_out;
        const float* sp = src + (pl * H_in + ymin[i]) * W + x;
        const float* wp = w + (size_t)i * kmax;
        const int n = ysize[i];
        float t = __fmul_rn(sp[0], wp[0]);
        for (int j = 1; j < n; ++j) t = __fadd_rn(t, __fmul_rn(sp[(size_t)j * W], wp[j]));
        if (round_u8) t = fminf(fmaxf(rintf(t), 0.f), 255.f);
        dst[idx] = t;
    }
}

extern "C" int o3v_resize_bicubic_aa(const void* src, int is_u8, float* tmp, float* dst, int planes, int H_in, int W_in,
                                     int H_out, int W_out, const int* xmin, const int* xsize, const float* xw, int xk,
                                     const int* ymin, const int* ysize, const float* yw, int yk, hipStream_t stream) {
    if (!src || !tmp || !dst || !xmin || !xsize || !xw || !ymin || !ysize || !yw || planes < 0 || H_in <= 0 || W_in <= 0 ||
        H_out <= 0 || W_out <= 0 || xk <= 0 || yk <= 0)
        return O3V_ERR_ARG;
    if (planes == 0) return O3V_OK;
    const long rows = (long)planes * H_in;
    long b1 = (rows * W_out + 255) / 256, b2 = ((long)planes * H_out * W_out + 255) / 256;
    b1 = b1 < 16384 ? b1 : 16384;
    b2 = b2 < 16384 ? b2 : 16384;
    if (is_u8)
        O3V_KLAUNCH(resize_aa_w_kernel<uint8_t>, dim3((int)b1), dim3(256), 0, stream, (const uint8_t*)src, tmp, rows, W_in, W_out,
                    xmin, xsize, xw, xk);
    else
        O3V_KLAUNCH(resize_aa_w_kernel<float>, dim3((int)b1), dim3(256), 0, stream, (const float*)src, tmp, rows, W_in, W_out, xmin,
                    xsize, xw, xk);
    O3V_KLAUNCH(resize_aa_h_kernel, dim3((int)b2), dim3(256), 0, stream, (const float*)tmp, dst, planes, H_in, H_out, W_out, ymin,
                ysize, yw, yk, is_u8);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Test-time-scaling crops (R:eval/tts.py:54-75 crop_box): box [x1,x2) x [y1,y2) of frame f, resized back to the
// full frame size W x H with OpenCV's INTER_LINEAR on float32 (pixel centres: src = (dst + 0.5) * scale - 0.5,
// edge replicate, horizontal pass then vertical pass, each a two-term float sum) and truncated to uint8.
// The frames are already resident on the device for the ViT; nothing goes back to the host.
// boxes: int32 [n][5] = {frame, x1, y1, x2, y2}, clipped to the frame, x2 > x1, y2 > y1.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_coeff(int d, double scale, int ssize, int& s0, int& s1, float& a0, float& a1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) {
        f = 0.f;
        si = 0;
    }
    if (si >= ssize - 1) {
        f = 0.f;
        si = ssize - 1;
    }
    s0 = si;
    s1 = si + 1 < ssize ? si + 1 : si;
    a0 = 1.f - f;
    a1 = f;
}

__global__ __launch_bounds__(256) void crop_resize_bilinear_kernel(const uint8_t* __restrict__ frames,
                                                                   const int* __restrict__ boxes, uint8_t* __restrict__ out,
                                                                   int T, int H, int W) {
    const int n = blockIdx.z, ch = blockIdx.y;
    const int f = boxes[n * 5], x1 = boxes[n * 5 + 1], y1 = boxes[n * 5 + 2], x2 = boxes[n * 5 + 3], y2 = boxes[n * 5 + 4];
    const int cw = x2 - x1, chh = y2 - y1;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W || f < 0 || f >= T || cw <= 0 || chh <= 0) return;
    const int dy = p / W, dx = p - dy * W;
    const double sx = (double)cw / (double)W, sy = (double)chh / (double)H;
    int xa, xb, ya, yb;
    float ax0, ax1, by0, by1;
    linear_coeff(dx, sx, cw, xa, xb, ax0, ax1);
    linear_coeff(dy, sy, chh, ya, yb, by0, by1);
    const uint8_t* src = frames + ((size_t)f * 3 + ch) * H * W + (size_t)y1 * W + x1;
    const float r0 = __fadd_rn(__fmul_rn((float)src[(size_t)ya * W + xa], ax0), __fmul_rn((float)src[(size_t)ya * W + xb], ax1));
    const float r1 = __fadd_rn(__fmul_rn((float)src[(size_t)yb * W + xa], ax0), __fmul_rn((float)src[(size_t)yb * W + xb], ax1));
    const float v = __fadd_rn(__fmul_rn(r0, by0), __fmul_rn(r1, by1));
    out[((size_t)n * 3 + ch) * H * W + p] = (uint8_t)(int)v;  // numpy astype(uint8) of a value in [0, 255]: truncation
}

extern "C" int o3v_crop_resize_bilinear(const void* frames, const int* boxes, void* out, int n, int T, int H, int W,
                                        hipStream_t stream) {
    if (!frames || !boxes || !out || n < 0 || T <= 0 || H <= 0 || W <= 0) return O3V_ERR_ARG;
    if (n == 0) return O3V_OK;
    if (n > 65535) return O3V_ERR_SHAPE;
    O3V_KLAUNCH(crop_resize_bilinear_kernel, dim3((H * W + 255) / 256, 3, n), dim3(256), 0, stream, (const uint8_t*)frames, boxes,
                (uint8_t*)out, T, H, W);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Greedy sampler (TF:generation/utils.py:2894-2929 + logits_process.py:404-414): fp32 view of the bf16
// last-token logits, repetition penalty over every id seen so far (prompt + generated; `seen` is a
// byte map [B,V]), argmax with lowest-index tie break, pad after EOS, margin = top1 - top2.
// One 1024-thread block per sequence.
// ------------------------------------------------------------------------------------------------
constexpr int GREEDY_NB = 64;

__device__ __forceinline__ void top2_merge(float& best, int& bi, float& second, float ob, int oi, float os) {
    if (ob > best || (ob == best && oi < bi)) {
        second = fmaxf(best, os);
        best = ob;
        bi = oi;
    } else {
        second = fmaxf(second, ob);
    }
}

// stage 1: grid (NB, B) x 256 threads, each block reduces a contiguous slice of the vocabulary to (best, idx, second)
__global__ __launch_bounds__(256) void sample_greedy_partial_kernel(const bf16_t* __restrict__ logits,
                                                                    const uint8_t* __restrict__ seen,
                                                                    float* __restrict__ part, int V, int ldl,
                                                                    float rep_penalty) {
    const int b = blockIdx.y, nb = blockIdx.x;
    const bf16_t* lr = logits + (size_t)b * ldl;
    const uint8_t* sr = seen + (size_t)b * V;
    const int per = (V + GREEDY_NB - 1) / GREEDY_NB;
    const int i0 = nb * per, i1 = (i0 + per < V) ? i0 + per : V;
    float best = -INFINITY, second = -INFINITY;
    int bi = 0x7fffffff;
    auto visit = [&](int i, float s, unsigned seen_byte) {
        if (rep_penalty != 1.0f && seen_byte) s = (s < 0.f) ? s * rep_penalty : s / rep_penalty;
        if (s > best) {
            second = best;
            best = s;
            bi = i;
        } else if (s > second) {
            second = s;
        }
    };
    const bool vec = ((per & 7) == 0) && ((V & 7) == 0) && ((ldl & 7) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(seen) & 7) == 0);
    if (vec) {
        // 16 bytes of logits + 8 bytes of the seen map per lane and trip: the slice is two trips instead of ten dependent ones
        for (int g = (i0 >> 3) + threadIdx.x; g < (i1 >> 3); g += 256) {
            const uint4 lv = *reinterpret_cast<const uint4*>(lr + g * 8);
            const uint2 sv = *reinterpret_cast<const uint2*>(sr + g * 8);
            const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)   // ascending index inside the lane: the first maximum wins, like torch.argmax
                visit(g * 8 + k, (k & 1) ? bf_hi(lw[k >> 1]) : bf_lo(lw[k >> 1]), ((k < 4 ? sv.x : sv.y) >> (8 * (k & 3))) & 0xffu);
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += 256) visit(i, bf2f(lr[i]), sr[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        top2_merge(best, bi, second, ob, oi, os);
    }
    __shared__ float sb[4], ss[4];
    __shared__ int si[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sb[w] = best;
        ss[w] = second;
        si[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) top2_merge(best, bi, second, sb[k], si[k], ss[k]);
        float* p = part + ((size_t)b * GREEDY_NB + nb) * 4;
        p[0] = best;
        p[1] = __int_as_float(bi);
        p[2] = second;
    }
}

// stage 2: one wave per sequence merges the NB partials and applies the token bookkeeping
__global__ __launch_bounds__(64) void sample_greedy_final_kernel(const float* __restrict__ part, uint8_t* __restrict__ seen,
                                                                 int* __restrict__ cur_tok, int* __restrict__ finished,
                                                                 int* __restrict__ out_ids, float* __restrict__ margins,
                                                                 const int* __restrict__ eos_ids, int n_eos, int pad_id,
                                                                 int V, int step, int out_stride,
                                                                 const uint4* __restrict__ embed, uint4* __restrict__ x_out,
                                                                 int hidden) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* p = part + ((size_t)b * GREEDY_NB + lane) * 4;
    float best = -INFINITY, second = -INFINITY;
    int bi = 0x7fffffff;
    if (lane < GREEDY_NB) {
        best = p[0];
        bi = __float_as_int(p[1]);
        second = p[2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        top2_merge(best, bi, second, ob, oi, os);
    }
    if (lane == 0) {
        int tok = finished[b] ? pad_id : bi;
        if (!finished[b]) {
            for (int k = 0; k < n_eos; ++k)
                if (tok == eos_ids[k]) finished[b] = 1;
        }
        if (tok >= 0 && tok < V) seen[(size_t)b * V + tok] = 1;
        cur_tok[b] = tok;
        out_ids[(size_t)b * out_stride + step] = tok;
        if (margins) margins[(size_t)b * out_stride + step] = best - second;
        bi = tok;
    }
    if (embed) {  // the next decode forward starts from this token's embedding row: gather it here, one launch less per step
        const int tok = __shfl(bi, 0, 64);
        const int cpr = hidden >> 3;
        const uint4* src = embed + (size_t)(tok >= 0 ? tok : 0) * cpr;
        for (int c = lane; c < cpr; c += 64) x_out[(size_t)b * cpr + c] = src[c];
    }
}

// scratch: f32 [B, 64, 4] partials
static int sample_greedy_impl(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* margins,
                              const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty, int step,
                              int out_stride, float* scratch, const void* embed, void* x_out, int hidden, hipStream_t stream) {
    if (!logits || !seen || !cur_tok || !finished || !out_ids || !scratch || B < 0 || V <= 0 || step < 0 ||
        step >= out_stride)
        return O3V_ERR_ARG;
    if (embed && (!x_out || hidden <= 0 || (hidden & 7))) return O3V_ERR_ARG;
    if (B == 0) return O3V_OK;
    O3V_KLAUNCH(sample_greedy_partial_kernel, dim3(GREEDY_NB, B), dim3(256), 0, stream, (const bf16_t*)logits,
                (const uint8_t*)seen, scratch, V, ldl, rep_penalty);
    O3V_KLAUNCH(sample_greedy_final_kernel, dim3(B), dim3(64), 0, stream, scratch, (uint8_t*)seen, cur_tok, finished, out_ids,
                margins, eos_ids, n_eos, pad_id, V, step, out_stride, (const uint4*)embed, (uint4*)x_out, hidden);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_sample_greedy(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* margins,
                                 const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty,
                                 int step, int out_stride, float* scratch, hipStream_t stream) {
    return sample_greedy_impl(logits, seen, cur_tok, finished, out_ids, margins, eos_ids, n_eos, pad_id, B, V, ldl, rep_penalty,
                              step, out_stride, scratch, nullptr, nullptr, 0, stream);
}

// The same, and the chosen token's embedding row goes to x_out[b] (bf16 [B, hidden]) -- what the next decode forward
// (TF:1206-1207 embed_tokens of the new token) starts from.
extern "C" int o3v_sample_greedy_embed(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids,
                                       float* margins, const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl,
                                       float rep_penalty, int step, int out_stride, float* scratch, const void* embed,
                                       void* x_out, int hidden, hipStream_t stream) {
    if (!embed) return O3V_ERR_ARG;
    return sample_greedy_impl(logits, seen, cur_tok, finished, out_ids, margins, eos_ids, n_eos, pad_id, B, V, ldl, rep_penalty,
                              step, out_stride, scratch, embed, x_out, hidden, stream);
}

// mark prompt ids as seen (repetition penalty covers prompt + generated, TF:logits_process.py:404-414)
__global__ void mark_seen_kernel(const int* __restrict__ ids, uint8_t* __restrict__ seen, int B, int S, int V) {
    long total = (long)B * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / S);
        int id = ids[i];
        if (id >= 0 && id < V) seen[(size_t)b * V + id] = 1;
    }
}

extern "C" int o3v_mark_seen(const int* ids, void* seen, int B, int S, int V, hipStream_t stream) {
    if (!ids || !seen || B < 0 || S < 0 || V <= 0) return O3V_ERR_ARG;
    if (B * (long)S == 0) return O3V_OK;
    int blocks = (int)(((long)B * S + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    O3V_KLAUNCH(mark_seen_kernel, dim3(blocks), dim3(256), 0, stream, ids, (uint8_t*)seen, B, S, V);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// decode-step embedding lookup: x[b,:] = table[cur_tok[b],:]
__global__ __launch_bounds__(256) void embed_tokens_kernel(const uint4* __restrict__ table, const int* __restrict__ tok,
                                                           uint4* __restrict__ out, int B, int chunks) {
    long total = (long)B * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % chunks);
        int b = (int)(i / chunks);
        out[(size_t)b * chunks + c] = table[(size_t)tok[b] * chunks + c];
    }
}

extern "C" int o3v_embed_tokens(const void* table, const int* tok, void* out, int B, int hidden, hipStream_t stream) {
    if (!table || !tok || !out || B < 0 || hidden <= 0 || (hidden & 7)) return O3V_ERR_ARG;
    if (B == 0) return O3V_OK;
    long total = (long)B * (hidden >> 3);
    int blocks = (int)((total + 255) / 256);
    O3V_KLAUNCH(embed_tokens_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)table, tok, (uint4*)out, B,
                       hidden >> 3);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-token log-prob gather (R:grpo_trainer.py:371-384): logits bf16 [R, V] row r predicts target[r];
// out[r] = logits[r,target] - logsumexp(logits[r,:]) computed in fp32 (log_softmax of the bf16 logits
// is done by torch in bf16->fp32 internally; the reference then gathers).  One block per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logprob_gather_kernel(const bf16_t* __restrict__ logits, const int* __restrict__ target,
                                                             float* __restrict__ out, int V, int ldl) {
    const int r = blockIdx.x;
    const bf16_t* lr = logits + (size_t)r * ldl;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) m = fmaxf(m, bf2f(lr[i]));
    __shared__ float red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) s += expf(bf2f(lr[i]) - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = red[0] + red[1] + red[2] + red[3];
        int t = target[r];
        out[r] = (t >= 0 && t < V) ? (bf2f(lr[t]) - m) - logf(s) : 0.f;
    }
}

extern "C" int o3v_logprob_gather(const void* logits, const int* target, float* out, int R, int V, int ldl,
                                  hipStream_t stream) {
    if (!logits || !target || !out || R < 0 || V <= 0) return O3V_ERR_ARG;
    if (R == 0) return O3V_OK;
    O3V_KLAUNCH(logprob_gather_kernel, dim3(R), dim3(256), 0, stream, (const bf16_t*)logits, target, out, V, ldl);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------
// 128-bit content hash of a device buffer (cache keys of the visual-token / prefix-K/V caches: two clips must never share
// a key by accident).  h = (sum_i mix(w_i + c1 i), sum_i mix(w_i * c2 ^ (i + c3))) over the 8-byte words w_i, mix =
// splitmix64's finaliser: sums commute, so the blocks combine with integer atomics in any order; the tail bytes and the
// length enter as extra words.  Not cryptographic; a chance collision is ~2^-128.
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned long long hmix64(unsigned long long x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void content_hash_kernel(const unsigned long long* __restrict__ p, size_t nwords,
                                                            const unsigned char* __restrict__ tail, int ntail, size_t nbytes,
                                                            unsigned long long* __restrict__ out) {
    unsigned long long h1 = 0, h2 = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) {
        const unsigned long long w = p[i];
        h1 += hmix64(w + 0x9E3779B97F4A7C15ull * (i + 1));
        h2 += hmix64((w * 0xD1B54A32D192ED03ull) ^ (i + 0x2545F4914F6CDD1Dull));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < ntail; ++k) t |= (unsigned long long)tail[k] << (8 * k);
        h1 += hmix64(t + 0x9E3779B97F4A7C15ull * (nwords + 1)) + hmix64(nbytes ^ 0xA0761D6478BD642Full);
        h2 += hmix64((t * 0xD1B54A32D192ED03ull) ^ (nwords + 0x2545F4914F6CDD1Dull)) + hmix64(nbytes * 0xE7037ED1A0B428DBull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        h1 += ((unsigned long long)__shfl_xor((unsigned)(h1 >> 32), o, 64) << 32) | __shfl_xor((unsigned)h1, o, 64);
        h2 += ((unsigned long long)__shfl_xor((unsigned)(h2 >> 32), o, 64) << 32) | __shfl_xor((unsigned)h2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, h1);
        atomicAdd(out + 1, h2);
    }
}
}  // namespace

// out: 2 x u64, zeroed by the caller before the call; data 8-byte aligned
extern "C" int o3v_content_hash128(const void* data, size_t bytes, unsigned long long* out, hipStream_t stream) {
    if (!data || !out || (reinterpret_cast<uintptr_t>(data) & 7)) return O3V_ERR_ARG;
    const size_t nwords = bytes / 8;
    int blocks = (int)((nwords + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    O3V_KLAUNCH(content_hash_kernel, dim3(blocks), dim3(256), 0, stream, (const unsigned long long*)data, nwords,
                (const unsigned char*)data + nwords * 8, (int)(bytes - nwords * 8), bytes, out);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
