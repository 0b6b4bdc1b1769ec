// Attention kernels of the Qwen2.5-VL generate path on gfx950 (MI355X).
//
//   o3v_attn_tiles  : flash-style attention over a list of 64-row query tiles.  Serves the ViT
//                     (non-causal, ragged segments from cu_seqlens: windows <= 64 patches and whole
//                     frames, TF:modeling_qwen2_5_vl.py:248-287) and the LLM prefill (causal, GQA, left
//                     padding, K/V read from the cache, TF:602-689).
//   o3v_attn_decode : one query token per sequence against the whole cache (HBM-bound KV read), split
//                     over the context; o3v_attn_decode_combine merges the splits.
//
// Flash kernel layout ("swapped QK^T", so nothing ever crosses lanes between the two MFMA chains):
//   S^T[key][q]  = K_tile . Q^T   A = K rows from LDS (ds_read_b128), B = Q rows held in VGPRs
//   accumulator: lane (q = lane&15, g = lane>>4) holds keys 16*kb + 4g + r  -> softmax state (m, l) of
//   query q lives in the lane; the row max needs two shuffles (xor 16, 32), the row sum none.
//   O^T[d][q]   += V^T . P^T      B = P^T straight from the S accumulators (cvt to bf16, k-slot order
//   permuted consistently), A = V^T fetched with ds_read_b64_tr_b16 from the row-major V tile.
// fp32 scores/softmax/accumulators; P is rounded to bf16 before P.V as the reference's bf16 attention does.
#include "o3v_common.h"
#include "o3v_attn_decode_body.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

struct TileDesc {  // 8 ints, built on the host (open_o3_video_amd/indexing.py)
    int q_row0;      // first query token (row of Q / O)
    int q_rows;      // valid rows in this tile (1..64)
    int k_row0;      // first key token of the segment inside the (batch-offset) K / V arrays
    int k_len;       // keys in the segment / cache length
    int causal_off;  // >= 0: key j allowed iff j <= causal_off + row_in_tile ; < 0: no causal mask
    int k_lo;        // first valid key (left padding)
    int batch;       // batch row (selects the K/V batch stride)
    int pad_;
};

template <int D>
struct AttnCfg {
    static constexpr int DPAD = (D + 31) / 32 * 32;       // QK^T k-steps of 32
    static constexpr int KS = DPAD / 32;
    static constexpr int DB = D / 16;                      // 16-wide output d blocks
    // row strides (bytes): K rows read by ds_read_b128 want stride/16 odd, V rows read by
    // ds_read_b64_tr_b16 (8 rows x 32 B per half-wave) want stride/32 odd -> conflict-free.
    static constexpr int KSTRIDE = DPAD * 2 + 16;
    static constexpr int VSTRIDE = ((D * 2 / 32) % 2 == 1) ? D * 2 : D * 2 + 32;
    static constexpr int K_BYTES = 64 * KSTRIDE;
    static constexpr int V_BYTES = 64 * VSTRIDE;
};

// RQ = 16-row query blocks per wave (1: 64-row tiles for the ragged ViT windows; 2: 128-row tiles for the LLM prefill,
// where every K fragment read and every transposed V read from LDS feeds two MFMAs instead of one -- the kernel is
// LDS-bandwidth-bound at RQ = 1).
template <int D, int RQ, bool PFX = false>
__global__ __launch_bounds__(256, 2) void attn_tiles_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                         const TileDesc* __restrict__ tiles, long q_ts, long k_ts,
                                                         long k_hs, long k_bs, long v_ts, long v_hs, long v_bs, long o_ts,
                                                         int n_rep, float scale_log2e, PrefixRef pf, int P) {
    using C = AttnCfg<D>;
    // head_dim 128: K/V tiles go global -> LDS by DMA (global_load_lds, no registers), double buffered, so the loads of
    // tile kt+1 run under the MFMAs of tile kt and a tile costs one barrier instead of two.  Rows are 256 B = 16 chunks
    // with no padding (a DMA instruction fills 1 KiB of consecutive LDS); the chunk a lane fetches is XOR-swizzled with the
    // row number instead, which keeps both the K fragment reads and the transposed V reads conflict-free.
    constexpr bool DMA = (D == 128);
    constexpr int STAGE = 2 * 64 * 256;  // K tile + V tile of one DMA stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + (DMA ? 64 * 256 : C::K_BYTES);

    const TileDesc td = tiles[blockIdx.x];
    const int h = blockIdx.y, hk = h / n_rep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = fr][ks*32 + 8*fg .. +7] of each of its RQ row blocks
    int qrow_in[RQ];
    bf16x8 qf[RQ][C::KS];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        qrow_in[rq] = (wave * RQ + rq) * 16 + fr;
        const int qrow_ld = td.q_row0 + (qrow_in[rq] < td.q_rows ? qrow_in[rq] : td.q_rows - 1);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 32 + fg * 8;
            if (d0 < D)
                qf[rq][ks] = *reinterpret_cast<const bf16x8*>(Q + (size_t)qrow_ld * q_ts + (size_t)h * D + d0);
            else
                qf[rq][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    float m_run[RQ], l_run[RQ];
    f32x4 o[RQ][C::DB];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        m_run[rq] = -1e30f;
        l_run[rq] = 0.f;
#pragma unroll
        for (int i = 0; i < C::DB; ++i) o[rq][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const bf16_t* Kb = K + (size_t)td.batch * k_bs + (size_t)hk * k_hs + (size_t)td.k_row0 * k_ts;
    const bf16_t* Vb = V + (size_t)td.batch * v_bs + (size_t)hk * v_hs + (size_t)td.k_row0 * v_ts;
    // PFX: keys below P come from the prompt entry shared by pf.rows batch rows; K / V then hold only the tokens behind it
    // (logical key kr >= P in row kr - P).  Same token strides in both.
    const bf16_t *Kp = nullptr, *Vp = nullptr;
    if (PFX) {
        const size_t po = (size_t)(td.batch / pf.rows) * pf.bs + (size_t)hk * pf.hs;
        Kp = pf.K + po;
        Vp = pf.V + po;
        Kb -= (size_t)P * k_ts;
        Vb -= (size_t)P * v_ts;
    }
    // (macros, not lambdas: a lambda called from the staging lambdas below keeps clang from emitting this kernel's host stub)
#define krow(kr) (((PFX && (kr) < P) ? Kp : Kb) + (size_t)(kr) * k_ts)
#define vrow(kr) (((PFX && (kr) < P) ? Vp : Vb) + (size_t)(kr) * v_ts)

    int k_end = td.k_len;
    if (td.causal_off >= 0) {
        const int lim = td.causal_off + td.q_rows;  // exclusive
        k_end = lim < k_end ? lim : k_end;
    }
    const int kt_lo = td.k_lo >> 6;
    const int kt_hi = (k_end + 63) >> 6;

    // K/V staging.  A split form (global loads of tile kt+1 issued right after the barrier that publishes tile kt, written
    // to LDS after the next barrier) is kept behind SPLIT: it costs 32 VGPRs and with them a block of occupancy, which
    // measured slower than the plain load-store staging whose latency the co-resident blocks hide.
    constexpr int CPR = C::DPAD / 8, VPR = D / 8;
    constexpr int NKC = (64 * CPR + 255) / 256, NVC = (64 * VPR + 255) / 256;
    constexpr bool SPLIT = false;  // measured on MI355X: the 32 extra VGPRs drop RQ = 2 to one block per CU: S=20k prefill 550 -> 700 ms
    uint4 kreg[SPLIT ? NKC : 1], vreg[SPLIT ? NVC : 1];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = threadIdx.x + i * 256;
            const int row = c / CPR, ch = c % CPR;
            int kr = kt * 64 + row;
            kr = kr < td.k_len ? kr : td.k_len - 1;
            kreg[SPLIT ? i : 0] = make_uint4(0, 0, 0, 0);
            if (c < 64 * CPR && ch * 8 < D) kreg[SPLIT ? i : 0] = *reinterpret_cast<const uint4*>(krow(kr) + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = threadIdx.x + i * 256;
            const int row = c / VPR, ch = c % VPR;
            int kr = kt * 64 + row;
            kr = kr < td.k_len ? kr : td.k_len - 1;
            if (c < 64 * VPR) vreg[SPLIT ? i : 0] = *reinterpret_cast<const uint4*>(vrow(kr) + ch * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < 64 * CPR) *reinterpret_cast<uint4*>(Kl + (c / CPR) * C::KSTRIDE + (c % CPR) * 16) = kreg[SPLIT ? i : 0];
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < 64 * VPR) *reinterpret_cast<uint4*>(Vl + (c / VPR) * C::VSTRIDE + (c % VPR) * 16) = vreg[SPLIT ? i : 0];
        }
    };
    if (SPLIT && kt_lo < kt_hi) load_tile(kt_lo);
    auto dma_tile = [&](int kt, char* stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int instr = wave * 4 + j;              // 16 instructions of 4 rows per tile, 4 per wave
            const int row = instr * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);      // logical 16-byte chunk stored at slot (lane & 15) of the row
            int kr = kt * 64 + row;
            kr = kr < td.k_len ? kr : td.k_len - 1;
            __builtin_amdgcn_global_load_lds(krow(kr) + c * 8, (lds_void_t*)(stage + instr * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(vrow(kr) + c * 8, (lds_void_t*)(stage + 64 * 256 + instr * 1024), 16, 0, 0);
        }
    };
    if (DMA && kt_lo < kt_hi) dma_tile(kt_lo, smem);

    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        __syncthreads();  // DMA: tile kt has landed (vmcnt(0) + barrier) and tile kt-1 is fully consumed
        if (DMA) {
            const int st = (kt - kt_lo) & 1;
            Kl = smem + st * STAGE;
            Vl = Kl + 64 * 256;
            if (kt + 1 < kt_hi) dma_tile(kt + 1, smem + (st ^ 1) * STAGE);
        } else if (SPLIT) {
            store_tile();
        } else {
            // ---- stage K (zero padded to DPAD) and V tiles through registers
            for (int c = threadIdx.x; c < 64 * CPR; c += 256) {
                const int row = c / CPR, ch = c % CPR;
                int kr = kt * 64 + row;
                kr = kr < td.k_len ? kr : td.k_len - 1;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ch * 8 < D) v = *reinterpret_cast<const uint4*>(krow(kr) + ch * 8);
                *reinterpret_cast<uint4*>(Kl + row * C::KSTRIDE + ch * 16) = v;
            }
            for (int c = threadIdx.x; c < 64 * VPR; c += 256) {
                const int row = c / VPR, ch = c % VPR;
                int kr = kt * 64 + row;
                kr = kr < td.k_len ? kr : td.k_len - 1;
                const uint4 v = *reinterpret_cast<const uint4*>(vrow(kr) + ch * 8);
                *reinterpret_cast<uint4*>(Vl + row * C::VSTRIDE + ch * 16) = v;
            }
        }
        if (!DMA) __syncthreads();
        if (SPLIT && kt + 1 < kt_hi) load_tile(kt + 1);
        // wave-level skip: with a causal mask a wave whose rows all precede this key tile has nothing to do
        if (td.causal_off >= 0 && kt * 64 > td.causal_off + (wave * RQ + RQ) * 16 - 1) continue;

        // ---- S^T = K . Q^T : 4 key blocks of 16, every K fragment feeds RQ MFMAs
        f32x4 s[RQ][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) s[rq][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const bf16x8 kf =
                    DMA ? *reinterpret_cast<const bf16x8*>(Kl + (kb * 16 + fr) * 256 + (((ks * 4 + fg) ^ fr) << 4))
                        : *reinterpret_cast<const bf16x8*>(Kl + (kb * 16 + fr) * C::KSTRIDE + (ks * 32 + fg * 8) * 2);
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) s[rq][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[rq][ks], s[rq][kb], 0, 0, 0);
            }
        }
        // ---- online softmax on the raw scores (running max kept unscaled; p = exp2(s*c - m*c) is one fma + one exp2).
        // Masking costs ~3x the arithmetic of the softmax itself, so it only runs on boundary tiles (left padding, the
        // causal diagonal of this wave, the ragged tail); interior tiles take the mask-free path (wave-uniform branch).
        bf16x8 pb[RQ][2];
        const bool boundary = (kt * 64 < td.k_lo) || (kt * 64 + 63 >= td.k_len) ||
                              (td.causal_off >= 0 && kt * 64 + 63 > td.causal_off + wave * RQ * 16);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
            float mx = -1e30f;
            if (boundary) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = kt * 64 + kb * 16 + fg * 4 + r;
                        bool v = (j < td.k_len) && (j >= td.k_lo);
                        if (td.causal_off >= 0) v = v && (j <= td.causal_off + qrow_in[rq]);
                        s[rq][kb][r] = v ? s[rq][kb][r] : -1e30f;
                        mx = fmaxf(mx, s[rq][kb][r]);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[rq][kb][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[rq], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[rq] - m_new) * scale_log2e);
            const float mc = -m_new * scale_log2e;
            m_run[rq] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // masked entries sit at -1e30: exp2 underflows to exactly 0 unless the whole row is still masked
                    // (m_new = -1e30), where the explicit select keeps them out
                    float p = __builtin_amdgcn_exp2f(fmaf(s[rq][kb][r], scale_log2e, mc));
                    if (boundary) p = (s[rq][kb][r] > -1e29f) ? p : 0.f;
                    psum += p;
                    pb[rq][kb >> 1][(kb & 1) * 4 + r] = (short)f2bf(p);
                }
            l_run[rq] = l_run[rq] * alpha + psum;
#pragma unroll
            for (int i = 0; i < C::DB; ++i) o[rq][i] *= alpha;
        }

        // ---- O^T += V^T . P^T.  k-slot j of lane group fg: j<4 -> key 32kk + 4fg + j ; j>=4 -> key 32kk + 16 + 4fg + (j-4)
        const int tq = fr >> 2, tp = fr & 3;  // tr-read: lane 4q+p of the 16-lane group addresses row q, cols 4p..4p+3
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // DMA layout: row * 256 + ((chunk ^ (row & 15)) << 4) + half * 8 with chunk = 2 db + (tp >> 1), half = tp & 1;
            // rows of r0 and r1 differ by 16, so they share (row & 15)
            const int vrow = kk * 32 + fg * 4 + tq, vx = vrow & 15;
            const char* r0 = DMA ? Vl + vrow * 256 + (tp & 1) * 8 : Vl + vrow * C::VSTRIDE + tp * 8;
            const char* r1 = r0 + 16 * (DMA ? 256 : C::VSTRIDE);
#pragma unroll
            for (int db = 0; db < C::DB; ++db) {
                const int voff = DMA ? (((2 * db + (tp >> 1)) ^ vx) << 4) : db * 32;
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r0 + voff));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r1 + voff));
                const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int rq = 0; rq < RQ; ++rq) o[rq][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[rq][kk], o[rq][db], 0, 0, 0);
            }
        }
    }

    // ---- finalize: l over the 4 lane groups, O[q][d = db*16 + 4fg + r]
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
        float l_tot = l_run[rq] + __shfl_xor(l_run[rq], 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (qrow_in[rq] < td.q_rows) {
            bf16_t* orow = O + (size_t)(td.q_row0 + qrow_in[rq]) * o_ts + (size_t)h * D + fg * 4;
#pragma unroll
            for (int db = 0; db < C::DB; ++db) {
                u32x2 pk;
                pk[0] = pack_bf2(o[rq][db][0] * inv, o[rq][db][1] * inv);
                pk[1] = pack_bf2(o[rq][db][2] * inv, o[rq][db][3] * inv);
                *reinterpret_cast<u32x2*>(orow + db * 16) = pk;
            }
        }
    }
}
#undef krow
#undef vrow

// ------------------------------------------------------------------------------------------------
// Decode attention.  grid (splits, Hkv, B), 4 waves.  A wave-iteration covers KPW = 64/(D/8) keys:
// lane (slot = lane / LPK, part = lane % LPK) loads 16 B of K and of V of key slot `slot`; the n_rep
// query heads of the kv head share every K/V byte (GQA).  Each (wave, slot) is an independent online
// softmax stream; streams are merged at the end, splits by o3v_attn_decode_combine.
// ------------------------------------------------------------------------------------------------

template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                          const bf16_t* __restrict__ Vc, float* __restrict__ part_o,
                                                          float* __restrict__ part_ml, const int* __restrict__ k_lo_arr,
                                                          int ctx, int Hq, int Hkv, int n_rep, long k_hs, long k_bs,
                                                          float scale_log2e) {
    constexpr int LPK = D / 8, KPW = 64 / LPK;
    const int split = blockIdx.x, nsplit = gridDim.x, hk = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane / LPK, part = lane % LPK;
    const int k_lo = k_lo_arr ? k_lo_arr[b] : 0;

    int chunk = (ctx + nsplit - 1) / nsplit;
    chunk = (chunk + 16 * KPW - 1) / (16 * KPW) * (16 * KPW);  // whole trips: 4 waves x KPW slots x U keys
    const int k0 = split * chunk;
    int k1 = k0 + chunk;
    k1 = k1 < ctx ? k1 : ctx;

    float q[NREP_MAX][8];
#pragma unroll
    for (int r = 0; r < NREP_MAX; ++r) {
        const int hh = hk * n_rep + (r < n_rep ? r : n_rep - 1);
        const u32x4 v = *reinterpret_cast<const u32x4*>(Q + ((size_t)b * Hq + hh) * D + part * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q[r][2 * j] = bf_lo(v[j]) * scale_log2e;
            q[r][2 * j + 1] = bf_hi(v[j]) * scale_log2e;
        }
    }
    float m[NREP_MAX], l[NREP_MAX], o[NREP_MAX][8];
#pragma unroll
    for (int r = 0; r < NREP_MAX; ++r) {
        m[r] = -1e30f;
        l[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[r][j] = 0.f;
    }
    const bf16_t* Kb = Kc + (size_t)b * k_bs + (size_t)hk * k_hs;
    const bf16_t* Vb = Vc + (size_t)b * k_bs + (size_t)hk * k_hs;

    // U keys per lane group and trip: all K/V loads of a trip are in flight before the first use, and the online
    // softmax rescales once per trip (max over the U scores), not once per key.
    constexpr int U = 4;
    for (int kb0 = k0 + wave * KPW; kb0 < k1; kb0 += 4 * KPW * U) {
        u32x4 kv[U], vv[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = kb0 + u * 4 * KPW + slot;
            valid[u] = key < k1 && key >= k_lo;
            const int kl = key < ctx ? key : ctx - 1;
            kv[u] = *reinterpret_cast<const u32x4*>(Kb + (size_t)kl * D + part * 8);
            vv[u] = *reinterpret_cast<const u32x4*>(Vb + (size_t)kl * D + part * 8);
        }
#pragma unroll
        for (int r = 0; r < NREP_MAX; ++r) {
            if (r >= n_rep) break;  // wave-uniform
            float sc[U];
            float mn = m[r];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s = fmaf(q[r][2 * j], bf_lo(kv[u][j]), s);
                    s = fmaf(q[r][2 * j + 1], bf_hi(kv[u][j]), s);
                }
#pragma unroll
                for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                sc[u] = valid[u] ? s : -1e30f;
                mn = fmaxf(mn, sc[u]);
            }
            const float alpha = __builtin_amdgcn_exp2f(m[r] - mn);
            m[r] = mn;
            float pu[U], ps = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = valid[u] ? __builtin_amdgcn_exp2f(sc[u] - mn) : 0.f;
                pu[u] = bf2f(f2bf(p));  // the reference rounds the probabilities to bf16 before P.V
                ps += pu[u];
            }
            l[r] = l[r] * alpha + ps;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a0 = o[r][2 * j] * alpha, a1 = o[r][2 * j + 1] * alpha;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    a0 = fmaf(pu[u], bf_lo(vv[u][j]), a0);
                    a1 = fmaf(pu[u], bf_hi(vv[u][j]), a1);
                }
                o[r][2 * j] = a0;
                o[r][2 * j + 1] = a1;
            }
        }
    }
    // ---- merge the KPW slot streams of this wave (lanes with equal `part`)
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
#pragma unroll
        for (int r = 0; r < NREP_MAX; ++r) {
            const float mo = __shfl_xor(m[r], off, 64), lo = __shfl_xor(l[r], off, 64);
            const float mn = fmaxf(m[r], mo);
            const float a = __builtin_amdgcn_exp2f(m[r] - mn), bsc = __builtin_amdgcn_exp2f(mo - mn);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[r][j] = o[r][j] * a + __shfl_xor(o[r][j], off, 64) * bsc;
            l[r] = l[r] * a + lo * bsc;
            m[r] = mn;
        }
    }
    // ---- merge the 4 waves through LDS
    __shared__ float sm[4][NREP_MAX], sl[4][NREP_MAX], so[4][NREP_MAX][D];
    if (slot == 0) {
#pragma unroll
        for (int r = 0; r < NREP_MAX; ++r) {
            if (part == 0) {
                sm[wave][r] = m[r];
                sl[wave][r] = l[r];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) so[wave][r][part * 8 + j] = o[r][j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_rep * D; i += 256) {
        const int r = i / D, d = i % D;
        float mn = fmaxf(fmaxf(sm[0][r], sm[1][r]), fmaxf(sm[2][r], sm[3][r]));
        float acc = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __builtin_amdgcn_exp2f(sm[w][r] - mn);
            acc += so[w][r][d] * sc;
            lt += sl[w][r] * sc;
        }
        const size_t idx = (((size_t)b * Hq + hk * n_rep + r) * nsplit + split);
        part_o[idx * D + d] = acc;
        if (d == 0) {
            part_ml[idx * 2] = mn;
            part_ml[idx * 2 + 1] = lt;
        }
    }
}

// Decode attention on the matrix cores: attn_decode_mfma_body in o3v_attn_decode_body.h (shared with o3v_fused.hip)
__global__ __launch_bounds__(256) void attn_decode_mfma_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                               const bf16_t* __restrict__ Vc, float* __restrict__ part_o,
                                                               float* __restrict__ part_ml, const int* __restrict__ k_lo_arr,
                                                               int ctx, int Hq, int Hkv, int n_rep, long k_hs, long k_bs,
                                                               float scale_log2e, int kbeg, int nsplit_tot, int split_off,
                                                               int G, int P, PrefixRef pf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 x V slice, reused for the merge
    attn_decode_mfma_body<false>(Q, Kc, Vc, part_o, part_ml, k_lo_arr, ctx, Hq, Hkv, n_rep, k_hs, k_bs, scale_log2e, kbeg,
                                 nsplit_tot, split_off, G, P, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, smem, AttnHandoff{}, pf);
}

// ------------------------------------------------------------------------------------------------
// Group decode attention over a SHARED prompt prefix (head_dim 128).  The G completions of one prompt
// (num_return_sequences, R:grpo_trainer.py:306-313; n samples, R:eval/tts.py:47-123) attend to the same first
// `P` keys, so those K/V bytes are read once for the whole group instead of G times: the G * n_rep query rows
// (sequence-major) fill NQB 16-column blocks of the swapped QK^T product.  Keys past the prefix (each row's own
// generated tokens) are handled by attn_decode_mfma_kernel with kbeg = P; attn_decode_combine_kernel merges both.
// K/V of the prefix are read from the cache row of the group's first sequence.
// ------------------------------------------------------------------------------------------------
template <int NQB>
__device__ __forceinline__ void attn_decode_group_body(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                       const bf16_t* __restrict__ Vc, float* __restrict__ part_o,
                                                       float* __restrict__ part_ml, const int* __restrict__ k_lo_arr, int P, int G,
                                                       int Hq, int Hkv, int n_rep, long k_hs, long k_bs, float scale_log2e,
                                                       int nsplit_tot, const int split, const int nsplit, const int hk,
                                                       const int grp, const PrefixRef pf) {
    constexpr int D = 128, KT = 32, VSTRIDE = 288, V_BYTES = KT * VSTRIDE, QSTRIDE = 272, NQ = NQB * 16;
    constexpr int Q_BYTES = NQ * QSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // Q rows | 4 x V slice ; reused: so[NQ][128] | sm | sl
    const int b0 = grp * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int k_lo = k_lo_arr ? k_lo_arr[b0] : 0;
    const int nq = G * n_rep;
    char* Qs = smem;
    char* Vl = smem + Q_BYTES + wave * V_BYTES;

    // ---- Q rows of the group -> LDS (row q = seq * n_rep + r; rows past nq are zero)
    for (int c = threadIdx.x; c < NQ * 16; c += 256) {
        const int q = c >> 4, ch = c & 15;
        u32x4 v = (u32x4){0u, 0u, 0u, 0u};
        if (q < nq) {
            const int seq = q / n_rep, r = q - seq * n_rep;
            v = *reinterpret_cast<const u32x4*>(Q + ((size_t)(b0 + seq) * Hq + hk * n_rep + r) * D + ch * 8);
        }
        *reinterpret_cast<u32x4*>(Qs + q * QSTRIDE + ch * 16) = v;
    }
    __syncthreads();

    int chunk = (P + nsplit - 1) / nsplit;
    chunk = (chunk + 4 * KT - 1) / (4 * KT) * (4 * KT);
    const int per_wave = chunk >> 2;
    const int kw0 = split * chunk + wave * per_wave;
    int kw1 = kw0 + per_wave;
    kw1 = kw1 < P ? kw1 : P;
    // the prefix: from the prompt's shared entry, or (legacy layout) from the cache row of the group's first sequence
    const bf16_t* Kb = pf.K ? pf.K + (size_t)(b0 / pf.rows) * pf.bs + (size_t)hk * pf.hs : Kc + (size_t)b0 * k_bs + (size_t)hk * k_hs;
    const bf16_t* Vb = pf.K ? pf.V + (size_t)(b0 / pf.rows) * pf.bs + (size_t)hk * pf.hs : Vc + (size_t)b0 * k_bs + (size_t)hk * k_hs;

    float m_run[NQB], l_run[NQB];
    f32x4 o[NQB][8];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        m_run[qb] = -1e30f;
        l_run[qb] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[qb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int tq = fr >> 2, tp = fr & 3;

    for (int key0 = kw0; key0 < kw1; key0 += KT) {
        bf16x8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            int kr = key0 + kb * 16 + fr;
            kr = kr < P ? kr : P - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                kf[kb][ks] = *reinterpret_cast<const bf16x8*>(Kb + (size_t)kr * D + ks * 32 + fg * 8);
        }
        u32x4 vreg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane, row = c >> 4, ch = c & 15;
            int kr = key0 + row;
            kr = kr < P ? kr : P - 1;
            vreg[i] = *reinterpret_cast<const u32x4*>(Vb + (size_t)kr * D + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane, row = c >> 4, ch = c & 15;
            *reinterpret_cast<u32x4*>(Vl + row * VSTRIDE + ch * 16) = vreg[i];
        }
        bool ok[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = key0 + kb * 16 + fg * 4 + r;
                ok[kb][r] = (j < kw1) && (j >= k_lo);
            }
        const char* r0 = Vl + (fg * 4 + tq) * VSTRIDE + tp * 8;
        const char* r1 = r0 + 16 * VSTRIDE;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            // ---- S^T = K . Q^T for this block of 16 query rows
            f32x4 sacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + (qb * 16 + fr) * QSTRIDE + ks * 64 + fg * 16);
                sacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][ks], qf, sacc[0], 0, 0, 0);
                sacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][ks], qf, sacc[1], 0, 0, 0);
            }
            float mx = -1e30f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = ok[kb][r] ? sacc[kb][r] * scale_log2e : -1e30f;
                    sacc[kb][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
            m_run[qb] = m_new;
            float psum = 0.f;
            bf16x8 pb;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = ok[kb][r] ? __builtin_amdgcn_exp2f(sacc[kb][r] - m_new) : 0.f;
                    const bf16_t pq = f2bf(p);
                    psum += bf2f(pq);
                    pb[kb * 4 + r] = (short)pq;
                }
            l_run[qb] = l_run[qb] * alpha + psum;
            // ---- O^T += V^T . P^T
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r0 + db * 32));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r1 + db * 32));
                const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                o[qb][db] *= alpha;
                o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, o[qb][db], 0, 0, 0);
            }
        }
    }
    // ---- merge the 4 waves in wave order (deterministic): global max per query row, then scaled accumulation in LDS
    __syncthreads();                                        // Q rows and V slices are dead
    float* so = reinterpret_cast<float*>(smem);             // [NQ][128]
    float* sm = so + NQ * D;                                // [4][NQ]
    float* sl = sm + 4 * NQ;                                // [NQ]
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
        if (fg == 0) sm[wave * NQ + qb * 16 + fr] = m_run[qb];
    __syncthreads();
    float wsc[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int q = qb * 16 + fr;
        const float mg = fmaxf(fmaxf(sm[q], sm[NQ + q]), fmaxf(sm[2 * NQ + q], sm[3 * NQ + q]));
        wsc[qb] = __builtin_amdgcn_exp2f(m_run[qb] - mg);
        float lt = l_run[qb] + __shfl_xor(l_run[qb], 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        l_run[qb] = lt * wsc[qb];
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
                for (int db = 0; db < 8; ++db) {
                    float* dst = so + ((size_t)(qb * 16 + fr) * D + db * 16 + fg * 4);
                    f32x4 v = o[qb][db] * wsc[qb];
                    if (w > 0) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
                if (fg == 0) sl[qb * 16 + fr] = (w > 0 ? sl[qb * 16 + fr] : 0.f) + l_run[qb];
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < nq * D; i += 256) {
        const int q = i / D, d = i % D;
        const int seq = q / n_rep, r = q - seq * n_rep;
        const size_t idx = (((size_t)(b0 + seq) * Hq + hk * n_rep + r) * nsplit_tot + split);
        part_o[idx * D + d] = so[(size_t)q * D + d];
        if (d == 0) {
            part_ml[idx * 2] = fmaxf(fmaxf(sm[q], sm[NQ + q]), fmaxf(sm[2 * NQ + q], sm[3 * NQ + q]));
            part_ml[idx * 2 + 1] = sl[q];
        }
    }
}

// One launch for a decode step's group attention: blocks [0, n_prefix) run the shared-prefix role for (split, kv head,
// sub-group), the rest the own-keys role for (split, kv head, row) -- both only write partials, the combine follows.
template <int NQB>
__global__ __launch_bounds__(256) void attn_decode_group_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                                const bf16_t* __restrict__ Vc, float* __restrict__ part_o,
                                                                float* __restrict__ part_ml, const int* __restrict__ k_lo_arr,
                                                                int P, int G, int ctx, int Hq, int Hkv, int n_rep, long k_hs,
                                                                long k_bs, float scale_log2e, int nsplit_prefix, int nsplit_own,
                                                                int n_prefix, PrefixRef pf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nsplit_tot = nsplit_prefix + nsplit_own;
    int L = blockIdx.x;
    if (L < n_prefix) {
        attn_decode_group_body<NQB>(Q, Kc, Vc, part_o, part_ml, k_lo_arr, P, G, Hq, Hkv, n_rep, k_hs, k_bs, scale_log2e, nsplit_tot,
                                    L % nsplit_prefix, nsplit_prefix, (L / nsplit_prefix) % Hkv, L / (nsplit_prefix * Hkv), pf);
    } else {
        L -= n_prefix;
        // own keys P..ctx-1 (with a shared prompt entry the row's cache starts at logical key P: the body offsets it by its P)
        attn_decode_mfma_body<false>(Q, Kc, Vc, part_o, part_ml, k_lo_arr, ctx, Hq, Hkv, n_rep, k_hs, k_bs, scale_log2e, P,
                                     nsplit_tot, nsplit_prefix, 1, pf.K ? P : 0, L % nsplit_own, nsplit_own, (L / nsplit_own) % Hkv,
                                     L / (nsplit_own * Hkv), smem, AttnHandoff{}, pf);
    }
}

// Merge the context splits: block per (b, head); split weights are computed once into LDS, then every thread sums its
// output dimension over the splits with independent loads.
template <int D>
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml,
                                                                  bf16_t* __restrict__ out, int nsplit) {
    __shared__ float sw[64];
    __shared__ float s_inv;
    const int bh = blockIdx.x;  // b*Hq + h
    const int t = threadIdx.x;
    if (t < 64) combine_weights<false>(part_ml, bh, nsplit, sw, &s_inv, t);
    __syncthreads();
    if (t >= D) return;
    const float acc = combine_dim<D, false>(part_o, bh, nsplit, sw, t);
    out[(size_t)bh * D + t] = f2bf(acc * s_inv);
}

}  // namespace

extern "C" int o3v_attn_tiles_prefix(const void* Q, const void* K, const void* V, const void* Kpre, const void* Vpre, long p_hs,
                                     long p_bs, int prefix_len, int rows_per_prefix, void* O, const int* tiles, int n_tiles,
                                     int rows_per_tile, int Hq, int n_rep, int D, long q_ts, long k_ts, long k_hs, long k_bs,
                                     long v_ts, long v_hs, long v_bs, long o_ts, float scale, hipStream_t stream);
extern "C" int o3v_attn_decode_group_prefix(const void* Q, const void* Kc, const void* Vc, const void* Kpre, const void* Vpre,
                                            int prefix_cap, int rows_per_prompt, void* out, float* part_o, float* part_ml,
                                            const int* k_lo, int B, int G, int Hq, int Hkv, int D, int prefix_len, int ctx, int Tmax,
                                            int nsplit_prefix, float scale, hipStream_t stream);

extern "C" int o3v_attn_tiles(const void* Q, const void* K, const void* V, void* O, const int* tiles, int n_tiles,
                              int rows_per_tile, int Hq, int n_rep, int D, long q_ts, long k_ts, long k_hs, long k_bs,
                              long v_ts, long v_hs, long v_bs, long o_ts, float scale, hipStream_t stream) {
    return o3v_attn_tiles_prefix(Q, K, V, nullptr, nullptr, 0, 0, 0, 1, O, tiles, n_tiles, rows_per_tile, Hq, n_rep, D, q_ts, k_ts, k_hs,
                                 k_bs, v_ts, v_hs, v_bs, o_ts, scale, stream);
}

// Kpre/Vpre != NULL: keys 0..prefix_len-1 of a tile are read from the shared prompt entry of its batch row
// ([batch / rows_per_prefix] at stride p_bs, kv head at stride p_hs, token stride k_ts / v_ts), keys from prefix_len on from
// K / V, whose rows then start at logical key prefix_len (a completion's own tokens behind a prompt kept once).
extern "C" int o3v_attn_tiles_prefix(const void* Q, const void* K, const void* V, const void* Kpre, const void* Vpre, long p_hs,
                                     long p_bs, int prefix_len, int rows_per_prefix, void* O, const int* tiles, int n_tiles,
                                     int rows_per_tile, int Hq, int n_rep, int D, long q_ts, long k_ts, long k_hs, long k_bs,
                                     long v_ts, long v_hs, long v_bs, long o_ts, float scale, hipStream_t stream) {
    if (rows_per_tile != 64 && rows_per_tile != 128) return O3V_ERR_ARG;
    if (!Q || !K || !V || !O || !tiles || n_tiles < 0 || Hq <= 0 || n_rep <= 0 || (Hq % n_rep)) return O3V_ERR_ARG;
    if ((Kpre == nullptr) != (Vpre == nullptr) || (Kpre && (prefix_len <= 0 || rows_per_prefix <= 0))) return O3V_ERR_ARG;
    if ((q_ts & 7) || (k_ts & 7) || (v_ts & 7) || (k_hs & 7) || (v_hs & 7) || (o_ts & 3) || (p_hs & 7)) return O3V_ERR_SHAPE;
    if (n_tiles == 0) return O3V_OK;
    const float sl2 = scale * 1.4426950408889634f;
    const PrefixRef pf{(const bf16_t*)Kpre, (const bf16_t*)Vpre, p_hs, p_bs, Kpre ? rows_per_prefix : 1};
    const bool pfx = Kpre != nullptr;
    dim3 grid(n_tiles, Hq), block(256);
#define O3V_AT2(DD, RQ, PF)                                                                                           \
    O3V_KLAUNCH((attn_tiles_kernel<DD, RQ, PF>), grid, block, (DD == 128 ? 4 * 64 * 256 : AttnCfg<DD>::K_BYTES + AttnCfg<DD>::V_BYTES), stream, \
                       (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, (const TileDesc*)tiles, q_ts, \
                       k_ts, k_hs, k_bs, v_ts, v_hs, v_bs, o_ts, n_rep, sl2, pf, prefix_len)
#define O3V_AT1(DD, RQ)          \
    do {                         \
        if (pfx)                 \
            O3V_AT2(DD, RQ, true);  \
        else                     \
            O3V_AT2(DD, RQ, false); \
    } while (0)
#define O3V_AT(DD)                       \
    do {                                 \
        if (rows_per_tile == 128)        \
            O3V_AT1(DD, 2);              \
        else                             \
            O3V_AT1(DD, 1);              \
    } while (0)
    switch (D) {
        case 32: O3V_AT(32); break;
        case 64: O3V_AT(64); break;
        case 80: O3V_AT(80); break;
        case 128: O3V_AT(128); break;
        default: return O3V_ERR_SHAPE;
    }
#undef O3V_AT
#undef O3V_AT1
#undef O3V_AT2
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_attn_decode(const void* Q, const void* Kc, const void* Vc, void* out, float* part_o, float* part_ml,
                               const int* k_lo, int B, int Hq, int Hkv, int D, int ctx, int Tmax, int nsplit, float scale,
                               hipStream_t stream) {
    if (!Q || !Kc || !Vc || !out || !part_o || !part_ml || B < 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || ctx <= 0 ||
        ctx > Tmax || nsplit == 0 || nsplit > 64 || nsplit < -64)
        return O3V_ERR_ARG;
    const int n_rep = Hq / Hkv;
    if (n_rep > NREP_MAX) return O3V_ERR_SHAPE;
    if (B == 0) return O3V_OK;
    const float sl2 = scale * 1.4426950408889634f;
    const long k_hs = (long)Tmax * D, k_bs = (long)Hkv * Tmax * D;
    const bool use_valu = nsplit < 0 ? true : false;  // negative nsplit selects the scalar kernel (tests / A-B)
    if (nsplit < 0) nsplit = -nsplit;
    dim3 grid(nsplit, Hkv, B), block(256);
#define O3V_AD(DD)                                                                                                     \
    O3V_KLAUNCH((attn_decode_kernel<DD>), grid, block, 0, stream, (const bf16_t*)Q, (const bf16_t*)Kc,           \
                       (const bf16_t*)Vc, part_o, part_ml, k_lo, ctx, Hq, Hkv, n_rep, k_hs, k_bs, sl2);                \
    O3V_KLAUNCH((attn_decode_combine_kernel<DD>), dim3(B* Hq), dim3(128), 0, stream, part_o,      \
                       part_ml, (bf16_t*)out, nsplit)
    switch (D) {
        case 32: O3V_AD(32); break;
        case 64: O3V_AD(64); break;
        case 128:
            if (use_valu) {
                O3V_AD(128);
            } else {
                O3V_KLAUNCH(attn_decode_mfma_kernel, grid, block, 4 * 32 * 288, stream, (const bf16_t*)Q, (const bf16_t*)Kc,
                            (const bf16_t*)Vc, part_o, part_ml, k_lo, ctx, Hq, Hkv, n_rep, k_hs, k_bs, sl2, 0, nsplit, 0, 1, 0,
                            PrefixRef{nullptr, nullptr, 0, 0, 1});
                O3V_KLAUNCH((attn_decode_combine_kernel<128>), dim3(B * Hq), dim3(128), 0, stream, part_o, part_ml,
                            (bf16_t*)out, nsplit);
            }
            break;
        default: return O3V_ERR_SHAPE;
    }
#undef O3V_AD
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_attn_decode_group(const void* Q, const void* Kc, const void* Vc, void* out, float* part_o, float* part_ml,
                                     const int* k_lo, int B, int G, int Hq, int Hkv, int D, int prefix_len, int ctx, int Tmax,
                                     int nsplit_prefix, float scale, hipStream_t stream) {
    return o3v_attn_decode_group_prefix(Q, Kc, Vc, nullptr, nullptr, 0, 1, out, part_o, part_ml, k_lo, B, G, Hq, Hkv, D, prefix_len, ctx,
                                        Tmax, nsplit_prefix, scale, stream);
}

// Kpre/Vpre != NULL: the prompts' K/V are kept once, [B / rows_per_prompt][Hkv][prefix_cap][D]; Kc/Vc [B][Hkv][Tmax][D] then
// hold only each row's generated tokens (logical key prefix_len + j in slot j).  G (the sub-group one workgroup serves) must
// divide rows_per_prompt.
extern "C" int o3v_attn_decode_group_prefix(const void* Q, const void* Kc, const void* Vc, const void* Kpre, const void* Vpre,
                                            int prefix_cap, int rows_per_prompt, void* out, float* part_o, float* part_ml,
                                            const int* k_lo, int B, int G, int Hq, int Hkv, int D, int prefix_len, int ctx, int Tmax,
                                            int nsplit_prefix, float scale, hipStream_t stream) {
    if (!Q || !Kc || !Vc || !out || !part_o || !part_ml || B <= 0 || G <= 1 || (B % G) || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) ||
        prefix_len <= 0 || ctx <= prefix_len || nsplit_prefix == 0 || (Kpre == nullptr) != (Vpre == nullptr))
        return O3V_ERR_ARG;
    if (Kpre ? (ctx - prefix_len > Tmax || prefix_len > prefix_cap || rows_per_prompt <= 0 || (rows_per_prompt % G) || (B % rows_per_prompt))
             : ctx > Tmax)
        return O3V_ERR_ARG;
    const PrefixRef pf{(const bf16_t*)Kpre, (const bf16_t*)Vpre, (long)prefix_cap * D, (long)Hkv * prefix_cap * D,
                       Kpre ? rows_per_prompt : 1};
    const int n_rep = Hq / Hkv;
    if (D != 128) return O3V_ERR_SHAPE;
    if (nsplit_prefix > 0 && G * n_rep > 64) return O3V_ERR_SHAPE;  // the one-pass kernel holds the group in 64 MFMA columns
    if (nsplit_prefix < 0) {
        // shared-read form: the per-row kernel, with every row of a group reading the prefix from the group's first row
        const int ns = -nsplit_prefix;
        if (ns > 64) return O3V_ERR_ARG;
        O3V_KLAUNCH(attn_decode_mfma_kernel, dim3(ns, Hkv, B), dim3(256), 4 * 32 * 288, stream, (const bf16_t*)Q, (const bf16_t*)Kc,
                    (const bf16_t*)Vc, part_o, part_ml, k_lo, ctx, Hq, Hkv, n_rep, (long)Tmax * D, (long)Hkv * Tmax * D,
                    scale * 1.4426950408889634f, 0, ns, 0, G, prefix_len, pf);
        O3V_KLAUNCH((attn_decode_combine_kernel<128>), dim3(B * Hq), dim3(128), 0, stream, part_o, part_ml, (bf16_t*)out, ns);
        O3V_CHECK_LAUNCH();
        return O3V_OK;
    }
    const int nsplit_own = (ctx - prefix_len + 127) / 128;
    const int nsplit_tot = nsplit_prefix + nsplit_own;
    if (nsplit_tot > 64) return O3V_ERR_ARG;
    const float sl2 = scale * 1.4426950408889634f;
    const long k_hs = (long)Tmax * D, k_bs = (long)Hkv * Tmax * D;
    const int nqb = (G * n_rep + 15) / 16;
    const int n_prefix = nsplit_prefix * Hkv * (B / G), n_own = nsplit_own * Hkv * B;
    dim3 grid(n_prefix + n_own), block(256);
#define O3V_AG(NQB)                                                                                                         \
    O3V_KLAUNCH((attn_decode_group_kernel<NQB>), grid, block,                                                                \
                (size_t)((NQB * 16 * 272 + 4 * 32 * 288) > (NQB * 16 * 128 * 4 + 5 * NQB * 16 * 4)                         \
                             ? (NQB * 16 * 272 + 4 * 32 * 288)                                                              \
                             : (NQB * 16 * 128 * 4 + 5 * NQB * 16 * 4)),                                                    \
                stream, (const bf16_t*)Q, (const bf16_t*)Kc, (const bf16_t*)Vc, part_o, part_ml, k_lo, prefix_len, G, ctx, Hq, \
                Hkv, n_rep, k_hs, k_bs, sl2, nsplit_prefix, nsplit_own, n_prefix, pf)
    if (nqb == 1)
        O3V_AG(1);
    else if (nqb == 2)
        O3V_AG(2);
    else if (nqb == 3)
        O3V_AG(3);
    else
        O3V_AG(4);
#undef O3V_AG
    O3V_KLAUNCH((attn_decode_combine_kernel<128>), dim3(B * Hq), dim3(128), 0, stream, part_o, part_ml, (bf16_t*)out,
                nsplit_tot);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
