// Decode attention on the matrix cores (head_dim 128) and the merge of its context splits: device bodies shared by the
// stand-alone kernels of o3v_attn.hip and the role-fused decode launch of o3v_fused.hip -- both instantiate THIS code, so
// their results are bit-identical.  Arithmetic: TF:modeling_qwen2_5_vl.py:186-208 (softmax in fp32, P rounded to bf16
// before P.V as the bf16 reference does), GQA via TF:174-183.
#pragma once
#include "o3v_common.h"
#include "o3v_handoff.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int NREP_MAX = 8;

// ------------------------------------------------------------------------------------------------
// Merge of the context splits of one (b, head).  combine_weights: ONE wave; sw[64] / s_inv in LDS.
// ------------------------------------------------------------------------------------------------
// the math of one (b, head): mv / lv = this lane's split (m, l) or (-1e30, 0) past nsplit
__device__ __forceinline__ void combine_weights_math(float mv, float lv, int nsplit, float* sw, float* s_inv, int lane) {
    const float mn = wave_max(mv);
    const float w = lane < nsplit ? __builtin_amdgcn_exp2f(mv - mn) : 0.f;
    const float lt = wave_sum(lv * w);
    sw[lane] = w;
    if (lane == 0) *s_inv = lt > 0.f ? 1.0f / lt : 0.f;
}
template <bool SC1>
__device__ __forceinline__ void combine_weights(const float* part_ml, size_t bh, int nsplit, float* sw, float* s_inv, int lane) {
    const float mv = lane < nsplit ? ldf<SC1>(part_ml + (bh * nsplit + lane) * 2) : -1e30f;
    const float lv = lane < nsplit ? ldf<SC1>(part_ml + (bh * nsplit + lane) * 2 + 1) : 0.f;
    combine_weights_math(mv, lv, nsplit, sw, s_inv, lane);
}

// output dimension d of (b, head): sum over the splits, 32 independent loads per round (a clamped index, weight 0 past the end)
template <int D, bool SC1>
__device__ __forceinline__ float combine_dim(const float* part_o, size_t bh, int nsplit, const float* sw, int d) {
    const float* po = part_o + bh * nsplit * D + d;
    float acc = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 32) {
        float pv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int s = s0 + u;
            pv[u] = ldf<SC1>(po + (size_t)(s < nsplit ? s : nsplit - 1) * D);
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = fmaf(pv[u], sw[(s0 + u) & 63], acc);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------
// Decode attention on the matrix cores (head_dim 128).  Same swapped-QK^T scheme as attn_tiles_kernel with the
// n_rep query heads of a kv head as the 16 "query rows" (GQA: K/V bytes are read once for the whole group):
// each wave walks its own key range in 32-key tiles -- K fragments straight from global memory to VGPRs (a key row
// is consumed by the 4 k-steps of one lane quad), V staged by the wave into its private LDS slice and read back
// transposed (ds_read_b64_tr_b16).  ~32 MFMA + ~100 VALU per 32 keys instead of ~1600 VALU in the scalar kernel.
//
// FUSED (o3v_fused.hip, one sequence, no shared prefix): q and the K/V row of the newest key (ctx-1) are produced by
// other workgroups of the same launch.  The wave requests its first K/V tile at once, then the workgroup waits for the
// q/k/v counter of its kv head, loads q and re-reads the newest key's row with sc1 loads, and goes on as the stand-alone
// kernel does; the partials are stored write-through, the last split of a kv head (ticket) tells the head's workgroups, and
// each of them merges a slice of the head's n_rep x D outputs (the chains of combine_dim); the last slice (second ticket)
// tells every o_proj workgroup's mailbox.
// ------------------------------------------------------------------------------------------------
struct AttnHandoff {
    uint32_t* mailbox;    // this workgroup's line: word 0 <- epoch once q and the new K/V row of its kv head are in memory,
                          // word 1 <- epoch once every split of the head has stored its partials
    uint32_t* head_boxes; // mailbox lines of the nsplit workgroups of this kv head
    uint32_t* att_ticket; // ticket counters of this kv head: partials stored / slice merged
    uint32_t* att_ticket2;
    uint32_t* o_boxes;    // mailbox lines of the o_proj workgroups (word hk <- epoch: kv head hk's output is in memory)
    int n_o_boxes;
    uint32_t epoch;
    uint32_t* tmo;        // sticky time-out word
    bf16_t* att_out;      // [B, Hq, D] attention output (input of o_proj)
#ifdef O3V_STAMPS
    unsigned long long* stamp;  // diagnostic build (tools/probes/probe_fused.py): this workgroup's {start, wait over, partials out, end, all partials known, slice stored}
    bool no_prefetch;           // ablation: request the first K/V tile only after the wait
#endif
};

#ifdef O3V_STAMPS
#define O3V_STAMP(p, i)                                                            \
    do {                                                                           \
        if (threadIdx.x == 0 && (p)) (p)[i] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define O3V_STAMP(p, i) \
    do {                \
    } while (0)
#endif

// A prompt's K/V kept ONCE for the rows that share it (the G completions of a prompt): [prompts][Hkv][cap][D].  K == nullptr:
// every row's cache holds its own copy of the prompt (legacy layout).  With it, a row's own cache holds only the tokens
// generated after the prompt: logical key kr >= P lives in row kr - P of the row's cache.
struct PrefixRef {
    const bf16_t *K, *V;
    long hs, bs;  // head / prompt strides in elements
    int rows;     // decode rows per prompt: row b reads prompt b / rows
};

// Qwen3-VL (TF3:480-484) inside the one-launch block: the q/k/v role publishes the projection outputs un-normalised and
// un-rotated (q_raw, k_raw, v_raw); the q/k/v workgroup that takes a kv head's LAST ticket then finishes that head group -- its
// n_rep q heads and the new key's k row normalised + rotated, the v row copied -- and only then tells the mailboxes, so the
// attention and o_proj roles run unchanged.  Same arithmetic as o3v_qkv_norm_rope_cache (qkn_* / rope_share).
struct QkNormRef {
    const bf16_t *q_raw, *k_raw, *v_raw;  // [Hq, D], [Hkv, D], [Hkv, D]
    const bf16_t *q_norm, *k_norm;        // [D]
    const bf16_t *cosr, *sinr;            // rotary row of this token, [D]
    float eps;
};

// One head per 16-lane row in the MFMA operand layout: the four lanes fg = 0..3 hold x[ks] = dims 32 ks + 8 fg .. +7, i.e. the lane
// shares fg = (x[0], x[2]) and 4 + fg = (x[1], x[3]).  The eight share sums are added pairwise over share distance 1, 2, 4 exactly
// as o3v_qkv_norm_rope_cache adds them; then w * bf16(x * rstd) and the rotation.
__device__ __forceinline__ void qkn_norm_rope_mfma(u32x4 (&x)[4], const bf16_t* nw, const QkNormRef& qk, const int fg) {
    float s_lo = qkn_chain(x[0], x[2]), s_hi = qkn_chain(x[1], x[3]);
    s_lo += __shfl_xor(s_lo, 16, 64);
    s_hi += __shfl_xor(s_hi, 16, 64);
    s_lo += __shfl_xor(s_lo, 32, 64);
    s_hi += __shfl_xor(s_hi, 32, 64);
    const float rstd = qkn_rstd(s_lo + s_hi, 128, qk.eps);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) x[ks] = qkn_scale(x[ks], *reinterpret_cast<const u32x4*>(nw + ks * 32 + fg * 8), rstd);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        u32x4 ol, oh;
        rope_share(x[p], x[p + 2], *reinterpret_cast<const u32x4*>(qk.cosr + p * 32 + fg * 8),
                   *reinterpret_cast<const u32x4*>(qk.sinr + p * 32 + fg * 8), ol, oh);
        x[p] = ol;
        x[p + 2] = oh;
    }
}

// Head group g finished by ONE wave (all 64 lanes active): rows fr < n_rep of the lane grid are the group's q heads, then row 0
// is its k head; v is copied by lanes 0..15.  Everything is read past L1, written through and drained before the return.
__device__ __forceinline__ void qkn_finish_head_group(const QkNormRef& qk, const int g, const int n_rep, bf16_t* q_out, bf16_t* krow,
                                                      bf16_t* vrow) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const __amdgpu_buffer_rsrc_t qr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(qk.q_raw + (size_t)g * n_rep * 128), 0, n_rep * 128 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)(qk.k_raw + (size_t)g * 128), 0, 128 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)(qk.v_raw + (size_t)g * 128), 0, 128 * 2, 0x00020000);
    u32x4 xq[4], xk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // rows past the descriptors read zeros
        xq[ks] = load16_sc1(qr, (uint32_t)(fr * 128 + ks * 32 + fg * 8) * 2);
        xk[ks] = load16_sc1(kr, fr == 0 ? (uint32_t)(ks * 32 + fg * 8) * 2 : O3V_OOB);
    }
    const u32x4 tv = load16_sc1(vr, lane < 16 ? (uint32_t)lane * 16 : O3V_OOB);
    qkn_norm_rope_mfma(xq, qk.q_norm, qk, fg);
    qkn_norm_rope_mfma(xk, qk.k_norm, qk, fg);
    if (fr < n_rep) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(q_out + ((size_t)g * n_rep + fr) * 128);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) __hip_atomic_store(dst + (ks * 32 + fg * 8) / 2 + j, xq[ks][j], O3V_RLX_AGENT);
    }
    if (fr == 0) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(krow);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) __hip_atomic_store(dst + (ks * 32 + fg * 8) / 2 + j, xk[ks][j], O3V_RLX_AGENT);
    }
    if (lane < 16) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(vrow) + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) __hip_atomic_store(dst + j, tv[j], O3V_RLX_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FUSED>
__device__ __forceinline__ void attn_decode_mfma_body(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                      const bf16_t* __restrict__ Vc, float* __restrict__ part_o,
                                                      float* __restrict__ part_ml, const int* __restrict__ k_lo_arr, int ctx,
                                                      int Hq, int Hkv, int n_rep, long k_hs, long k_bs, float scale_log2e,
                                                      int kbeg, int nsplit_tot, int split_off, int G, int P, const int split,
                                                      const int nsplit, const int hk, const int b, char* smem,
                                                      const AttnHandoff& ho, const PrefixRef pf = PrefixRef{nullptr, nullptr, 0, 0, 1}) {
    // keys kbeg..ctx-1 of every row; partials go to slots split_off.. of the row's nsplit_tot (the slots before
    // split_off belong to attn_decode_group_kernel when the rows of a group share their first kbeg keys)
    constexpr int D = 128, KT = 32, VSTRIDE = 288, V_BYTES = KT * VSTRIDE;  // 9216 B per wave; smem: 4 x V slice, reused for the merge
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int k_lo = k_lo_arr ? k_lo_arr[b] : 0;
    char* Vl = smem + wave * V_BYTES;

    int chunk = (ctx - kbeg + nsplit - 1) / nsplit;
    chunk = (chunk + 4 * KT - 1) / (4 * KT) * (4 * KT);  // whole 32-key tiles per wave
    const int per_wave = chunk >> 2;
    const int kw0 = kbeg + split * chunk + wave * per_wave;
    int kw1 = kw0 + per_wave;
    kw1 = kw1 < ctx ? kw1 : ctx;

    const bf16_t* Kb = Kc + (size_t)b * k_bs + (size_t)hk * k_hs;
    const bf16_t* Vb = Vc + (size_t)b * k_bs + (size_t)hk * k_hs;
    // G > 1: keys below P are read from the cache row of the group's first sequence (identical bytes for the G rows of a
    // group, whose blocks share an XCD and hence an L2: one HBM read serves the group)
    const size_t lead = G > 1 ? (size_t)(b - (b / G) * G) * k_bs : 0;
    const bf16_t* Kl = Kb - lead;
    const bf16_t* Vl0 = Vb - lead;
    if (pf.K) {  // shared prompt K/V: keys below P from the prompt's entry, the row's own cache starts at logical key P
        const size_t po = (size_t)(b / pf.rows) * pf.bs + (size_t)hk * pf.hs;
        Kl = pf.K + po;
        Vl0 = pf.V + po;
        Kb -= (size_t)P * D;
        Vb -= (size_t)P * D;
    }

    bf16x8 kf[2][4];
    u32x4 vreg[8];
    // one 32-key tile: K fragments (A operand: lane (key = fr, quad fg) loads 16 B of row key0 + kb*16 + fr) and the V
    // rows (32 rows x 16 chunks of 16 B, 8 per lane)
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            int kr = key0 + kb * 16 + fr;
            kr = kr < ctx ? kr : ctx - 1;
            const bf16_t* Kr = (kr < P ? Kl : Kb) + (size_t)kr * D;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(Kr + ks * 32 + fg * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane, row = c >> 4, ch = c & 15;
            int kr = key0 + row;
            kr = kr < ctx ? kr : ctx - 1;
            vreg[i] = *reinterpret_cast<const u32x4*>((kr < P ? Vl0 : Vb) + (size_t)kr * D + ch * 8);
        }
    };
    // FUSED: every fragment that belongs to (or is clamped onto) the newest key ctx-1 is read again, past L1
    auto patch_newest = [&](int key0, __amdgpu_buffer_rsrc_t krs, __amdgpu_buffer_rsrc_t vrs) {
        const uint32_t row_off = (uint32_t)(ctx - 1) * (D * 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bool hit = key0 + kb * 16 + fr >= ctx - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 t = load16_sc1(krs, hit ? row_off + (uint32_t)(ks * 32 + fg * 8) * 2 : O3V_OOB);
                if (hit) kf[kb][ks] = __builtin_bit_cast(bf16x8, t);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane, row = c >> 4, ch = c & 15;
            const bool hit = key0 + row >= ctx - 1;
            const u32x4 t = load16_sc1(vrs, hit ? row_off + (uint32_t)ch * 16 : O3V_OOB);
            if (hit) vreg[i] = t;
        }
    };

    bf16x8 qf[4];
    __amdgpu_buffer_rsrc_t krs, vrs;
    if (FUSED) {
#ifdef O3V_STAMPS
        if (!ho.no_prefetch)
#endif
            if (kw0 < kw1) load_tile(kw0);  // K/V of the first tile travel while q is still being computed
        if (wave == 0) spin_until<2>(ho.mailbox, 1, ho.epoch, ho.tmo, 0x100u + (uint32_t)hk);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __syncthreads();
        O3V_STAMP(ho.stamp, 1);
        const __amdgpu_buffer_rsrc_t qrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(Q + ((size_t)b * Hq + (size_t)hk * n_rep) * D), 0, n_rep * D * 2, 0x00020000);
        krs = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, ctx * D * 2, 0x00020000);
        vrs = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, ctx * D * 2, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)  // rows fr >= n_rep lie past the descriptor: zeros
            qf[ks] = __builtin_bit_cast(bf16x8, load16_sc1(qrs, (uint32_t)(fr * D + ks * 32 + fg * 8) * 2));
#ifdef O3V_STAMPS
        if (ho.no_prefetch && kw0 < kw1) load_tile(kw0);
#endif
        if (kw0 < kw1 && kw0 + KT > ctx - 1) patch_newest(kw0, krs, vrs);
    } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (fr < n_rep)
                qf[ks] = *reinterpret_cast<const bf16x8*>(Q + ((size_t)b * Hq + hk * n_rep + fr) * D + ks * 32 + fg * 8);
            else
                qf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    float m_run = -1e30f, l_run = 0.f;
    f32x4 o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tq = fr >> 2, tp = fr & 3;

    for (int key0 = kw0; key0 < kw1; key0 += KT) {
        if (!FUSED || key0 != kw0) {
            load_tile(key0);
            if (FUSED && key0 + KT > ctx - 1) patch_newest(key0, krs, vrs);
        }
        // ---- V tile -> this wave's LDS slice
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 64 + lane, row = c >> 4, ch = c & 15;
            *reinterpret_cast<u32x4*>(Vl + row * VSTRIDE + ch * 16) = vreg[i];
        }
        // ---- S^T = K . Q^T
        f32x4 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
        }
        float mx = -1e30f;
        bool ok[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = key0 + kb * 16 + fg * 4 + r;
                ok[kb][r] = (j < kw1) && (j >= k_lo);
                const float sv = ok[kb][r] ? s[kb][r] * scale_log2e : -1e30f;
                s[kb][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = ok[kb][r] ? __builtin_amdgcn_exp2f(s[kb][r] - m_new) : 0.f;
                const bf16_t pq = f2bf(p);
                psum += bf2f(pq);
                pb[kb * 4 + r] = (short)pq;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= alpha;
        // ---- O^T += V^T . P^T   (LDS ops of one wave execute in order: the tr-reads see the stores above)
        const char* r0 = Vl + (fg * 4 + tq) * VSTRIDE + tp * 8;
        const char* r1 = r0 + 16 * VSTRIDE;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r0 + db * 32));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(r1 + db * 32));
            const bf16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, o[db], 0, 0, 0);
        }
    }
    // ---- merge the 4 waves: so[w][q][d], sm/sl[w][q] in LDS (the V slices are dead)
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    __syncthreads();
    float* so = reinterpret_cast<float*>(smem);           // [4][16][128] = 32 KiB
    float* sm = so + 4 * 16 * D;                           // [4][16]
    float* sl = sm + 64;                                   // [4][16]
#pragma unroll
    for (int db = 0; db < 8; ++db)
        *reinterpret_cast<f32x4*>(so + ((size_t)(wave * 16 + fr) * D + db * 16 + fg * 4)) = o[db];
    if (fg == 0) {
        sm[wave * 16 + fr] = m_run;
        sl[wave * 16 + fr] = l_tot;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_rep * D; i += 256) {
        const int q = i / D, d = i % D;
        const float mn = fmaxf(fmaxf(sm[q], sm[16 + q]), fmaxf(sm[32 + q], sm[48 + q]));
        float acc = 0.f, lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __builtin_amdgcn_exp2f(sm[w * 16 + q] - mn);
            acc += so[(size_t)(w * 16 + q) * D + d] * sc;
            lt += sl[w * 16 + q] * sc;
        }
        const size_t idx = (((size_t)b * Hq + hk * n_rep + q) * nsplit_tot + split_off + split);
        stf<FUSED>(part_o + idx * D + d, acc);
        if (d == 0) {
            stf<FUSED>(part_ml + idx * 2, mn);
            stf<FUSED>(part_ml + idx * 2 + 1, lt);
        }
    }
    if (!FUSED) return;

    // ---- merge of the context splits, spread over the kv head's nsplit workgroups.  One CU moves ~25 GB/s, so a single
    // workgroup reading all n_rep x nsplit x D partials (143 KB at 7B / 4.6k keys) needs 5+ us; a slice of
    // ceil(n_rep*D / nsplit) output elements per workgroup needs one round trip.  Ticket 1: the last split tells every
    // workgroup of the head (mailbox word 1) that all partials are in memory.
    float* sw = reinterpret_cast<float*>(smem + 33536);           // [NREP_MAX][64], behind so / sm / sl
    float* sinv = sw + NREP_MAX * 64;                             // [NREP_MAX]
    float* pvt = reinterpret_cast<float*>(smem);                  // [slice][nsplit | 1] partials of this slice (so is dead)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its sc1 stores
    __syncthreads();
    O3V_STAMP(ho.stamp, 2);
    if (wave == 0) {
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(ho.att_ticket, 1u, O3V_RLX_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == ho.epoch * (uint32_t)nsplit - 1u) notify_mailboxes(ho.head_boxes, nsplit, 1, ho.epoch);
        spin_until<2>(ho.mailbox + 1, 1, ho.epoch, ho.tmo, 0x180u + (uint32_t)hk);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    O3V_STAMP(ho.stamp, 4);
    const size_t bh0 = (size_t)b * Hq + (size_t)hk * n_rep;
    const int n_out = n_rep * D, per = (n_out + nsplit - 1) / nsplit;
    const int o0 = split * per, o1 = (o0 + per) < n_out ? (o0 + per) : n_out, cnt = o1 - o0;
    if (cnt > 0) {
        const int q_first = o0 / D, q_last = (o1 - 1) / D;
        // partials of the slice: element fastest across lanes (runs of consecutive d), all requested before any is used
        constexpr int NLD = (NREP_MAX * D + 64 + 255) / 256;  // loads per thread for the largest slice x splits
        float pr[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = threadIdx.x + k * 256;
            const int ii = idx % cnt, s2 = idx / cnt;
            const int e = o0 + ii;
            pr[k] = s2 < nsplit_tot ? ldf<true>(part_o + ((bh0 + e / D) * nsplit_tot + s2) * D + (e % D)) : 0.f;
        }
        for (int q = q_first + wave; q <= q_last; q += 4) combine_weights<true>(part_ml, bh0 + q, nsplit_tot, sw + q * 64, sinv + q, lane);
        const int ldp = nsplit_tot | 1;  // odd row stride: lanes on consecutive elements hit distinct banks
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = threadIdx.x + k * 256;
            const int ii = idx % cnt, s2 = idx / cnt;
            if (s2 < nsplit_tot) pvt[ii * ldp + s2] = pr[k];
        }
        __syncthreads();
        // per element the fmaf chain of combine_dim over the splits in order (its clamped tail terms have weight zero);
        // eight LDS reads in flight per step of the chain
        for (int ii = threadIdx.x; ii < cnt; ii += 256) {  // cnt <= 256 unless there are fewer splits than query heads
            const int e = o0 + ii, q = e / D;
            const float* pp = pvt + ii * ldp;
            const float* ww = sw + q * 64;
            float acc = 0.f;
            for (int s0 = 0; s0 < nsplit_tot; s0 += 8) {
                float pv8[8], w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s2 = s0 + u < nsplit_tot ? s0 + u : nsplit_tot - 1;
                    pv8[u] = pp[s2];
                    w8[u] = s0 + u < nsplit_tot ? ww[s2] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < nsplit_tot) acc = fmaf(pv8[u], w8[u], acc);
            }
            __hip_atomic_store(ho.att_out + bh0 * D + e, f2bf(acc * sinv[q]), O3V_RLX_AGENT);
        }
    }
    // ---- ticket 2: the last slice tells every o_proj workgroup that this kv head's output is in memory
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    O3V_STAMP(ho.stamp, 5);
    if (wave == 0) {
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(ho.att_ticket2, 1u, O3V_RLX_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == ho.epoch * (uint32_t)nsplit - 1u) notify_mailboxes(ho.o_boxes, ho.n_o_boxes, hk, ho.epoch);
    }
    O3V_STAMP(ho.stamp, 3);
}

}  // namespace
