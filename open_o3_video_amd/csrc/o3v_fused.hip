// One launch for the attention half of a decode layer at batch 1 (TF:modeling_qwen2_5_vl.py:692-757, first half):
//
//     RMSNorm + q/k/v projection + bias + M-RoPE + cache append  ->  attention over the cache  ->  merge of the context
//     splits  ->  o_proj + residual
//
// As four launches this chain moves 68 MB (7B) in ~29 us: every kernel is latency-bound (33 / 9.6 / 0.6 / 25.7 MB) and the
// HBM idles across each boundary.  Here the three stages are ROLES of the workgroups of one grid, in block-index order
//   [0, nb_qkv)            the weight-streaming GEMV of o3v_gemv_body.h (q/k/v rows, outputs stored write-through)
//   [nb_qkv, +nb_attn)     attn_decode_mfma_body<FUSED>: requests its K/V tile at once, waits for its kv head's q/k/v
//                          counter, computes; the last split of a kv head merges the splits and publishes the output
//   [.., +nb_o)            o_proj: a wave requests ALL weight bytes of its two rows (56 VGPRs at K = 3584) at once, waits
//                          for the attention output, then multiplies
// so the weights of o_proj and the K/V cache stream while the q/k/v projection runs, and the hand-offs (counters in HBM,
// protocol in o3v_attn_decode_body.h) replace two kernel boundaries and the combine launch.  Every role instantiates the
// same device code as the stand-alone kernels: results are bit-identical to the four-launch path.
//
// Progress without any assumption on dispatch order: only the attention and o_proj roles wait, each only on roles that
// never wait on them (qkv <- attn <- o); the launcher refuses the fused form unless nb_attn + nb_o is smaller than the
// number of workgroups the chip holds at once (occupancy query x CUs), so waiting workgroups can never fill every slot
// and the q/k/v workgroups always find room.  Spins are bounded and the give-up is sticky.
#include <hip/hip_runtime.h>

#include "o3v_attn_decode_body.h"
#include "o3v_common.h"
#include "o3v_gemv_body.h"

namespace {

// sync buffer (32-bit words; zeroed once per generate call, see the protocol in o3v_attn_decode_body.h)
constexpr int SYNC_QKV = 0;                            // [8] ticket lines: kv head g <- workgroups of its q heads, k head, v head
constexpr int SYNC_ATT = 8 * O3V_SYNC_STRIDE;          // [8] ticket lines: kv head g <- its context splits
constexpr int SYNC_TMO = 16 * O3V_SYNC_STRIDE;         // sticky time-out word (byte 2048 = O3V_SYNC_TMO_BYTE)
constexpr int SYNC_ATT2 = 24 * O3V_SYNC_STRIDE;        // [8] ticket lines: kv head g <- its merged slices
constexpr int SYNC_BOX_ATT = 32 * O3V_SYNC_STRIDE;     // [512] mailbox lines of the attention workgroups (kv head, split)
constexpr int SYNC_BOX_O = (32 + 512) * O3V_SYNC_STRIDE;  // [1024] mailbox lines of the o_proj workgroups, word hk per kv head
constexpr int SYNC_WORDS = (32 + 512 + 1024) * O3V_SYNC_STRIDE;
constexpr int ATTN_LDS = 4 * 32 * 288;                 // attn_decode_mfma_body: 4 V slices (merge + combine scratch inside)

struct FusedArgs {
    const bf16_t *x, *ln_w, *qkv_w, *qkv_b, *o_w;   // qkv_w / o_w: bf16 rows, or fp8 rows (WB = 1) with the scales below
    const float *qkv_s, *o_s;
    bf16_t *att, *xout;
    float* part_o;
    float* part_ml;
    const int* k_lo;
    uint32_t* sync;
    uint32_t epoch;
    RopeArgs ra;        // destinations and rotary row of this token (true cache geometry)
    RopeArgs ra_qkv;    // what the q/k/v role writes through: == ra, or (QKN) the raw scratch with the rotation switched off
    QkNormRef qk;       // QKN only
    float eps, scale_log2e;
    int H, ctx, nsplit, nb_qkv, nb_attn, nb_o;
#ifdef O3V_STAMPS
    unsigned long long* stamps;  // [grid][8] s_memrealtime ticks (100 MHz)
    int knob;                    // ablations: 1 attention + o_proj roles exit at once, 2 o_proj role exits at once,
                                 // 4 no K/V request ahead of the wait, 8 o_proj weights requested after the wait
#endif
};
#ifdef O3V_STAMPS
#define O3V_STAMP_PTR(a) ((a).stamps ? (a).stamps + (size_t)blockIdx.x * 8 : nullptr)
#else
#define O3V_STAMP_PTR(a) nullptr
#endif

// o_proj + residual for two rows per wave; NSTEP = steps of 64 16-byte weight chunks per row (all held in registers), WB =
// bytes per weight (2 bf16, 1 fp8 + row scale).  Same accumulation order as gemv_body<1, 2, 1, EPI_RESIDUAL, false, ..., WB>:
// chunk lane + 64 i, i ascending, then the wave sum.
template <int NSTEP, int WB>
__device__ __forceinline__ void oproj_role(const FusedArgs& a, const int bid) {
    constexpr int XPC = 2 / WB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.ra.Hq * a.ra.D, N = a.H, nch = K >> (WB == 1 ? 4 : 3);
    int rows[2];
    const u32x4* wp[2];
    float e_res[2], e_scale[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        rows[r] = (bid * 4 + wave) * 2 + r;
        const int rr = rows[r] < N ? rows[r] : N - 1;
        wp[r] = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.o_w) + (size_t)rr * K * WB);
        e_res[r] = bf2f(a.x[rr]);  // residual stream: written by an earlier launch
        e_scale[r] = WB == 1 ? a.o_s[rr] : 1.0f;
    }
    u32x4 wv[NSTEP][2];
    auto load_weights = [&]() {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int c = i * 64 + lane;
            const bool in = c < nch;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                wv[i][r] = __builtin_nontemporal_load(wp[r] + (in ? c : 0));
                if (!in) wv[i][r] = (u32x4){0, 0, 0, 0};
            }
#ifdef O3V_STAMPS
            for (int p = 0; p < ((a.knob >> 5) & 7); ++p) __builtin_amdgcn_s_sleep(16);  // ablation: paced weight stream
#endif
        }
    };
    uint32_t* box = a.sync + SYNC_BOX_O + (size_t)bid * O3V_SYNC_STRIDE;
    // The weight stream starts once every q/k/v workgroup has published (words 8.. of the mailbox): requested earlier it
    // shares the CUs' memory pipelines with the q/k/v rows and delays the head of the chain by 2.4 us (measured); from here
    // on it runs beside the attention role, whose K/V tiles are already in registers.
    if (wave == 0) spin_until<8>(box + 8, a.ra.Hkv, a.epoch, a.sync + SYNC_TMO, 0x300u);
    __syncthreads();
#ifdef O3V_STAMPS
    if (!(a.knob & 8))
#endif
        load_weights();
    if (wave == 0) spin_until<8>(box, a.ra.Hkv, a.epoch, a.sync + SYNC_TMO, 0x200u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    O3V_STAMP(O3V_STAMP_PTR(a), 1);
#ifdef O3V_STAMPS
    if (a.knob & 8) load_weights();
#endif
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.att, 0, K * 2, 0x00020000);
    u32x4 xv[NSTEP][XPC];
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        const int c = i * 64 + lane;
#pragma unroll
        for (int h = 0; h < XPC; ++h)
            xv[i][h] = load16_sc1(xrs, (uint32_t)((c < nch ? c : 0) * XPC + h) * 16);  // lanes past the row meet zeroed weights
    }
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NSTEP; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (WB == 1)
                fma16_fp8(wv[i][r], xv[i][0], xv[i][XPC - 1], acc[r]);
            else
                fma8(wv[i][r], xv[i][0], acc[r]);
        }
#pragma unroll
    for (int r = 0; r < 2; ++r) acc[r] = wave_sum(acc[r]);
    if (lane != 0) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (rows[r] >= N) continue;
        float v = acc[r] * e_scale[r] + 0.f;  // no bias (TF:620); bf16 rows: scale 1
        v = rbf(v) + e_res[r];
        a.xout[rows[r]] = f2bf(v);
    }
}

// NSTEP: steps of 64 16-byte weight chunks of an o_proj row (K = Hq*D; 512 k per step in bf16, 1024 in fp8); WB: bytes per
// weight of the two projections.  (Requesting a q/k/v row pair's whole weight stream in one trip was measured and dropped:
// no faster, and 170+ VGPRs.)
// QKN: Qwen3-VL's per-head q/k RMSNorm sits between the projection and the rotation (QkNormRef in o3v_attn_decode_body.h)
template <int NSTEP, int WB, bool QKN = false>
__global__ __launch_bounds__(256, 3) void decode_attn_block_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int Hq = a.ra.Hq, Hkv = a.ra.Hkv, D = a.ra.D, n_rep = Hq / Hkv;
    O3V_STAMP(O3V_STAMP_PTR(a), 0);
    if (bid < a.nb_qkv) {
        gemv_body<1, 2, 1, EPI_QKVROPE, true, true, 0, 4, WB>(a.x, a.qkv_w, a.qkv_b, nullptr, nullptr, a.ln_w, a.eps,
                                                              (Hq + 2 * Hkv) * D, a.H, a.H, a.H, 0, 0, a.ra_qkv, bid, smem, a.qkv_s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x < 64) {  // wave 0: ticket; the last workgroup of the kv head tells that head's attention workgroups
            const int head = (bid * 4) / (D >> 1);  // a workgroup's four rotary pairs lie in one head (D/2 % 4 == 0)
            const int g = head < Hq ? head / n_rep : (head - Hq) % Hkv;
            uint32_t old = 0;
            if (threadIdx.x == 0) old = __hip_atomic_fetch_add(a.sync + SYNC_QKV + g * O3V_SYNC_STRIDE, 1u, O3V_RLX_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == a.epoch * (uint32_t)((n_rep + 2) * (D >> 3)) - 1u) {
                if (QKN)  // every raw q/k/v row of kv head g is in memory: finish the head group, then tell the consumers
                    qkn_finish_head_group(a.qk, g, n_rep, a.ra.qout, a.ra.kc + ((size_t)g * a.ra.Tmax + a.ra.slot) * D,
                                          a.ra.vc + ((size_t)g * a.ra.Tmax + a.ra.slot) * D);
                notify_mailboxes(a.sync + SYNC_BOX_ATT + (size_t)g * a.nsplit * O3V_SYNC_STRIDE, a.nsplit, 0, a.epoch);
                notify_mailboxes(a.sync + SYNC_BOX_O, a.nb_o, 8 + g, a.epoch);  // o_proj: "q/k/v of kv head g are done"
            }
        }
        O3V_STAMP(O3V_STAMP_PTR(a), 3);
        return;
    }
#ifdef O3V_STAMPS
    if ((a.knob & 1) || ((a.knob & 2) && bid >= a.nb_qkv + a.nb_attn)) return;
#endif
    if (bid < a.nb_qkv + a.nb_attn) {
        const int t = bid - a.nb_qkv, split = t % a.nsplit, hk = t / a.nsplit;
        AttnHandoff ho{a.sync + SYNC_BOX_ATT + (size_t)t * O3V_SYNC_STRIDE,
                       a.sync + SYNC_BOX_ATT + (size_t)hk * a.nsplit * O3V_SYNC_STRIDE,
                       a.sync + SYNC_ATT + hk * O3V_SYNC_STRIDE,
                       a.sync + SYNC_ATT2 + hk * O3V_SYNC_STRIDE,
                       a.sync + SYNC_BOX_O,
                       a.nb_o,
                       a.epoch,
                       a.sync + SYNC_TMO,
                       a.att};
#ifdef O3V_STAMPS
        ho.stamp = O3V_STAMP_PTR(a);
        ho.no_prefetch = (a.knob & 4) != 0;
#endif
        attn_decode_mfma_body<true>(a.ra.qout, a.ra.kc, a.ra.vc, a.part_o, a.part_ml, a.k_lo, a.ctx, Hq, Hkv, n_rep,
                                    (long)a.ra.Tmax * D, (long)Hkv * a.ra.Tmax * D, a.scale_log2e, 0, a.nsplit, 0, 1, 0, split,
                                    a.nsplit, hk, 0, smem, ho);
        return;
    }
    oproj_role<NSTEP, WB>(a, bid - a.nb_qkv - a.nb_attn);
    O3V_STAMP(O3V_STAMP_PTR(a), 3);
}

// workgroups of this kernel the chip holds at once (0: query failed)
template <int NSTEP, int WB, bool QKN = false>
int fused_capacity(size_t shmem) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_attn_block_kernel<NSTEP, WB, QKN>, 256, shmem) != hipSuccess) return 0;
    // the query can over-report for SGPR-heavy kernels, never below 6 workgroups per CU (MI355X_MICROARCH.md, Residency)
    if (per_cu > 6) per_cu = 6;
    return per_cu * prop.multiProcessorCount;
}

}  // namespace

// ================================================================================================
// Layer block: q/k/v -> attention -> merge -> o_proj -> RMSNorm -> gate/up (SwiGLU) of ONE decode layer as ONE launch whose
// workgroups are PERSISTENT and IDENTICAL: the grid is exactly what the chip holds (3 workgroups of 256 threads per CU), every
// wave owns a static list of weight rows across the three projections and keeps its next three rows (3 x 7 KiB at K = 3584) in
// flight in registers at all times -- also while the workgroup waits at a hand-off.  That is what the role-per-workgroup block
// above cannot do: there the HBM idles for the ~13 us of the dependent attention chain (only o_proj's 26 MB are preloaded) and
// the gate/up launch pays a boundary and a ramp; here ~64 MB of o_proj / gate/up rows stream under the chain and the gate/up
// rows simply continue.  Rows and arithmetic are those of gemv_body (chunk lane + 64 i, i ascending, wave sum; the epilogues and
// the RMSNorm prologues copied term by term), the attention is attn_decode_mfma_body<true> itself: results are bit-identical to
// o3v_decode_attn_block + o3v_linear_decode(gate/up).
//
// Workgroups [0, nb_attn) also run one (kv head, context split) attention item between their q/k/v rows and their gate/up rows
// (they own no o_proj row and start streaming again after the chain); the others own the o_proj rows.  Hand-offs (protocol of
// o3v_handoff.h): E1 all workgroups -> attention items (q, new K/V row), E2 / E3 inside the attention body, E4 o_proj owners ->
// everybody (the residual stream x').  A workgroup takes its ticket only when its waves have no young loads in flight (the rows
// requested before are long there), so the store drain (vmcnt(0)) costs no memory latency.
// ================================================================================================
constexpr int SYNC_O_DONE = 17 * O3V_SYNC_STRIDE;   // ticket line: workgroups that own o_proj rows
constexpr int SYNC_QKV_ALL = 18 * O3V_SYNC_STRIDE;  // ticket line: every workgroup's q/k/v rows
constexpr int BOX_XREADY = 12;                      // mailbox word: the residual stream after o_proj is in memory
constexpr int LB_RED = 8192;                        // LDS: [0, 8192) the operand vector (K <= 4096), then the norm's scratch

struct LayerArgs {
    FusedArgs f;  // q/k/v, attention and o_proj operands as in the block above (nb_qkv / nb_o unused)
    const bf16_t *ln2, *gu_w;
    bf16_t* act;  // [I] SwiGLU output (input of down_proj, the next launch)
    int I, n_wg;
};

// rows a wave consumes after its q/k/v pair, in order: its o_proj rows (at most two), then its (gate, up) pairs
enum { SU_NONE = 0, SU_O = 1, SU_G = 2, SU_U = 3 };
struct SubUnit {
    int kind, row;
};
struct RowEnum {
    int G, NW, Go, NWo, No, Pg, k, cur, phase;  // phase 0: o_proj rows, 1: gate row next, 2: up row next, 3: done
    __device__ __forceinline__ int gu_pair(int kk) const {
        const int full = Pg / NW, rem = Pg - full * NW;
        if (kk < full) return kk * NW + G;
        if (kk == full && rem > 0) {  // the last, partial round goes to every stride-th wave: no workgroup gets a whole extra round
            const int stride = NW / rem;
            if (G % stride == 0 && G / stride < rem) return full * NW + G / stride;
        }
        return -1;
    }
    __device__ __forceinline__ SubUnit next() {
        if (phase == 0) {
            const int r = Go >= 0 ? Go + k * NWo : No;
            if (r < No) {
                ++k;
                return SubUnit{SU_O, r};
            }
            phase = 1;
            k = 0;
        }
        if (phase == 1) {
            const int p = gu_pair(k);
            if (p < 0) {
                phase = 3;
                return SubUnit{SU_NONE, 0};
            }
            cur = (p >> 4) * 32 + (p & 15);  // gate row of output column p in the 16-row interleaved weight; its up row is + 16
            phase = 2;
            return SubUnit{SU_G, cur};
        }
        if (phase == 2) {
            phase = 1;
            ++k;
            return SubUnit{SU_U, cur + 16};
        }
        return SubUnit{SU_NONE, 0};
    }
};

// RMSNorm of a K-vector into LDS exactly as gemv_body's NORM prologue does it for M = 1 with NT threads (NT / 64 waves): the
// per-thread fmaf chains, the wave sums and the order in which the wave sums are added.  SC1: x was produced in this launch.
template <int NT, bool SC1>
__device__ __forceinline__ void rmsnorm_to_lds(const bf16_t* __restrict__ x, const bf16_t* __restrict__ norm_w, float eps, int K, char* smem) {
    constexpr int XC = (512 + NT - 1) / NT, NWN = NT / 64;
    const int nxc = K >> 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* red = reinterpret_cast<float*>(smem + LB_RED);
    u32x4 xr[XC];  // (the norm weights are fetched at store time: this code runs with a wave's weight rows live in registers)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, K * 2, 0x00020000);
    if (tid < NT) {
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int c = tid + i * NT;
            if (SC1)
                xr[i] = load16_sc1(xrs, c < nxc ? (uint32_t)c * 16 : O3V_OOB);
            else
                xr[i] = c < nxc ? *reinterpret_cast<const u32x4*>(x + (size_t)c * 8) : (u32x4){0, 0, 0, 0};
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < XC; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ss = fmaf(bf_lo(xr[i][j]), bf_lo(xr[i][j]), ss);
                ss = fmaf(bf_hi(xr[i][j]), bf_hi(xr[i][j]), ss);
            }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    if (tid < NT) {
        float t = red[0];
#pragma unroll
        for (int w2 = 1; w2 < NWN; ++w2) t += red[w2];
        const float rstd = 1.0f / sqrtf(t / (float)K + eps);
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int c = tid + i * NT;
            if (c < nxc) {
                const u32x4 wn = *reinterpret_cast<const u32x4*>(norm_w + (size_t)c * 8);
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf2(bf_lo(wn[j]) * rbf(bf_lo(xr[i][j]) * rstd), bf_hi(wn[j]) * rbf(bf_hi(xr[i][j]) * rstd));
                *reinterpret_cast<u32x4*>(smem + (size_t)c * 16) = o;
            }
        }
    }
    __syncthreads();
}

// one weight row of NS whole steps (K = 512 NS) in registers: requested with non-temporal loads, multiplied with the operand
// vector in LDS in gemv_body's order (chunk lane + 64 i, i ascending), then the wave sum
template <int NS>
__device__ __forceinline__ void lb_load(u32x4 (&b)[NS], const bf16_t* __restrict__ W, int row) {
    const u32x4* wp = reinterpret_cast<const u32x4*>(W + (size_t)row * (NS * 512)) + (threadIdx.x & 63);
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) b[s2] = __builtin_nontemporal_load(wp + s2 * 64);
}
template <int NS>
__device__ __forceinline__ float lb_dot(const u32x4 (&b)[NS], const char* vec) {
    const char* vp = vec + (size_t)(threadIdx.x & 63) * 16;
    float acc = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(vp + s2 * 1024);
        fma8(b[s2], xv, acc);
        if (s2 & 1) asm volatile("" ::: "memory");  // two operand chunks in flight, not all of them: the registers hold weight rows
    }
    return wave_sum(acc);
}

// the residual stream after o_proj is complete -> RMSNorm(x') into LDS as the 2-wave gate/up gemv does it
__device__ __forceinline__ void lb_enter_gu(const LayerArgs& a, uint32_t* box, char* smem) {
    if (threadIdx.x < 64) spin_until<8>(box + BOX_XREADY, 1, a.f.epoch, a.f.sync + SYNC_TMO, 0x400u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    rmsnorm_to_lds<128, true>(a.f.xout, a.ln2, a.f.eps, a.f.H, smem);
}

// a workgroup's rows of an op are stored: drain, meet, one ticket; the last ticket of the episode tells the consumers
template <bool QKV>
__device__ __forceinline__ void lb_ticket(const LayerArgs& a) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t old = 0;
        const int n_o_wg = a.n_wg - a.f.nb_attn;
        uint32_t* ticket = a.f.sync + (QKV ? SYNC_QKV_ALL : SYNC_O_DONE);
        const uint32_t want = QKV ? (uint32_t)a.n_wg : (uint32_t)n_o_wg;
        if (threadIdx.x == 0) old = __hip_atomic_fetch_add(ticket, 1u, O3V_RLX_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == a.f.epoch * want - 1u) {
            if (QKV) {
                notify_mailboxes(a.f.sync + SYNC_BOX_ATT, a.f.nb_attn, 0, a.f.epoch);
            } else {
                notify_mailboxes(a.f.sync + SYNC_BOX_ATT, a.f.nb_attn, BOX_XREADY, a.f.epoch);
                notify_mailboxes(a.f.sync + SYNC_BOX_O, n_o_wg, BOX_XREADY, a.f.epoch);
            }
        }
    }
}

// q/k/v rotary pair of this wave from buffers b0 (row j) and b1 (row j + D/2): bias, rotation, q out / new K,V row (gemv_body's
// EPI_QKVROPE epilogue), stored write-through
template <int NS>
__device__ __forceinline__ void lb_qkv_pair(const LayerArgs& a, const u32x4 (&b0)[NS], const u32x4 (&b1)[NS], int row0, float bias0, float bias1,
                                            float cs, float sn, const char* vec) {
    const float acc0 = lb_dot<NS>(b0, vec), acc1 = lb_dot<NS>(b1, vec);
    if ((threadIdx.x & 63) != 0) return;
    const RopeArgs& ra = a.f.ra;
    const int half = ra.D >> 1, head = row0 / ra.D, j = row0 % ra.D;
    const float v0 = rbf(acc0 + bias0), v1 = rbf(acc1 + bias1);
    if (head >= ra.Hq + ra.Hkv) {  // v: no rotation
        bf16_t* dst = ra.vc + ((size_t)(head - ra.Hq - ra.Hkv) * ra.Tmax + ra.slot) * ra.D;
        gemv_store_bf16<true>(dst + j, f2bf(v0));
        gemv_store_bf16<true>(dst + j + half, f2bf(v1));
        return;
    }
    const float o0 = __fadd_rn(rbf(__fmul_rn(v0, cs)), rbf(__fmul_rn(-v1, sn)));
    const float o1 = __fadd_rn(rbf(__fmul_rn(v1, cs)), rbf(__fmul_rn(v0, sn)));
    bf16_t* dst = head < ra.Hq ? ra.qout + (size_t)head * ra.D : ra.kc + ((size_t)(head - ra.Hq) * ra.Tmax + ra.slot) * ra.D;
    gemv_store_bf16<true>(dst + j, f2bf(o0));
    gemv_store_bf16<true>(dst + j + half, f2bf(o1));
}

template <int NS>
struct RowPipe {
    u32x4 b[3][NS];
    float res[3];  // o_proj rows: the residual, requested with the row
    SubUnit d[3];
};

template <int I3, int NS>
__device__ __forceinline__ void lb_issue(RowPipe<NS>& P, const LayerArgs& a, const SubUnit u) {
    P.d[I3] = SubUnit{__builtin_amdgcn_readfirstlane(u.kind), __builtin_amdgcn_readfirstlane(u.row)};
    P.res[I3] = 0.f;
    if (P.d[I3].kind == SU_NONE) {  // (defined on every path: an old row must not stay live in the compiler's eyes)
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) P.b[I3][s2] = (u32x4){0, 0, 0, 0};
        return;
    }
    if (P.d[I3].kind == SU_O) {
        lb_load<NS>(P.b[I3], a.f.o_w, P.d[I3].row);
        P.res[I3] = bf2f(a.f.x[P.d[I3].row]);  // residual: written by an earlier launch
    } else {
        lb_load<NS>(P.b[I3], a.gu_w, P.d[I3].row);
    }
}

// The row stream of one wave behind its q/k/v pair: [o_proj rows] [gate, up, gate, up, ...], three rows in flight.  `in_o`: the
// workgroup owns o_proj rows (every wave of it then enters and leaves the o_proj op, also one that owns none).  The ticket of the
// o_proj op is taken BEFORE the freed buffer is refilled (the loads still in flight are old: the drain costs no latency), the
// refill goes out before the wait for x'.
template <int NS>
__device__ __forceinline__ void lb_stream(RowPipe<NS>& P, const LayerArgs& a, RowEnum& E, bool in_o, uint32_t* box, char* smem) {
    float gate_acc = 0.f;
    bool done = false;
    const int lane = threadIdx.x & 63;
#define O3V_LB_STEP(I3, NX)                                                                        \
    if (!done) {                                                                                   \
        const SubUnit u = P.d[I3];                                                                 \
        if (u.kind == SU_NONE) {                                                                   \
            done = true;                                                                           \
        } else {                                                                                   \
            if (in_o && u.kind != SU_O) { /* a wave without o_proj rows: it only meets the others */ \
                lb_ticket<false>(a);                                                               \
                lb_enter_gu(a, box, smem);                                                         \
                in_o = false;                                                                      \
            }                                                                                      \
            const float acc = lb_dot<NS>(P.b[I3], smem);                                           \
            if (u.kind == SU_O) {                                                                  \
                if (lane == 0) {                                                                   \
                    float v = acc + 0.f; /* no bias (TF:620) */                                    \
                    v = rbf(v) + P.res[I3];                                                        \
                    gemv_store_bf16<true>(a.f.xout + u.row, f2bf(v));                              \
                }                                                                                  \
                if (P.d[NX].kind != SU_O) { /* this wave's last o_proj row */                      \
                    lb_ticket<false>(a);                                                           \
                    lb_issue<I3, NS>(P, a, E.next());                                              \
                    lb_enter_gu(a, box, smem);                                                     \
                    in_o = false;                                                                  \
                } else {                                                                           \
                    lb_issue<I3, NS>(P, a, E.next());                                              \
                }                                                                                  \
            } else {                                                                               \
                if (u.kind == SU_G) {                                                              \
                    gate_acc = acc;                                                                \
                } else if (lane == 0) {                                                            \
                    const int grow = u.row - 16, no = (grow >> 5) * 16 + (grow & 15);              \
                    const float g = rbf(gate_acc + 0.f), uu = rbf(acc + 0.f);                      \
                    a.act[no] = f2bf(rbf(silu_f(g)) * uu);                                         \
                }                                                                                  \
                lb_issue<I3, NS>(P, a, E.next());                                                  \
            }                                                                                      \
        }                                                                                          \
    }
    while (!done) {
        O3V_LB_STEP(2, 0)   // the stream starts in buffer 2: buffers 0 and 1 carried the q/k/v pair
        O3V_LB_STEP(0, 1)
        O3V_LB_STEP(1, 2)
    }
#undef O3V_LB_STEP
    if (in_o) {  // (cannot happen: every wave owns gate/up rows; kept so that the workgroup's barriers always pair up)
        lb_ticket<false>(a);
        lb_enter_gu(a, box, smem);
    }
}

template <int NS>
__global__ __launch_bounds__(256, 3) void decode_layer_block_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // wave-uniform values are told to the compiler as such (readfirstlane): the row descriptors, their pointers and every branch on
    // them then live in SGPRs / scalar branches instead of per-lane registers and exec masks
    const int bid = blockIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int Hq = a.f.ra.Hq, Hkv = a.f.ra.Hkv, D = a.f.ra.D, n_rep = Hq / Hkv;
    const bool attn_wg = bid < a.f.nb_attn;
    uint32_t* box = attn_wg ? a.f.sync + SYNC_BOX_ATT + (size_t)bid * O3V_SYNC_STRIDE
                            : a.f.sync + SYNC_BOX_O + (size_t)(bid - a.f.nb_attn) * O3V_SYNC_STRIDE;
    RowEnum E;
    E.G = bid * 4 + wave;
    E.NW = a.n_wg * 4;
    E.NWo = (a.n_wg - a.f.nb_attn) * 4;
    E.Go = attn_wg ? -1 : (bid - a.f.nb_attn) * 4 + wave;
    E.No = a.f.H;
    E.Pg = a.I;
    E.k = 0;
    E.cur = 0;
    E.phase = 0;
    RowPipe<NS> P;
    // ---- t = 0: the q/k/v rotary pair of this wave (at most one: pairs <= waves) into buffers 0 / 1; an o_proj owner also
    // requests the first row of its stream into buffer 2, and a wave without a pair fills buffers 0 / 1 from its stream as well
    const int Pq = (Hq + 2 * Hkv) * D / 2, half = D >> 1;
    const bool has_q = E.G < Pq;
    const int row0 = has_q ? (E.G / half) * D + (E.G % half) : 0;
    float bias0 = 0.f, bias1 = 0.f, cs = 1.f, sn = 0.f;
    lb_issue<0, NS>(P, a, SubUnit{SU_NONE, 0});
    lb_issue<1, NS>(P, a, SubUnit{SU_NONE, 0});
    lb_issue<2, NS>(P, a, SubUnit{SU_NONE, 0});
    if (has_q) {
        lb_load<NS>(P.b[0], a.f.qkv_w, row0);
        lb_load<NS>(P.b[1], a.f.qkv_w, row0 + half);
        if (a.f.qkv_b) {
            bias0 = bf2f(a.f.qkv_b[row0]);
            bias1 = bf2f(a.f.qkv_b[row0 + half]);
        }
        const size_t ci = (size_t)a.f.ra.cs_off * D + (row0 % D);
        cs = bf2f(a.f.ra.cosT[ci]);
        sn = bf2f(a.f.ra.sinT[ci]);
    }
    if (!attn_wg) {
        lb_issue<2, NS>(P, a, E.next());
        if (!has_q) {
            lb_issue<0, NS>(P, a, E.next());
            lb_issue<1, NS>(P, a, E.next());
        }
    }
    rmsnorm_to_lds<256, false>(a.f.x, a.f.ln_w, a.f.eps, a.f.H, smem);
    if (has_q) lb_qkv_pair<NS>(a, P.b[0], P.b[1], row0, bias0, bias1, cs, sn, smem);
    lb_ticket<true>(a);  // E1: every workgroup's q / k / v rows are in memory -> the attention items
    if (!attn_wg) {
        if (has_q) {
            lb_issue<0, NS>(P, a, E.next());
            lb_issue<1, NS>(P, a, E.next());
        }
        // o_proj operand: the attention output, once every kv head's merged slices are in memory (E3)
        if (wave == 0) spin_until<8>(box, Hkv, a.f.epoch, a.f.sync + SYNC_TMO, 0x200u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __syncthreads();
        {
            const int nch = (Hq * D) >> 3;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.f.att, 0, Hq * D * 2, 0x00020000);
            for (int c = threadIdx.x; c < nch; c += 256) *reinterpret_cast<u32x4*>(smem + (size_t)c * 16) = load16_sc1(rs, (uint32_t)c * 16);
        }
        __syncthreads();
        lb_stream<NS>(P, a, E, true, box, smem);
        return;
    }
    // ---- attention workgroup: one (kv head, context split) item -- nothing of the row pipeline is live across it -- then its
    // share of the gate/up rows
    {
        const int t = bid, split = t % a.f.nsplit, hk = t / a.f.nsplit;
        AttnHandoff ho{box,
                       a.f.sync + SYNC_BOX_ATT + (size_t)hk * a.f.nsplit * O3V_SYNC_STRIDE,
                       a.f.sync + SYNC_ATT + hk * O3V_SYNC_STRIDE,
                       a.f.sync + SYNC_ATT2 + hk * O3V_SYNC_STRIDE,
                       a.f.sync + SYNC_BOX_O,
                       a.n_wg - a.f.nb_attn,
                       a.f.epoch,
                       a.f.sync + SYNC_TMO,
                       a.f.att};
#ifdef O3V_STAMPS
        ho.stamp = nullptr;
        ho.no_prefetch = false;
#endif
        attn_decode_mfma_body<true>(a.f.ra.qout, a.f.ra.kc, a.f.ra.vc, a.f.part_o, a.f.part_ml, a.f.k_lo, a.f.ctx, Hq, Hkv, n_rep,
                                    (long)a.f.ra.Tmax * D, (long)Hkv * a.f.ra.Tmax * D, a.f.scale_log2e, 0, a.f.nsplit, 0, 1, 0, split,
                                    a.f.nsplit, hk, 0, smem, ho);
    }
    __syncthreads();  // the attention scratch is dead: the operand vector of gate/up goes there
    RowPipe<NS> P2;
    lb_issue<2, NS>(P2, a, E.next());
    lb_issue<0, NS>(P2, a, E.next());
    lb_issue<1, NS>(P2, a, E.next());
    lb_enter_gu(a, box, smem);
    lb_stream<NS>(P2, a, E, false, box, smem);
}


// o_proj row lengths built (steps of 64 chunks): bf16 rows of 1792 (fixtures) / 2048 (3B) / 3584 (7B) / 4096 (8B class) take
// 4 / 4 / 7 / 8 steps, their fp8 forms 2 / 2 / 4 / 4
#define O3V_FUSED_SHAPES(X) X(7, 2) X(4, 2) X(8, 2) X(2, 1) X(4, 1)
// with the q/k norm (Qwen3-VL): rows of 4096 (8B) and 1024 (fixture) in bf16 and fp8
#define O3V_FUSED_SHAPES_QKN(X) X(8, 2) X(2, 2) X(4, 1) X(1, 1)

// workgroups of the fused kernel the chip holds at once for o_proj rows of qd = Hq*D elements of wb bytes (diagnostics / tests)
extern "C" int o3v_decode_attn_block_capacity(int qd, int wb) {
    const int nq = (qd / (wb == 1 ? 16 : 8) + 63) / 64;
#define O3V_X(A, B) \
    if (nq == A && wb == B) return fused_capacity<A, B>(ATTN_LDS);
    O3V_FUSED_SHAPES(O3V_X)
#undef O3V_X
    return 0;
}

#ifdef O3V_STAMPS
static unsigned long long* g_stamps = nullptr;  // diagnostic build only
static int g_knob = 0;
extern "C" void o3v_fused_set_stamps(unsigned long long* p) { g_stamps = p; }
extern "C" void o3v_fused_set_knob(int k) { g_knob = k; }
#endif

extern "C" size_t o3v_decode_sync_bytes(void) { return (size_t)SYNC_WORDS * 4; }


static int attn_block_launch(void* x, const void* ln_w, float eps, const void* qkv_w, const float* qkv_s, const void* qkv_b,
                             const void* o_w, const float* o_s, const void* q_norm, const void* k_norm, void* kv_raw,
                             const void* cosT, const void* sinT, void* q_buf, void* att_buf,
                             void* kcache, void* vcache, float* part_o, float* part_ml, const int* k_lo, int H, int Hq, int Hkv,
                             int D, int slot, int Tmax, int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync,
                             uint32_t epoch, hipStream_t stream) {
    const int wb = qkv_s ? 1 : 2;
    if ((qkv_s == nullptr) != (o_s == nullptr)) return O3V_ERR_ARG;
    const bool qkn = q_norm != nullptr;
    if (qkn && (!k_norm || !kv_raw)) return O3V_ERR_ARG;
    if (!x || !ln_w || !qkv_w || !o_w || !cosT || !sinT || !q_buf || !att_buf || !kcache || !vcache || !part_o || !part_ml ||
        !sync || epoch == 0 || slot < 0 || slot >= Tmax || H <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || nsplit <= 0 || nsplit > 64)
        return O3V_ERR_ARG;
    const int n_rep = Hq / Hkv, QD = Hq * D, NQKV = (Hq + 2 * Hkv) * D;
    // shapes the roles are written for: head_dim 128 matrix-core attention, <= 8 kv heads (counter slots), whole
    // 16-byte chunks, the fused-norm GEMV's x in registers (K <= 4096), o_proj rows in <= 8 register steps
    if (D != 128 || Hkv > 8 || n_rep > NREP_MAX || (H & 7) || H > 4096 || QD > 4096 || (NQKV % 8)) return O3V_ERR_SHAPE;
    const int nb_qkv = NQKV / 8, nb_attn = nsplit * Hkv, nb_o = (H + 7) / 8;
    if (nb_attn > 512 || nb_o > 512) return O3V_ERR_SHAPE;  // mailbox lines
    const size_t lds_qkv = (size_t)H * 2 + 4 * 2 * 4 + 4 * 4;
    const size_t shmem = lds_qkv > (size_t)ATTN_LDS ? lds_qkv : (size_t)ATTN_LDS;
    if (wb == 1 && ((H & 15) || (QD & 15))) return O3V_ERR_SHAPE;
    const int nstep = (QD / (wb == 1 ? 16 : 8) + 63) / 64;
    FusedArgs a;
    a.qkv_s = qkv_s;
    a.o_s = o_s;
    a.x = (const bf16_t*)x;
    a.ln_w = (const bf16_t*)ln_w;
    a.qkv_w = (const bf16_t*)qkv_w;
    a.qkv_b = (const bf16_t*)qkv_b;
    a.o_w = (const bf16_t*)o_w;
    a.att = (bf16_t*)att_buf;
    a.xout = (bf16_t*)x;
    a.part_o = part_o;
    a.part_ml = part_ml;
    a.k_lo = k_lo;
    a.sync = sync;
    a.epoch = epoch;
    a.ra = RopeArgs{(const bf16_t*)cosT, (const bf16_t*)sinT, (bf16_t*)q_buf, (bf16_t*)kcache, (bf16_t*)vcache,
                    slot, Hq, Hkv, D, Tmax, cs_stride_row, cs_off};
    a.ra_qkv = a.ra;
    a.qk = QkNormRef{};
    if (qkn) {
        // the q/k/v role stores bf16(acc + bias) as it is (rotation off) into the scratch [k: Hkv*D | v: Hkv*D | q: Hq*D] (cache
        // geometry "one slot per head"); the last arriver of a kv head finishes the head group into q_buf and the cache row
        bf16_t* kr = (bf16_t*)kv_raw;
        bf16_t* vr = kr + (size_t)Hkv * D;
        bf16_t* qr = vr + (size_t)Hkv * D;
        a.ra_qkv = RopeArgs{nullptr, nullptr, qr, kr, vr, 0, Hq, Hkv, D, 1, 0, 0, 1};
        const bf16_t* cr = (const bf16_t*)cosT + (size_t)cs_off * D;
        const bf16_t* sr = (const bf16_t*)sinT + (size_t)cs_off * D;
        a.qk = QkNormRef{qr, kr, vr, (const bf16_t*)q_norm, (const bf16_t*)k_norm, cr, sr, eps};
    }
    a.eps = eps;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.H = H;
    a.ctx = slot + 1;
    a.nsplit = nsplit;
    a.nb_qkv = nb_qkv;
    a.nb_attn = nb_attn;
    a.nb_o = nb_o;
#ifdef O3V_STAMPS
    a.stamps = g_stamps;
    a.knob = g_knob;
#endif
    const dim3 grid(nb_qkv + nb_attn + nb_o), block(256);
    bool launched = false;
#define O3V_X(A, B)                                                                                      \
    if (!launched && nstep == A && wb == B) {                                                            \
        static const int cap = fused_capacity<A, B>(shmem);                                              \
        if (nb_attn + nb_o >= cap) return O3V_ERR_SHAPE; /* waiting workgroups must not fill the chip */ \
        O3V_KLAUNCH((decode_attn_block_kernel<A, B>), grid, block, shmem, stream, a);                    \
        launched = true;                                                                                 \
    }
    if (!qkn) {
        O3V_FUSED_SHAPES(O3V_X)
    }
#undef O3V_X
#define O3V_X(A, B)                                                                                      \
    if (!launched && nstep == A && wb == B) {                                                            \
        static const int cap = fused_capacity<A, B, true>(shmem);                                        \
        if (nb_attn + nb_o >= cap) return O3V_ERR_SHAPE;                                                 \
        O3V_KLAUNCH((decode_attn_block_kernel<A, B, true>), grid, block, shmem, stream, a);              \
        launched = true;                                                                                 \
    }
    if (qkn) {
        O3V_FUSED_SHAPES_QKN(O3V_X)
    }
#undef O3V_X
    if (!launched) return O3V_ERR_SHAPE;
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_decode_attn_block(void* x, const void* ln_w, float eps, const void* qkv_w, const void* qkv_b, const void* o_w,
                                     const void* cosT, const void* sinT, void* q_buf, void* att_buf, void* kcache, void* vcache,
                                     float* part_o, float* part_ml, const int* k_lo, int H, int Hq, int Hkv, int D, int slot,
                                     int Tmax, int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync,
                                     uint32_t epoch, hipStream_t stream) {
    return attn_block_launch(x, ln_w, eps, qkv_w, nullptr, qkv_b, o_w, nullptr, nullptr, nullptr, nullptr, cosT, sinT, q_buf, att_buf,
                             kcache, vcache, part_o, part_ml, k_lo, H, Hq, Hkv, D, slot, Tmax, cs_stride_row, cs_off, nsplit, scale, sync,
                             epoch, stream);
}

// ---- the persistent layer block (decode_layer_block_kernel): bf16 rows, no q/k norm; head_dim 128, hidden = Hq * D a whole number
// of 512-element steps (2048: 3B, 3584: 7B), at most one q/k/v rotary pair and two o_proj rows per wave of the resident grid
template <int NS>
static int layer_block_capacity() {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_layer_block_kernel<NS>, 256, ATTN_LDS) != hipSuccess) return 0;
    if (per_cu < 3) return 0;
    return 3 * prop.multiProcessorCount;  // the grid: three workgroups per CU, all resident
}

extern "C" int o3v_decode_layer_block(void* x, const void* ln1, float eps, const void* qkv_w, const void* qkv_b, const void* o_w,
                                      const void* ln2, const void* gu_w, void* act, const void* cosT, const void* sinT, void* q_buf,
                                      void* att_buf, void* kcache, void* vcache, float* part_o, float* part_ml, const int* k_lo, int H,
                                      int I, int Hq, int Hkv, int D, int slot, int Tmax, int cs_stride_row, int cs_off, int nsplit,
                                      float scale, uint32_t* sync, uint32_t epoch, hipStream_t stream) {
    if (!x || !ln1 || !qkv_w || !o_w || !ln2 || !gu_w || !act || !cosT || !sinT || !q_buf || !att_buf || !kcache || !vcache || !part_o ||
        !part_ml || !sync || epoch == 0 || slot < 0 || slot >= Tmax || H <= 0 || I <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || nsplit <= 0 ||
        nsplit > 64)
        return O3V_ERR_ARG;
    const int n_rep = Hq / Hkv, QD = Hq * D, NQKV = (Hq + 2 * Hkv) * D;
    if (D != 128 || Hkv > 8 || n_rep > NREP_MAX || (H % 512) || QD != H || (I & 15)) return O3V_ERR_SHAPE;
    const int ns = H / 512;
    int n_wg = 0;
    if (ns == 4) {
        static const int cap = layer_block_capacity<4>();
        n_wg = cap;
    } else if (ns == 7) {
        static const int cap = layer_block_capacity<7>();
        n_wg = cap;
    } else {
        return O3V_ERR_SHAPE;
    }
    const int nb_attn = nsplit * Hkv;
    if (n_wg <= 0 || nb_attn > 512 || nb_attn >= n_wg) return O3V_ERR_SHAPE;
    const int n_o_wg = n_wg - nb_attn;
    if (n_o_wg > 1024 || NQKV / 2 > n_wg * 4 || H > 2 * n_o_wg * 4) return O3V_ERR_SHAPE;  // one rotary pair, two o_proj rows per wave
    LayerArgs a;
    a.f.qkv_s = nullptr;
    a.f.o_s = nullptr;
    a.f.x = (const bf16_t*)x;
    a.f.ln_w = (const bf16_t*)ln1;
    a.f.qkv_w = (const bf16_t*)qkv_w;
    a.f.qkv_b = (const bf16_t*)qkv_b;
    a.f.o_w = (const bf16_t*)o_w;
    a.f.att = (bf16_t*)att_buf;
    a.f.xout = (bf16_t*)x;
    a.f.part_o = part_o;
    a.f.part_ml = part_ml;
    a.f.k_lo = k_lo;
    a.f.sync = sync;
    a.f.epoch = epoch;
    a.f.ra = RopeArgs{(const bf16_t*)cosT, (const bf16_t*)sinT, (bf16_t*)q_buf, (bf16_t*)kcache, (bf16_t*)vcache,
                      slot, Hq, Hkv, D, Tmax, cs_stride_row, cs_off};
    a.f.ra_qkv = a.f.ra;
    a.f.qk = QkNormRef{};
    a.f.eps = eps;
    a.f.scale_log2e = scale * 1.4426950408889634f;
    a.f.H = H;
    a.f.ctx = slot + 1;
    a.f.nsplit = nsplit;
    a.f.nb_qkv = 0;
    a.f.nb_attn = nb_attn;
    a.f.nb_o = 0;
#ifdef O3V_STAMPS
    a.f.stamps = nullptr;
    a.f.knob = 0;
#endif
    a.ln2 = (const bf16_t*)ln2;
    a.gu_w = (const bf16_t*)gu_w;
    a.act = (bf16_t*)act;
    a.I = I;
    a.n_wg = n_wg;
    const dim3 grid(n_wg), block(256);
    if (ns == 4)
        O3V_KLAUNCH((decode_layer_block_kernel<4>), grid, block, ATTN_LDS, stream, a);
    else
        O3V_KLAUNCH((decode_layer_block_kernel<7>), grid, block, ATTN_LDS, stream, a);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

// the same with fp8 (OCP e4m3fn) rows + per-row scales for the two projections (o3v_linear_decode_fp8's weight format)
extern "C" int o3v_decode_attn_block_fp8(void* x, const void* ln_w, float eps, const void* qkv_w8, const float* qkv_s,
                                         const void* qkv_b, const void* o_w8, const float* o_s, const void* cosT, const void* sinT,
                                         void* q_buf, void* att_buf, void* kcache, void* vcache, float* part_o, float* part_ml,
                                         const int* k_lo, int H, int Hq, int Hkv, int D, int slot, int Tmax, int cs_stride_row,
                                         int cs_off, int nsplit, float scale, uint32_t* sync, uint32_t epoch, hipStream_t stream) {
    if (!qkv_s || !o_s) return O3V_ERR_ARG;
    return attn_block_launch(x, ln_w, eps, qkv_w8, qkv_s, qkv_b, o_w8, o_s, nullptr, nullptr, nullptr, cosT, sinT, q_buf, att_buf,
                             kcache, vcache, part_o, part_ml, k_lo, H, Hq, Hkv, D, slot, Tmax, cs_stride_row, cs_off, nsplit, scale, sync,
                             epoch, stream);
}

// Qwen3-VL form (TF3:438-500): no q/k/v bias, RMSNorm weights q_norm / k_norm [D] on every q and k head between the projection
// and the rotation.  kv_raw: (Hq + 2 * Hkv) * D bf16 of scratch (the un-normalised k, v and q of this token); qkv_s / o_s non-NULL select
// fp8 rows as in o3v_decode_attn_block_fp8.  Bit-identical to o3v_linear_decode(q/k/v) + o3v_qkv_norm_rope_cache +
// o3v_attn_decode + o3v_linear_decode(o_proj, RESIDUAL).
extern "C" int o3v_decode_attn_block_qknorm(void* x, const void* ln_w, float eps, const void* qkv_w, const float* qkv_s,
                                            const void* o_w, const float* o_s, const void* q_norm, const void* k_norm, void* kv_raw,
                                            const void* cosT, const void* sinT, void* q_buf, void* att_buf, void* kcache, void* vcache,
                                            float* part_o, float* part_ml, const int* k_lo, int H, int Hq, int Hkv, int D, int slot,
                                            int Tmax, int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync,
                                            uint32_t epoch, hipStream_t stream) {
    if (!q_norm || !k_norm || !kv_raw) return O3V_ERR_ARG;
    return attn_block_launch(x, ln_w, eps, qkv_w, qkv_s, nullptr, o_w, o_s, q_norm, k_norm, kv_raw, cosT, sinT, q_buf, att_buf, kcache,
                             vcache, part_o, part_ml, k_lo, H, Hq, Hkv, D, slot, Tmax, cs_stride_row, cs_off, nsplit, scale, sync, epoch,
                             stream);
}
