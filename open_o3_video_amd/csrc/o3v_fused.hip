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
constexpr int SYNC_BOX_O = (32 + 512) * O3V_SYNC_STRIDE;  // [512] mailbox lines of the o_proj workgroups, word hk per kv head
constexpr int SYNC_WORDS = (32 + 512 + 512) * O3V_SYNC_STRIDE;
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
