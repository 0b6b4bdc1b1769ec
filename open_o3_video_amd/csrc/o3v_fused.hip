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
#include <string.h>

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

// The in-launch hand-offs rest on gfx950 behaviour that was measured, not on a portable memory-model guarantee (o3v_handoff.h): refuse
// the one-launch forms on anything else (the callers then take the stand-alone kernels)
static bool device_is_gfx950(const hipDeviceProp_t& prop) { return strncmp(prop.gcnArchName, "gfx950", 6) == 0; }

// workgroups of this kernel the chip holds at once (0: query failed)
template <int NSTEP, int WB, bool QKN = false>
int fused_capacity(size_t shmem) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (!device_is_gfx950(prop)) return 0;
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

// The (gate, up) rows of wave G of NW: output column (pair) kk * NW + G in round kk; the last, partial round goes to every
// stride-th wave, so that no workgroup gets a whole extra round.  Row 2 kk is the pair's gate row, 2 kk + 1 its up row (16-row
// interleaved weight: gate row of column p at (p >> 4) * 32 + (p & 15), up row + 16).
struct GuRows {
    int G, NW, Pg;
    __device__ __forceinline__ int pairs() const {
        const int full = Pg / NW, rem = Pg - full * NW;
        if (rem == 0) return full;
        const int stride = NW / rem;
        return full + ((G % stride == 0 && G / stride < rem) ? 1 : 0);
    }
    __device__ __forceinline__ int row(int k) const {
        const int kk = k >> 1, full = Pg / NW;
        const int p = kk < full ? kk * NW + G : full * NW + G / (NW / (Pg - full * NW));
        return (p >> 4) * 32 + (p & 15) + (k & 1) * 16;
    }
};

// workgroup barrier that orders LDS only: __syncthreads() carries a fence, and the compiler drains vmcnt(0) for it -- i.e. it would
// wait for every weight row the waves keep in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RMSNorm of a K-vector into LDS exactly as gemv_body's NORM prologue does it for M = 1 with NT threads (NT / 64 waves): the
// per-thread fmaf chains, the wave sums and the order in which the wave sums are added.  The caller has loaded the x chunks
// (thread t: chunks t + i NT, zeros past the row) and the matching norm-weight chunks -- it decides when those loads are issued
// relative to the weight rows.
template <int NT, int XC>
__device__ __forceinline__ void norm_finish(const u32x4 (&xr)[XC], const u32x4 (&wn)[XC], float eps, int K, char* smem) {
    constexpr int NWN = NT / 64;
    const int nxc = K >> 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* red = reinterpret_cast<float*>(smem + LB_RED);
    if (tid < NT) {
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
    lds_barrier();
    if (tid < NT) {
        float t = red[0];
#pragma unroll
        for (int w2 = 1; w2 < NWN; ++w2) t += red[w2];
        const float rstd = 1.0f / sqrtf(t / (float)K + eps);
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int c = tid + i * NT;
            if (c < nxc) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf2(bf_lo(wn[i][j]) * rbf(bf_lo(xr[i][j]) * rstd), bf_hi(wn[i][j]) * rbf(bf_hi(xr[i][j]) * rstd));
                *reinterpret_cast<u32x4*>(smem + (size_t)c * 16) = o;
            }
        }
    }
    lds_barrier();
}

// loads the compiler does not track (see lb_load): 16 bytes from a per-lane address, a bf16 scalar from a wave-uniform address
__device__ __forceinline__ u32x4 asm_load16(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t asm_load_bf16(const bf16_t* p) {
    uint32_t v;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// The row loads are written as asm and waited for by hand.  With compiler-visible loads the waitcnt pass loses the count of
// outstanding loads at the (wave-uniform) branches of the stream loop and waits for vmcnt(0..6) before it touches the OLDEST of the
// three rows in flight -- i.e. for all three -- which leaves one row in flight per wave instead of three (measured: the block was
// 14 us per layer slower than the launches it replaces).  Loads return in order, so "at most NS * (rows requested after this one)
// operations outstanding" means this row has landed; stores or other loads issued in between only make that wait stricter.
template <int NS>
__device__ __forceinline__ void lb_load(u32x4 (&b)[NS], const bf16_t* __restrict__ W, int row) {
    const char* base = reinterpret_cast<const char*>(W) + (size_t)row * (NS * 1024);  // wave-uniform: an SGPR pair
    const uint32_t v0 = (uint32_t)(threadIdx.x & 63) * 16u, v1 = v0 + 4096u;           // the immediate offset reaches 4095
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        if (s2 < 4)
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(b[s2]) : "v"(v0), "s"(base), "n"(s2 * 1024) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(b[s2]) : "v"(v1), "s"(base), "n"((s2 - 4) * 1024) : "memory");
    }
}
// wait until at most NS * `younger` vector-memory operations are outstanding; the row's registers pass through the asm so that no
// use of them can be scheduled above the wait
template <int NS>
__device__ __forceinline__ void lb_wait(u32x4 (&b)[NS], int younger) {
    if (younger >= 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS) : "memory");
    else if (younger == 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) asm volatile("" : "+v"(b[s2]));
}
__device__ __forceinline__ void asm_tie(uint32_t& v) { asm volatile("" : "+v"(v)); }
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

// the residual stream after o_proj is complete -> RMSNorm(x') into LDS as the 2-wave gate/up gemv does it (threads 0..127).  The
// polls and the x' chunks are ordinary loads: they return behind the row this wave requested last, which has had the whole
// hand-off to land.
__device__ __forceinline__ void lb_enter_gu(const LayerArgs& a, uint32_t* box, char* smem) {
    if (threadIdx.x < 64) spin_until<8>(box + BOX_XREADY, 1, a.f.epoch, a.f.sync + SYNC_TMO, 0x400u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    lds_barrier();
    constexpr int XC = 4;
    const int nxc = a.f.H >> 3, tid = threadIdx.x;
    u32x4 xr[XC], wn[XC];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.f.xout, 0, a.f.H * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < XC; ++i) {
        const int c = tid + i * 128;
        const bool in = tid < 128 && c < nxc;
        xr[i] = load16_sc1(xrs, in ? (uint32_t)c * 16 : O3V_OOB);
        wn[i] = in ? *reinterpret_cast<const u32x4*>(a.ln2 + (size_t)c * 8) : (u32x4){0, 0, 0, 0};
    }
    norm_finish<128, XC>(xr, wn, a.f.eps, a.f.H, smem);
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

// The gate/up rows of one wave, three in flight: row k lives in buffer (k + 1) % 3 (buffer 1 first: buffers 2 and 0 carry the
// wave's o_proj rows before).  Nothing but the row loads (asm), the LDS reads, the dot products and the SwiGLU store is in this
// loop: no hand-off, no compiler-visible load -- so no wait but the hand-written one.  `y0` / `y1`: rows known to be younger than
// row 0 / row 1 when the loop starts (then 2).
template <int NS>
__device__ __forceinline__ void lb_gu_loop(u32x4 (&b0)[NS], u32x4 (&b1)[NS], u32x4 (&b2)[NS], const LayerArgs& a, const GuRows& R,
                                           const int n_rows, const int y0, const int y1, const char* smem) {
    const int lane = threadIdx.x & 63;
    float gate_acc = 0.f;
    int k = 0;
#define O3V_GU_STEP(BUF, GATE)                                                                  \
    {                                                                                           \
        const int left = n_rows - 1 - k;                                                        \
        int y = k == 0 ? y0 : (k == 1 ? y1 : 2);                                                \
        y = y < left ? y : left;                                                                \
        lb_wait<NS>(BUF, y);                                                                    \
        const float acc = lb_dot<NS>(BUF, smem);                                                \
        if (GATE) {                                                                             \
            gate_acc = acc;                                                                     \
        } else if (lane == 0) {                                                                 \
            const int grow = R.row(k) - 16, no = (grow >> 5) * 16 + (grow & 15);                \
            const float g = rbf(gate_acc + 0.f), uu = rbf(acc + 0.f);                           \
            a.act[no] = f2bf(rbf(silu_f(g)) * uu);                                              \
        }                                                                                       \
        if (k + 3 < n_rows) lb_load<NS>(BUF, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(k + 3))); \
        ++k;                                                                                    \
    }
    while (k < n_rows) {  // n_rows is even: a (gate, up) pair is never split by the loop's exit
        O3V_GU_STEP(b1, true)
        O3V_GU_STEP(b2, false)
        if (k >= n_rows) break;
        O3V_GU_STEP(b0, true)
        O3V_GU_STEP(b1, false)
        if (k >= n_rows) break;
        O3V_GU_STEP(b2, true)
        O3V_GU_STEP(b0, false)
    }
#undef O3V_GU_STEP
}

#ifdef O3V_STAMPS
#define O3V_LSTAMP(i)                                                                                              \
    do {                                                                                                           \
        if (threadIdx.x == 0 && a.f.stamps) a.f.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define O3V_LSTAMP(i) \
    do {              \
    } while (0)
#endif

template <int NS>
__global__ __launch_bounds__(256, 3) void decode_layer_block_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    O3V_LSTAMP(0);
    // wave-uniform values are told to the compiler as such (readfirstlane): row numbers, their pointers and every branch on them
    // then live in SGPRs / scalar branches instead of per-lane registers and exec masks
    const int bid = blockIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int Hq = a.f.ra.Hq, Hkv = a.f.ra.Hkv, D = a.f.ra.D, n_rep = Hq / Hkv;
    const bool attn_wg = bid < a.f.nb_attn;
    uint32_t* box = attn_wg ? a.f.sync + SYNC_BOX_ATT + (size_t)bid * O3V_SYNC_STRIDE
                            : a.f.sync + SYNC_BOX_O + (size_t)(bid - a.f.nb_attn) * O3V_SYNC_STRIDE;
    const int G = bid * 4 + wave, NW = a.n_wg * 4, NWo = (a.n_wg - a.f.nb_attn) * 4;
    const GuRows R{G, NW, a.I};
    const int n_rows = 2 * R.pairs();  // >= 6 (the launcher checks I >= 3 * NW)
    // this wave's o_proj rows (none in an attention workgroup): Go, Go + NWo
    const int Go = attn_wg ? a.f.H : (bid - a.f.nb_attn) * 4 + wave;
    const int n_o = Go >= a.f.H ? 0 : (Go + NWo < a.f.H ? 2 : 1);
    u32x4 b0[NS], b1[NS], b2[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) b0[s2] = b1[s2] = b2[s2] = (u32x4){0, 0, 0, 0};
    uint32_t res0 = 0, res1 = 0;  // residuals of the o_proj rows (raw bf16)
    // ---- t = 0.  Every load of this phase is issued and waited for by hand (see lb_load), oldest first:
    //   x chunks + norm-weight chunks of the RMSNorm (2 + 2 per thread), the epilogue scalars of the q/k/v pair (bias, cos, sin),
    //   the pair's two weight rows -- then the sum of squares, and only then the first rows of the wave's stream (they have the
    //   whole attention chain to land; the pair is on the critical path).
    const int Pq = (Hq + 2 * Hkv) * D / 2, half = D >> 1;
    const bool has_q = G < Pq;  // uniform per workgroup: Pq % 4 == 0 (the launcher checks)
    const int row0 = has_q ? (G / half) * D + (G % half) : 0;
    constexpr int XCQ = 2;
    u32x4 xr[XCQ], wn[XCQ];
    {
        const int nxc = a.f.H >> 3, tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < XCQ; ++i) {
            const int c = tid + i * 256;
            xr[i] = asm_load16(a.f.x + (size_t)(c < nxc ? c : 0) * 8);
            wn[i] = asm_load16(a.f.ln_w + (size_t)(c < nxc ? c : 0) * 8);
        }
    }
    uint32_t bias0 = 0, bias1 = 0, cs = 0x3f80u, sn = 0;  // raw bf16; cos = 1, sin = 0
    if (has_q) {
        if (a.f.qkv_b) {
            bias0 = asm_load_bf16(a.f.qkv_b + row0);
            bias1 = asm_load_bf16(a.f.qkv_b + row0 + half);
        }
        const size_t ci = (size_t)a.f.ra.cs_off * D + (row0 % D);
        cs = asm_load_bf16(a.f.ra.cosT + ci);
        sn = asm_load_bf16(a.f.ra.sinT + ci);
        // A CU's vector-memory pipeline serves its waves' requests in arrival order: the few KB of x / norm weights that the OTHER
        // workgroups of this CU are about to ask for must not queue behind this workgroup's 56 KiB of rows (measured: x landed after
        // 4.6 us in the median, 13 us at worst, instead of 1.8).  All workgroups start within 0.3 us: half a microsecond of sleep.
        __builtin_amdgcn_s_sleep(20);
        lb_load<NS>(b0, a.f.qkv_w, row0);
        lb_load<NS>(b1, a.f.qkv_w, row0 + half);
    }
    // x / norm weights have landed once at most the younger operations are outstanding: the pair's 2 NS rows (its scalars, older
    // than the rows, are then there as well)
    if (has_q)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < XCQ; ++i) {
        asm volatile("" : "+v"(xr[i]), "+v"(wn[i]));
        if ((int)threadIdx.x + i * 256 >= (a.f.H >> 3)) {
            xr[i] = (u32x4){0, 0, 0, 0};
            wn[i] = (u32x4){0, 0, 0, 0};
        }
    }
    O3V_LSTAMP(1);
    norm_finish<256, XCQ>(xr, wn, a.f.eps, a.f.H, smem);
    if (has_q) {
        lb_wait<NS>(b0, 1);  // the pair's second row is younger
        lb_wait<NS>(b1, 0);
        asm_tie(bias0);
        asm_tie(bias1);
        asm_tie(cs);
        asm_tie(sn);
        lb_qkv_pair<NS>(a, b0, b1, row0, bf2f((bf16_t)bias0), bf2f((bf16_t)bias1), bf2f((bf16_t)cs), bf2f((bf16_t)sn), smem);
    }
    O3V_LSTAMP(2);
    // E1, per kv head as in the role-per-workgroup block: the 16 workgroups of a q / k / v head (64 rotary pairs) take tickets on the
    // head group's line; the last of its (n_rep + 2) * 16 tickets tells that kv head's attention items
    if (has_q) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            const int head = (bid * 4) / half;
            const int g = head < Hq ? head / n_rep : (head - Hq) % Hkv;
            uint32_t old = 0;
            if (threadIdx.x == 0) old = __hip_atomic_fetch_add(a.f.sync + SYNC_QKV + g * O3V_SYNC_STRIDE, 1u, O3V_RLX_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == a.f.epoch * (uint32_t)((n_rep + 2) * (D >> 3)) - 1u)
                notify_mailboxes(a.f.sync + SYNC_BOX_ATT + (size_t)g * a.f.nsplit * O3V_SYNC_STRIDE, a.f.nsplit, 0, a.f.epoch);
        }
    }
    O3V_LSTAMP(3);
    if (!attn_wg) {
        // The stream starts, PACED: a row is requested when the one before it has landed, so a wave has 7 KiB in flight, not 21 --
        // the CU's memory pipeline is a queue, and everything the attention chain does on this CU (K/V tiles, q, partials, tickets,
        // polls) waits behind whatever bulk is queued there.  Three rows per wave still land long before the chain ends.
        if (n_o >= 1) {
            res0 = asm_load_bf16(a.f.x + Go);  // requested ahead of its row: loads return in order, it is there when the row is
            lb_load<NS>(b2, a.f.o_w, Go);
        } else {
            lb_load<NS>(b2, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(1)));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lb_load<NS>(b1, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(0)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (n_o == 2) {
            res1 = asm_load_bf16(a.f.x + Go + NWo);
            lb_load<NS>(b0, a.f.o_w, Go + NWo);
        } else {
            lb_load<NS>(b0, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(2)));
        }
        // ---- o_proj: its operand is the attention output, once every kv head's merged slices are in memory (E3).  The rows in the
        // buffers were requested 10+ us ago: vmcnt(0) costs nothing here.
        if (wave == 0) spin_until<8>(box, Hkv, a.f.epoch, a.f.sync + SYNC_TMO, 0x200u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lds_barrier();
        {
            const int nch = (Hq * D) >> 3;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.f.att, 0, Hq * D * 2, 0x00020000);
            for (int c = threadIdx.x; c < nch; c += 256) *reinterpret_cast<u32x4*>(smem + (size_t)c * 16) = load16_sc1(rs, (uint32_t)c * 16);
        }
        lds_barrier();
        O3V_LSTAMP(4);
        lb_wait<NS>(b2, 0);
        lb_wait<NS>(b0, 0);
        asm_tie(res0);
        asm_tie(res1);
        if (n_o >= 1) {
            const float acc = lb_dot<NS>(b2, smem);
            if (lane == 0) {
                float v = acc + 0.f;  // no bias (TF:620)
                v = rbf(v) + bf2f((bf16_t)res0);
                gemv_store_bf16<true>(a.f.xout + Go, f2bf(v));
            }
        }
        if (n_o == 2) {
            const float acc = lb_dot<NS>(b0, smem);
            if (lane == 0) {
                float v = acc + 0.f;
                v = rbf(v) + bf2f((bf16_t)res1);
                gemv_store_bf16<true>(a.f.xout + Go + NWo, f2bf(v));
            }
        }
        lb_ticket<false>(a);  // E4: this workgroup's rows of the residual stream are in memory (every wave arrives, also one without rows)
        // one freed buffer takes its gate/up row BEFORE the wait for x' (it travels during the hand-off), the other after it (the
        // x' chunks must not queue behind two rows per wave)
        if (n_o >= 1) lb_load<NS>(b2, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(1)));
        O3V_LSTAMP(5);
        lb_enter_gu(a, box, smem);
        if (n_o == 2) lb_load<NS>(b0, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(2)));
        O3V_LSTAMP(6);
        // rows requested after row 0 (buffer 1): row 2 or its refill, and row 1's refill; after row 1 (buffer 2): see the order above
        lb_gu_loop<NS>(b0, b1, b2, a, R, n_rows, n_o >= 1 ? 2 : 1, n_o == 1 ? 1 : 2, smem);
        O3V_LSTAMP(7);
        return;
    }
    // ---- attention workgroup: one (kv head, context split) item -- no weight row is live across it -- then its gate/up rows
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
    O3V_LSTAMP(4);
    u32x4 c0[NS], c1[NS], c2[NS];
    lb_load<NS>(c1, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(0)));
    lb_enter_gu(a, box, smem);
    lb_load<NS>(c2, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(1)));
    lb_load<NS>(c0, a.gu_w, __builtin_amdgcn_readfirstlane(R.row(2)));
    O3V_LSTAMP(6);
    lb_gu_loop<NS>(c0, c1, c2, a, R, n_rows, 2, 2, smem);
    O3V_LSTAMP(7);
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
    if (!device_is_gfx950(prop)) return 0;
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
    if (n_o_wg > 1024 || NQKV / 2 > n_wg * 4 || ((NQKV / 2) & 3) || H > 2 * n_o_wg * 4 || I < 2 * n_wg * 4) return O3V_ERR_SHAPE;  // one rotary pair, two o_proj rows per wave; the waves of a workgroup all own a pair or none
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
    a.f.stamps = g_stamps;
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
