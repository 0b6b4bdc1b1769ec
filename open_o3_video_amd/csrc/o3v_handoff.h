// In-launch hand-offs between the workgroups of the role-fused decode launches (o3v_fused.hip).
#pragma once
#include "o3v_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Hand-offs between workgroups of ONE launch (o3v_fused.hip).  Protocol = cdna_hip_programming.md Guideline 16, R1:
//   producers   store their payload write-through (sc1); every storing wave drains its stores (s_waitcnt vmcnt(0)), the
//               workgroup meets at a barrier and ONE lane draws a TICKET (returning agent-scope add).  The workgroup
//               that draws the last ticket of the episode knows every payload byte is in memory; it stores the episode's
//               EPOCH into the MAILBOX line of every consumer workgroup (one sc1 store per consumer).
//   consumers   ONE wave polls the workgroup's OWN mailbox line (no other poller, no atomic on that line, so a poll is a
//               quiet L2 hit until the flag lands -- pollers sharing one counter line with its 576 adders saturated that
//               line's memory channel and stretched the q/k/v role's tail from 10 to 16 us), the workgroup meets at a
//               barrier, and EVERY load of the payload is an sc1 load (L1 bypassed: no stale line of this CU can be read;
//               the per-XCD L2s are kept coherent for local HBM by the memory probes).
// What this relies on (gfx950, measured on MI355X in SPX mode; NOT the HSA memory model's release / acquire): (a) an sc1 store is
// written through this XCD's L2 to memory, and the storing wave's `s_waitcnt vmcnt(0)` returns only when that write has been
// acknowledged; (b) an sc1 load bypasses the CU's L1 and is served by the L2 / memory side that holds the written-through line; (c)
// the ticket is a returning agent-scope atomic executed at the L2 / memory side, issued by program order after (a).  Every atomic
// is relaxed: an agent-scope RELEASE would add `buffer_wbl2` (a write-back of the whole L2) per hand-off, an ACQUIRE a `buffer_inv`.
// Because this is behaviour of one part in one partition mode, the launchers refuse the one-launch forms unless the device
// reports gfx950 (o3v_fused.hip: device_is_gfx950), every wait is bounded, and the engine re-runs a call on the stand-alone kernels
// when a wait gives up (engine.generate).
// Nothing is zeroed between launches: the buffer is zeroed once per generate call, the epoch e = 1, 2, ... is the launch's
// index in that call, the tickets count on (the last ticket of episode e is e * want - 1) and a mailbox holds the last
// epoch it was told.  Every spin is bounded and the give-up is sticky (one time-out makes every later wait return at once).
// ------------------------------------------------------------------------------------------------
constexpr int O3V_SYNC_STRIDE = 32;            // 32-bit words: every ticket / mailbox on a 128-byte line of its own
constexpr uint32_t O3V_SPIN_LIMIT = 1u << 16;  // polls of ~0.5-1 us each: a wait gives up after ~50 ms

#define O3V_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// lines of the sync buffer used outside o3v_fused.hip (its own lines: 0..18, 24.. -- see the SYNC_* constants there)
constexpr int O3V_SYNC_TMO_WORD = 16 * O3V_SYNC_STRIDE;     // sticky time-out word (byte 2048 = O3V_SYNC_TMO_BYTE)
constexpr int O3V_SYNC_TAIL_TICKET = 19 * O3V_SYNC_STRIDE;  // TailNorm (o3v_gemm.hip): tickets of the storing waves
constexpr int O3V_SYNC_TAIL_DONE = 20 * O3V_SYNC_STRIDE;    // TailNorm: epoch whose rows are all in memory

// wave-uniform: true once lanes 0..n-1 all read `want` from p[lane] (n <= 64 words of ONE mailbox line)
template <int SLEEP>
__device__ __forceinline__ bool spin_until(uint32_t* p, int n, uint32_t want, uint32_t* tmo, uint32_t code) {
    const int lane = threadIdx.x & 63;
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t v = lane < n ? __hip_atomic_load(p + lane, O3V_RLX_AGENT) : want;
        if (__all(v == want)) return true;
        if (spins >= O3V_SPIN_LIMIT || ((spins & 31u) == 31u && __hip_atomic_load(tmo, O3V_RLX_AGENT) != 0u)) {
            __hip_atomic_store(tmo, code, O3V_RLX_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// ONE wave: store `epoch` into word `word` of n consecutive mailbox lines
__device__ __forceinline__ void notify_mailboxes(uint32_t* box0, int n, int word, uint32_t epoch) {
    for (int i = threadIdx.x & 63; i < n; i += 64) __hip_atomic_store(box0 + (size_t)i * O3V_SYNC_STRIDE + word, epoch, O3V_RLX_AGENT);
}

// 16-byte load that bypasses this CU's L1 (buffer_load_dwordx4 ... sc1); an offset at or past the descriptor's size
// returns zeros without touching memory
__device__ __forceinline__ u32x4 load16_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
constexpr uint32_t O3V_OOB = 0x7ffffff0u;

template <bool SC1>
__device__ __forceinline__ float ldf(const float* p) {
    return SC1 ? __hip_atomic_load(p, O3V_RLX_AGENT) : *p;
}
template <bool SC1>
__device__ __forceinline__ void stf(float* p, float v) {
    if (SC1)
        __hip_atomic_store(p, v, O3V_RLX_AGENT);
    else
        *p = v;
}

}  // namespace
