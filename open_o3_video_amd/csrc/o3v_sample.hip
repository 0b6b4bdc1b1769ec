// Rollout sampler: repetition penalty -> temperature -> top-p -> multinomial, one 1024-thread block per
// sequence, no sort.  Restates TF:generation/logits_process.py:404-414 (penalty), :301-303 (temperature),
// :527-539 (top-p: drop the ascending-sorted prefix whose cumulative probability is <= 1-top_p, keep >= 1)
// and TF:generation/utils.py:2921-2923 (softmax -> multinomial) as used by the GSPO rollout
// (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:306-313: do_sample, top_p 0.95, temperature 1).
// The top-p cut is found by a radix select on the probability bit pattern (kept set {p >= tau}); the draw walks the
// kept mass in index order.  RNG: counter-based (splitmix64 of seed, completion id, step) so a completion is
// reproducible wherever it is generated.
#include "o3v_common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__device__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
    return t;
}

__global__ __launch_bounds__(1024) void sample_top_p_kernel(const bf16_t* __restrict__ logits, uint8_t* __restrict__ seen,
                                                            int* __restrict__ cur_tok, int* __restrict__ finished,
                                                            int* __restrict__ out_ids, float* __restrict__ chosen_lp,
                                                            const int* __restrict__ eos_ids, int n_eos, int pad_id, int V,
                                                            int ldl, float rep_penalty, float temperature, float top_p,
                                                            uint64_t seed, const int* __restrict__ row_id, int step,
                                                            int out_stride, float* __restrict__ scratch) {
    __shared__ float red[16];
    __shared__ float csum[1024];
    __shared__ float hist[4096];
    __shared__ int s_pick;
    __shared__ float s_below;
    const int b = blockIdx.x, tid = threadIdx.x;
    const bf16_t* lr = logits + (size_t)b * ldl;
    uint8_t* sr = seen + (size_t)b * V;
    float* pr = scratch + (size_t)b * V;

    // 1. processed scores, max
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) {
        float s = bf2f(lr[i]);
        if (rep_penalty != 1.0f && sr[i]) s = (s < 0.f) ? s * rep_penalty : s / rep_penalty;
        s = s / temperature;
        pr[i] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, red);
    // 2. probabilities
    float z = 0.f;
    for (int i = tid; i < V; i += 1024) {
        const float e = expf(pr[i] - mx);
        pr[i] = e;
        z += e;
    }
    z = block_sum(z, red);
    const float invz = 1.0f / z;
    // 3. top-p threshold by a 3-level radix select on the bit pattern of p (positive floats order like their bits):
    //    smallest tau with M(tau) = sum_{p <= tau} p > 1 - top_p; the kept set is {p >= tau} (TF keeps the complement of
    //    the ascending prefix with cumulative probability <= 1 - top_p).  Each level histograms probability MASS over
    //    12 / 12 / 7 bits of the elements that match the prefix found so far: 3 passes instead of a 30-pass bisection.
    float lo = 0.f;  // kept: p > lo  (lo = largest representable value below tau)
    if (top_p < 1.0f) {
        const float cut = 1.0f - top_p;
        unsigned prefix = 0;       // bits of tau decided so far (high bits)
        float below = 0.f;         // mass of all p whose high bits are < prefix
        const int shifts[3] = {19, 7, 0};
        const int widths[3] = {12, 12, 7};
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int sh = shifts[lvl], nb = 1 << widths[lvl];
            for (int i = tid; i < nb; i += 1024) hist[i] = 0.f;
            __syncthreads();
            const unsigned hi_mask = (lvl == 0) ? 0u : (0xffffffffu << (sh + widths[lvl]));
            for (int i = tid; i < V; i += 1024) {
                const float p = pr[i] * invz;
                const unsigned bits = __float_as_uint(p);
                if ((bits & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(bits >> sh) & (nb - 1)], p);
            }
            __syncthreads();
            // exclusive scan of the bins in ascending order; every thread owns nb/1024 (>= 1) consecutive bins
            const int per = nb >= 1024 ? nb / 1024 : 1;
            float mine = 0.f;
            if (tid * per < nb)
                for (int k = 0; k < per; ++k) mine += hist[tid * per + k];
            float incl = mine;  // inclusive scan across the block
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float t = __shfl_up(incl, o, 64);
                if ((tid & 63) >= o) incl += t;
            }
            if ((tid & 63) == 63) red[tid >> 6] = incl;
            __syncthreads();
            float wave_off = 0.f;
            for (int w = 0; w < (tid >> 6); ++w) wave_off += red[w];
            const float before = below + wave_off + incl - mine;
            if (tid == 0) s_pick = -1;
            __syncthreads();
            if (tid * per < nb && mine > 0.f && before <= cut && before + mine > cut) {
                float run = before;
                for (int k = 0; k < per; ++k) {
                    const float h = hist[tid * per + k];
                    if (h > 0.f && run + h > cut) {
                        s_pick = tid * per + k;
                        s_below = run;
                        break;
                    }
                    run += h;
                }
            }
            __syncthreads();
            if (s_pick < 0) {  // rounding left the crossing undetected: take the highest non-empty bin
                if (tid == 0) {
                    float run = below;
                    int lastb = 0;
                    float lastrun = below;
                    for (int k = 0; k < nb; ++k)
                        if (hist[k] > 0.f) {
                            lastb = k;
                            lastrun = run;
                            run += hist[k];
                        }
                    s_pick = lastb;
                    s_below = lastrun;
                }
                __syncthreads();
            }
            prefix |= ((unsigned)s_pick) << sh;
            below = s_below;
            __syncthreads();
        }
        // tau = prefix (all 31 value bits decided); keep p >= tau  <=>  p > tau_minus
        lo = __uint_as_float(prefix > 0 ? prefix - 1 : 0u);
    }
    // 4. kept mass per contiguous index chunk, then the draw
    const int chunk = (V + 1023) / 1024;
    const int i0 = tid * chunk, i1 = (i0 + chunk < V) ? i0 + chunk : V;
    float mine = 0.f;
    for (int i = i0; i < i1; ++i) {
        const float p = pr[i] * invz;
        mine += (p > lo) ? p : 0.f;
    }
    csum[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < 1024; ++i) tot += csum[i];
        const uint64_t rid = row_id ? (uint64_t)row_id[b] : (uint64_t)b;
        const uint64_t r = splitmix64(splitmix64(seed ^ (rid * 0xD1B54A32D192ED03ull)) + (uint64_t)step);
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
        float target = u * tot;
        int t = 0;
        float run = 0.f;
        for (; t < 1023; ++t) {
            if (run + csum[t] > target) break;
            run += csum[t];
        }
        // walk thread t's chunk
        int pick = -1;
        const int a0 = t * chunk, a1 = (a0 + chunk < V) ? a0 + chunk : V;
        int last_kept = -1;
        for (int i = a0; i < a1; ++i) {
            const float p = pr[i] * invz;
            if (p > lo) {
                last_kept = i;
                run += p;
                if (run > target) {
                    pick = i;
                    break;
                }
            }
        }
        if (pick < 0) pick = last_kept;
        if (pick < 0) {  // numerical corner: fall back to the arg-max (always kept)
            float best = -1.f;
            for (int i = 0; i < V; ++i)
                if (pr[i] > best) {
                    best = pr[i];
                    pick = i;
                }
        }
        int tok = finished[b] ? pad_id : pick;
        if (!finished[b])
            for (int k = 0; k < n_eos; ++k)
                if (tok == eos_ids[k]) finished[b] = 1;
        if (tok >= 0 && tok < V) sr[tok] = 1;
        cur_tok[b] = tok;
        out_ids[(size_t)b * out_stride + step] = tok;
        if (chosen_lp) chosen_lp[(size_t)b * out_stride + step] = logf(pr[pick] * invz);
        s_pick = pick;
    }
}

}  // namespace

extern "C" int o3v_sample_top_p(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids,
                                float* chosen_logprob, const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl,
                                float rep_penalty, float temperature, float top_p, uint64_t seed, const int* row_id, int step,
                                int out_stride, float* scratch, hipStream_t stream) {
    if (!logits || !seen || !cur_tok || !finished || !out_ids || !scratch || B < 0 || V <= 0 || step < 0 ||
        step >= out_stride || !(temperature > 0.f) || !(top_p > 0.f))
        return O3V_ERR_ARG;
    if (B == 0) return O3V_OK;
    O3V_KLAUNCH(sample_top_p_kernel, dim3(B), dim3(1024), 0, stream, (const bf16_t*)logits, (uint8_t*)seen, cur_tok,
                       finished, out_ids, chosen_logprob, eos_ids, n_eos, pad_id, V, ldl, rep_penalty, temperature,
                       top_p, seed, row_id, step, out_stride, scratch);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}
