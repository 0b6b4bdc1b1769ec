// Rollout sampler: repetition penalty -> temperature -> top-k -> top-p -> multinomial, 32 blocks per sequence, no sort.
// Restates TF:generation/logits_process.py:404-414 (penalty), :301-303 (temperature), :590-594 (top-k: drop every score
// below the k-th largest; ties with it stay), :527-539 (top-p over what top-k kept: drop the ascending-sorted prefix whose
// cumulative probability is <= 1-top_p, keep >= 1)
// and TF:generation/utils.py:2921-2923 (softmax -> multinomial) as used by the GSPO rollout
// (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:306-313: do_sample, top_p 0.95, temperature 1).
//
// The row (152064 bf16 logits = 300 KB, L2-resident right after the lm_head) is streamed five times with 16-byte
// loads, eight groups in flight per thread; nothing is written back.  Probability mass is carried as 2^-40 fixed point
// (e = exp(s - max) <= 1), so every sum -- the normaliser, the radix-select histograms, the draw -- is an integer sum:
// exact and independent of the order the atomics land in, hence reproducible.
//   pass A  max of the processed scores
//   top-k   (top_k > 0 only) three count histograms over 12 / 12 / 8 bits of the score's order-preserving key, then one
//           block per row walks them from the top to the key of the k-th largest score; the passes below skip smaller keys
//   pass B  z = sum e and the level-0 histogram of mass over the top 12 bits of e (values below 2^-24 are lumped)
//   pass C/D levels 1 and 2 (12 and 7 more bits) of the radix select for the smallest tau with mass{e <= tau} > (1-top_p) z
//   pass E  kept mass {e >= tau} per thread, block scan, the thread holding the drawn target walks its own elements
// RNG: counter-based (splitmix64 of seed, completion id, step) so a completion is reproducible wherever it is generated
// (SURVEY.md section 8e).
#include "o3v_common.h"

namespace {

constexpr int NT = 256, NBLK = 32;                  // threads per block, blocks per row
constexpr float FX_SCALE = 1099511627776.0f;        // 2^40

typedef unsigned long long u64;

// per-row workspace (u64 units) inside the caller's scratch: O3V_SAMPLE_SCRATCH_FLOATS floats per row
constexpr int WS_H0 = 0, WS_H1 = 4096, WS_H2 = 8192, WS_BM = 8320, WS_Z = 8352, WS_PREFIX = 8353, WS_KTH = 8354;
constexpr int WS_C0 = 8356, WS_C1 = WS_C0 + 4096, WS_C2 = WS_C1 + 4096;   // top-k count histograms (descending key order)
constexpr int WS_PMAX = WS_C2 + 256;                                      // everything below is cleared by pass A
constexpr int WS_U64 = WS_PMAX + NBLK / 2;                                // pmax: NBLK floats
constexpr int WS_FLOATS = 40960;
static_assert(WS_U64 * 2 <= WS_FLOATS, "row workspace");

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m, 64), hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int d) {
    const unsigned lo = __shfl_up((unsigned)v, d, 64), hi = __shfl_up((unsigned)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}

__device__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}
__device__ u64 block_sum_u64(u64 v, u64* red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 t = 0;
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
// exclusive prefix of `v` over the block in thread order; *total = block sum
__device__ u64 block_scan_excl_u64(u64 v, u64* red, u64* total) {
    u64 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = shfl_up_u64(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 off = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < (int)(threadIdx.x >> 6)) off += red[w];
        tot += red[w];
    }
    *total = tot;
    return off + incl - v;
}

struct Row {
    const bf16_t* lr;
    const uint8_t* sr;
    int V;
    bool vec;
    float rep, temp;
};

__device__ __forceinline__ Row make_row(const bf16_t* logits, const uint8_t* seen, int b, int V, int ldl, float rep, float temp) {
    Row r;
    r.lr = logits + (size_t)b * ldl;
    r.sr = seen + (size_t)b * V;
    r.V = V;
    r.vec = ((V & 7) == 0) && ((ldl & 7) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) &&
            ((reinterpret_cast<uintptr_t>(seen) & 7) == 0);
    r.rep = rep;
    r.temp = temp;
    return r;
}

// processed scores of elements 8g..8g+7 (TF:404-414 penalty, :301-303 temperature); -inf past the vocabulary
__device__ __forceinline__ void load_group(const Row& r, int g, float s[8]) {
    const int base = g * 8;
    if (r.vec && base + 8 <= r.V) {
        const u32x4 lv = *reinterpret_cast<const u32x4*>(r.lr + base);
        const uint2 sv = *reinterpret_cast<const uint2*>(r.sr + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = bf2f((bf16_t)((lv[k >> 1] >> (16 * (k & 1))) & 0xffffu));
            const unsigned sb = ((k < 4 ? sv.x : sv.y) >> (8 * (k & 3))) & 0xffu;
            if (r.rep != 1.0f && sb) x = (x < 0.f) ? x * r.rep : x / r.rep;
            s[k] = x / r.temp;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + k;
            float x = -INFINITY;
            if (i < r.V) {
                x = bf2f(r.lr[i]);
                if (r.rep != 1.0f && r.sr[i]) x = (x < 0.f) ? x * r.rep : x / r.rep;
                x = x / r.temp;
            }
            s[k] = x;
        }
    }
}

// Visits the elements of block `blk`'s slice of the row: thread t owns groups g0+t, g0+t+NT, ...
template <typename F>
__device__ __forceinline__ void for_each_score(const Row& r, int blk, F&& f) {
    const int NG = (r.V + 7) / 8, per_blk = (NG + NBLK - 1) / NBLK;
    const int g0 = blk * per_blk, g1 = (g0 + per_blk < NG) ? g0 + per_blk : NG;
    for (int g = g0 + threadIdx.x; g < g1; g += NT) {
        float s[8];
        load_group(r, g, s);
#pragma unroll
        for (int k = 0; k < 8; ++k) f(g * 8 + k, s[k]);
    }
}

__device__ __forceinline__ u64 to_fx(float e) { return (u64)(e * FX_SCALE); }

// order-preserving 32-bit key of a score (larger score <=> larger key); kth key 0 keeps everything
__device__ __forceinline__ unsigned score_key(float s) {
    const unsigned b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float row_max(const u64* ws) {
    const float* pm = reinterpret_cast<const float*>(ws + WS_PMAX);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) m = fmaxf(m, pm[i]);
    return m;
}

// Bin of `hist[0..nb)` in which the running mass (starting at *below) first exceeds `cut`; *below becomes the mass
// before that bin.  Every thread of the block returns the same value.
__device__ int pick_bin(const u64* __restrict__ hist, int nb, u64 cut, u64* below, u64* red, int* s_pick, u64* s_below) {
    const int per = (nb + NT - 1) / NT, tid = threadIdx.x;
    const int k0 = tid * per, k1 = (k0 + per < nb) ? k0 + per : nb;
    u64 mine = 0;
    for (int k = k0; k < k1; ++k) mine += hist[k];
    u64 tot;
    const u64 before = *below + block_scan_excl_u64(mine, red, &tot);
    if (tid == 0) *s_pick = -1;
    __syncthreads();
    if (mine > 0 && before <= cut && before + mine > cut) {
        u64 run = before;
        for (int k = k0; k < k1; ++k) {
            const u64 h = hist[k];
            if (h > 0 && run + h > cut) {
                *s_pick = k;
                *s_below = run;
                break;
            }
            run += h;
        }
    }
    __syncthreads();
    if (*s_pick < 0) {  // cut >= all the mass at this level (top_p ~ 0): take the highest non-empty bin
        if (tid == 0) {
            u64 run = *below, lastrun = *below;
            int lastb = 0;
            for (int k = 0; k < nb; ++k)
                if (hist[k] > 0) {
                    lastb = k;
                    lastrun = run;
                    run += hist[k];
                }
            *s_pick = lastb;
            *s_below = lastrun;
        }
        __syncthreads();
    }
    const int pick = *s_pick;
    *below = *s_below;
    __syncthreads();
    return pick;
}

// K1: slice maxima of the processed scores; clears the row's accumulators for the kernels that follow
__global__ __launch_bounds__(NT) void sample_max_kernel(const bf16_t* __restrict__ logits, const uint8_t* __restrict__ seen,
                                                        u64* __restrict__ wsp, int V, int ldl, float rep, float temp) {
    __shared__ float redf[NT / 64];
    const int blk = blockIdx.x, b = blockIdx.y;
    u64* ws = wsp + (size_t)b * (WS_FLOATS / 2);
    const Row r = make_row(logits, seen, b, V, ldl, rep, temp);
    for (int i = blk * NT + threadIdx.x; i < WS_PMAX; i += NBLK * NT) ws[i] = 0;
    float mx = -INFINITY;
    for_each_score(r, blk, [&](int, float s) { mx = fmaxf(mx, s); });
    mx = block_max(mx, redf);
    if (threadIdx.x == 0) reinterpret_cast<float*>(ws + WS_PMAX)[blk] = mx;
}

// top-k, levels 0..2: histogram of COUNTS over 12 / 12 / 8 bits of the inverted key (bin 0 = largest scores), restricted to
// the bins the earlier levels chose for the k-th largest element
template <int LVL>
__global__ __launch_bounds__(NT) void sample_count_kernel(const bf16_t* __restrict__ logits, const uint8_t* __restrict__ seen,
                                                          u64* __restrict__ wsp, int V, int ldl, float rep, float temp, int top_k) {
    constexpr int SH = LVL == 0 ? 20 : (LVL == 1 ? 8 : 0), NB = LVL == 2 ? 256 : 4096;
    __shared__ unsigned hist[NB];
    __shared__ u64 red[NT / 64];
    __shared__ int s_pick;
    __shared__ u64 s_below;
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    u64* ws = wsp + (size_t)b * (WS_FLOATS / 2);
    const Row r = make_row(logits, seen, b, V, ldl, rep, temp);
    for (int i = tid; i < NB; i += NT) hist[i] = 0;
    unsigned prefix = 0, hi_mask = 0;
    if (LVL > 0) {
        const u64 cut = (u64)(top_k - 1);
        u64 below = 0;
        prefix = (unsigned)pick_bin(ws + WS_C0, 4096, cut, &below, red, &s_pick, &s_below) << 20;
        hi_mask = 0xffffffffu << 20;
        if (LVL > 1) {
            prefix |= (unsigned)pick_bin(ws + WS_C1, 4096, cut, &below, red, &s_pick, &s_below) << 8;
            hi_mask = 0xffffffffu << 8;
        }
    }
    __syncthreads();
    for_each_score(r, blk, [&](int i, float s) {
        if (i >= V) return;
        const unsigned inv = ~score_key(s);
        if ((inv & hi_mask) == prefix) atomicAdd(&hist[(inv >> SH) & (NB - 1)], 1u);
    });
    __syncthreads();
    u64* gh = ws + (LVL == 0 ? WS_C0 : (LVL == 1 ? WS_C1 : WS_C2));
    for (int i = tid; i < NB; i += NT)
        if (hist[i]) atomicAdd(gh + i, (u64)hist[i]);
}

// top-k: one block per row resolves the key of the k-th largest score (fewer than k scores: the smallest one)
__global__ __launch_bounds__(NT) void sample_kth_kernel(u64* __restrict__ wsp, int top_k) {
    __shared__ u64 red[NT / 64];
    __shared__ int s_pick;
    __shared__ u64 s_below;
    u64* ws = wsp + (size_t)blockIdx.x * (WS_FLOATS / 2);
    const u64 cut = (u64)(top_k - 1);
    u64 below = 0;
    unsigned inv = (unsigned)pick_bin(ws + WS_C0, 4096, cut, &below, red, &s_pick, &s_below) << 20;
    inv |= (unsigned)pick_bin(ws + WS_C1, 4096, cut, &below, red, &s_pick, &s_below) << 8;
    inv |= (unsigned)pick_bin(ws + WS_C2, 256, cut, &below, red, &s_pick, &s_below);
    if (threadIdx.x == 0) ws[WS_KTH] = (u64)(~inv);
}

// K2..K4 (LVL 0,1,2): mass histogram over 12 / 12 / 7 bits of e = exp(s - max) for the elements that match the bits
// decided by the earlier levels.  Radix select of the smallest tau with mass{e <= tau} > (1 - top_p) z: the kept set
// is {e >= tau} (TF keeps the complement of the ascending prefix with cumulative probability <= 1 - top_p).
template <int LVL>
__global__ __launch_bounds__(NT) void sample_hist_kernel(const bf16_t* __restrict__ logits, const uint8_t* __restrict__ seen,
                                                         u64* __restrict__ wsp, int V, int ldl, float rep, float temp,
                                                         float top_p) {
    constexpr int SH = LVL == 0 ? 19 : (LVL == 1 ? 7 : 0), NB = LVL == 2 ? 128 : 4096;
    __shared__ u64 hist[NB];
    __shared__ u64 red[NT / 64];
    __shared__ int s_pick;
    __shared__ u64 s_below;
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    u64* ws = wsp + (size_t)b * (WS_FLOATS / 2);
    const Row r = make_row(logits, seen, b, V, ldl, rep, temp);
    const float mx = row_max(ws);
    for (int i = tid; i < NB; i += NT) hist[i] = 0;
    unsigned prefix = 0, hi_mask = 0;
    if (LVL > 0) {
        const u64 cut = (u64)((double)(1.0f - top_p) * (double)ws[WS_Z]);
        u64 below = 0;
        prefix = (unsigned)pick_bin(ws + WS_H0, 4096, cut, &below, red, &s_pick, &s_below) << 19;
        hi_mask = 0xffffffffu << 19;
        if (LVL > 1) {
            prefix |= (unsigned)pick_bin(ws + WS_H1, 4096, cut, &below, red, &s_pick, &s_below) << 7;
            hi_mask = 0xffffffffu << 7;
        }
    }
    __syncthreads();
    u64 zsum = 0;
    const unsigned kth = (unsigned)ws[WS_KTH];
    for_each_score(r, blk, [&](int, float s) {
        if (score_key(s) < kth) return;   // removed by top-k
        const float e = expf(s - mx);
        const u64 fx = to_fx(e);
        const unsigned bits = __float_as_uint(e);
        if (LVL == 0) zsum += fx;
        if (fx && (bits & hi_mask) == prefix) atomicAdd(&hist[(bits >> SH) & (NB - 1)], fx);
    });
    if (LVL == 0) {
        zsum = block_sum_u64(zsum, red);
        if (tid == 0) atomicAdd(ws + WS_Z, zsum);
    }
    __syncthreads();
    u64* gh = ws + (LVL == 0 ? WS_H0 : (LVL == 1 ? WS_H1 : WS_H2));
    for (int i = tid; i < NB; i += NT)
        if (hist[i]) atomicAdd(gh + i, hist[i]);
}

// K5: kept mass {bits(e) >= tau} of every slice; publishes tau (all blocks of a row compute the same value)
__global__ __launch_bounds__(NT) void sample_mass_kernel(const bf16_t* __restrict__ logits, const uint8_t* __restrict__ seen,
                                                         u64* __restrict__ wsp, int V, int ldl, float rep, float temp,
                                                         float top_p) {
    __shared__ u64 red[NT / 64];
    __shared__ int s_pick;
    __shared__ u64 s_below;
    const int blk = blockIdx.x, b = blockIdx.y;
    u64* ws = wsp + (size_t)b * (WS_FLOATS / 2);
    const Row r = make_row(logits, seen, b, V, ldl, rep, temp);
    const float mx = row_max(ws);
    unsigned prefix = 0;
    if (top_p < 1.0f) {
        const u64 cut = (u64)((double)(1.0f - top_p) * (double)ws[WS_Z]);
        u64 below = 0;
        prefix = (unsigned)pick_bin(ws + WS_H0, 4096, cut, &below, red, &s_pick, &s_below) << 19;
        prefix |= (unsigned)pick_bin(ws + WS_H1, 4096, cut, &below, red, &s_pick, &s_below) << 7;
        prefix |= (unsigned)pick_bin(ws + WS_H2, 128, cut, &below, red, &s_pick, &s_below);
    }
    u64 mine = 0;
    const unsigned kth = (unsigned)ws[WS_KTH];
    for_each_score(r, blk, [&](int, float s) {
        const float e = expf(s - mx);
        if (score_key(s) >= kth && __float_as_uint(e) >= prefix) mine += to_fx(e);
    });
    mine = block_sum_u64(mine, red);
    if (threadIdx.x == 0) {
        ws[WS_BM + blk] = mine;
        ws[WS_PREFIX] = prefix;
    }
}

// K6: the draw.  Elements are ordered by slice, then by owning thread, then by that thread's visiting order; the block
// and then the thread whose mass range holds the target walk to the token.
__global__ __launch_bounds__(NT) void sample_draw_kernel(const bf16_t* __restrict__ logits, uint8_t* __restrict__ seen,
                                                         const u64* __restrict__ wsp, int* __restrict__ cur_tok,
                                                         int* __restrict__ finished, int* __restrict__ out_ids,
                                                         float* __restrict__ chosen_lp, const int* __restrict__ eos_ids,
                                                         int n_eos, int pad_id, int V, int ldl, float rep, float temp,
                                                         uint64_t seed, const int* __restrict__ row_id, int step,
                                                         int out_stride) {
    __shared__ u64 red[NT / 64];
    const int blk = blockIdx.x, b = blockIdx.y;
    const u64* ws = wsp + (size_t)b * (WS_FLOATS / 2);
    u64 tot = 0, blk_before = 0;
    for (int i = 0; i < NBLK; ++i) {
        if (i < blk) blk_before += ws[WS_BM + i];
        tot += ws[WS_BM + i];
    }
    const uint64_t rid = row_id ? (uint64_t)row_id[b] : (uint64_t)b;
    const uint64_t rnd = splitmix64(splitmix64(seed ^ (rid * 0xD1B54A32D192ED03ull)) + (uint64_t)step);
    const double u = (double)(rnd >> 40) * (1.0 / 16777216.0);
    u64 target = (u64)(u * (double)tot);
    if (tot > 0 && target >= tot) target = tot - 1;
    const u64 blk_mass = ws[WS_BM + blk];
    if (!(blk_mass > 0 && blk_before <= target && target < blk_before + blk_mass)) return;  // block-uniform
    const Row r = make_row(logits, seen, b, V, ldl, rep, temp);
    const float mx = row_max(ws);
    const unsigned prefix = (unsigned)ws[WS_PREFIX], kth = (unsigned)ws[WS_KTH];
    u64 mine = 0;
    for_each_score(r, blk, [&](int, float s) {
        const float e = expf(s - mx);
        if (score_key(s) >= kth && __float_as_uint(e) >= prefix) mine += to_fx(e);
    });
    u64 btot;
    const u64 before = blk_before + block_scan_excl_u64(mine, red, &btot);
    if (mine > 0 && before <= target && target < before + mine) {
        u64 run = before;
        int pick = -1;
        float e_pick = 0.f;
        for_each_score(r, blk, [&](int i, float s) {
            if (pick >= 0) return;
            const float e = expf(s - mx);
            if (score_key(s) >= kth && __float_as_uint(e) >= prefix) {
                run += to_fx(e);
                if (run > target) {
                    pick = i;
                    e_pick = e;
                }
            }
        });
        const bool fin = finished[b] != 0;
        const int tok = fin ? pad_id : pick;
        if (!fin)
            for (int k = 0; k < n_eos; ++k)
                if (tok == eos_ids[k]) finished[b] = 1;
        if (tok >= 0 && tok < V) seen[(size_t)b * V + tok] = 1;
        cur_tok[b] = tok;
        out_ids[(size_t)b * out_stride + step] = tok;
        if (chosen_lp) chosen_lp[(size_t)b * out_stride + step] = logf(e_pick / ((float)ws[WS_Z] * (1.0f / FX_SCALE)));
    }
}

}  // namespace

extern "C" int o3v_sample_top_k_top_p(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids,
                                      float* chosen_logprob, const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl,
                                      float rep_penalty, float temperature, int top_k, float top_p, uint64_t seed,
                                      const int* row_id, int step, int out_stride, float* scratch, hipStream_t stream) {
    if (!logits || !seen || !cur_tok || !finished || !out_ids || !scratch || B < 0 || V <= 0 || step < 0 ||
        step >= out_stride || !(temperature > 0.f) || !(top_p > 0.f) || top_k < 0 || (reinterpret_cast<uintptr_t>(scratch) & 7))
        return O3V_ERR_ARG;
    if (B == 0) return O3V_OK;
    const bf16_t* lg = (const bf16_t*)logits;
    uint8_t* sn = (uint8_t*)seen;
    u64* ws = (u64*)scratch;
    const dim3 grid(NBLK, B), block(NT);
    O3V_KLAUNCH(sample_max_kernel, grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature);
    if (top_k > 0 && top_k < V) {  // TF:591: top_k = min(top_k, vocabulary); k == V removes nothing
        O3V_KLAUNCH((sample_count_kernel<0>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_k);
        O3V_KLAUNCH((sample_count_kernel<1>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_k);
        O3V_KLAUNCH((sample_count_kernel<2>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_k);
        O3V_KLAUNCH(sample_kth_kernel, dim3(B), block, 0, stream, ws, top_k);
    }
    O3V_KLAUNCH((sample_hist_kernel<0>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_p);
    if (top_p < 1.0f) {
        O3V_KLAUNCH((sample_hist_kernel<1>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_p);
        O3V_KLAUNCH((sample_hist_kernel<2>), grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_p);
    }
    O3V_KLAUNCH(sample_mass_kernel, grid, block, 0, stream, lg, sn, ws, V, ldl, rep_penalty, temperature, top_p);
    O3V_KLAUNCH(sample_draw_kernel, grid, block, 0, stream, lg, sn, ws, cur_tok, finished, out_ids, chosen_logprob, eos_ids, n_eos,
                pad_id, V, ldl, rep_penalty, temperature, seed, row_id, step, out_stride);
    O3V_CHECK_LAUNCH();
    return O3V_OK;
}

extern "C" int o3v_sample_top_p(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids,
                                float* chosen_logprob, const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl,
                                float rep_penalty, float temperature, float top_p, uint64_t seed, const int* row_id, int step,
                                int out_stride, float* scratch, hipStream_t stream) {
    return o3v_sample_top_k_top_p(logits, seen, cur_tok, finished, out_ids, chosen_logprob, eos_ids, n_eos, pad_id, B, V, ldl,
                                  rep_penalty, temperature, 0, top_p, seed, row_id, step, out_stride, scratch, stream);
}
