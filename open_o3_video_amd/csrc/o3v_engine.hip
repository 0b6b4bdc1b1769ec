// Host-side runtime of libo3v_hip.so: the layer loops of the Qwen2.5-VL ViT, LLM prefill and decode,
// enqueued kernel by kernel on one HIP stream from C++ (no Python between launches, no allocation, no
// sync -- the whole forward is graph-capturable).  Arithmetic restated from transformers 5.15.0
// models/qwen2_5_vl/modeling_qwen2_5_vl.py (TF:) -- see include/o3v.h for the per-entry citations.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "../../include/o3v.h"
#include "o3v_common.h"

#define TRY(expr)                 \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != O3V_OK) return rc__; \
    } while (0)

namespace {

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carver {
    char* base;
    size_t off, cap;
    void* take(size_t bytes) {
        size_t o = off;
        off = align256(off + bytes);
        return off <= cap ? base + o : nullptr;
    }
};

constexpr int SPLITK_MAX_ROWS = 128;   // one row tile: the GEMM grid is N/128 blocks, far fewer than the 256 CUs
constexpr int SPLITK_MAX_N = 8192;

// nn.Linear for any row count: MFMA GEMM above 8 rows, weight-streaming GEMV (in groups of 8 rows) otherwise.
// splitk_ws (optional, SPLITK floats): lets a 9..128-row GEMM with few output tiles split K over more blocks.
int linear(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K, int lda,
           int ldo, int ldr, int epi, o3v_stream_t s, float* splitk_ws = nullptr, size_t splitk_bytes = 0, int tile = 0) {
    if (M > 8) {
        const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        if (splitk_ws && M <= SPLITK_MAX_ROWS && N <= SPLITK_MAX_N && tiles < 128 && epi != O3V_EPI_SWIGLU) {
            int splits = 256 / tiles;                       // ~one block per CU
            const int nk = K / 64;
            while (splits > 1 && nk / splits < 4) --splits;  // keep >= 4 k-steps per block
            if (splits > 1 && (size_t)splits * M * N * 4 <= splitk_bytes)
                return o3v_gemm_bf16_splitk(A, W, bias, res, out, M, N, K, lda, K, ldo, ldr, epi, splits, splitk_ws,
                                            splitk_bytes, s);
        }
        return o3v_gemm_bf16_tile(A, W, bias, res, out, M, N, K, lda, K, ldo, ldr, epi, tile, s);
    }
    return o3v_gemv_bf16(A, W, bias, res, out, M, N, K, lda, K, ldo, ldr, epi, s);
}

}  // namespace

extern "C" int o3v_abi_version(void) { return 6; }

// ------------------------------------------------------------------------------------------------ context handle
// Owning host-side copy of the descriptors ("no global mutable state except an explicit o3v_ctx").
#include <new>
#include <vector>
struct o3v_ctx {
    bool has_llm = false, has_vit = false, has_vit3 = false;
    o3v_llm_desc llm{};
    o3v_vit_desc vit{};
    o3v_vit3_desc vit3{};
    std::vector<o3v_llm_layer_w> layers;
    std::vector<o3v_vit_block_w> blocks;
    std::vector<o3v_vit3_block_w> blocks3;
};

extern "C" o3v_ctx* o3v_ctx_create(const o3v_llm_desc* llm, const o3v_vit_desc* vit, const o3v_vit3_desc* vit3) {
    if ((llm && (llm->layers < 0 || (llm->layers > 0 && !llm->layer))) || (vit && (vit->depth < 0 || (vit->depth > 0 && !vit->blocks))) ||
        (vit3 && (vit3->depth < 0 || (vit3->depth > 0 && !vit3->blocks))))
        return nullptr;
    o3v_ctx* c = new (std::nothrow) o3v_ctx();
    if (!c) return nullptr;
    if (llm) {
        c->has_llm = true;
        c->llm = *llm;
        c->layers.assign(llm->layer, llm->layer + llm->layers);
        c->llm.layer = c->layers.data();
    }
    if (vit) {
        c->has_vit = true;
        c->vit = *vit;
        c->blocks.assign(vit->blocks, vit->blocks + vit->depth);
        c->vit.blocks = c->blocks.data();
    }
    if (vit3) {
        c->has_vit3 = true;
        c->vit3 = *vit3;
        c->blocks3.assign(vit3->blocks, vit3->blocks + vit3->depth);
        c->vit3.blocks = c->blocks3.data();
    }
    return c;
}
extern "C" void o3v_ctx_destroy(o3v_ctx* ctx) { delete ctx; }
extern "C" const o3v_llm_desc* o3v_ctx_llm(const o3v_ctx* ctx) { return ctx && ctx->has_llm ? &ctx->llm : nullptr; }
extern "C" const o3v_vit_desc* o3v_ctx_vit(const o3v_ctx* ctx) { return ctx && ctx->has_vit ? &ctx->vit : nullptr; }
extern "C" const o3v_vit3_desc* o3v_ctx_vit3(const o3v_ctx* ctx) { return ctx && ctx->has_vit3 ? &ctx->vit3 : nullptr; }

// ------------------------------------------------------------------------------------------------ ViT
extern "C" size_t o3v_vit_workspace_bytes(const o3v_vit_desc* d, int P) {
    if (!d || P <= 0) return 0;
    const size_t hid = d->hidden, e = 2;
    const size_t Pm = (size_t)P / d->merge_unit;
    size_t n = 0;
    n += align256((size_t)P * hid * e);          // x
    n += align256((size_t)P * hid * e);          // h
    n += align256((size_t)P * 3 * hid * e);      // qkv
    n += align256((size_t)P * hid * e);          // att
    n += align256((size_t)P * d->inter_pad * e); // mlp
    n += align256(Pm * hid * d->merge_unit * e); // m1
    n += align256(Pm * d->out_hidden * e);       // m2
    return n;
}

extern "C" int o3v_vit_forward(const o3v_vit_desc* d, const void* pixels, int P, const int* win_idx, const int* rev_idx,
                               const float* cosT, const float* sinT, const int* tiles_win, int n_tiles_win,
                               const int* tiles_full, int n_tiles_full, void* workspace, size_t ws_bytes, void* out,
                               o3v_stream_t s) {
    if (!d || !pixels || !win_idx || !rev_idx || !cosT || !sinT || !tiles_win || !tiles_full || !workspace || !out)
        return O3V_ERR_ARG;
    if (P <= 0 || d->merge_unit <= 0 || (P % d->merge_unit) || d->heads <= 0 || (d->hidden % d->heads)) return O3V_ERR_ARG;
    if (ws_bytes < o3v_vit_workspace_bytes(d, P)) return O3V_ERR_WORKSPACE;
    const int hid = d->hidden, H = d->heads, D = hid / H, unit = d->merge_unit, Pm = P / unit, ip = d->inter_pad;
    Carver cv{(char*)workspace, 0, ws_bytes};
    char* x = (char*)cv.take((size_t)P * hid * 2);
    char* h = (char*)cv.take((size_t)P * hid * 2);
    char* qkv = (char*)cv.take((size_t)P * 3 * hid * 2);
    char* att = (char*)cv.take((size_t)P * hid * 2);
    char* mlp = (char*)cv.take((size_t)P * ip * 2);
    char* m1 = (char*)cv.take((size_t)Pm * hid * unit * 2);
    char* m2 = (char*)cv.take((size_t)Pm * d->out_hidden * 2);
    if (!m2) return O3V_ERR_WORKSPACE;
    const float scale = 1.0f / sqrtf((float)D);

    // patch embed (Conv3d k=s == GEMM over flattened patches), then window order at merge-unit granularity
    TRY(o3v_gemm_bf16_tile(pixels, d->patch_w, nullptr, nullptr, h, P, hid, d->patch_k_pad, d->patch_k_pad, d->patch_k_pad, hid,
                      0, O3V_EPI_NONE, d->gemm_tile, s));
    TRY(o3v_gather_rows(h, win_idx, x, Pm, unit * hid * 2, s));

    for (int i = 0; i < d->depth; ++i) {
        const o3v_vit_block_w& w = d->blocks[i];
        const bool full = (d->fullatt_mask >> i) & 1;
        TRY(o3v_rmsnorm(x, w.norm1, h, P, hid, hid, hid, 1e-6f, s));
        TRY(o3v_gemm_bf16_tile(h, w.qkv_w, w.qkv_b, nullptr, qkv, P, 3 * hid, hid, hid, hid, 3 * hid, 0, O3V_EPI_NONE, d->gemm_tile, s));
        TRY(o3v_vit_rope(qkv, cosT, sinT, P, H, D, s));
        TRY(o3v_attn_tiles(qkv, qkv + (size_t)hid * 2, qkv + (size_t)2 * hid * 2, att, full ? tiles_full : tiles_win,
                           full ? n_tiles_full : n_tiles_win, 64, H, 1, D, 3L * hid, 3L * hid, D, 0, 3L * hid, D, 0, hid, scale,
                           s));
        TRY(o3v_gemm_bf16_tile(att, w.proj_w, w.proj_b, x, x, P, hid, hid, hid, hid, hid, hid, O3V_EPI_RESIDUAL, d->gemm_tile, s));
        TRY(o3v_rmsnorm(x, w.norm2, h, P, hid, hid, hid, 1e-6f, s));
        TRY(o3v_gemm_bf16_tile(h, w.gu_w, w.gu_b, nullptr, mlp, P, 2 * ip, hid, hid, hid, ip, 0, O3V_EPI_SWIGLU, d->gemm_tile, s));
        TRY(o3v_gemm_bf16_tile(mlp, w.down_w, w.down_b, x, x, P, hid, ip, ip, ip, hid, hid, O3V_EPI_RESIDUAL, d->gemm_tile, s));
    }
    // merger: RMSNorm -> [P/4, 4*hid] -> Linear+GELU -> Linear -> original token order
    TRY(o3v_rmsnorm(x, d->ln_q, h, P, hid, hid, hid, 1e-6f, s));
    const int mh = hid * unit;
    TRY(linear(h, d->m0_w, d->m0_b, nullptr, m1, Pm, mh, mh, mh, mh, 0, O3V_EPI_GELU, s, nullptr, 0, d->gemm_tile));
    TRY(linear(m1, d->m2_w, d->m2_b, nullptr, m2, Pm, d->out_hidden, mh, mh, d->out_hidden, 0, O3V_EPI_NONE, s, nullptr, 0, d->gemm_tile));
    TRY(o3v_gather_rows(m2, rev_idx, out, Pm, d->out_hidden * 2, s));
    return O3V_OK;
}

// ------------------------------------------------------------------------------------------------ ViT, Qwen3-VL
// TF3 = transformers 5.15.0 models/qwen3_vl/modeling_qwen3_vl.py.  Differences from the Qwen2.5-VL tower: patch-embed bias and
// a learned position table (interpolated per grid on the host side, added in the patch GEMM's epilogue), LayerNorm with bias,
// plain fc1 -> GELU(tanh) -> fc2 MLP, full attention inside every temporal patch (no windows, no row permutation), and
// DeepStack: after the blocks in deep_index[] a second kind of merger (LayerNorm over the 4-patch group) taps the stream.
// Heads of 72 dims are stored 80 wide ([36 | 4 zeros | 36 | 4 zeros], weights packed on the host), which keeps the rotary
// halves aligned and reuses the 80-wide attention tiles; the softmax scale stays 72^-1/2.
namespace {
inline int pad64(int n) { return (n + 63) & ~63; }
}

extern "C" size_t o3v_vit3_workspace_bytes(const o3v_vit3_desc* d, int P) {
    if (!d || P <= 0 || d->merge_unit <= 0) return 0;
    const size_t hid = d->hidden, hp = pad64(d->hidden), W = (size_t)d->heads * d->head_dim_pad;
    const size_t Pm = (size_t)P / d->merge_unit, mh = hid * d->merge_unit;
    size_t n = 0;
    n += align256((size_t)P * hid * 2);          // x
    n += align256((size_t)P * hp * 2);           // h (K-padded LayerNorm output)
    n += align256((size_t)P * 3 * W * 2);        // qkv
    n += align256((size_t)P * W * 2);            // att
    n += align256((size_t)P * d->inter_pad * 2); // mlp
    n += align256(Pm * mh * 2);                  // m0
    n += align256(Pm * mh * 2);                  // m1
    return n;
}

namespace {
int vit3_merger(const o3v_vit3_merger_w& m, const char* x, char* m0, char* m1, void* out, int P, int hid, int unit, int out_hidden,
                int tile, o3v_stream_t s) {
    const int Pm = P / unit, mh = hid * unit;
    if (m.postshuffle)
        TRY(o3v_layernorm(x, m.norm_w, m.norm_b, m0, Pm, mh, mh, mh, 1e-6f, s));
    else
        TRY(o3v_layernorm(x, m.norm_w, m.norm_b, m0, P, hid, hid, hid, 1e-6f, s));
    TRY(o3v_gemm_bf16_tile(m0, m.fc1_w, m.fc1_b, nullptr, m1, Pm, mh, mh, mh, mh, mh, 0, O3V_EPI_GELU, tile, s));
    TRY(o3v_gemm_bf16_tile(m1, m.fc2_w, m.fc2_b, nullptr, out, Pm, out_hidden, mh, mh, mh, out_hidden, 0, O3V_EPI_NONE, tile, s));
    return O3V_OK;
}
}  // namespace

extern "C" int o3v_vit3_forward(const o3v_vit3_desc* d, const void* pixels, int P, const void* pos_embed, const float* cosT,
                                const float* sinT, const int* tiles, int n_tiles, void* workspace, size_t ws_bytes, void* out,
                                void* deep_out, o3v_stream_t s) {
    if (!d || !pixels || !pos_embed || !cosT || !sinT || !tiles || !workspace || !out) return O3V_ERR_ARG;
    if (P <= 0 || d->merge_unit <= 0 || (P % d->merge_unit) || d->heads <= 0 || d->n_deep < 0 || d->n_deep > O3V_MAX_DEEPSTACK ||
        (d->n_deep && !deep_out))
        return O3V_ERR_ARG;
    const int hid = d->hidden, hp = pad64(hid), H = d->heads, Dp = d->head_dim_pad, W = H * Dp, unit = d->merge_unit,
              Pm = P / unit, ip = d->inter_pad;
    if ((W % 64) || (ip % 64) || ((hid * unit) % 64) || (hid & 7)) return O3V_ERR_SHAPE;
    if (ws_bytes < o3v_vit3_workspace_bytes(d, P)) return O3V_ERR_WORKSPACE;
    Carver cv{(char*)workspace, 0, ws_bytes};
    char* x = (char*)cv.take((size_t)P * hid * 2);
    char* h = (char*)cv.take((size_t)P * hp * 2);
    char* qkv = (char*)cv.take((size_t)P * 3 * W * 2);
    char* att = (char*)cv.take((size_t)P * W * 2);
    char* mlp = (char*)cv.take((size_t)P * ip * 2);
    char* m0 = (char*)cv.take((size_t)Pm * hid * unit * 2);
    char* m1 = (char*)cv.take((size_t)Pm * hid * unit * 2);
    if (!m1) return O3V_ERR_WORKSPACE;
    const float scale = 1.0f / sqrtf((float)d->head_dim);
    if (hp != hid && hipMemsetAsync(h, 0, (size_t)P * hp * 2, (hipStream_t)s) != hipSuccess) return O3V_ERR_LAUNCH;

    // patch embed: Conv3d(k = s) == GEMM, + bias, + interpolated position rows (TF3:606-640, :700-710)
    TRY(o3v_gemm_bf16_tile(pixels, d->patch_w, d->patch_b, pos_embed, x, P, hid, d->patch_k_pad, d->patch_k_pad, d->patch_k_pad, hid,
                           hid, O3V_EPI_RESIDUAL, d->gemm_tile, s));
    int next_deep = 0;
    for (int i = 0; i < d->depth; ++i) {
        const o3v_vit3_block_w& w = d->blocks[i];
        TRY(o3v_layernorm(x, w.norm1_w, w.norm1_b, h, P, hid, hid, hp, 1e-6f, s));
        TRY(o3v_gemm_bf16_tile(h, w.qkv_w, w.qkv_b, nullptr, qkv, P, 3 * W, hp, hp, hp, 3 * W, 0, O3V_EPI_NONE, d->gemm_tile, s));
        TRY(o3v_vit_rope(qkv, cosT, sinT, P, H, Dp, s));
        TRY(o3v_attn_tiles(qkv, qkv + (size_t)W * 2, qkv + (size_t)2 * W * 2, att, tiles, n_tiles, 64, H, 1, Dp, 3L * W, 3L * W, Dp, 0,
                           3L * W, Dp, 0, W, scale, s));
        TRY(o3v_gemm_bf16_tile(att, w.proj_w, w.proj_b, x, x, P, hid, W, W, W, hid, hid, O3V_EPI_RESIDUAL, d->gemm_tile, s));
        TRY(o3v_layernorm(x, w.norm2_w, w.norm2_b, h, P, hid, hid, hp, 1e-6f, s));
        TRY(o3v_gemm_bf16_tile(h, w.fc1_w, w.fc1_b, nullptr, mlp, P, ip, hp, hp, hp, ip, 0, O3V_EPI_GELU_TANH, d->gemm_tile, s));
        TRY(o3v_gemm_bf16_tile(mlp, w.fc2_w, w.fc2_b, x, x, P, hid, ip, ip, ip, hid, hid, O3V_EPI_RESIDUAL, d->gemm_tile, s));
        if (next_deep < d->n_deep && d->deep_index[next_deep] == i) {
            TRY(vit3_merger(d->deep[next_deep], x, m0, m1, (char*)deep_out + (size_t)next_deep * Pm * d->out_hidden * 2, P, hid, unit,
                            d->out_hidden, d->gemm_tile, s));
            ++next_deep;
        }
    }
    if (next_deep != d->n_deep) return O3V_ERR_ARG;  // deep_index must be ascending and < depth
    return vit3_merger(d->merger, x, m0, m1, out, P, hid, unit, d->out_hidden, d->gemm_tile, s);
}

// ------------------------------------------------------------------------------------------------ LLM
namespace {
// fp32 partials of the widest split-K linear (q/k/v or hidden outputs; gate/up has enough tiles on its own)
size_t splitk_bytes(const o3v_llm_desc* d, int rows) {
    const size_t nq = (size_t)(d->heads + 2 * d->kv_heads) * d->head_dim, nh = d->hidden;
    const size_t n = nq > nh ? nq : nh;
    const size_t tiles = (n + 127) / 128;
    const size_t splits = tiles < 256 ? 256 / tiles + 1 : 1;
    return splits * rows * n * sizeof(float);
}
}  // namespace

extern "C" size_t o3v_llm_workspace_bytes(const o3v_llm_desc* d, int rows) {
    if (!d || rows <= 0) return 0;
    const size_t R = rows, H = d->hidden, QD = (size_t)d->heads * d->head_dim, KD = (size_t)d->kv_heads * d->head_dim;
    size_t n = 0;
    n += align256(R * H * 2);               // h
    n += align256(R * (QD + 2 * KD) * 2);   // qkv
    n += align256(R * QD * 2);              // q (roped)
    n += align256(R * QD * 2);              // att
    n += align256(R * d->inter * 2);        // mlp
    n += align256(R * H * 2);               // normed (head)
    if (rows > 8 && rows <= SPLITK_MAX_ROWS) n += align256(splitk_bytes(d, rows));  // split-K partials of the few-tile GEMMs
    if (d->layer && d->layer[0].gu_w8) {    // fp8 rows present: room for the W8A8 prefill (o3v_prefill_opts.w8a8)
        n += align256(R * (H > QD ? H : QD));   // a8: fp8 activations of the q/k/v, o and gate/up linears
        n += align256(R * d->inter);            // m8: fp8 SwiGLU output
        n += align256(R * sizeof(float));       // per-row scales
    }
    return n;
}

namespace {
struct LlmWs {
    char *h, *qkv, *q, *att, *mlp, *normed;
    float* splitk;
    size_t splitk_bytes;
    char *a8 = nullptr, *m8 = nullptr;  // W8A8 prefill only
    float* sa = nullptr;
};
bool carve_llm(const o3v_llm_desc* d, int rows, void* ws, size_t bytes, LlmWs& w) {
    const size_t R = rows, H = d->hidden, QD = (size_t)d->heads * d->head_dim, KD = (size_t)d->kv_heads * d->head_dim;
    Carver cv{(char*)ws, 0, bytes};
    w.h = (char*)cv.take(R * H * 2);
    w.qkv = (char*)cv.take(R * (QD + 2 * KD) * 2);
    w.q = (char*)cv.take(R * QD * 2);
    w.att = (char*)cv.take(R * QD * 2);
    w.mlp = (char*)cv.take(R * d->inter * 2);
    w.normed = (char*)cv.take(R * H * 2);
    w.splitk = nullptr;
    w.splitk_bytes = 0;
    if (rows > 8 && rows <= SPLITK_MAX_ROWS) {
        w.splitk_bytes = splitk_bytes(d, rows);
        w.splitk = (float*)cv.take(w.splitk_bytes);
        if (!w.splitk) return false;
    }
    if (d->layer && d->layer[0].gu_w8) {
        w.a8 = (char*)cv.take(R * (H > QD ? H : QD));
        w.m8 = (char*)cv.take(R * d->inter);
        w.sa = (float*)cv.take(R * sizeof(float));
        if (!w.sa) return false;
    }
    return w.normed != nullptr;
}
}  // namespace

extern "C" int o3v_llm_prefill(const o3v_llm_desc* d, void* x, const void* cosT, const void* sinT, const int* tiles,
                               int n_tiles, int rows_per_tile, void* kcache, void* vcache, int B, int S, int past,
                               int Tmax, void* workspace, size_t ws_bytes, o3v_stream_t s) {
    return o3v_llm_prefill_ex(d, x, cosT, sinT, tiles, n_tiles, rows_per_tile, kcache, vcache, B, S, past, Tmax, nullptr, workspace,
                              ws_bytes, s);
}

// opts (may be NULL), see o3v_prefill_opts in include/o3v.h:
//  * DeepStack (TF3:839-862): after decoder layer l < n_deep, x[ds_rows[i]] += ds_feat[l][ds_src[i]] for the n_ds visual rows.
//  * shared prompt entry: the first prefix_len of the `past` tokens live ONCE per prompt in kprefix / vprefix; kcache / vcache
//    then hold only the tokens behind them (the S new ones land in slots past - prefix_len ..).
extern "C" int o3v_llm_prefill_ex(const o3v_llm_desc* d, void* x, const void* cosT, const void* sinT, const int* tiles, int n_tiles,
                                  int rows_per_tile, void* kcache, void* vcache, int B, int S, int past, int Tmax,
                                  const o3v_prefill_opts* opts, void* workspace, size_t ws_bytes, o3v_stream_t s) {
    if (!d || !x || !cosT || !sinT || !tiles || !kcache || !vcache || !workspace) return O3V_ERR_ARG;
    const o3v_prefill_opts none = {};
    const o3v_prefill_opts& o = opts ? *opts : none;
    const int n_ds = o.n_ds, n_deep = o.n_deep;
    if (n_ds < 0 || n_deep < 0 || (n_ds > 0 && n_deep > 0 && (!o.ds_rows || !o.ds_src || !o.ds_feat))) return O3V_ERR_ARG;
    const bool pfx = o.kprefix != nullptr;
    if (pfx && (!o.vprefix || o.prefix_len <= 0 || o.prefix_len > past || o.prefix_len > o.prefix_cap || o.rows_per_prefix <= 0 ||
                (B % o.rows_per_prefix)))
        return O3V_ERR_ARG;
    const int own_past = pfx ? past - o.prefix_len : past;  // tokens already in this call's own caches
    if (B <= 0 || S <= 0 || past < 0 || own_past + S > Tmax || d->kv_heads <= 0 || (d->heads % d->kv_heads)) return O3V_ERR_ARG;
    const int rows = B * S, H = d->hidden, Hq = d->heads, Hkv = d->kv_heads, D = d->head_dim, I = d->inter;
    const int QD = Hq * D, NQKV = (Hq + 2 * Hkv) * D;
    LlmWs w;
    if (ws_bytes < o3v_llm_workspace_bytes(d, rows) || !carve_llm(d, rows, workspace, ws_bytes, w)) return O3V_ERR_WORKSPACE;
    const float scale = 1.0f / sqrtf((float)D);
    const size_t layer_stride = (size_t)B * Hkv * Tmax * D * 2;
    const size_t pre_stride = pfx ? (size_t)(B / o.rows_per_prefix) * Hkv * o.prefix_cap * D * 2 : 0;
    const long p_hs = (long)o.prefix_cap * D, p_bs = (long)Hkv * o.prefix_cap * D;
    // W8A8 (opt-in): the four linears of every layer as fp8 x fp8 on the matrix cores -- activations quantised per token on the fly
    // (the RMSNorm kernels emit fp8 + a row scale directly), weights = the fp8 rows of the decode; whole 128-byte k-tiles only
    const bool w8a8 = o.w8a8 && w.sa && d->layer[0].qkv_w8 && d->layer[0].o_w8 && d->layer[0].gu_w8 && d->layer[0].down_w8 &&
                      (H % 128) == 0 && (QD % 128) == 0 && (I % 128) == 0;  // any row count: a log-prob must not depend on its chunk's size
    if (o.w8a8 && !w8a8 && o.w8a8 > 1) return O3V_ERR_SHAPE;  // w8a8 = 2: required, not merely preferred
    for (int l = 0; l < d->layers; ++l) {
        const o3v_llm_layer_w& lw = d->layer[l];
        char* kc = (char*)kcache + l * layer_stride;
        char* vc = (char*)vcache + l * layer_stride;
        const char* kp = pfx ? (const char*)o.kprefix + l * pre_stride : nullptr;
        const char* vp = pfx ? (const char*)o.vprefix + l * pre_stride : nullptr;
        if (w8a8) {
            TRY(o3v_rmsnorm_quantize_fp8(x, lw.ln1, w.a8, w.sa, rows, H, H, H, d->rms_eps, s));
            TRY(o3v_gemm_fp8(w.a8, w.sa, lw.qkv_w8, lw.qkv_s, lw.qkv_b, nullptr, w.qkv, rows, NQKV, H, H, H, NQKV, 0, O3V_EPI_NONE, s));
            if (lw.q_norm)
                TRY(o3v_qkv_norm_rope_cache(w.qkv, lw.q_norm, lw.k_norm, d->rms_eps, cosT, sinT, w.q, kc, vc, own_past, rows, S, Hq, Hkv, D,
                                            Tmax, S, 0, s));
            else
                TRY(o3v_qkv_rope_cache(w.qkv, cosT, sinT, w.q, kc, vc, own_past, rows, S, Hq, Hkv, D, Tmax, S, 0, s));
            TRY(o3v_attn_tiles_prefix(w.q, kc, vc, kp, vp, p_hs, p_bs, o.prefix_len, pfx ? o.rows_per_prefix : 1, w.att, tiles, n_tiles,
                                      rows_per_tile, Hq, Hq / Hkv, D, QD, D, (long)Tmax * D, (long)Hkv * Tmax * D, D, (long)Tmax * D,
                                      (long)Hkv * Tmax * D, QD, scale, s));
            TRY(o3v_quantize_rows_fp8(w.att, w.a8, w.sa, rows, QD, QD, QD, s));
            TRY(o3v_gemm_fp8(w.a8, w.sa, lw.o_w8, lw.o_s, nullptr, x, x, rows, H, QD, QD, QD, H, H, O3V_EPI_RESIDUAL, s));
            TRY(o3v_rmsnorm_quantize_fp8(x, lw.ln2, w.a8, w.sa, rows, H, H, H, d->rms_eps, s));
            TRY(o3v_gemm_fp8(w.a8, w.sa, lw.gu_w8, lw.gu_s, nullptr, nullptr, w.mlp, rows, 2 * I, H, H, H, I, 0, O3V_EPI_SWIGLU, s));
            TRY(o3v_quantize_rows_fp8(w.mlp, w.m8, w.sa, rows, I, I, I, s));
            TRY(o3v_gemm_fp8(w.m8, w.sa, lw.down_w8, lw.down_s, nullptr, x, x, rows, H, I, I, I, H, H, O3V_EPI_RESIDUAL, s));
            if (l < n_deep && n_ds > 0)
                TRY(o3v_add_rows(x, o.ds_rows, o.ds_src, (const char*)o.ds_feat + (size_t)l * o.ds_stride * 2, n_ds, H, s));
            continue;
        }
        TRY(o3v_rmsnorm(x, lw.ln1, w.h, rows, H, H, H, d->rms_eps, s));
        TRY(linear(w.h, lw.qkv_w, lw.qkv_b, nullptr, w.qkv, rows, NQKV, H, H, NQKV, 0, O3V_EPI_NONE, s, w.splitk, w.splitk_bytes, d->gemm_tile));
        if (lw.q_norm)
            TRY(o3v_qkv_norm_rope_cache(w.qkv, lw.q_norm, lw.k_norm, d->rms_eps, cosT, sinT, w.q, kc, vc, own_past, rows, S, Hq, Hkv, D,
                                        Tmax, S, 0, s));
        else
            TRY(o3v_qkv_rope_cache(w.qkv, cosT, sinT, w.q, kc, vc, own_past, rows, S, Hq, Hkv, D, Tmax, S, 0, s));
        TRY(o3v_attn_tiles_prefix(w.q, kc, vc, kp, vp, p_hs, p_bs, o.prefix_len, pfx ? o.rows_per_prefix : 1, w.att, tiles, n_tiles,
                                  rows_per_tile, Hq, Hq / Hkv, D, QD, D, (long)Tmax * D, (long)Hkv * Tmax * D, D, (long)Tmax * D,
                                  (long)Hkv * Tmax * D, QD, scale, s));
        TRY(linear(w.att, lw.o_w, nullptr, x, x, rows, H, QD, QD, H, H, O3V_EPI_RESIDUAL, s, w.splitk, w.splitk_bytes, d->gemm_tile));
        TRY(o3v_rmsnorm(x, lw.ln2, w.h, rows, H, H, H, d->rms_eps, s));
        TRY(linear(w.h, lw.gu_w, nullptr, nullptr, w.mlp, rows, 2 * I, H, H, I, 0, O3V_EPI_SWIGLU, s, nullptr, 0, d->gemm_tile));
        TRY(linear(w.mlp, lw.down_w, nullptr, x, x, rows, H, I, I, H, H, O3V_EPI_RESIDUAL, s, w.splitk, w.splitk_bytes, d->gemm_tile));
        if (l < n_deep && n_ds > 0)
            TRY(o3v_add_rows(x, o.ds_rows, o.ds_src, (const char*)o.ds_feat + (size_t)l * o.ds_stride * 2, n_ds, H, s));
    }
    return O3V_OK;
}

namespace {
int llm_head(const o3v_llm_desc* d, const void* x, int ldx, int rows, void* normed, void* logits, bool fp8_rows, o3v_stream_t s,
             bool pre_normed = false);
}
// bf16 weights always: the head of the prefill, of forward_logits and of the log-prob pass (R:grpo_trainer.py:371-384) -- the fp8
// rows are DECODE rows (o3v_llm_decode streams them), a log-prob must not depend on how many rows its chunk happens to hold
extern "C" int o3v_llm_head(const o3v_llm_desc* d, const void* x, int ldx, int rows, void* normed, void* logits,
                            o3v_stream_t s) {
    return llm_head(d, x, ldx, rows, normed, logits, false, s, false);
}

namespace {
// pre_normed: `normed` already holds RMSNorm(x; final_norm) (the last down_proj of a batched decode step normalised it, TailNorm)
int llm_head(const o3v_llm_desc* d, const void* x, int ldx, int rows, void* normed, void* logits, bool fp8_rows, o3v_stream_t s,
             bool pre_normed) {
    if (!d || !x || !normed || !logits || rows <= 0) return O3V_ERR_ARG;
    const int H = d->hidden;
    if (fp8_rows && rows >= 4 && rows <= 32 && d->lm_head8p) {  // batched decode on fp8 rows: norm apart, fp8 fragments widened in registers
        TRY(o3v_rmsnorm(x, d->final_norm, normed, rows, H, ldx, H, d->rms_eps, s));
        return o3v_linear_decode_fp8_rows(normed, d->lm_head8p, d->lm_head_s, nullptr, nullptr, logits, rows, d->vocab, H, H, d->vocab, 0,
                                          O3V_EPI_NONE, s);
    }
    if (rows >= 8 && rows <= 32 && d->lm_head_p) {  // batched decode: norm apart, LDS-free matrix-core linear
        if (!pre_normed) TRY(o3v_rmsnorm(x, d->final_norm, normed, rows, H, ldx, H, d->rms_eps, s));
        return o3v_linear_decode(normed, nullptr, 0.f, d->lm_head, d->lm_head_p, nullptr, nullptr, logits, rows, d->vocab, H, H,
                                 d->vocab, 0, O3V_EPI_NONE, s);
    }
    if (fp8_rows && rows <= 3 && d->lm_head8)  // fp8 head (decode with fp8 weights): half the 1.09 GB
        return o3v_linear_decode_fp8(x, d->final_norm, d->rms_eps, d->lm_head8, d->lm_head_s, nullptr, nullptr, logits, rows,
                                     d->vocab, H, ldx, d->vocab, 0, O3V_EPI_NONE, s);
    if (rows <= 8)  // decode / last-token head: RMSNorm fused into the weight-streaming GEMV
        return o3v_linear_decode(x, d->final_norm, d->rms_eps, d->lm_head, d->lm_head_p, nullptr, nullptr, logits, rows,
                                 d->vocab, H, ldx, d->vocab, 0, O3V_EPI_NONE, s);
    TRY(o3v_rmsnorm(x, d->final_norm, normed, rows, H, ldx, H, d->rms_eps, s));
    for (int r0 = 0; r0 < rows;) {
        // GEMV groups of <= 8 rows for small row counts; one MFMA GEMM otherwise
        const int n = (rows > 8) ? rows : rows - r0;
        TRY(linear((const char*)normed + (size_t)r0 * H * 2, d->lm_head, nullptr, nullptr,
                   (char*)logits + (size_t)r0 * d->vocab * 2, n, d->vocab, H, H, d->vocab, 0, O3V_EPI_NONE, s));
        r0 += n;
    }
    return O3V_OK;
}
}  // namespace

extern "C" int o3v_llm_decode(const o3v_llm_desc* d, const o3v_decode_state* st, int step0, int n_steps,
                              int skip_last_forward, o3v_stream_t s) {
    if (!d || !st || !st->x || !st->kcache || !st->vcache || !st->cosT || !st->sinT || !st->logits || !st->seen ||
        !st->cur_tok || !st->finished || !st->out_ids || !st->part_o || !st->part_ml || !st->workspace)
        return O3V_ERR_ARG;
    const int B = st->B;
    // shared prompt entries: the caches hold only the generated tokens (slot = step), the prompt K/V live once per prompt
    const bool pfx = st->kprefix != nullptr;
    if (B <= 0 || B > 32 || step0 < 0 || n_steps < 0 || step0 + n_steps > st->Tnew ||
        (pfx ? st->Tnew : st->S + st->Tnew) > st->Tmax + 1)
        return O3V_ERR_ARG;
    if (pfx && (!st->vprefix || st->group <= 1 || st->rows_per_prompt <= 0 || (B % st->rows_per_prompt) || st->S > st->prefix_cap))
        return O3V_ERR_ARG;
    const int slot0 = pfx ? 0 : st->S;  // cache slot of generated token 0
    if (!st->sample_scratch) return O3V_ERR_ARG;
    const int H = d->hidden, Hq = d->heads, Hkv = d->kv_heads, D = d->head_dim, I = d->inter, V = d->vocab;
    const int QD = Hq * D, NQKV = (Hq + 2 * Hkv) * D;
    LlmWs w;
    if (st->ws_bytes < o3v_llm_workspace_bytes(d, B) || !carve_llm(d, B, st->workspace, st->ws_bytes, w))
        return O3V_ERR_WORKSPACE;
    const float scale = 1.0f / sqrtf((float)D);
    const size_t layer_stride = (size_t)B * Hkv * st->Tmax * D * 2;
    const size_t pre_stride = pfx ? (size_t)(B / st->rows_per_prompt) * Hkv * st->prefix_cap * D * 2 : 0;
    // B >= 4 rows run on the matrix-core linears, which stage the normalised x of a fused RMSNorm in LDS (57 KB at B=8, 115 KB
    // at B=16: two blocks, then one, per CU).  With a separate 3 us norm launch the linears are LDS-free: gate/up 54 -> 44 us
    // at B=8, 75 -> 50 us at B=16 (profiles/r01_m8_linear.txt).  Whole step, 7B: B=4 3.56 -> 3.61 ms (worse: two more
    // launches per layer), B=8 4.04 -> 3.97, B=16 5.63 -> 5.25: taken from B=8 on.
    const bool norm_apart = B >= 8;
    // ... except q/k/v at 8 rows: its grid is ~1 workgroup per CU (288 at 7B), so the LDS the fused norm stages x in costs no
    // residency, and the prologue replaces a 5.9 us launch.  Same sums, same rounding as o3v_rmsnorm: bit-identical.  Measured
    // (7B, G rows of one prompt, ms/step fused | apart): G=8 3.721 | 3.768, G=16 4.365 | 4.312 -- taken at 8 rows only.
    const bool qkv_fused_norm = norm_apart && B <= 8 && !(st->flags & 4) && (size_t)B * (H * 2 + 16) <= 144 * 1024;
    // batch 1: q/k/v + attention + merge + o_proj as ONE launch (o3v_fused.hip); epoch = index of the launch in this generate call
    // fp8 weights (batch <= 3): every decode linear streams the fp8 copy
    const bool fp8 = B <= 3 && d->layer[0].qkv_w8 && d->layer[0].o_w8 && d->layer[0].gu_w8 && d->layer[0].down_w8 && st->group <= 1;
    // fp8 rows at 4..32 rows (BASELINE config #5: N = 16 chains on fp8 weights): norms run apart, the linears stream the
    // fragment-major fp8 images
    const bool fp8b = B >= 4 && d->layer[0].gu_w8p && d->layer[0].down_w8p && d->layer[0].o_w8p;
    const bool qk_norm = d->layer[0].q_norm != nullptr;  // Qwen3-VL: per-head RMSNorm between the q/k/v linear and the rotation
    bool fused = st->sync && B == 1 && st->group <= 1 && st->nsplit > 0;
    // the persistent layer block (attention half + gate/up in one launch): bf16 rows, no q/k norm
    bool layer_block = fused && (st->flags & 1) && !fp8 && !qk_norm;
    int step = step0;
    // 8..32 bf16 rows: the residual linears normalise for the linear that follows (o3v_linear_decode_norm_next), so only layer 0's
    // first norm of a step is a launch of its own.  st->flags & 2 switches it off (A/B, tests).
    bool tail = norm_apart && !fp8b && st->sync && !(st->flags & 2) && d->lm_head_p && d->layer[0].o_wp && d->layer[0].down_wp;
    bool h_ready = false;       // w.h holds RMSNorm(x; ln1 of the layer about to run)
    bool h2_ready = false;      // w.h holds RMSNorm(x; ln2 of the running layer)
    bool normed_ready = false;  // w.normed holds RMSNorm(x; final_norm)
    // host_stats (optional, host memory): [0] += decode forwards, [1] += kernel launches inside their layer loops, [2] += layers whose
    // attention half ran as the one-launch block, [3] += layers whose attention half ran on the stand-alone kernels
    long long* hs = (long long*)st->host_stats;
    // attention half of layer l as stand-alone launches: q/k/v (+norm, rope, cache append), attention + merge, o_proj + residual
    auto attention_half = [&](int l) -> int {
        const o3v_llm_layer_w& lw = d->layer[l];
        char* kc = (char*)st->kcache + l * layer_stride;
        char* vc = (char*)st->vcache + l * layer_stride;
        if (qk_norm) {
            if (fp8)
                TRY(o3v_linear_decode_fp8(st->x, lw.ln1, d->rms_eps, lw.qkv_w8, lw.qkv_s, lw.qkv_b, nullptr, w.qkv, B, NQKV, H, H, NQKV, 0,
                                          O3V_EPI_NONE, s));
            else if (fp8b && lw.qkv_w8p) {
                TRY(o3v_rmsnorm(st->x, lw.ln1, w.h, B, H, H, H, d->rms_eps, s));
                TRY(o3v_linear_decode_fp8_rows(w.h, lw.qkv_w8p, lw.qkv_s, lw.qkv_b, nullptr, w.qkv, B, NQKV, H, H, NQKV, 0, O3V_EPI_NONE, s));
            } else if (norm_apart && !qkv_fused_norm) {  // (above 16 rows the linears take no fused norm at all)
                if (!h_ready) TRY(o3v_rmsnorm(st->x, lw.ln1, w.h, B, H, H, H, d->rms_eps, s));
                TRY(o3v_linear_decode(w.h, nullptr, 0.f, lw.qkv_w, lw.qkv_wp, lw.qkv_b, nullptr, w.qkv, B, NQKV, H, H, NQKV, 0, O3V_EPI_NONE, s));
            } else
                TRY(o3v_linear_decode(st->x, lw.ln1, d->rms_eps, lw.qkv_w, lw.qkv_wp, lw.qkv_b, nullptr, w.qkv, B, NQKV, H, H, NQKV, 0,
                                      O3V_EPI_NONE, s));
            TRY(o3v_qkv_norm_rope_cache(w.qkv, lw.q_norm, lw.k_norm, d->rms_eps, st->cosT, st->sinT, w.q, kc, vc, slot0 + step, B, 1, Hq,
                                        Hkv, D, st->Tmax, st->Tnew, step, s));
        } else if (fp8b && lw.qkv_w8p) {
            TRY(o3v_rmsnorm(st->x, lw.ln1, w.h, B, H, H, H, d->rms_eps, s));
            TRY(o3v_qkv_rope_fp8_rows(w.h, lw.qkv_w8p, lw.qkv_s, lw.qkv_b, B, H, H, st->cosT, st->sinT, w.q, kc, vc, slot0 + step, Hq, Hkv, D,
                                      st->Tmax, st->Tnew, step, s));
        } else if (fp8) {
            TRY(o3v_gemv_norm_qkv_rope_fp8(st->x, lw.ln1, d->rms_eps, lw.qkv_w8, lw.qkv_s, lw.qkv_b, B, H, H, st->cosT, st->sinT, w.q,
                                           kc, vc, slot0 + step, Hq, Hkv, D, st->Tmax, st->Tnew, step, s));
        } else if (norm_apart && !qkv_fused_norm) {
            if (!h_ready) TRY(o3v_rmsnorm(st->x, lw.ln1, w.h, B, H, H, H, d->rms_eps, s));
            TRY(o3v_gemv_norm_qkv_rope(w.h, nullptr, 0.f, lw.qkv_w, lw.qkv_wp, lw.qkv_b, B, H, H, st->cosT, st->sinT, w.q, kc, vc,
                                       slot0 + step, Hq, Hkv, D, st->Tmax, st->Tnew, step, s));
        } else {
            TRY(o3v_gemv_norm_qkv_rope(st->x, lw.ln1, d->rms_eps, lw.qkv_w, lw.qkv_wp, lw.qkv_b, B, H, H, st->cosT, st->sinT, w.q,
                                       kc, vc, slot0 + step, Hq, Hkv, D, st->Tmax, st->Tnew, step, s));
        }
        if (st->group > 1)  // the rows of a group share the prompt K/V: read it once per group
            TRY(o3v_attn_decode_group_prefix(w.q, kc, vc, pfx ? (const char*)st->kprefix + l * pre_stride : nullptr,
                                             pfx ? (const char*)st->vprefix + l * pre_stride : nullptr, st->prefix_cap,
                                             st->rows_per_prompt, w.att, st->part_o, st->part_ml, st->k_lo, B, st->group, Hq, Hkv, D,
                                             st->S, st->S + step + 1, st->Tmax, st->nsplit, scale, s));
        else
            TRY(o3v_attn_decode(w.q, kc, vc, w.att, st->part_o, st->part_ml, st->k_lo, B, Hq, Hkv, D, st->S + step + 1, st->Tmax,
                                st->nsplit, scale, s));
        if (fp8)
            TRY(o3v_linear_decode_fp8(w.att, nullptr, 0.f, lw.o_w8, lw.o_s, nullptr, st->x, st->x, B, H, QD, QD, H, H, O3V_EPI_RESIDUAL, s));
        else if (fp8b)
            TRY(o3v_linear_decode_fp8_rows(w.att, lw.o_w8p, lw.o_s, nullptr, st->x, st->x, B, H, QD, QD, H, H, O3V_EPI_RESIDUAL, s));
        else {
            h_ready = h2_ready = false;
            if (tail) {
                const uint32_t epoch = (uint32_t)(step * d->layers + l) * 2u + 1u;
                const int rc = o3v_linear_decode_norm_next(w.att, lw.o_wp, st->x, st->x, B, H, QD, QD, H, H, lw.ln2, d->rms_eps, w.h, H,
                                                           st->sync, epoch, s);
                if (rc == O3V_OK) {
                    h2_ready = true;
                    return O3V_OK;
                }
                if (rc != O3V_ERR_SHAPE || l != 0 || step != step0) return rc;
                tail = false;  // not this device / these shapes: two launches, for the whole call
            }
            TRY(o3v_linear_decode(w.att, nullptr, 0.f, lw.o_w, lw.o_wp, nullptr, st->x, st->x, B, H, QD, QD, H, H, O3V_EPI_RESIDUAL, s));
        }
        return O3V_OK;
    };
    for (int i = 0; i < n_steps; ++i) {
        step = step0 + i;
        if (st->do_sample)
            TRY(o3v_sample_top_k_top_p(st->logits, st->seen, st->cur_tok, st->finished, st->out_ids, st->margins, st->eos_ids,
                                       st->n_eos, st->pad_id, B, V, V, st->rep_penalty, st->temperature, st->top_k, st->top_p,
                                       st->seed, st->row_id, step, st->Tnew, st->sample_scratch, s));
        else  // greedy: the chosen token's embedding row is gathered by the sampler's last stage
            TRY(o3v_sample_greedy_embed(st->logits, st->seen, st->cur_tok, st->finished, st->out_ids, st->margins, st->eos_ids,
                                        st->n_eos, st->pad_id, B, V, V, st->rep_penalty, step, st->Tnew, st->sample_scratch,
                                        d->embed, st->x, H, s));
        if (skip_last_forward && i == n_steps - 1) break;
        if (slot0 + step >= st->Tmax) return O3V_ERR_ARG;
        // one decode forward: token `step` sits in cache slot S+step, context = S+step+1 keys
        if (st->do_sample) TRY(o3v_embed_tokens(d->embed, st->cur_tok, st->x, B, H, s));
        h_ready = h2_ready = normed_ready = false;  // x is the sampler's embedding row
        const long long launches0 = o3v_tl_launches;
        for (int l = 0; l < d->layers; ++l) {
            const o3v_llm_layer_w& lw = d->layer[l];
            char* kc = (char*)st->kcache + l * layer_stride;
            char* vc = (char*)st->vcache + l * layer_stride;
            if (layer_block) {
                const uint32_t epoch = (uint32_t)(step * d->layers + l + 1);
                const int rc = o3v_decode_layer_block(st->x, lw.ln1, d->rms_eps, lw.qkv_w, lw.qkv_b, lw.o_w, lw.ln2, lw.gu_w, w.mlp, st->cosT,
                                                      st->sinT, w.q, w.att, kc, vc, st->part_o, st->part_ml, st->k_lo, H, I, Hq, Hkv, D,
                                                      st->S + step, st->Tmax, st->Tnew, step, st->nsplit, scale, st->sync, epoch, s);
                if (rc == O3V_ERR_SHAPE && l == 0 && step == step0) {
                    layer_block = false;  // shapes or residency: the role-per-workgroup block (or the stand-alone kernels) instead
                } else if (rc != O3V_OK) {
                    return rc;
                } else {
                    if (hs) ++hs[2];
                    TRY(o3v_linear_decode(w.mlp, nullptr, 0.f, lw.down_w, lw.down_wp, nullptr, st->x, st->x, B, H, I, I, H, H,
                                          O3V_EPI_RESIDUAL, s));
                    continue;
                }
            }
            if (fused) {
                const uint32_t epoch = (uint32_t)(step * d->layers + l + 1);
                const int rc =
                    qk_norm ? o3v_decode_attn_block_qknorm(st->x, lw.ln1, d->rms_eps, fp8 ? lw.qkv_w8 : lw.qkv_w, fp8 ? lw.qkv_s : nullptr,
                                                           fp8 ? lw.o_w8 : lw.o_w, fp8 ? lw.o_s : nullptr, lw.q_norm, lw.k_norm, w.qkv,
                                                           st->cosT, st->sinT, w.q, w.att, kc, vc, st->part_o, st->part_ml, st->k_lo, H, Hq,
                                                           Hkv, D, st->S + step, st->Tmax, st->Tnew, step, st->nsplit, scale, st->sync,
                                                           epoch, s)
                    : fp8 ? o3v_decode_attn_block_fp8(st->x, lw.ln1, d->rms_eps, lw.qkv_w8, lw.qkv_s, lw.qkv_b, lw.o_w8, lw.o_s, st->cosT,
                                                    st->sinT, w.q, w.att, kc, vc, st->part_o, st->part_ml, st->k_lo, H, Hq, Hkv, D,
                                                    st->S + step, st->Tmax, st->Tnew, step, st->nsplit, scale, st->sync, epoch, s)
                        : o3v_decode_attn_block(st->x, lw.ln1, d->rms_eps, lw.qkv_w, lw.qkv_b, lw.o_w, st->cosT, st->sinT, w.q, w.att, kc,
                                                vc, st->part_o, st->part_ml, st->k_lo, H, Hq, Hkv, D, st->S + step, st->Tmax, st->Tnew,
                                                step, st->nsplit, scale, st->sync, epoch, s);
                if (rc == O3V_ERR_SHAPE && l == 0)
                    fused = false;  // shapes or residency do not allow the one-launch form: the stand-alone kernels instead
                else if (rc != O3V_OK)
                    return rc;
            }
            if (!fused) TRY(attention_half(l));
            if (hs) ++hs[fused ? 2 : 3];
            if (fp8) {
                TRY(o3v_linear_decode_fp8(st->x, lw.ln2, d->rms_eps, lw.gu_w8, lw.gu_s, nullptr, nullptr, w.mlp, B, 2 * I, H, H, I, 0,
                                          O3V_EPI_SWIGLU, s));
                TRY(o3v_linear_decode_fp8(w.mlp, nullptr, 0.f, lw.down_w8, lw.down_s, nullptr, st->x, st->x, B, H, I, I, H, H,
                                          O3V_EPI_RESIDUAL, s));
                continue;
            }
            if (fp8b) {
                TRY(o3v_rmsnorm(st->x, lw.ln2, w.h, B, H, H, H, d->rms_eps, s));
                TRY(o3v_linear_decode_fp8_rows(w.h, lw.gu_w8p, lw.gu_s, nullptr, nullptr, w.mlp, B, 2 * I, H, H, I, 0, O3V_EPI_SWIGLU, s));
                TRY(o3v_linear_decode_fp8_rows(w.mlp, lw.down_w8p, lw.down_s, nullptr, st->x, st->x, B, H, I, I, H, H, O3V_EPI_RESIDUAL, s));
                continue;
            }
            if (norm_apart) {
                if (!h2_ready) TRY(o3v_rmsnorm(st->x, lw.ln2, w.h, B, H, H, H, d->rms_eps, s));
                TRY(o3v_linear_decode(w.h, nullptr, 0.f, lw.gu_w, lw.gu_wp, nullptr, nullptr, w.mlp, B, 2 * I, H, H, I, 0,
                                      O3V_EPI_SWIGLU, s));
                h2_ready = false;
                if (tail) {  // down_proj + the next layer's first norm (the head's norm after the last layer)
                    const bool last = l + 1 == d->layers;
                    const uint32_t epoch = (uint32_t)(step * d->layers + l) * 2u + 2u;
                    TRY(o3v_linear_decode_norm_next(w.mlp, lw.down_wp, st->x, st->x, B, H, I, I, H, H, last ? d->final_norm : d->layer[l + 1].ln1,
                                                    d->rms_eps, last ? w.normed : w.h, H, st->sync, epoch, s));
                    (last ? normed_ready : h_ready) = true;
                    continue;
                }
            } else {
                TRY(o3v_linear_decode(st->x, lw.ln2, d->rms_eps, lw.gu_w, lw.gu_wp, nullptr, nullptr, w.mlp, B, 2 * I, H, H, I, 0,
                                      O3V_EPI_SWIGLU, s));
            }
            TRY(o3v_linear_decode(w.mlp, nullptr, 0.f, lw.down_w, lw.down_wp, nullptr, st->x, st->x, B, H, I, I, H, H,
                                  O3V_EPI_RESIDUAL, s));
        }
        if (hs) {
            ++hs[0];
            hs[1] += o3v_tl_launches - launches0;
        }
        // the head follows the layers: fp8 rows only when the layer stack of this call streams fp8 rows
        TRY(llm_head(d, st->x, H, B, w.normed, st->logits, fp8 || fp8b, s, normed_ready));
    }
    return O3V_OK;
}
