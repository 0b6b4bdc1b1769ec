/* o3v.h -- C ABI of libo3v_hip.so: the MI355X (gfx950) hot path of Open-o3-Video's generate loop.
 *
 * The reference (marinero4972/Open-o3-Video) is pure Python and has no FFI of its own: its boundary for this
 * path is two Python call sites,
 *   R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:581-582   model.generate(**prompt_inputs, generation_config)
 *   R:eval/inference_example.py:81 / R:eval/models/model_vllm.py:103,117,125   llm.generate(inputs, sampling_params)
 * and the arithmetic behind them lives in transformers (TF: = transformers 5.15.0,
 * models/qwen2_5_vl/modeling_qwen2_5_vl.py).  This header is the boundary a maintainer binds instead
 * (ctypes stub in INTEGRATION.md); each entry point names the reference code it replaces.
 *
 * Conventions: plain pointers + sizes, no torch types.  All device buffers are caller-allocated, bf16 is a raw
 * uint16 pattern, every call only ENQUEUES work on `stream` (no sync, no allocation, graph-capturable) and returns
 * 0 or a negative O3V_ERR_* code; nothing throws.  Thread-safe for distinct streams: the library keeps no mutable state
 * (every tunable travels in an argument or a descriptor; the only statics are occupancy figures queried once).
 */
#ifndef O3V_H
#define O3V_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* o3v_stream_t; /* hipStream_t */

#define O3V_OK 0
#define O3V_ERR_ARG (-1)    /* null pointer / negative size / inconsistent arguments */
#define O3V_ERR_SHAPE (-2)  /* shape not supported by the kernels (alignment, head_dim, M > 32 for the decode linears ...) */
#define O3V_ERR_LAUNCH (-3) /* HIP reported a launch failure */
#define O3V_ERR_WORKSPACE (-4)

#define O3V_EPI_NONE 0     /* out = bf16(acc + bias) */
#define O3V_EPI_RESIDUAL 1 /* out = bf16(bf16(acc + bias) + res)                          TF:311-321, :733-755 */
#define O3V_EPI_GELU 2     /* out = bf16(gelu_erf(bf16(acc + bias)))                       TF:141-145 merger */
#define O3V_EPI_GELU_TANH 6 /* out = bf16(gelu_tanh(bf16(acc + bias)))   MFMA GEMM only   TF3:101-112 Qwen3-VL vision MLP */
#define O3V_EPI_SWIGLU 3   /* out[:, j] = bf16(bf16(silu(gate_j)) * up_j); W rows packed by o3v 16-row interleave  TF:85-96,541-554 */

int o3v_abi_version(void);

/* ---- elementwise / layout (HBM-bound) ------------------------------------------------------------------- */
/* Qwen2_5_VLRMSNorm.forward, TF:65-79 */
int o3v_rmsnorm(const void* x, const void* w, void* out, int rows, int cols, int ld_in, int ld_out, float eps,
                o3v_stream_t stream);
/* nn.LayerNorm with bias over `cols` (TF3:122-135, :268-284): fp32 statistics, one rounding.  x,out bf16 [rows, ld]. */
int o3v_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int cols, int ld_in, int ld_out, float eps,
                  o3v_stream_t stream);
/* apply_rotary_pos_emb_vision, TF:160-171, in place on the q,k thirds of qkv[P,3*H*D]; cos/sin fp32 [P,D/2] */
int o3v_vit_rope(void* qkv, const float* cosT, const float* sinT, int P, int H, int D, o3v_stream_t stream);
/* Qwen2_5_VLRotaryEmbedding.forward + mrope section select, TF:525-538, :590-596; pos int32 [3,T] */
int o3v_mrope_table(const int* pos, const float* inv_freq, const int* axis_of, void* cosT, void* sinT, int T, int D,
                    o3v_stream_t stream);
/* apply_multimodal_rotary_pos_emb + DynamicCache.update, TF:557-599, :652-664 */
int o3v_qkv_rope_cache(const void* qkv, const void* cosT, const void* sinT, void* qout, void* kcache, void* vcache,
                       int slot_base, int T, int tokens_per_row, int Hq, int Hkv, int D, int Tmax, int cs_stride_row,
                       int cs_off, o3v_stream_t stream);
/* The same with Qwen3-VL's per-head RMSNorm (weights q_norm, k_norm bf16 [D], eps) applied to the q and k heads before the
 * rotation (TF3:480-484); D/16 must be a power of two. */
int o3v_qkv_norm_rope_cache(const void* qkv, const void* q_norm, const void* k_norm, float eps, const void* cosT,
                            const void* sinT, void* qout, void* kcache, void* vcache, int slot_base, int T, int tokens_per_row,
                            int Hq, int Hkv, int D, int Tmax, int cs_stride_row, int cs_off, o3v_stream_t stream);
/* DeepStack add (TF3:839-862): x[rows[i], :] = bf16(x[rows[i], :] + feat[src[i], :]), i < n; x, feat bf16 rows of `hidden`. */
int o3v_add_rows(void* x, const int* rows, const int* src, const void* feat, int n, int hidden, o3v_stream_t stream);
/* hidden_states[window_index] / [reverse_indices], TF:436-439, :464-466 */
int o3v_gather_rows(const void* src, const int* idx, void* dst, int rows, int row_bytes, o3v_stream_t stream);
/* embed_tokens + masked_scatter of image embeds, TF:1206-1215 */
int o3v_embed_scatter(const void* table, const void* vis, const int* src_row, void* out, int T, int hidden,
                      o3v_stream_t stream);
int o3v_embed_tokens(const void* table, const int* tok, void* out, int B, int hidden, o3v_stream_t stream);
/* pixel_values.type(visual.dtype), TF:1090, with zero padding of the patch row to Kp */
int o3v_cast_pad_f32_bf16(const float* src, void* dst, int P, int K0, int Kp, o3v_stream_t stream);
/* Qwen2VLImageProcessor rescale+normalize+patchify, TF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:152-233,
 * on frames [T,3,H,W] (uint8 or f32 0..255) as R:vision_process.py:279-318 returns them */
int o3v_patchify(const void* frames, int is_u8, void* dst, int T, int H, int W, int Kp, const float* mean3,
                 const float* std3, o3v_stream_t stream);
/* The same for a `patch`-pixel patch (Qwen3-VL: 16; H, W multiples of 2*patch; Kp >= 6*patch*patch). */
int o3v_patchify_ps(const void* frames, int is_u8, void* dst, int T, int H, int W, int Kp, int patch, const float* mean3,
                    const float* std3, o3v_stream_t stream);
/* Native video input (the "video" entry of R:eval/models/model_vllm.py:72-88, pixel_values_videos of R:…/grpo_trainer.py:555-564):
 * Qwen2VLVideoProcessor's patchify, TF:models/qwen2_vl/video_processing_qwen2_vl.py:236-274 -- temporal patch k of the ONE
 * video `frames` [n_frames,3,H,W] holds frames 2k and 2k+1 (an odd count repeats the last frame); dst bf16
 * [ceil(n_frames/2)*(H/patch)*(W/patch), Kp], rows (grid_t, gh/2, gw/2, 2, 2), columns (channel, t, py, px). */
int o3v_patchify_video(const void* frames, int is_u8, void* dst, int n_frames, int H, int W, int Kp, int patch,
                       const float* mean3, const float* std3, o3v_stream_t stream);

/* fetch_video's frame resize, R:src/r1-v/src/open_r1/vision_process.py:310-315 (torchvision resize, BICUBIC, antialias ==
 * ATen _upsample_bicubic2d_aa): src [planes,H_in,W_in] uint8 or f32 -> dst f32 [planes,H_out,W_out]; tmp f32
 * [planes,H_in,W_out].  Tap tables per output column / row ({x,y}min, {x,y}size, normalised weights [out][k]) are built on
 * the host (open_o3_video_amd/vision_process.py aa_tables).  uint8 sources are rounded and clamped like torchvision. */
int o3v_resize_bicubic_aa(const void* src, int is_u8, float* tmp, float* dst, int planes, int H_in, int W_in, int H_out,
                          int W_out, const int* xmin, const int* xsize, const float* xw, int xk, const int* ymin,
                          const int* ysize, const float* yw, int yk, o3v_stream_t stream);
/* crop_box of the test-time-scaling loop, R:eval/tts.py:54-75: boxes int32 [n][5] = {frame, x1, y1, x2, y2} (clipped,
 * non-empty) of uint8 frames [T,3,H,W] -> uint8 [n,3,H,W], cv2.resize(float32 crop, (W,H), INTER_LINEAR).astype(uint8) */
int o3v_crop_resize_bilinear(const void* frames, const int* boxes, void* out, int n, int T, int H, int W,
                             o3v_stream_t stream);

/* 128-bit content hash of `bytes` bytes at `data` (8-byte aligned) into out[2] (u64, ZEROED by the caller): cache keys of
 * the visual-token / prefix-K/V caches.  Order-independent sums of mixed words: exact and reproducible. */
int o3v_content_hash128(const void* data, size_t bytes, unsigned long long* out, o3v_stream_t stream);

/* ---- GEMMs ------------------------------------------------------------------------------------------------ */
/* nn.Linear: out[M,N] = epi(A[M,K] . W[N,K]^T + bias); K % 64 == 0.  MFMA path (ViT, merger, prefill). */
int o3v_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K, int lda,
                  int ldw, int ldo, int ldr, int epilogue, o3v_stream_t stream);
/* the same with the tile forced: 0 = per shape (what o3v_gemm_bf16 does), 128 or 256 (tests and A/B measurements; both
 * kernels give bit-identical results).  The model-level entries take the choice from their descriptor's gemm_tile. */
int o3v_gemm_bf16_tile(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K, int lda,
                       int ldw, int ldo, int ldr, int epilogue, int tile, o3v_stream_t stream);
/* The 256 x 256 x 64 tile on the PHASED schedule (csrc/o3v_gemm8p.hip): 16-MFMA phases between raw barriers, the global->LDS copies
 * of half-tiles kept 6 deep in flight across them (counted vmcnt), the two wave rows one barrier apart so that a SIMD always has
 * one wave multiplying while the other reads LDS.  K % 128 == 0, K >= 256; bit-identical to o3v_gemm_bf16 / o3v_gemm_bf16_tile;
 * O3V_ERR_SHAPE otherwise.  o3v_gemm_bf16 picks it by itself where it applies. */
int o3v_gemm_bf16_phased(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K, int lda,
                         int ldw, int ldo, int ldr, int epilogue, o3v_stream_t stream);
/* o3v_gemm_bf16 for row counts too small to fill the chip with 128x128 output tiles (a prompt suffix behind a cached
 * prefix: 9..128 rows): K split over `splits` blocks per tile, fp32 partials in `workspace` (splits*M*N floats), reduced
 * in split order.  Epilogues NONE / RESIDUAL / GELU. */
int o3v_gemm_bf16_splitk(const void* A, const void* W, const void* bias, const void* res, void* out, int M, int N, int K,
                         int lda, int ldw, int ldo, int ldr, int epilogue, int splits, float* workspace, size_t ws_bytes,
                         o3v_stream_t stream);
/* same contract for M <= 32 rows (decode): weight-streaming GEMV up to 3 rows, matrix-core skinny GEMM from 4 rows on
 * (9..32 rows need K % 32 == 0 and N % 16 == 0; 17..32 rows run two 16-row column blocks per streamed weight fragment and
 * take no fused norm: normalise with o3v_rmsnorm first). */
int o3v_gemv_bf16(const void* X, const void* W, const void* bias, const void* res, void* out, int M, int N, int K, int ldx,
                  int ldw, int ldo, int ldr, int epilogue, o3v_stream_t stream);

/* RMSNorm (TF:65-79) fused into the projection that consumes it: out = epi(rmsnorm(X; norm_w, eps) . W^T + bias), M <= 16
 * (the normalised rows are staged in LDS: M * (2 K + 16) bytes <= 144 KiB).
 * Removes one launch per q/k/v, gate/up and lm_head projection of a decode step. */
int o3v_gemv_norm_bf16(const void* X, const void* norm_w, float eps, const void* W, const void* bias, const void* res,
                       void* out, int M, int N, int K, int ldx, int ldw, int ldo, int ldr, int epilogue,
                       o3v_stream_t stream);

/* Decode q/k/v projection fully fused: RMSNorm -> Linear(+bias) -> M-RoPE -> q to qout[M,Hq,D], k,v appended to the
 * cache [M,Hkv,Tmax,D] at `slot` (TF:733-736, :636-664); cos/sin row of sequence m = m*cs_stride_row + cs_off.
 * norm_w == NULL (M >= 4 only): X is already normalised (the batched decode runs o3v_rmsnorm apart, which frees the
 * LDS the fused form stages x in and doubles the resident waves). */
int o3v_gemv_norm_qkv_rope(const void* X, const void* norm_w, float eps, const void* W, const void* Wp, const void* bias,
                           int M, int K, int ldx, const void* cosT, const void* sinT, void* qout, void* kcache, void* vcache,
                           int slot, int Hq, int Hkv, int D, int Tmax, int cs_stride_row, int cs_off, o3v_stream_t stream);
/* Decode nn.Linear with both weight images: W row-major [N,K] (M == 1) and Wp = the same weights in MFMA-fragment-major
 * order [N/16][K/32][64][8] (4 <= M <= 16, group rollout / batched eval; 9..16 rows need Wp or an MFMA-shaped W); norm_w (fused RMSNorm) and Wp may be NULL. */
int o3v_linear_decode(const void* X, const void* norm_w, float eps, const void* W, const void* Wp, const void* bias,
                      const void* res, void* out, int M, int N, int K, int ldx, int ldo, int ldr, int epilogue,
                      o3v_stream_t stream);
/* The residual linears of a batched decode layer (o_proj, down_proj; TF:692-757 `hidden = residual + ...` followed by the next
 * RMSNorm, TF:65-79) with the NEXT norm folded in: out = X . W^T + res and h = RMSNorm(out; next_norm_w) in ONE launch, 8 <= M <= 32
 * rows of already normalised X, Wp the fragment-major image, N <= 4096.  The waves that draw the last M tickets of the launch
 * normalise one row each once every part of `out` is in memory; h is bit-identical to o3v_rmsnorm(out, next_norm_w).  sync: the
 * zeroed buffer of o3v_decode_sync_bytes() (lines of its own; the sticky time-out word at O3V_SYNC_TMO_BYTE is shared), epoch = 1,
 * 2, ... counts the calls made on it.  O3V_ERR_SHAPE when the form does not apply (not gfx950, shapes): issue o3v_linear_decode +
 * o3v_rmsnorm instead. */
int o3v_linear_decode_norm_next(const void* X, const void* Wp, const void* res, void* out, int M, int N, int K, int ldx, int ldo,
                                int ldr, const void* next_norm_w, float eps, void* h, int ldh, uint32_t* sync, uint32_t epoch,
                                o3v_stream_t stream);

/* Decode linears on fp8 (OCP e4m3fn) weights, M <= 3 rows: W8 uint8 [N, K] (K % 16 == 0) with one fp32 scale per output
 * row; out = epi(scale[n] * (rmsnorm(X; norm_w) . fp8(W8[n])) + bias), x and the accumulation as in the bf16 path (the fp8
 * values are widened to bf16 exactly, v_cvt_scalef32_pk_bf16_fp8, and meet x in v_dot2c_f32_bf16).  Same epilogues. */
int o3v_linear_decode_fp8(const void* X, const void* norm_w, float eps, const void* W8, const float* scale, const void* bias,
                          const void* res, void* out, int M, int N, int K, int ldx, int ldo, int ldr, int epilogue,
                          o3v_stream_t stream);
/* fp8 rows for 4 <= M <= 32 rows of already normalised x (batched decode, N = 16 self-consistency chains of BASELINE config #5):
 * W8p is the fragment-major fp8 image [N/16][K/64][64][16 B] of the matrix; the weights are widened exactly to bf16 in registers
 * and multiplied on the bf16 matrix cores, the row scale multiplies the fp32 sum.  Epilogues NONE / RESIDUAL / SWIGLU.
 * K % 64 == 0, N % 16 == 0 (SWIGLU: N % 32 == 0). */
int o3v_linear_decode_fp8_rows(const void* X, const void* W8p, const float* scale, const void* bias, const void* res, void* out,
                               int M, int N, int K, int ldx, int ldo, int ldr, int epilogue, o3v_stream_t stream);
/* q/k/v on fp8 rows for 4..32 rows of already normalised x: bias, M-RoPE and the cache append in the epilogue (the arguments of
 * o3v_gemv_norm_qkv_rope without the fused norm; W8p fragment-major as above). */
int o3v_qkv_rope_fp8_rows(const void* X, const void* W8p, const float* scale, const void* bias, int M, int K, int ldx, const void* cosT,
                          const void* sinT, void* qout, void* kcache, void* vcache, int slot, int Hq, int Hkv, int D, int Tmax,
                          int cs_stride_row, int cs_off, o3v_stream_t stream);
int o3v_gemv_norm_qkv_rope_fp8(const void* X, const void* norm_w, float eps, const void* W8, const float* scale,
                               const void* bias, int M, int K, int ldx, const void* cosT, const void* sinT, void* qout,
                               void* kcache, void* vcache, int slot, int Hq, int Hkv, int D, int Tmax, int cs_stride_row,
                               int cs_off, o3v_stream_t stream);

/* ---- attention ---------------------------------------------------------------------------------------------- */
/* tiles: int32[n_tiles][8] = {q_row0, q_rows (<= rows_per_tile), k_row0, k_len, causal_off, k_lo, batch, 0};
 * rows_per_tile 64 (ragged ViT windows) or 128 (prefill).
 * ViT varlen attention (TF:248-287) and causal GQA prefill attention (TF:186-208, :602-689). */
int o3v_attn_tiles(const void* Q, const void* K, const void* V, void* O, const int* tiles, int n_tiles, int rows_per_tile,
                   int Hq, int n_rep, int D, long q_ts, long k_ts, long k_hs, long k_bs, long v_ts, long v_hs, long v_bs,
                   long o_ts, float scale, o3v_stream_t stream);
/* The same with a shared prompt entry: keys 0..prefix_len-1 of a tile come from Kpre / Vpre ([batch / rows_per_prefix] at
 * stride p_bs, kv head at stride p_hs, token strides k_ts / v_ts), later keys from K / V whose rows then start at logical key
 * prefix_len -- the completions of one prompt run behind ONE copy of its K/V (R:grpo_trainer.py:601-632 log-prob pass). */
int o3v_attn_tiles_prefix(const void* Q, const void* K, const void* V, const void* Kpre, const void* Vpre, long p_hs, long p_bs,
                          int prefix_len, int rows_per_prefix, void* O, const int* tiles, int n_tiles, int rows_per_tile, int Hq,
                          int n_rep, int D, long q_ts, long k_ts, long k_hs, long k_bs, long v_ts, long v_hs, long v_bs, long o_ts,
                          float scale, o3v_stream_t stream);
/* q_len == 1 attention against the cache [B,Hkv,Tmax,D]; part_o: f32[B*Hq*nsplit*D], part_ml: f32[B*Hq*nsplit*2] */
int o3v_attn_decode(const void* Q, const void* Kc, const void* Vc, void* out, float* part_o, float* part_ml,
                    const int* k_lo, int B, int Hq, int Hkv, int D, int ctx, int Tmax, int nsplit, float scale,
                    o3v_stream_t stream);
/* The same for B = (B/G) groups of G rows whose first `prefix_len` keys are identical (the G completions of one
 * prompt: `num_return_sequences`, R:grpo_trainer.py:306-313, TF:1493-1579 `_expand_inputs_for_generation`; the n
 * samples of R:eval/tts.py:47-123).  The prefix K/V are read ONCE per group, from the cache row of the group's first
 * sequence, with the G*Hq/Hkv (<= 64) query rows of a kv head sharing every key tile; each row's own keys
 * prefix_len..ctx-1 come from its own cache row.  head_dim 128 only.  part_o/part_ml: 64 splits per (row, head). */
int o3v_attn_decode_group(const void* Q, const void* Kc, const void* Vc, void* out, float* part_o, float* part_ml,
                          const int* k_lo, int B, int G, int Hq, int Hkv, int D, int prefix_len, int ctx, int Tmax,
                          int nsplit_prefix, float scale, o3v_stream_t stream);
/* The same without per-row copies of the prompt: Kpre / Vpre [B / rows_per_prompt][Hkv][prefix_cap][D] hold each prompt's
 * K/V once, Kc / Vc [B][Hkv][Tmax][D] only the rows' generated tokens (logical key prefix_len + j in slot j).  G must divide
 * rows_per_prompt.  Kpre == NULL: exactly o3v_attn_decode_group. */
int o3v_attn_decode_group_prefix(const void* Q, const void* Kc, const void* Vc, const void* Kpre, const void* Vpre, int prefix_cap,
                                 int rows_per_prompt, void* out, float* part_o, float* part_ml, const int* k_lo, int B, int G,
                                 int Hq, int Hkv, int D, int prefix_len, int ctx, int Tmax, int nsplit_prefix, float scale,
                                 o3v_stream_t stream);

/* One launch for the attention half of a batch-1 decode layer (TF:692-757 first half; replaces o3v_gemv_norm_qkv_rope +
 * o3v_attn_decode + o3v_linear_decode(o_proj, RESIDUAL), bit-identical to them): the three stages are roles of the
 * workgroups of one grid, handing q / the new K,V row / the split partials / the attention output to one another through
 * write-through stores, tickets and per-workgroup mailbox lines in `sync` (o3v_decode_sync_bytes() bytes, 128-byte aligned,
 * zeroed ONCE by the caller; `epoch` = 1, 2, 3, ... counts the launches made on that buffer, with the same nsplit, Hq, Hkv).
 * The word at byte O3V_SYNC_TMO_BYTE of `sync` is a sticky time-out flag (non-zero: a bounded wait gave up, results invalid).
 * x bf16 [H]: residual stream, normalised by ln_w for q/k/v and updated in place by o_proj.  Returns O3V_ERR_SHAPE when
 * the shapes (head_dim 128, Hkv <= 8, H and Hq*D <= 4096 in a built combination) or the chip's residency do not allow
 * the fused form: call the three stand-alone entries instead. */
#define O3V_SYNC_TMO_BYTE 2048
size_t o3v_decode_sync_bytes(void);
int o3v_decode_attn_block_capacity(int qd, int wb); /* workgroups of the fused kernel resident at once; 0: shape not built */
int o3v_decode_attn_block(void* x, const void* ln_w, float eps, const void* qkv_w, const void* qkv_b, const void* o_w,
                          const void* cosT, const void* sinT, void* q_buf, void* att_buf, void* kcache, void* vcache,
                          float* part_o, float* part_ml, const int* k_lo, int H, int Hq, int Hkv, int D, int slot, int Tmax,
                          int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync, uint32_t epoch,
                          o3v_stream_t stream);

/* ---- fp8 x fp8 on the matrix cores (o3v_fp8.hip): the compute-bound linears as W8A8 (vLLM's fp8 linear method for the checkpoints the
 * reference serves with quantization="fp8"; BASELINE config #5).  OCP e4m3fn; every scale is a power of two. ---- */
/* q[r, :] = fp8(x[r, :] / scale[r]) with scale[r] = the smallest power of two that brings row r into +-448 (1 for a zero row). */
int o3v_quantize_rows_fp8(const void* x, void* q, float* scale, int rows, int cols, int ld_in, int ld_q, o3v_stream_t stream);
/* The same on y = RMSNorm(x) * w with o3v_rmsnorm's arithmetic (TF:65-79): the bf16 values the bf16 path would feed its linears. */
int o3v_rmsnorm_quantize_fp8(const void* x, const void* w, void* q, float* scale, int rows, int cols, int ld_in, int ld_q, float eps,
                             o3v_stream_t stream);
/* out[M, N] = epilogue((A8[M, K] . W8[N, K]^T) * sa[m] * sw[n] + bias) on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), fp32
 * accumulation; epilogue O3V_EPI_NONE / _RESIDUAL / _SWIGLU as o3v_gemm_bf16.  K % 128 == 0, lda / ldw in bytes (= elements). */
int o3v_gemm_fp8(const void* A8, const float* sa, const void* W8, const float* sw, const void* bias, const void* res, void* out, int M,
                 int N, int K, int lda, int ldw, int ldo, int ldr, int epilogue, o3v_stream_t stream);
/* o3v_gemm_fp8 with the schedule chosen by the caller (tests, A/B): 0 = the library's choice (the phased kernel wherever K has an even
 * number >= 4 of 128-byte K-tiles), 1 = the kernel with one __syncthreads() per K-tile, 2 = the phased kernel of csrc/o3v_fp8.hip
 * (o3v_gemm8p.hip's schedule; O3V_ERR_SHAPE where it does not apply).  Same bits either way. */
int o3v_gemm_fp8_sched(const void* A8, const float* sa, const void* W8, const float* sw, const void* bias, const void* res, void* out,
                       int M, int N, int K, int lda, int ldw, int ldo, int ldr, int epilogue, int schedule, o3v_stream_t stream);

/* The PERSISTENT layer block: RMSNorm + q/k/v (+bias, M-RoPE, cache append) -> attention -> merge -> o_proj + residual -> RMSNorm +
 * gate/up + SwiGLU of one decode layer at batch 1 as ONE launch (TF:692-757 up to the SwiGLU; down_proj + residual is the next
 * launch).  The grid is what the chip holds (three 256-thread workgroups per CU); every wave owns a static list of weight rows over
 * the three projections and keeps its next three rows in flight in registers -- also across the in-launch hand-offs, so ~60 MB of
 * o_proj / gate/up rows stream while the dependent attention chain runs.  act: bf16 [I] (input of down_proj).  Same arithmetic as
 * o3v_decode_attn_block + o3v_linear_decode(gate/up): bit-identical.  O3V_ERR_SHAPE unless head_dim 128, hidden == Hq*D a multiple of
 * 512 (2048, 3584) and the grid is resident: call the other entries instead.  sync: as o3v_decode_attn_block (its own ticket lines). */
int o3v_decode_layer_block(void* x, const void* ln1, float eps, const void* qkv_w, const void* qkv_b, const void* o_w, const void* ln2,
                           const void* gu_w, void* act, const void* cosT, const void* sinT, void* q_buf, void* att_buf, void* kcache,
                           void* vcache, float* part_o, float* part_ml, const int* k_lo, int H, int I, int Hq, int Hkv, int D, int slot,
                           int Tmax, int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync, uint32_t epoch,
                           o3v_stream_t stream);

/* o3v_decode_attn_block on fp8 (OCP e4m3fn) rows + per-row scales for the q/k/v and o projections (bit-identical to
 * o3v_gemv_norm_qkv_rope_fp8 + o3v_attn_decode + o3v_linear_decode_fp8) */
int o3v_decode_attn_block_fp8(void* x, const void* ln_w, float eps, const void* qkv_w8, const float* qkv_s, const void* qkv_b,
                              const void* o_w8, const float* o_s, const void* cosT, const void* sinT, void* q_buf, void* att_buf,
                              void* kcache, void* vcache, float* part_o, float* part_ml, const int* k_lo, int H, int Hq, int Hkv,
                              int D, int slot, int Tmax, int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync,
                              uint32_t epoch, o3v_stream_t stream);
/* Qwen3-VL form (TF3:438-500): no q/k/v bias; RMSNorm weights q_norm / k_norm bf16 [D] on every q and k head between the
 * projection and the rotation.  kv_raw: (Hq + 2 * Hkv) * D bf16 of scratch.  qkv_s / o_s NULL: bf16 rows; non-NULL: fp8 rows + scales.
 * Bit-identical to o3v_linear_decode (q/k/v) + o3v_qkv_norm_rope_cache + o3v_attn_decode + o3v_linear_decode (o_proj, RESIDUAL). */
int o3v_decode_attn_block_qknorm(void* x, const void* ln_w, float eps, const void* qkv_w, const float* qkv_s, const void* o_w,
                                 const float* o_s, const void* q_norm, const void* k_norm, void* kv_raw, const void* cosT,
                                 const void* sinT, void* q_buf, void* att_buf, void* kcache, void* vcache, float* part_o,
                                 float* part_ml, const int* k_lo, int H, int Hq, int Hkv, int D, int slot, int Tmax,
                                 int cs_stride_row, int cs_off, int nsplit, float scale, uint32_t* sync, uint32_t epoch,
                                 o3v_stream_t stream);

/* ---- sampling / log-probs ------------------------------------------------------------------------------------ */
/* GenerationMixin._sample greedy branch + RepetitionPenaltyLogitsProcessor,
 * TF:generation/utils.py:2894-2929, TF:generation/logits_process.py:404-414 */
#define O3V_SAMPLE_SCRATCH_FLOATS 40960 /* per-row accumulators of o3v_sample_top_k_top_p (histograms, slice maxima / masses) */
int o3v_sample_greedy(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* margins,
                      const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty, int step,
                      int out_stride, float* scratch /* f32[B*256] */, o3v_stream_t stream);
/* o3v_sample_greedy + embed_tokens of the chosen token (TF:1206-1207) in the same launch: x_out bf16 [B, hidden] */
int o3v_sample_greedy_embed(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* margins,
                            const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty, int step,
                            int out_stride, float* scratch, const void* embed, void* x_out, int hidden, o3v_stream_t stream);
/* temperature + top-p + multinomial (TF:logits_process.py:301-303, :527-539; utils.py:2921-2923), counter-based RNG
 * keyed by (seed, row_id[b], step) so a completion does not depend on which rank/batch slot produced it */
int o3v_sample_top_p(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* chosen_logprob,
                     const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty, float temperature,
                     float top_p, uint64_t seed, const int* row_id, int step, int out_stride,
                     float* scratch /* f32[B * O3V_SAMPLE_SCRATCH_FLOATS], 8-byte aligned */, o3v_stream_t stream);
/* the same with TopKLogitsWarper (TF:logits_process.py:590-594) between temperature and top-p: scores below the k-th
 * largest are removed (ties with it stay), top-p then acts on the softmax over what is left.  top_k == 0: no top-k. */
int o3v_sample_top_k_top_p(const void* logits, void* seen, int* cur_tok, int* finished, int* out_ids, float* chosen_logprob,
                           const int* eos_ids, int n_eos, int pad_id, int B, int V, int ldl, float rep_penalty,
                           float temperature, int top_k, float top_p, uint64_t seed, const int* row_id, int step,
                           int out_stride, float* scratch, o3v_stream_t stream);
int o3v_mark_seen(const int* ids, void* seen, int B, int S, int V, o3v_stream_t stream);
/* _get_per_token_logps, R:grpo_trainer.py:371-384: out[r] = log_softmax(logits[r])[target[r]] */
int o3v_logprob_gather(const void* logits, const int* target, float* out, int R, int V, int ldl, o3v_stream_t stream);

/* ---- model-level engine (layer loops in C++, no Python per kernel) ------------------------------------------- */
typedef struct {
    const void *norm1, *norm2;     /* [hid] */
    const void *qkv_w, *qkv_b;     /* [3hid,hid], [3hid] */
    const void *proj_w, *proj_b;   /* [hid,hid], [hid] */
    const void *gu_w, *gu_b;       /* packed gate/up [2*ipad,hid], [2*ipad] (16-row interleave, zero pad rows) */
    const void *down_w, *down_b;   /* [hid,ipad] (zero pad cols), [hid] */
} o3v_vit_block_w;

typedef struct {
    int depth, hidden, heads, inter_pad, out_hidden, patch_k_pad, merge_unit;
    uint64_t fullatt_mask;         /* bit i set: block i attends over whole frames (TF:448-454) */
    const void* patch_w;           /* [hidden, patch_k_pad] (Conv3d weight flattened, zero pad cols)  TF:99-122 */
    const o3v_vit_block_w* blocks; /* [depth] */
    const void *ln_q, *m0_w, *m0_b, *m2_w, *m2_b; /* merger TF:137-150 */
    int gemm_tile;                 /* 0 = per shape; 128 / 256 = force that GEMM kernel (tests) */
} o3v_vit_desc;

typedef struct {
    const void *ln1, *ln2;       /* [H] */
    const void *qkv_w, *qkv_b;   /* fused q|k|v [(Hq+2Hkv)*D, H] */
    const void* o_w;             /* [H, Hq*D] */
    const void* gu_w;            /* packed gate/up [2*I, H] */
    const void* down_w;          /* [H, I] */
    /* optional MFMA-fragment-major copies of the four matrices for the M >= 2 decode path (NULL = row-major only) */
    const void *qkv_wp, *o_wp, *gu_wp, *down_wp;
    /* optional fp8 (OCP e4m3fn) copies of the four matrices, same row layouts, with one fp32 scale per output row
     * (NULL = bf16 only): the batch <= 3 decode streams these instead -- half the bytes (o3v_linear_decode_fp8) */
    const void *qkv_w8, *o_w8, *gu_w8, *down_w8;
    const float *qkv_s, *o_s, *gu_s, *down_s;
    /* Qwen3-VL (TF3:438-500): RMSNorm weights [D] applied per head to q and k before the rotation; NULL = none (Qwen2.5-VL).
     * With them qkv_b is NULL (attention_bias false) and the decode runs q/k/v as linear + o3v_qkv_norm_rope_cache. */
    const void *q_norm, *k_norm;
    /* optional fragment-major fp8 images [N/16][K/64][64][16 B] of the four matrices (same scales): 4..32 decode rows stream
     * these (o3v_linear_decode_fp8_rows); q/k/v only without the fused rotation (Qwen3-VL) */
    const void *qkv_w8p, *o_w8p, *gu_w8p, *down_w8p;
} o3v_llm_layer_w;

typedef struct {
    int hidden, layers, heads, kv_heads, head_dim, inter, vocab;
    float rms_eps;
    const void* embed;           /* [vocab, H] */
    const o3v_llm_layer_w* layer;/* [layers] */
    const void* final_norm;      /* [H] */
    const void* lm_head;         /* [vocab, H] (== embed when tied) */
    const void* lm_head_p;       /* optional fragment-major copy of lm_head */
    int gemm_tile;               /* 0 = per shape; 128 / 256 = force that GEMM kernel in the prefill (tests) */
    const void* lm_head8;        /* optional fp8 copy of lm_head + per-row scales (decode, <= 3 rows) */
    const float* lm_head_s;
    const void* lm_head8p;       /* optional fragment-major fp8 image of lm_head (4..32 rows) */
} o3v_llm_desc;

size_t o3v_vit_workspace_bytes(const o3v_vit_desc* d, int P);
/* Qwen2_5_VisionTransformerPretrainedModel.forward, TF:408-471.  pixels: bf16 [P, patch_k_pad]; win_idx/rev_idx
 * int32 [P/merge_unit]; cos/sin f32 [P, head_dim/2] in window order; out bf16 [P/merge_unit, out_hidden]. */
int o3v_vit_forward(const o3v_vit_desc* d, const void* pixels, int P, const int* win_idx, const int* rev_idx,
                    const float* cosT, const float* sinT, const int* tiles_win, int n_tiles_win, const int* tiles_full,
                    int n_tiles_full, void* workspace, size_t ws_bytes, void* out, o3v_stream_t stream);

/* ---- Qwen3-VL vision tower (Qwen3VLVisionModel.forward, TF3:606-737) */
#define O3V_MAX_DEEPSTACK 8
typedef struct {
    const void *norm1_w, *norm1_b;  /* LayerNorm [hidden] */
    const void *qkv_w, *qkv_b;      /* [3*heads*head_dim_pad, pad64(hidden)]: per head [36|0000|36|0000]-style padded rows */
    const void *proj_w, *proj_b;    /* [hidden, heads*head_dim_pad] (zero columns at the padded dims) */
    const void *norm2_w, *norm2_b;
    const void *fc1_w, *fc1_b;      /* [inter_pad, pad64(hidden)] (zero rows past intermediate_size) */
    const void *fc2_w, *fc2_b;      /* [hidden, inter_pad] */
} o3v_vit3_block_w;
typedef struct {
    const void *norm_w, *norm_b;    /* LayerNorm over hidden (postshuffle 0) or hidden*merge_unit (postshuffle 1) */
    const void *fc1_w, *fc1_b;      /* [hidden*unit, hidden*unit], GELU(erf) */
    const void *fc2_w, *fc2_b;      /* [out_hidden, hidden*unit] */
    int postshuffle;
} o3v_vit3_merger_w;
typedef struct {
    int depth, hidden, heads, head_dim, head_dim_pad, inter_pad, out_hidden, patch_k_pad, merge_unit;
    int n_deep;                       /* DeepStack taps */
    int deep_index[O3V_MAX_DEEPSTACK];/* ascending block indexes (deepstack_visual_indexes) */
    int gemm_tile;
    const void *patch_w, *patch_b;    /* [hidden, patch_k_pad], [hidden] */
    const o3v_vit3_block_w* blocks;   /* [depth] */
    o3v_vit3_merger_w merger;
    o3v_vit3_merger_w deep[O3V_MAX_DEEPSTACK];
} o3v_vit3_desc;
size_t o3v_vit3_workspace_bytes(const o3v_vit3_desc* d, int P);
/* pixels bf16 [P, patch_k_pad] in merge-block order; pos_embed bf16 [P, hidden]: the learned position table interpolated to
 * this grid (TF3:643-710, host side); cos/sin f32 [P, head_dim_pad/2]; tiles: one non-causal segment per temporal patch.
 * out bf16 [P/unit, out_hidden]; deep_out bf16 [n_deep][P/unit, out_hidden]. */
int o3v_vit3_forward(const o3v_vit3_desc* d, const void* pixels, int P, const void* pos_embed, const float* cosT,
                     const float* sinT, const int* tiles, int n_tiles, void* workspace, size_t ws_bytes, void* out, void* deep_out,
                     o3v_stream_t stream);

/* ---- context handle: the library keeps NO mutable state of its own; a context is an owning host-side copy of
 * the model descriptors (the structs above with their block / layer arrays; the device buffers they point at stay the
 * caller's), so that an FFI caller can drop its own structs after create.  Any descriptor may be NULL.  Host code only: no
 * HIP call, usable without a GPU.  The model-level entries take the descriptor pointers the accessors return. */
typedef struct o3v_ctx o3v_ctx;
o3v_ctx* o3v_ctx_create(const o3v_llm_desc* llm, const o3v_vit_desc* vit, const o3v_vit3_desc* vit3);
void o3v_ctx_destroy(o3v_ctx* ctx);
const o3v_llm_desc* o3v_ctx_llm(const o3v_ctx* ctx);
const o3v_vit_desc* o3v_ctx_vit(const o3v_ctx* ctx);
const o3v_vit3_desc* o3v_ctx_vit3(const o3v_ctx* ctx);

size_t o3v_llm_workspace_bytes(const o3v_llm_desc* d, int rows);
/* Qwen2_5_VLTextModel.forward over the prompt, TF:790-872: x bf16 [B*S,H] holds inputs_embeds on entry and the
 * last layer's residual stream on return; K/V caches [layers][B][Hkv][Tmax][D].
 * past > 0: the caches already hold `past` tokens per row (a shared prompt prefix, TF cache semantics of
 * `past_key_values`: DynamicCache.update appends); the S new tokens land in slots past..past+S-1 and the causal
 * tiles carry causal_off = past + q0, k_len = past + S. */
int o3v_llm_prefill(const o3v_llm_desc* d, void* x, const void* cosT, const void* sinT, const int* tiles, int n_tiles,
                    int rows_per_tile, void* kcache, void* vcache, int B, int S, int past, int Tmax, void* workspace,
                    size_t ws_bytes, o3v_stream_t stream);
/* Options of o3v_llm_prefill_ex (all optional; a zeroed struct == o3v_llm_prefill):
 *  * DeepStack (Qwen3-VL, TF3:839-862): after decoder layer l < n_deep, x[ds_rows[i]] += ds_feat[l][ds_src[i]] for i < n_ds
 *    (the visual rows among this call's B*S rows); ds_feat: n_deep tables, ds_stride bf16 elements apart.
 *  * shared prompt entry: the first prefix_len of the `past` tokens of every row live ONCE per prompt in kprefix / vprefix
 *    ([layers][B / rows_per_prefix][Hkv][prefix_cap][D]); kcache / vcache [layers][B][Hkv][Tmax][D] then hold only the tokens
 *    behind them (this call's S tokens land in slots past - prefix_len ..) -- the G completions of one prompt run behind one
 *    copy of its K/V (R:grpo_trainer.py:601-632).  Tile descriptors keep logical key indexes (k_len = past + S).
 *  * w8a8 (BASELINE config #5, "fp8 weights on CDNA4 fp8 MFMA"): 1 = run the four linears of every layer as fp8 x fp8 on the matrix
 *    cores (o3v_gemm_fp8: per-token activation scales, the per-row weight scales of the fp8 rows) when the descriptor carries fp8
 *    rows and the widths are whole 128-byte k-tiles, else the bf16 path; 2 = the same but O3V_ERR_SHAPE instead of the bf16 path. */
typedef struct {
    const int *ds_rows, *ds_src;
    int n_ds;
    const void* ds_feat;
    int n_deep;
    long ds_stride;
    const void *kprefix, *vprefix;
    int prefix_len, prefix_cap, rows_per_prefix;
    int w8a8;
} o3v_prefill_opts;
int o3v_llm_prefill_ex(const o3v_llm_desc* d, void* x, const void* cosT, const void* sinT, const int* tiles, int n_tiles,
                       int rows_per_tile, void* kcache, void* vcache, int B, int S, int past, int Tmax,
                       const o3v_prefill_opts* opts, void* workspace, size_t ws_bytes, o3v_stream_t stream);
/* final norm + lm_head on `rows` rows of x (row stride ldx): logits bf16 [rows, vocab]  TF:867, :1386-1387.  Always the bf16
 * head (prefill, forward_logits, the log-prob pass); the fp8 head is used inside o3v_llm_decode only. */
int o3v_llm_head(const o3v_llm_desc* d, const void* x, int ldx, int rows, void* normed, void* logits,
                 o3v_stream_t stream);

typedef struct {
    int B, S, Tmax, Tnew;        /* rows, prompt slots already in the cache, cache capacity, decode table length */
    int nsplit;                  /* context splits of the decode attention */
    int pad_id, n_eos;
    int do_sample;               /* 0 greedy, 1 temperature/top-p multinomial */
    float rep_penalty, temperature, top_p;
    uint64_t seed;
    void *x;                     /* bf16 [B,H] scratch: residual stream of the current token */
    void *kcache, *vcache;       /* [layers][B][Hkv][Tmax][D] */
    const void *cosT, *sinT;     /* bf16 [B,Tnew,D]: M-RoPE table of the decode positions (TF:1164-1174) */
    void *logits;                /* bf16 [B,vocab]; holds the logits to sample from on entry */
    void *seen;                  /* u8 [B,vocab] */
    int *cur_tok, *finished;     /* [B] */
    int *out_ids;                /* [B,Tnew] */
    float *margins;              /* [B,Tnew] top1-top2 (greedy) or chosen log-prob (sampling); may be NULL */
    const int *eos_ids;          /* [n_eos] */
    const int *k_lo;             /* [B] left-pad counts or NULL */
    const int *row_id;           /* [B] global completion index for the RNG or NULL */
    float *part_o, *part_ml;     /* decode-attention split buffers */
    float *sample_scratch;       /* f32 [B, O3V_SAMPLE_SCRATCH_FLOATS] when sampling, f32 [B,256] for greedy */
    void *workspace; size_t ws_bytes;
    int group;                   /* > 1: rows g*group..g*group+group-1 share their first S keys -> o3v_attn_decode_group
                                    (nsplit is then the prefix split count; part_o/part_ml hold 64 splits) */
    uint32_t *sync;              /* optional: o3v_decode_sync_bytes() bytes, 128-byte aligned, zeroed by the caller before the
                                    first step of a generate call.  Non-NULL selects the one-launch attention block
                                    (o3v_decode_attn_block) where its shapes allow (B == 1) */
    int top_k;                   /* sampling only: TopKLogitsWarper's k, 0 = off */
    /* optional shared prompt entries (needs group > 1): kprefix / vprefix [layers][B / rows_per_prompt][Hkv][prefix_cap][D] hold
     * each prompt's S keys ONCE; kcache / vcache [layers][B][Hkv][Tmax][D] then hold only the generated tokens (token `step` in
     * slot `step`, Tmax >= Tnew - 1) instead of a copy of the prompt per row. */
    const void *kprefix, *vprefix;
    int prefix_cap, rows_per_prompt;
    /* optional HOST array of 4 counters the call adds to: decode forwards, kernel launches inside their layer loops, layers whose
     * attention half ran as the one-launch block, layers whose attention half ran on the stand-alone kernels */
    long long* host_stats;
    /* bit 0: batch-1 bf16 decode may use the persistent layer block (o3v_decode_layer_block) where its shapes allow, then one
     * launch for the attention half + gate/up and one for down_proj per layer; otherwise o3v_decode_attn_block + two launches */
    int flags;
} o3v_decode_state;

/* GenerationMixin._sample loop, TF:generation/utils.py:2783-2942, steps [step0, step0+n_steps): sample from
 * st->logits, then (unless it is the globally last step) run one decode forward to refill st->logits. */
int o3v_llm_decode(const o3v_llm_desc* d, const o3v_decode_state* st, int step0, int n_steps, int skip_last_forward,
                   o3v_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* O3V_H */
