"""ctypes binding of libo3v_hip.so (include/o3v.h).  The library is built in-tree by
``open_o3_video_amd.build``; there is NO fallback: if it is missing or a call fails we raise."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libo3v_hip.so")

OK, ERR_ARG, ERR_SHAPE, ERR_LAUNCH, ERR_WORKSPACE = 0, -1, -2, -3, -4
_ERR = {ERR_ARG: "bad argument", ERR_SHAPE: "unsupported shape", ERR_LAUNCH: "HIP launch failure",
        ERR_WORKSPACE: "workspace too small"}
EPI_NONE, EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU, EPI_GELU_TANH = 0, 1, 2, 3, 6
MAX_DEEPSTACK = 8

vp, ip, fp, i32, i64, f32, u64, sz = (C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_long,
                                      C.c_float, C.c_uint64, C.c_size_t)


class VitBlockW(C.Structure):
    _fields_ = [(n, vp) for n in ("norm1", "norm2", "qkv_w", "qkv_b", "proj_w", "proj_b", "gu_w", "gu_b", "down_w", "down_b")]


class VitDesc(C.Structure):
    _fields_ = [("depth", i32), ("hidden", i32), ("heads", i32), ("inter_pad", i32), ("out_hidden", i32),
                ("patch_k_pad", i32), ("merge_unit", i32), ("fullatt_mask", u64), ("patch_w", vp),
                ("blocks", C.POINTER(VitBlockW)), ("ln_q", vp), ("m0_w", vp), ("m0_b", vp), ("m2_w", vp), ("m2_b", vp), ("gemm_tile", i32)]


class LlmLayerW(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1", "ln2", "qkv_w", "qkv_b", "o_w", "gu_w", "down_w", "qkv_wp", "o_wp", "gu_wp", "down_wp",
                                  "qkv_w8", "o_w8", "gu_w8", "down_w8", "qkv_s", "o_s", "gu_s", "down_s", "q_norm", "k_norm",
                                  "qkv_w8p", "o_w8p", "gu_w8p", "down_w8p")]


class Vit3BlockW(C.Structure):
    _fields_ = [(n, vp) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b",
                                  "fc2_w", "fc2_b")]


class Vit3MergerW(C.Structure):
    _fields_ = [(n, vp) for n in ("norm_w", "norm_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")] + [("postshuffle", i32)]


class Vit3Desc(C.Structure):
    _fields_ = [("depth", i32), ("hidden", i32), ("heads", i32), ("head_dim", i32), ("head_dim_pad", i32), ("inter_pad", i32),
                ("out_hidden", i32), ("patch_k_pad", i32), ("merge_unit", i32), ("n_deep", i32), ("deep_index", i32 * MAX_DEEPSTACK),
                ("gemm_tile", i32), ("patch_w", vp), ("patch_b", vp), ("blocks", C.POINTER(Vit3BlockW)), ("merger", Vit3MergerW),
                ("deep", Vit3MergerW * MAX_DEEPSTACK)]


class LlmDesc(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("kv_heads", i32), ("head_dim", i32), ("inter", i32),
                ("vocab", i32), ("rms_eps", f32), ("embed", vp), ("layer", C.POINTER(LlmLayerW)), ("final_norm", vp),
                ("lm_head", vp), ("lm_head_p", vp), ("gemm_tile", i32), ("lm_head8", vp), ("lm_head_s", vp), ("lm_head8p", vp)]


class DecodeState(C.Structure):
    _fields_ = [("B", i32), ("S", i32), ("Tmax", i32), ("Tnew", i32), ("nsplit", i32), ("pad_id", i32), ("n_eos", i32),
                ("do_sample", i32), ("rep_penalty", f32), ("temperature", f32), ("top_p", f32), ("seed", u64),
                ("x", vp), ("kcache", vp), ("vcache", vp), ("cosT", vp), ("sinT", vp), ("logits", vp), ("seen", vp),
                ("cur_tok", vp), ("finished", vp), ("out_ids", vp), ("margins", vp), ("eos_ids", vp), ("k_lo", vp),
                ("row_id", vp), ("part_o", vp), ("part_ml", vp), ("sample_scratch", vp), ("workspace", vp),
                ("ws_bytes", sz), ("group", i32), ("sync", vp), ("top_k", i32), ("kprefix", vp), ("vprefix", vp),
                ("prefix_cap", i32), ("rows_per_prompt", i32), ("host_stats", vp), ("flags", i32)]


class PrefillOpts(C.Structure):
    _fields_ = [("ds_rows", vp), ("ds_src", vp), ("n_ds", i32), ("ds_feat", vp), ("n_deep", i32), ("ds_stride", i64),
                ("kprefix", vp), ("vprefix", vp), ("prefix_len", i32), ("prefix_cap", i32), ("rows_per_prefix", i32), ("w8a8", i32)]


# name -> argtypes (return type int unless listed in _RET)
SIGNATURES = {
    "o3v_abi_version": [],
    "o3v_ctx_create": [C.POINTER(LlmDesc), C.POINTER(VitDesc), C.POINTER(Vit3Desc)],
    "o3v_ctx_destroy": [vp],
    "o3v_ctx_llm": [vp],
    "o3v_ctx_vit": [vp],
    "o3v_ctx_vit3": [vp],
    "o3v_rmsnorm": [vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "o3v_layernorm": [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "o3v_qkv_norm_rope_cache": [vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_add_rows": [vp, vp, vp, vp, i32, i32, vp],
    "o3v_patchify_ps": [vp, i32, vp, i32, i32, i32, i32, i32, fp, fp, vp],
    "o3v_patchify_video": [vp, i32, vp, i32, i32, i32, i32, i32, fp, fp, vp],
    "o3v_vit_rope": [vp, vp, vp, i32, i32, i32, vp],
    "o3v_mrope_table": [vp, vp, vp, vp, vp, i32, i32, vp],
    "o3v_qkv_rope_cache": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gather_rows": [vp, vp, vp, i32, i32, vp],
    "o3v_embed_scatter": [vp, vp, vp, vp, i32, i32, vp],
    "o3v_embed_tokens": [vp, vp, vp, i32, i32, vp],
    "o3v_cast_pad_f32_bf16": [vp, vp, i32, i32, i32, vp],
    "o3v_patchify": [vp, i32, vp, i32, i32, i32, i32, fp, fp, vp],
    "o3v_content_hash128": [vp, sz, vp, vp],
    "o3v_gemm_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gemv_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_attn_tiles": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i64, i64, f32, vp],
    "o3v_attn_decode": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "o3v_gemm_bf16_tile": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gemm_bf16_phased": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gemm_bf16_splitk": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp],
    "o3v_resize_bicubic_aa": [vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp],
    "o3v_crop_resize_bilinear": [vp, vp, vp, i32, i32, i32, i32, vp],
    "o3v_attn_decode_group": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "o3v_decode_sync_bytes": [],
    "o3v_decode_attn_block_capacity": [i32, i32],
    "o3v_decode_attn_block_fp8": [vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                  i32, i32, i32, f32, vp, C.c_uint32, vp],
    "o3v_decode_attn_block_qknorm": [vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32,
                                     i32, i32, i32, i32, i32, f32, vp, C.c_uint32, vp],
    "o3v_decode_attn_block": [vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                              i32, f32, vp, C.c_uint32, vp],
    "o3v_quantize_rows_fp8": [vp, vp, vp, i32, i32, i32, i32, vp],
    "o3v_rmsnorm_quantize_fp8": [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "o3v_gemm_fp8": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gemm_fp8_sched": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_decode_layer_block": [vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32,
                               i32, i32, i32, f32, vp, C.c_uint32, vp],
    "o3v_sample_greedy": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp],
    "o3v_sample_greedy_embed": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, i32, vp],
    "o3v_gemv_norm_qkv_rope": [vp, vp, f32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_qkv_rope_fp8_rows": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_linear_decode_fp8_rows": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_linear_decode_fp8": [vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_gemv_norm_qkv_rope_fp8": [vp, vp, f32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_linear_decode": [vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_linear_decode_norm_next": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, f32, vp, i32, vp, C.c_uint32, vp],
    "o3v_gemv_norm_bf16": [vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "o3v_sample_top_p": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, f32, u64, vp, i32, i32, vp, vp],
    "o3v_sample_top_k_top_p": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, i32, f32, u64, vp, i32, i32, vp, vp],
    "o3v_mark_seen": [vp, vp, i32, i32, i32, vp],
    "o3v_logprob_gather": [vp, vp, vp, i32, i32, i32, vp],
    "o3v_vit_workspace_bytes": [C.POINTER(VitDesc), i32],
    "o3v_vit_forward": [C.POINTER(VitDesc), vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, sz, vp, vp],
    "o3v_llm_workspace_bytes": [C.POINTER(LlmDesc), i32],
    "o3v_llm_prefill": [C.POINTER(LlmDesc), vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, sz, vp],
    "o3v_vit3_workspace_bytes": [C.POINTER(Vit3Desc), i32],
    "o3v_vit3_forward": [C.POINTER(Vit3Desc), vp, i32, vp, vp, vp, vp, i32, vp, sz, vp, vp, vp],
    "o3v_llm_prefill_ex": [C.POINTER(LlmDesc), vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, C.POINTER(PrefillOpts), vp, sz, vp],
    "o3v_attn_tiles_prefix": [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i64,
                              i64, f32, vp],
    "o3v_attn_decode_group_prefix": [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "o3v_llm_head": [C.POINTER(LlmDesc), vp, i32, i32, vp, vp, vp],
    "o3v_llm_decode": [C.POINTER(LlmDesc), C.POINTER(DecodeState), i32, i32, i32, vp],
}
_RET = {"o3v_ctx_create": vp, "o3v_ctx_destroy": None, "o3v_ctx_llm": C.POINTER(LlmDesc), "o3v_ctx_vit": C.POINTER(VitDesc),
        "o3v_ctx_vit3": C.POINTER(Vit3Desc), "o3v_vit_workspace_bytes": sz, "o3v_vit3_workspace_bytes": sz, "o3v_llm_workspace_bytes": sz, "o3v_decode_sync_bytes": sz}

SYNC_TMO_BYTE = 2048            # O3V_SYNC_TMO_BYTE in include/o3v.h
SAMPLE_SCRATCH_FLOATS = 40960   # O3V_SAMPLE_SCRATCH_FLOATS in include/o3v.h

_lib = None


class O3VError(RuntimeError):
    pass


def load():
    """Load the HIP library (raises if it has not been built: there is no CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise O3VError(f"{LIB_PATH} not found: run `python -m open_o3_video_amd.build` (hipcc, gfx950)")
        # torch-ROCm bundles its own libamdhip64.so.7; import it FIRST so our DT_NEEDED entry resolves to the runtime
        # that owns torch's streams and allocations (two HIP runtimes in one process cannot launch on each other's
        # streams: "HIP launch failure").
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and this table diverge
            fn.argtypes = args
            fn.restype = _RET.get(name, i32)
        if lib.o3v_abi_version() != 6:
            raise O3VError("libo3v_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != OK:
        raise O3VError(f"{what} failed: {_ERR.get(rc, rc)} ({rc})")


def call(name: str, *args):
    check(getattr(load(), name)(*args), name)
