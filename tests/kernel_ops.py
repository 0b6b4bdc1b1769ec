"""Test helpers: thin torch-tensor wrappers over the per-op entry points of the C ABI (the engine calls the model-level
entry points instead).  Every wrapper launches on the current torch stream.  Used by tests/ and tools/ only."""
from __future__ import annotations

import ctypes as C

import torch

from open_o3_video_amd import _lib
from open_o3_video_amd._lib import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU  # noqa: F401


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rmsnorm(x, w, eps):
    out = torch.empty_like(x)
    _lib.call("o3v_rmsnorm", _p(x), _p(w), _p(out), x.shape[0], x.shape[1], x.stride(0), out.stride(0), float(eps), _s())
    return out


def gemm(a, w, bias=None, res=None, epi=EPI_NONE, force=None, tile=0):
    """out = epi(a @ w.T + bias).  force in {None,'gemm','gemv'}; tile 0 / 128 / 256 picks the MFMA GEMM kernel."""
    M, K = a.shape
    N = w.shape[0]
    No = N // 2 if epi == EPI_SWIGLU else N
    out = torch.empty((M, No), dtype=torch.bfloat16, device=a.device)
    ldr = 0 if res is None else res.stride(0)
    if force == "gemv" or (force is None and M <= 8):
        _lib.call("o3v_gemv_bf16", _p(a), _p(w), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr, epi, _s())
    else:
        _lib.call("o3v_gemm_bf16_tile", _p(a), _p(w), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr,
                  epi, tile, _s())
    return out


def gemm_splitk(a, w, bias=None, res=None, epi=EPI_NONE, splits=8):
    """Split-K form of `gemm` for few-tile shapes (epilogues NONE / RESIDUAL / GELU)."""
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    ws = torch.empty(splits * M * N, dtype=torch.float32, device=a.device)
    _lib.call("o3v_gemm_bf16_splitk", _p(a), _p(w), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0),
              0 if res is None else res.stride(0), epi, splits, _p(ws), ws.numel() * 4, _s())
    return out


def vit_rope_(qkv, cos, sin, H, D):
    _lib.call("o3v_vit_rope", _p(qkv), _p(cos), _p(sin), qkv.shape[0], H, D, _s())
    return qkv


def attn_tiles(q, k, v, tiles, Hq, n_rep, D, q_ts, k_ts, k_hs, k_bs, v_ts, v_hs, v_bs, out, o_ts, scale, rows_per_tile=64):
    _lib.call("o3v_attn_tiles", _p(q), _p(k), _p(v), _p(out), _p(tiles), tiles.shape[0], rows_per_tile, Hq, n_rep, D, q_ts, k_ts, k_hs,
              k_bs, v_ts, v_hs, v_bs, o_ts, float(scale), _s())
    return out


def attn_decode(q, kc, vc, k_lo, ctx, nsplit, scale):
    B, Hq, D = q.shape
    _, Hkv, Tmax, _ = kc.shape
    out = torch.empty_like(q)
    ns = abs(nsplit)  # negative nsplit selects the scalar (non-MFMA) kernel
    po = torch.empty(B * Hq * ns * D, dtype=torch.float32, device=q.device)
    pm = torch.empty(B * Hq * ns * 2, dtype=torch.float32, device=q.device)
    _lib.call("o3v_attn_decode", _p(q), _p(kc), _p(vc), _p(out), _p(po), _p(pm), _p(k_lo), B, Hq, Hkv, D, ctx, Tmax, nsplit,
              float(scale), _s())
    return out


def attn_decode_group(q, kc, vc, k_lo, G, prefix_len, ctx, nsplit_prefix, scale):
    B, Hq, D = q.shape
    _, Hkv, Tmax, _ = kc.shape
    out = torch.empty_like(q)
    po = torch.empty(B * Hq * 64 * D, dtype=torch.float32, device=q.device)
    pm = torch.empty(B * Hq * 64 * 2, dtype=torch.float32, device=q.device)
    _lib.call("o3v_attn_decode_group", _p(q), _p(kc), _p(vc), _p(out), _p(po), _p(pm), _p(k_lo), B, G, Hq, Hkv, D, prefix_len, ctx,
              Tmax, nsplit_prefix, float(scale), _s())
    return out


def attn_decode_group_prefix(q, kc_own, vc_own, kpre, vpre, rows_per_prompt, k_lo, G, prefix_len, ctx, nsplit_prefix, scale):
    """kc_own / vc_own [B, Hkv, Tmax_own, D]: generated tokens only; kpre / vpre [B / rows_per_prompt, Hkv, cap, D]."""
    B, Hq, D = q.shape
    _, Hkv, Tmax, _ = kc_own.shape
    out = torch.empty_like(q)
    po = torch.empty(B * Hq * 64 * D, dtype=torch.float32, device=q.device)
    pm = torch.empty(B * Hq * 64 * 2, dtype=torch.float32, device=q.device)
    _lib.call("o3v_attn_decode_group_prefix", _p(q), _p(kc_own), _p(vc_own), _p(kpre), _p(vpre), kpre.shape[2], rows_per_prompt, _p(out),
              _p(po), _p(pm), _p(k_lo), B, G, Hq, Hkv, D, prefix_len, ctx, Tmax, nsplit_prefix, float(scale), _s())
    return out


def attn_tiles_prefix(q, k, v, kpre, vpre, p_hs, p_bs, prefix_len, rows_per_prefix, tiles, Hq, n_rep, D, q_ts, k_ts, k_hs, k_bs, v_ts,
                      v_hs, v_bs, out, o_ts, scale, rows_per_tile=64):
    _lib.call("o3v_attn_tiles_prefix", _p(q), _p(k), _p(v), _p(kpre), _p(vpre), p_hs, p_bs, prefix_len, rows_per_prefix, _p(out),
              _p(tiles), tiles.shape[0], rows_per_tile, Hq, n_rep, D, q_ts, k_ts, k_hs, k_bs, v_ts, v_hs, v_bs, o_ts, float(scale), _s())
    return out


def gather_rows(src, idx):
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=src.dtype, device=src.device)
    _lib.call("o3v_gather_rows", _p(src), _p(idx), _p(out), idx.shape[0], src.shape[1] * src.element_size(), _s())
    return out
