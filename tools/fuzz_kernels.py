#!/usr/bin/env python3
"""Randomised shape sweep of the GEMM / skinny-GEMM / decode-attention entry points against fp32 torch references
(tolerances of tests/test_gpu_kernels.py).  Not part of the test suite: a longer soak to run on a GPU box by hand."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from open_o3_video_amd import _lib, indexing  # noqa: E402
import kernel_ops as ops  # noqa: E402  (tests/kernel_ops.py: thin ctypes wrappers of the C ABI)
from test_gpu_kernels import _attn_ref, _epi_ref, close_bf16  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
t0, n = time.time(), {"gemm": 0, "gemv": 0, "fp8_rows": 0, "attn": 0, "prefill_attn": 0}
t_print = t0
while time.time() - t0 < budget:
    if time.time() - t_print > 30:      # a line every half minute: a silent GPU job is taken for hung
        t_print = time.time()
        print(f"... {n} after {t_print - t0:.0f} s", flush=True)
    # ---- GEMM (both tilings, launcher's choice included)
    M, N, K = ri(1, 900), 8 * ri(1, 200), 64 * ri(1, 12)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    epi, b, r = [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, None, res), (ops.EPI_GELU, bias, None)][ri(0, 2)]
    outs = []
    for tile in (0, 128, 256):
        outs.append(ops.gemm(a, w, b, r, epi, force="gemm", tile=tile))
    close_bf16(outs[0], _epi_ref(acc, b, r, epi))
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[0], outs[1]), (M, N, K)
    if M <= 128:
        close_bf16(ops.gemm_splitk(a, w, b, r, epi, ri(1, 9)), _epi_ref(acc, b, r, epi))
    n["gemm"] += 1
    # ---- the phased 256-tile kernel (even number of K-tiles): bit-identical to the others, also when launched repeatedly
    M, N, K = ri(1, 2600), 8 * ri(1, 400), 128 * ri(2, 40)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    epi, b, r = [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, None, res), (ops.EPI_GELU, bias, None)][ri(0, 2)]
    want = ops.gemm(a, w, b, r, epi, force="gemm", tile=256)
    for _ in range(3):
        got = ops.gemm(a, w, b, r, epi, force="gemm", tile=257)
        assert torch.equal(got, want), ("phased", M, N, K, epi, int((got != want).sum()))
    n["gemm_phased"] = n.get("gemm_phased", 0) + 1
    # ---- skinny GEMM / GEMV, 1..32 rows (17..32: two MFMA column blocks)
    M, N, K = ri(1, 32), 16 * ri(1, 300), 32 * ri(1, 64)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    close_bf16(ops.gemm(a, w, None, res, ops.EPI_RESIDUAL, force="gemv"), _epi_ref(acc, None, res, ops.EPI_RESIDUAL))
    n["gemv"] += 1
    # ---- fp8 rows at 4..32 rows (fragment-major image, exact widening) vs fp32 on the dequantised weights
    from open_o3_video_amd.weights import dequantize_rows_fp8, pack_mfma_fragments_fp8, quantize_rows_fp8
    import ctypes as C
    M, N, K = ri(4, 32), 16 * ri(1, 200), 64 * ri(1, 40)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    q8, sc = quantize_rows_fp8(w)
    q8p = pack_mfma_fragments_fp8(q8)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    o8 = torch.empty(M, N, dtype=BF, device=dev)
    P_ = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    _lib.call("o3v_linear_decode_fp8_rows", P_(a), P_(q8p), P_(sc), None, P_(res), P_(o8), M, N, K, K, N, N, ops.EPI_RESIDUAL,
              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    close_bf16(o8, _epi_ref(a.float() @ dequantize_rows_fp8(q8, sc).t(), None, res, ops.EPI_RESIDUAL))
    n["fp8_rows"] += 1
    # ---- decode attention: per-row and group forms
    G, groups, Hkv, rep = ri(2, 8), ri(1, 2), ri(1, 4), ri(1, 7)
    B, Hq, D = G * groups, Hkv * rep, 128
    P, own, pad = ri(1, 700), ri(1, 200), ri(0, 40)
    pad = min(pad, P - 1)
    ctx, Tmax = P + own, P + own + ri(0, 9)
    q = torch.randn(B, Hq, D, generator=g).to(BF)
    k = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    v = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    k[:, :, :P] = k[::G, :, :P].repeat_interleave(G, dim=0)
    v[:, :, :P] = v[::G, :, :P].repeat_interleave(G, dim=0)
    pads = torch.full((B,), pad, dtype=torch.int32)
    std = ops.attn_decode(q.to(dev), k.to(dev), v.to(dev), pads.to(dev), ctx, max(1, min(64, ri(1, 12))), D ** -0.5).cpu()
    for b_ in range(B):
        kk = k[b_, :, pad:ctx].repeat_interleave(rep, dim=0)
        vv = v[b_, :, pad:ctx].repeat_interleave(rep, dim=0)
        close_bf16(std[b_], _attn_ref(q[b_][:, None, :], kk, vv, D ** -0.5)[:, 0], ulps=3, atol=4e-3)
    own_splits = (own + 127) // 128
    for nsp in ((ri(1, 20),) if G * rep > 64 else (ri(1, 20), -ri(1, 20))):
        if abs(nsp) + own_splits > 64:
            continue
        grp = ops.attn_decode_group(q.to(dev), k.to(dev), v.to(dev), pads.to(dev), G, P, ctx, nsp if G * rep <= 64 else -abs(nsp),
                                    D ** -0.5).cpu()
        close_bf16(grp, std, ulps=3, atol=4e-3)
        # the same with the prompts' K/V kept once and the rows' caches holding only their own keys: bit-identical
        kpre, vpre = k[::G, :, :P].contiguous(), v[::G, :, :P].contiguous()
        kown, vown = k[:, :, P:].contiguous(), v[:, :, P:].contiguous()
        grp2 = ops.attn_decode_group_prefix(q.to(dev), kown.to(dev), vown.to(dev), kpre.to(dev), vpre.to(dev), G, pads.to(dev), G, P, ctx,
                                            nsp if G * rep <= 64 else -abs(nsp), D ** -0.5).cpu()
        assert torch.equal(grp2, grp)
    n["attn"] += 1
    # ---- causal GQA prefill attention behind a cached prefix (head_dim 128: the DMA path; 64: the register path)
    D = (128, 128, 64)[ri(0, 2)]
    Hkv, rep, Bp = ri(1, 3), ri(1, 4), ri(1, 2)
    Hq = Hkv * rep
    past, Sn, tile = (0, ri(1, 400), (64, 128)[ri(0, 1)]) if ri(0, 1) else (ri(1, 300), ri(1, 200), 128)
    pads = [min(ri(0, 30), (past if past else Sn) - 1) for _ in range(Bp)]
    Tmax = past + Sn + ri(0, 5)
    q = torch.randn(Bp, Sn, Hq, D, generator=g).to(BF)
    k = torch.randn(Bp, Hkv, Tmax, D, generator=g).to(BF)
    v = torch.randn(Bp, Hkv, Tmax, D, generator=g).to(BF)
    tiles = torch.from_numpy(indexing.prefill_tiles(Bp, Sn, pads, tile, past=past)).to(dev)
    out = torch.zeros(Bp * Sn, Hq * D, dtype=BF, device=dev)
    ops.attn_tiles(q.reshape(Bp * Sn, -1).to(dev), k.to(dev), v.to(dev), tiles, Hq, rep, D, Hq * D, D, Tmax * D, Hkv * Tmax * D, D,
                   Tmax * D, Hkv * Tmax * D, out, Hq * D, D ** -0.5, rows_per_tile=tile)
    out = out.view(Bp, Sn, Hq, D).cpu()
    for b_, pad in enumerate(pads):
        qi = past + torch.arange(Sn)[:, None]
        kj = torch.arange(past + Sn)[None, :]
        mask = (kj <= qi) & (kj >= pad)
        kk = k[b_, :, :past + Sn].repeat_interleave(rep, dim=0)
        vv = v[b_, :, :past + Sn].repeat_interleave(rep, dim=0)
        ref = _attn_ref(q[b_].transpose(0, 1), kk, vv, D ** -0.5, mask[None]).transpose(0, 1)
        lo = 0 if past else pad
        close_bf16(out[b_, lo:], ref[lo:], ulps=3, atol=4e-3)
    n["prefill_attn"] += 1
torch.cuda.synchronize()
print("fuzz ok:", n, f"in {time.time() - t0:.0f} s")
