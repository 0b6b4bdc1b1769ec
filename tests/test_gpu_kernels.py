"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes) on torch-ROCm tensors and
compared with plain fp32 torch references that apply the reference's bf16 rounding points.
Tolerances: bf16 outputs may differ by accumulation order -> 1-2 bf16 ulp (rtol 2^-7) unless stated."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open_o3_video_amd import _lib
    _lib.load()
    return torch.device("cuda")


def rb(x):
    """value after a bf16 materialisation"""
    return x.to(BF).float()


def close_bf16(got, ref, ulps=2, atol=1e-3, frac=0.999):
    """Element-wise: |err| <= atol + ulps * bf16_ulp(|ref|) for at least `frac` of the elements (a different
    accumulation order flips a bf16 rounding now and then, and a residual add can cancel), and NO element further
    than 8 bf16 ulps of the tensor's magnitude."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + ulps * (2.0 ** -8) * ref.abs()
    bad = err > tol
    rate = bad.float().mean().item()
    assert rate <= 1.0 - frac, f"{bad.sum().item()} / {bad.numel()} outside tolerance; max err {err.max().item()}"
    cap = 8 * (2.0 ** -8) * max(1.0, ref.abs().max().item())
    assert err.max().item() <= cap, f"max err {err.max().item()} > cap {cap}"


def test_rmsnorm(dev):
    import kernel_ops as ops
    from oracle import model_ref
    g = torch.Generator().manual_seed(0)
    for rows, cols in [(1, 64), (5, 128), (37, 1280), (130, 3584), (3, 5120)]:
        x = (torch.randn(rows, cols, generator=g) * 3).to(BF)
        w = (1 + 0.1 * torch.randn(cols, generator=g)).to(BF)
        out = ops.rmsnorm(x.to(dev), w.to(dev), 1e-6)
        ref = model_ref.rmsnorm(x, w, 1e-6)
        close_bf16(out, ref, ulps=1, atol=0, frac=1.0)


def test_layernorm(dev):
    """o3v_layernorm (Qwen3-VL vision blocks and mergers) vs F.layer_norm on the bf16 inputs (fp32 statistics, one rounding), with a
    padded output stride."""
    import ctypes as C
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(2)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rows, cols, ldo in [(1, 64, 64), (5, 288, 320), (37, 1152, 1152), (130, 4608, 4608), (3, 72, 128)]:
        x = (torch.randn(rows, cols, generator=g) * 3 + 0.5).to(BF)
        w = (1 + 0.1 * torch.randn(cols, generator=g)).to(BF)
        b = (0.2 * torch.randn(cols, generator=g)).to(BF)
        out = torch.full((rows, ldo), 7.0, dtype=BF, device=dev)
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        _lib.call("o3v_layernorm", P(xd), P(wd), P(bd), P(out), rows, cols, cols, ldo, 1e-6, st)
        ref = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-6)
        close_bf16(out[:, :cols], ref, ulps=1, atol=1e-3, frac=0.999)
        assert (out[:, cols:] == 7.0).all()


def test_gemm_gelu_tanh_epilogue(dev):
    """EPI_GELU_TANH of the MFMA GEMM (Qwen3-VL vision fc1) vs F.gelu(approximate="tanh") of the bf16-rounded linear output."""
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(4)
    for M, N, K in [(72, 96, 64), (160, 4352, 1152), (520, 448, 320), (4, 128, 64)]:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
        bias = (0.3 * torch.randn(N, generator=g)).to(BF).to(dev)
        for tile in (128, 256):
            out = ops.gemm(a, w, bias, None, _lib.EPI_GELU_TANH, force="gemm", tile=tile)
            lin = rb(a.float() @ w.float().t() + bias.float())
            close_bf16(out, torch.nn.functional.gelu(lin, approximate="tanh"), ulps=1, atol=1e-3, frac=0.999)


@pytest.mark.parametrize("D,Hq,Hkv,T,tpr", [(128, 8, 2, 37, 37), (128, 32, 8, 3, 1), (32, 4, 2, 21, 7), (64, 4, 4, 5, 5)])
def test_qkv_norm_rope_cache(dev, D, Hq, Hkv, T, tpr):
    """o3v_qkv_norm_rope_cache (Qwen3-VL: RMSNorm on every q / k head, then the rotation, then the cache append) vs the oracle's
    rmsnorm + rotate_half arithmetic in bf16 (TF3:480-484).  The sum of squares is taken in another order than torch's: 1 bf16 ulp
    on a small fraction of entries; v rows and untouched slots exactly."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from oracle import model_ref
    g = torch.Generator().manual_seed(D + T)
    B, Tmax, slot0 = T // tpr, tpr + 9, 4
    HT = Hq + 2 * Hkv
    qkv = (torch.randn(T, HT * D, generator=g) * 1.5).to(BF)
    qn = (1 + 0.2 * torch.randn(D, generator=g)).to(BF)
    kn = (1 + 0.2 * torch.randn(D, generator=g)).to(BF)
    ang = torch.rand(T, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF)
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qo = torch.zeros(T, Hq, D, dtype=BF, device=dev)
    kc = torch.zeros(B, Hkv, Tmax, D, dtype=BF, device=dev)
    vc = torch.zeros_like(kc)
    qkvd, qnd, knd, cosd, sind = qkv.to(dev), qn.to(dev), kn.to(dev), cos.to(dev), sin.to(dev)
    _lib.call("o3v_qkv_norm_rope_cache", P(qkvd), P(qnd), P(knd), 1e-6, P(cosd), P(sind), P(qo), P(kc), P(vc), slot0, T, tpr, Hq, Hkv, D,
              Tmax, tpr, 0, st)
    q, k, v = qkv.view(T, HT, D).split([Hq, Hkv, Hkv], dim=1)
    c4, s4 = cos.view(T, 1, D), sin.view(T, 1, D)
    qr = model_ref.rmsnorm(q, qn, 1e-6)
    kr = model_ref.rmsnorm(k, kn, 1e-6)
    q_ref = (qr * c4) + (model_ref.rotate_half(qr) * s4)
    k_ref = (kr * c4) + (model_ref.rotate_half(kr) * s4)
    close_bf16(qo, q_ref, ulps=1, atol=1e-3, frac=0.995)
    kgot = kc.cpu()[:, :, slot0:slot0 + tpr].transpose(1, 2).reshape(T, Hkv, D)
    close_bf16(kgot, k_ref, ulps=1, atol=1e-3, frac=0.995)
    assert torch.equal(vc.cpu()[:, :, slot0:slot0 + tpr].transpose(1, 2).reshape(T, Hkv, D), v)
    assert (kc[:, :, :slot0] == 0).all() and (kc[:, :, slot0 + tpr:] == 0).all()


def test_add_rows(dev):
    """o3v_add_rows (DeepStack, TF3:839-862): x[rows[i]] = bf16(x[rows[i]] + feat[src[i]]), other rows untouched."""
    import ctypes as C
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(9)
    H, T, n = 1024, 50, 17
    x = torch.randn(T, H, generator=g).to(BF)
    feat = torch.randn(30, H, generator=g).to(BF)
    rows = torch.randperm(T, generator=g)[:n].to(torch.int32)
    src = torch.randint(0, 30, (n,), generator=g).to(torch.int32)
    xd = x.to(dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    rd, sd, fd = rows.to(dev), src.to(dev), feat.to(dev)
    _lib.call("o3v_add_rows", P(xd), P(rd), P(sd), P(fd), n, H, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ref = x.clone()
    ref[rows.long()] = (x[rows.long()].float() + feat[src.long()].float()).to(BF)
    assert torch.equal(xd.cpu(), ref)


def _epi_ref(acc, bias, res, epi):
    import kernel_ops as ops
    v = acc + (bias.float() if bias is not None else 0)
    if epi == ops.EPI_NONE:
        return rb(v)
    if epi == ops.EPI_RESIDUAL:
        return rb(rb(v) + res.float())
    if epi == ops.EPI_GELU:
        return rb(torch.nn.functional.gelu(rb(v)))
    raise AssertionError


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 192, 320), (1, 64, 64), (77, 1280, 1216),
                                    (515, 3456, 1280), (300, 4608, 3584), (130, 128, 1152)])
def test_gemm_epilogues(dev, M, N, K):
    import kernel_ops as ops
    g = torch.Generator().manual_seed(M * 131 + N)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    for epi, b, r in [(ops.EPI_NONE, None, None), (ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, bias, res),
                      (ops.EPI_GELU, bias, None)]:
        out = ops.gemm(a, w, b, r, epi, force="gemm")
        close_bf16(out, _epi_ref(acc, b, r, epi))
    # in-place residual (out aliases res), as the engine uses it
    r2 = res.clone()
    from open_o3_video_amd import _lib
    import ctypes as C
    _lib.call("o3v_gemm_bf16", C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), None, C.c_void_p(r2.data_ptr()),
              C.c_void_p(r2.data_ptr()), M, N, K, K, K, N, N, ops.EPI_RESIDUAL,
              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    close_bf16(r2, _epi_ref(acc, None, res, ops.EPI_RESIDUAL))


@pytest.mark.parametrize("M,N,K,splits", [(64, 3584, 3584, 8), (9, 4608, 3584, 7), (128, 3584, 18944, 9), (40, 200, 448, 3),
                                           (17, 128, 64, 4), (100, 1280, 1280, 1)])
def test_gemm_splitk(dev, M, N, K, splits):
    """Split-K form used for a prompt suffix behind a cached prefix (9..128 rows): same epilogues, same tolerance, and
    bit-identical from run to run (the partials are reduced in split order, no atomics)."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    for epi, b, r in [(ops.EPI_NONE, None, None), (ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, bias, res),
                      (ops.EPI_GELU, bias, None)]:
        out = ops.gemm_splitk(a, w, b, r, epi, splits)
        close_bf16(out, _epi_ref(acc, b, r, epi))
        assert torch.equal(out, ops.gemm_splitk(a, w, b, r, epi, splits))
        close_bf16(out, ops.gemm(a, w, b, r, epi, force="gemm").float(), ulps=1, atol=1e-3, frac=0.999)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (700, 1280, 192), (515, 3456, 1280), (1030, 520, 128), (300, 4608, 3584)])
def test_gemm_256_tile_kernel(dev, M, N, K):
    """The 8-wave 256x256 kernel (forced here; the launcher picks it for the big prefill shapes): every epilogue, ragged M
    and N edges, against the fp32 reference and bit-identical to the 128-tile kernel (same k order per output element)."""
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_gate_up
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    cases = [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, bias, res), (ops.EPI_GELU, None, None)]
    outs = {}
    for tile in (256, 128):
        outs[tile] = [ops.gemm(a, w, b, r, epi, force="gemm", tile=tile) for epi, b, r in cases]
    for o256, o128, (epi, b, r) in zip(outs[256], outs[128], cases):
        close_bf16(o256, _epi_ref(acc, b, r, epi))
        assert torch.equal(o256, o128)
    if N % 32 == 0:
        I = N // 2
        wg, wu = w[:I].contiguous(), w[I:].contiguous()
        packed = pack_gate_up(wg, wu, I)
        sw = [ops.gemm(a, packed, None, None, ops.EPI_SWIGLU, force="gemm", tile=tile) for tile in (256, 128)]
        assert torch.equal(sw[0], sw[1])
        gv = rb(a.float() @ wg.float().t())
        uv = rb(a.float() @ wu.float().t())
        close_bf16(sw[0], rb(torch.nn.functional.silu(gv)) * uv)


def _swiglu_case(dev, M, I, K, ipad, seed):
    import kernel_ops as ops
    from open_o3_video_amd.weights import pack_gate_up
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    wg = (torch.randn(I, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wu = (torch.randn(I, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bg = (0.1 * torch.randn(I, generator=g)).to(BF).to(dev)
    bu = (0.1 * torch.randn(I, generator=g)).to(BF).to(dev)
    gate = rb(a.float() @ wg.float().t() + bg.float())
    up = rb(a.float() @ wu.float().t() + bu.float())
    ref = rb(rb(torch.nn.functional.silu(gate)) * up)
    return a, pack_gate_up(wg, wu, ipad), pack_gate_up(bg, bu, ipad), ref


@pytest.mark.parametrize("M,I,K", [(130, 428, 320), (64, 96, 64), (257, 1152, 896), (5, 256, 128), (1, 1152, 896), (8, 428, 320)])
def test_swiglu_gemm_and_gemv(dev, M, I, K):
    import kernel_ops as ops
    ipad = (I + 63) // 64 * 64
    a, w, b, ref = _swiglu_case(dev, M, I, K, ipad, M + I)
    for force in (["gemm", "gemv"] if M <= 8 else ["gemm"]):
        out = ops.gemm(a, w, b, None, ops.EPI_SWIGLU, force=force)
        assert out.shape == (M, ipad)
        close_bf16(out[:, :I], ref)
        assert (out[:, I:] == 0).all()  # zero weight/bias pad rows -> silu(0)*0 = 0 exactly


@pytest.mark.parametrize("M", [1, 2, 3, 5, 8, 12, 16])
@pytest.mark.parametrize("N,K", [(64, 128), (4608, 3584), (3584, 18944), (1000, 896), (8192, 1280)])
def test_gemv(dev, M, N, K):
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(M * 7 + N)
    if M > 8 and N % 16:
        # 9..16 rows exist only on the matrix-core path, which needs whole 16-row weight blocks: refused, not mis-computed
        with pytest.raises(_lib.O3VError):
            ops.gemm(torch.zeros(M, K, dtype=BF, device=dev), torch.zeros(N, K, dtype=BF, device=dev), force="gemv")
        return
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    for epi, b, r in [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, None, res), (ops.EPI_GELU, bias, None)]:
        out = ops.gemm(a, w, b, r, epi, force="gemv")
        close_bf16(out, _epi_ref(acc, b, r, epi))


@pytest.mark.parametrize("M", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("N,K,epi_name", [(4608, 3584, "none"), (37888, 3584, "swiglu"), (152064, 3584, "none"),
                                          (512, 128, "none"), (2304, 896, "swiglu")])
def test_gemv_fused_rmsnorm(dev, M, N, K, epi_name):
    """RMSNorm fused into the decode projections must equal rmsnorm kernel -> gemv (same rounding points)."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from oracle import model_ref
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 2).to(BF)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    epi = ops.EPI_SWIGLU if epi_name == "swiglu" else ops.EPI_NONE
    xn = model_ref.rmsnorm(x, nw, 1e-6).to(dev)
    ref = ops.gemm(xn, w, bias, None, epi, force="gemv")
    out = torch.empty_like(ref)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    xd, nd = x.to(dev), nw.to(dev)
    _lib.call("o3v_gemv_norm_bf16", P(xd), P(nd), 1e-6, P(w), P(bias), None, P(out), M, N, K, K, K, out.stride(0), 0, epi,
              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    close_bf16(out, ref, ulps=1, atol=1e-3, frac=0.9995)


@pytest.mark.parametrize("M,Hq,Hkv,D,K", [(1, 28, 4, 128, 3584), (3, 7, 1, 128, 896), (8, 4, 2, 32, 128), (16, 28, 4, 128, 3584),
                                           (11, 14, 2, 128, 896)])
def test_gemv_fused_norm_qkv_rope_cache(dev, M, Hq, Hkv, D, K):
    """Fully fused decode q/k/v projection == rmsnorm -> gemv(+bias) -> qkv_rope_cache (bit-identical outputs)."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(M + Hq)
    N, Tmax, Tnew, step, slot = (Hq + 2 * Hkv) * D, 40, 6, 4, 33
    x = (torch.randn(M, K, generator=g) * 2).to(BF).to(dev)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    ang = torch.rand(M, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # unfused chain
    qkv = torch.empty(M, N, dtype=BF, device=dev)
    _lib.call("o3v_gemv_norm_bf16", P(x), P(nw), 1e-6, P(w), P(bias), None, P(qkv), M, N, K, K, K, N, 0, ops.EPI_NONE, st)
    q1 = torch.zeros(M, Hq, D, dtype=BF, device=dev)
    k1 = torch.zeros(M, Hkv, Tmax, D, dtype=BF, device=dev)
    v1 = torch.zeros_like(k1)
    _lib.call("o3v_qkv_rope_cache", P(qkv), P(cos), P(sin), P(q1), P(k1), P(v1), slot, M, 1, Hq, Hkv, D, Tmax, Tnew, step, st)
    # fused
    q2, k2, v2 = torch.zeros_like(q1), torch.zeros_like(k1), torch.zeros_like(v1)
    from open_o3_video_amd.weights import pack_mfma_fragments
    _lib.call("o3v_gemv_norm_qkv_rope", P(x), P(nw), 1e-6, P(w), None, P(bias), M, K, K, P(cos), P(sin), P(q2), P(k2), P(v2),
              slot, Hq, Hkv, D, Tmax, Tnew, step, st)
    # same with the fragment-major weight image (matrix-core path from M = 2)
    wp = pack_mfma_fragments(w)
    q3, k3, v3 = torch.zeros_like(q1), torch.zeros_like(k1), torch.zeros_like(v1)
    _lib.call("o3v_gemv_norm_qkv_rope", P(x), P(nw), 1e-6, P(w), P(wp), P(bias), M, K, K, P(cos), P(sin), P(q3), P(k3), P(v3),
              slot, Hq, Hkv, D, Tmax, Tnew, step, st)
    close_bf16(q3, q1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(k3, k1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(v3, v1.float(), ulps=1, atol=2e-3, frac=0.999)
    # the fused kernel splits K over 2 waves, the unfused gemv may not: allow 1 bf16 ulp on the projection
    close_bf16(q2, q1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(k2, k1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(v2, v1.float(), ulps=1, atol=2e-3, frac=0.999)
    assert (k2[:, :, :slot] == 0).all() and (k2[:, :, slot + 1:] == 0).all()


@pytest.mark.parametrize("M", [1, 2, 3, 4, 8, 12, 16])
@pytest.mark.parametrize("N,K,epi_name,norm", [(4608, 3584, "none", True), (37888, 3584, "swiglu", True), (3584, 18944, "res", False),
                                               (3584, 3584, "res", False), (152064, 3584, "none", True), (512, 128, "gelu", False),
                                               (2304, 896, "swiglu", True)])
def test_linear_decode_packed_weights(dev, M, N, K, epi_name, norm):
    """o3v_linear_decode with the MFMA-fragment-major weight copy == the row-major path (all epilogues, fused norm)."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_mfma_fragments
    g = torch.Generator().manual_seed(M * 3 + N)
    x = (torch.randn(M, K, generator=g) * 2).to(BF).to(dev)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wp = pack_mfma_fragments(w)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    epi = {"none": ops.EPI_NONE, "swiglu": ops.EPI_SWIGLU, "res": ops.EPI_RESIDUAL, "gelu": ops.EPI_GELU}[epi_name]
    No = N // 2 if epi == ops.EPI_SWIGLU else N
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for use_p in (False, True):
        out = torch.empty(M, No, dtype=BF, device=dev)
        _lib.call("o3v_linear_decode", P(x), P(nw) if norm else None, 1e-6, P(w), P(wp) if use_p else None, P(bias),
                  P(res) if epi == ops.EPI_RESIDUAL else None, P(out), M, N, K, K, No, N, epi, st)
        outs.append(out)
    # reference: fp32 matmul with the same rounding points
    from oracle import model_ref
    xn = model_ref.rmsnorm(x.cpu(), nw.cpu(), 1e-6).to(dev) if norm else x
    if epi == ops.EPI_SWIGLU:
        ref = ops.gemm(xn, w, bias, None, epi, force="gemv" if M <= 8 else "gemm")
        close_bf16(outs[1], ref.float(), ulps=1, atol=1e-3, frac=0.999)
    else:
        acc = xn.float() @ w.float().t()
        close_bf16(outs[1], _epi_ref(acc, bias, res if epi == ops.EPI_RESIDUAL else None, epi))
    close_bf16(outs[1], outs[0].float(), ulps=1, atol=1e-3, frac=0.999)


@pytest.mark.parametrize("M", [17, 24, 32])
@pytest.mark.parametrize("N,K,epi_name", [(4608, 3584, "none"), (37888, 3584, "swiglu"), (3584, 18944, "res"), (3584, 3584, "res"),
                                          (512, 128, "gelu"), (2304, 896, "swiglu"), (8192, 1024, "none")])
def test_linear_decode_two_column_blocks(dev, M, N, K, epi_name):
    """17..32 decode rows: two 16-row column blocks of x per streamed weight fragment (gemv_mfma_kernel<CB = 2>, x already
    normalised) -- against the fp32 reference, and rows 0..15 against the one-block kernel on the same rows."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_mfma_fragments
    g = torch.Generator().manual_seed(M * 5 + N)
    x = (torch.randn(M, K, generator=g) * 2).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wp = pack_mfma_fragments(w)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    epi = {"none": ops.EPI_NONE, "swiglu": ops.EPI_SWIGLU, "res": ops.EPI_RESIDUAL, "gelu": ops.EPI_GELU}[epi_name]
    No = N // 2 if epi == ops.EPI_SWIGLU else N
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty(M, No, dtype=BF, device=dev)
    _lib.call("o3v_linear_decode", P(x), None, 0.0, P(w), P(wp), P(bias), P(res) if epi == ops.EPI_RESIDUAL else None, P(out), M, N, K, K,
              No, N, epi, st)
    if epi == ops.EPI_SWIGLU:
        ref = ops.gemm(x, w, bias, None, epi, force="gemm")
        close_bf16(out, ref.float(), ulps=1, atol=1e-3, frac=0.999)
    else:
        close_bf16(out, _epi_ref(x.float() @ w.float().t(), bias, res if epi == ops.EPI_RESIDUAL else None, epi))
    out16 = torch.empty(16, No, dtype=BF, device=dev)
    _lib.call("o3v_linear_decode", P(x), None, 0.0, P(w), P(wp), P(bias), P(res) if epi == ops.EPI_RESIDUAL else None, P(out16), 16, N, K,
              K, No, N, epi, st)
    close_bf16(out[:16], out16.float(), ulps=1, atol=1e-3, frac=0.999)
    # a fused norm is not offered above 16 rows (the normalised rows would not fit in LDS): the engine runs the norm apart
    nw = torch.ones(K, dtype=BF, device=dev)
    assert _lib.load().o3v_linear_decode(P(x), P(nw), 1e-6, P(w), P(wp), P(bias), None, P(out), M, N, K, K, No, N, ops.EPI_NONE, st) == _lib.ERR_SHAPE


@pytest.mark.parametrize("M,Hq,Hkv,D,K", [(32, 28, 4, 128, 3584), (19, 8, 2, 128, 1024), (24, 4, 2, 32, 128)])
def test_qkv_rope_cache_two_column_blocks(dev, M, Hq, Hkv, D, K):
    """Decode q/k/v (+bias, M-RoPE, cache append) at 17..32 rows on already normalised x == GEMM -> o3v_qkv_rope_cache."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_mfma_fragments
    g = torch.Generator().manual_seed(M + Hq)
    N, Tmax, Tnew, step, slot = (Hq + 2 * Hkv) * D, 40, 6, 4, 33
    x = (torch.randn(M, K, generator=g)).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wp = pack_mfma_fragments(w)
    bias = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    ang = torch.rand(M, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qkv = ops.gemm(x, w, bias, None, ops.EPI_NONE, force="gemm")
    q1 = torch.zeros(M, Hq, D, dtype=BF, device=dev)
    k1 = torch.zeros(M, Hkv, Tmax, D, dtype=BF, device=dev)
    v1 = torch.zeros_like(k1)
    _lib.call("o3v_qkv_rope_cache", P(qkv), P(cos), P(sin), P(q1), P(k1), P(v1), slot, M, 1, Hq, Hkv, D, Tmax, Tnew, step, st)
    q2, k2, v2 = torch.zeros_like(q1), torch.zeros_like(k1), torch.zeros_like(v1)
    _lib.call("o3v_gemv_norm_qkv_rope", P(x), None, 0.0, P(w), P(wp), P(bias), M, K, K, P(cos), P(sin), P(q2), P(k2), P(v2), slot, Hq,
              Hkv, D, Tmax, Tnew, step, st)
    close_bf16(q2, q1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(k2, k1.float(), ulps=1, atol=2e-3, frac=0.999)
    close_bf16(v2, v1.float(), ulps=1, atol=2e-3, frac=0.999)
    assert (k2[:, :, :slot] == 0).all() and (k2[:, :, slot + 1:] == 0).all()


@pytest.mark.parametrize("M", [4, 8, 16, 21, 32])
@pytest.mark.parametrize("N,K,epi_name", [(37888, 3584, "swiglu"), (3584, 18944, "res"), (6144, 4096, "none"), (4096, 4096, "res"),
                                          (24576, 4096, "swiglu"), (8192, 1024, "none"), (128, 64, "none")])
def test_linear_decode_fp8_rows(dev, M, N, K, epi_name):
    """o3v_linear_decode_fp8_rows (4..32 rows of x against fp8 rows + per-row scales, fragment-major image, widened exactly to
    bf16 and multiplied on the matrix cores) against the bf16 kernels run on the DEQUANTISED weights (exact in bf16: power-of-two
    scales) and the fp32 reference: same values, another summation order."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import dequantize_rows_fp8, pack_mfma_fragments_fp8, quantize_rows_fp8
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = (torch.randn(M, K, generator=g) * 2).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    q8, sc = quantize_rows_fp8(w)
    wq = dequantize_rows_fp8(q8, sc).to(BF)
    assert torch.equal(wq.float(), dequantize_rows_fp8(q8, sc))       # exact in bf16
    q8p = pack_mfma_fragments_fp8(q8)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    epi = {"none": ops.EPI_NONE, "swiglu": ops.EPI_SWIGLU, "res": ops.EPI_RESIDUAL}[epi_name]
    No = N // 2 if epi == ops.EPI_SWIGLU else N
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty(M, No, dtype=BF, device=dev)
    _lib.call("o3v_linear_decode_fp8_rows", P(x), P(q8p), P(sc), P(bias), P(res) if epi == ops.EPI_RESIDUAL else None, P(out), M, N, K, K,
              No, N, epi, st)
    if epi == ops.EPI_SWIGLU:
        ref = ops.gemm(x, wq, bias, None, epi, force="gemm")
        close_bf16(out, ref.float(), ulps=1, atol=1e-3, frac=0.998)
    else:
        close_bf16(out, _epi_ref(x.float() @ wq.float().t(), bias, res if epi == ops.EPI_RESIDUAL else None, epi))
    assert _lib.load().o3v_linear_decode_fp8_rows(P(x), P(q8p), P(sc), None, None, P(out), 3, N, K, K, No, N, ops.EPI_NONE, st) == _lib.ERR_ARG


@pytest.mark.parametrize("M,Hq,Hkv,D,K", [(4, 28, 4, 128, 3584), (16, 28, 4, 128, 3584), (32, 16, 2, 128, 2048), (9, 4, 2, 32, 128), (24, 8, 2, 64, 256)])
def test_qkv_rope_fp8_rows(dev, M, Hq, Hkv, D, K):
    """q/k/v on fp8 rows at 4..32 rows with bias, M-RoPE and the cache append in the epilogue == the bf16 matrix-core kernel on the
    dequantised weights (exact in bf16) up to the summation order."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import dequantize_rows_fp8, pack_mfma_fragments, pack_mfma_fragments_fp8, quantize_rows_fp8
    g = torch.Generator().manual_seed(M + Hq + K)
    N, Tmax, Tnew, step, slot = (Hq + 2 * Hkv) * D, 40, 6, 4, 33
    x = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    q8, sc = quantize_rows_fp8(w)
    wq = dequantize_rows_fp8(q8, sc).to(BF)
    q8p = pack_mfma_fragments_fp8(q8)
    bias = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    ang = torch.rand(M, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for fp8 in (False, True):
        q = torch.zeros(M, Hq, D, dtype=BF, device=dev)
        kc = torch.zeros(M, Hkv, Tmax, D, dtype=BF, device=dev)
        vc = torch.zeros_like(kc)
        if fp8:
            _lib.call("o3v_qkv_rope_fp8_rows", P(x), P(q8p), P(sc), P(bias), M, K, K, P(cos), P(sin), P(q), P(kc), P(vc), slot, Hq, Hkv, D,
                      Tmax, Tnew, step, st)
        else:
            wqp = pack_mfma_fragments(wq) if K % 32 == 0 else None
            _lib.call("o3v_gemv_norm_qkv_rope", P(x), None, 0.0, P(wq), P(wqp), P(bias), M, K, K, P(cos), P(sin), P(q), P(kc), P(vc), slot,
                      Hq, Hkv, D, Tmax, Tnew, step, st)
        outs.append((q, kc, vc))
    for t1, t2 in zip(*outs):
        close_bf16(t2, t1.float(), ulps=1, atol=2e-3, frac=0.998)
    assert (outs[1][1][:, :, :slot] == 0).all() and (outs[1][1][:, :, slot + 1:] == 0).all()


def test_gemm_rejects_bad_shapes(dev):
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    a = torch.zeros(4, 100, dtype=BF, device=dev)
    w = torch.zeros(8, 100, dtype=BF, device=dev)
    with pytest.raises(_lib.O3VError):
        ops.gemm(a, w, force="gemm")  # K % 64 != 0
    with pytest.raises(_lib.O3VError):
        ops.gemm(torch.zeros(9, 128, dtype=BF, device=dev), torch.zeros(8, 128, dtype=BF, device=dev), force="gemv")


def test_vit_rope(dev):
    import kernel_ops as ops
    from oracle.model_ref import rotate_half
    g = torch.Generator().manual_seed(3)
    for P, H, D in [(48, 2, 32), (100, 4, 80), (33, 3, 64)]:
        qkv = torch.randn(P, 3, H, D, generator=g).to(BF)
        ang = torch.rand(P, D // 2, generator=g) * 20
        cos, sin = ang.cos(), ang.sin()
        c = torch.cat([cos, cos], -1)[:, None, :]
        s = torch.cat([sin, sin], -1)[:, None, :]
        ref = qkv.clone()
        for i in (0, 1):
            x = qkv[:, i].float()
            ref[:, i] = ((x * c) + (rotate_half(x) * s)).to(BF)
        out = ops.vit_rope_(qkv.reshape(P, -1).clone().to(dev), cos.contiguous().to(dev), sin.contiguous().to(dev), H, D)
        assert torch.equal(out.cpu().view(P, 3, H, D), ref)  # same fp32 op order -> bit exact


def test_mrope_table_and_qkv_rope_cache(dev):
    import ctypes as C
    from open_o3_video_amd import _lib, indexing
    from oracle import model_ref
    cfg = {"text_config": {"hidden_size": 896, "num_attention_heads": 7, "rope_theta": 1e6, "mrope_section": [16, 24, 24]}}
    D, Hq, Hkv, B, S, Tmax = 128, 7, 1, 2, 37, 50
    g = torch.Generator().manual_seed(5)
    pos = torch.randint(0, 6000, (3, B, S), generator=g)
    cos_ref, sin_ref = model_ref.mrope_cos_sin(cfg, pos, BF)          # [B,S,D]
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).to(dev)
    axis = torch.from_numpy(indexing.mrope_axis_table([16, 24, 24])).to(dev)
    p32 = pos.reshape(3, B * S).to(torch.int32).contiguous().to(dev)
    cos = torch.empty(B * S, D, dtype=BF, device=dev)
    sin = torch.empty_like(cos)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.call("o3v_mrope_table", P(p32), P(inv), P(axis), P(cos), P(sin), B * S, D, st)
    # device cosf/sinf vs torch CPU: allow 1 bf16 ulp on a tiny fraction of entries
    close_bf16(cos.view(B, S, D), cos_ref, ulps=1, atol=2e-3)
    close_bf16(sin.view(B, S, D), sin_ref, ulps=1, atol=2e-3)
    # rope + cache write, using the device table so the comparison is exact
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g).to(BF)
    q, k, v = qkv.view(B, S, Hq + 2 * Hkv, D).split([Hq, Hkv, Hkv], dim=2)
    c4, s4 = cos.cpu().view(B, S, 1, D), sin.cpu().view(B, S, 1, D)
    q_ref = (q * c4) + (model_ref.rotate_half(q) * s4)
    k_ref = (k * c4) + (model_ref.rotate_half(k) * s4)
    qo = torch.zeros(B * S, Hq, D, dtype=BF, device=dev)
    kc = torch.zeros(B, Hkv, Tmax, D, dtype=BF, device=dev)
    vc = torch.zeros_like(kc)
    qd = qkv.to(dev)
    _lib.call("o3v_qkv_rope_cache", P(qd), P(cos), P(sin), P(qo), P(kc), P(vc), 0, B * S, S, Hq, Hkv, D, Tmax, S, 0, st)
    assert torch.equal(qo.cpu().view(B, S, Hq, D), q_ref)
    assert torch.equal(kc.cpu()[:, :, :S].transpose(1, 2), k_ref)
    assert torch.equal(vc.cpu()[:, :, :S].transpose(1, 2), v)
    assert (kc[:, :, S:] == 0).all()
    # decode form: one token per row at slot S+3, table row offset 3 of a [B,Tnew,D] table
    Tnew = 5
    cosd = cos.view(B, S, D)[:, :Tnew].contiguous()
    sind = sin.view(B, S, D)[:, :Tnew].contiguous()
    one = qkv.view(B, S, -1)[:, 3].contiguous().to(dev)
    q1 = torch.zeros(B, Hq, D, dtype=BF, device=dev)
    _lib.call("o3v_qkv_rope_cache", P(one), P(cosd), P(sind), P(q1), P(kc), P(vc), S + 3, B, 1, Hq, Hkv, D, Tmax, Tnew, 3, st)
    assert torch.equal(q1.cpu(), q_ref[:, 3])
    assert torch.equal(kc.cpu()[:, :, S + 3], k_ref[:, 3])


def _attn_ref(q, k, v, scale, mask=None):
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return p @ v.float()


@pytest.mark.parametrize("H,D,lens", [(4, 80, [12, 16, 48, 64, 100, 480, 1]), (2, 32, [24, 64, 65, 7]), (3, 64, [130, 64]),
                                      (2, 128, [200, 33])])
def test_attn_varlen_noncausal(dev, H, D, lens):
    """ViT attention over ragged segments, q/k/v read in place from the fused qkv buffer."""
    import kernel_ops as ops
    from open_o3_video_amd import indexing
    g = torch.Generator().manual_seed(sum(lens) + D)
    P = sum(lens)
    qkv = torch.randn(P, 3, H, D, generator=g).to(BF)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tiles = torch.from_numpy(indexing.segment_tiles(cu)).to(dev)
    qd = qkv.reshape(P, -1).contiguous().to(dev)
    out = torch.zeros(P, H * D, dtype=BF, device=dev)
    hid = H * D
    ops.attn_tiles(qd, qd[:, hid:], qd[:, 2 * hid:], tiles, H, 1, D, 3 * hid, 3 * hid, D, 0, 3 * hid, D, 0, out, hid, D ** -0.5)
    ref = torch.zeros(P, H, D)
    for a, e in zip(cu[:-1], cu[1:]):
        q, k, v = (qkv[a:e, i].transpose(0, 1) for i in range(3))
        ref[a:e] = _attn_ref(q, k, v, D ** -0.5).transpose(0, 1)
    close_bf16(out.view(P, H, D), ref, ulps=3, atol=4e-3)


@pytest.mark.parametrize("Hq,Hkv,D,S,pads", [(7, 1, 128, 150, [0, 13]), (4, 2, 32, 70, [5]), (16, 2, 128, 333, [0]),
                                             (4, 4, 64, 129, [64, 0, 127])])
@pytest.mark.parametrize("tile", [64, 128])
def test_attn_causal_gqa_prefill(dev, Hq, Hkv, D, S, pads, tile):
    import kernel_ops as ops
    from open_o3_video_amd import indexing
    B, Tmax = len(pads), S + 9
    g = torch.Generator().manual_seed(S + D)
    q = torch.randn(B, S, Hq, D, generator=g).to(BF)
    k = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    v = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    tiles = torch.from_numpy(indexing.prefill_tiles(B, S, pads, tile)).to(dev)
    out = torch.zeros(B * S, Hq * D, dtype=BF, device=dev)
    ops.attn_tiles(q.reshape(B * S, -1).to(dev), k.to(dev), v.to(dev), tiles, Hq, Hq // Hkv, D, Hq * D, D, Tmax * D,
                   Hkv * Tmax * D, D, Tmax * D, Hkv * Tmax * D, out, Hq * D, D ** -0.5, rows_per_tile=tile)
    out = out.view(B, S, Hq, D).cpu()
    rep = Hq // Hkv
    for b, pad in enumerate(pads):
        qi = torch.arange(S)[:, None]
        kj = torch.arange(S)[None, :]
        mask = (kj <= qi) & (kj >= pad)
        kk = k[b, :, :S].repeat_interleave(rep, dim=0)
        vv = v[b, :, :S].repeat_interleave(rep, dim=0)
        ref = _attn_ref(q[b].transpose(0, 1), kk, vv, D ** -0.5, mask[None]).transpose(0, 1)  # [S,Hq,D]
        close_bf16(out[b, pad:], ref[pad:], ulps=3, atol=4e-3)


def test_attn_online_softmax_rescale_branch(dev):
    """Force the running max to jump late (spiked key in the last tile) -- the rescale path must be exact."""
    import kernel_ops as ops
    from open_o3_video_amd import indexing
    H, D, L = 1, 128, 256
    g = torch.Generator().manual_seed(11)
    q = torch.randn(L, 1, D, generator=g).to(BF)
    k = (0.05 * torch.randn(1, L, D, generator=g)).to(BF)
    v = torch.randn(1, L, D, generator=g).to(BF)
    k[0, L - 3] = (q[7, 0].float() * 3).to(BF)  # huge score for query 7 in the last kv tile
    tiles = torch.from_numpy(indexing.segment_tiles(np.asarray([0, L], dtype=np.int32))).to(dev)
    out = torch.zeros(L, D, dtype=BF, device=dev)
    ops.attn_tiles(q.reshape(L, D).to(dev), k.to(dev), v.to(dev), tiles, 1, 1, D, D, D, L * D, 0, D, L * D, 0, out, D, D ** -0.5)
    ref = _attn_ref(q.transpose(0, 1), k, v, D ** -0.5)[0]
    close_bf16(out, ref, ulps=3, atol=4e-3)


@pytest.mark.parametrize("B,Hq,Hkv,D,ctx,pads,nsplit", [(1, 28, 4, 128, 4700, [0], 64), (2, 14, 2, 128, 333, [0, 40], 5),
                                                         (3, 4, 2, 32, 17, [0, 3, 16], 1), (2, 16, 2, 128, 64, [0, 0], 3),
                                                         (8, 4, 4, 64, 200, [0] * 8, 4),
                                                         # negative nsplit: scalar kernel for head_dim 128 (MFMA kernel otherwise)
                                                         (1, 28, 4, 128, 4700, [0], -36), (2, 14, 2, 128, 333, [0, 40], -5),
                                                         (1, 28, 4, 128, 5001, [17], 37), (4, 7, 1, 128, 1, [0] * 4, 2),
                                                         (2, 8, 1, 128, 130, [129, 0], 1)])
def test_attn_decode(dev, B, Hq, Hkv, D, ctx, pads, nsplit):
    import kernel_ops as ops
    Tmax = ctx + 11
    g = torch.Generator().manual_seed(ctx)
    q = torch.randn(B, Hq, D, generator=g).to(BF)
    k = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    v = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    out = ops.attn_decode(q.to(dev), k.to(dev), v.to(dev), torch.tensor(pads, dtype=torch.int32, device=dev), ctx, nsplit,
                          D ** -0.5).cpu()
    rep = Hq // Hkv
    for b in range(B):
        kk = k[b, :, pads[b]:ctx].repeat_interleave(rep, dim=0)
        vv = v[b, :, pads[b]:ctx].repeat_interleave(rep, dim=0)
        ref = _attn_ref(q[b][:, None, :], kk, vv, D ** -0.5)[:, 0]
        close_bf16(out[b], ref, ulps=3, atol=4e-3)


@pytest.mark.parametrize("B,G,Hq,Hkv,P,own,pad,nsp", [
    (8, 8, 28, 4, 1000, 1, 0, 8), (8, 8, 28, 4, 333, 77, 5, 3), (8, 4, 28, 4, 515, 130, 0, 5), (6, 2, 28, 4, 97, 300, 40, 1),
    (3, 3, 16, 2, 260, 33, 0, 4), (6, 6, 28, 4, 4490, 511, 0, 36), (4, 4, 8, 2, 64, 1, 63, 2),
    # negative split count: the shared-read form (per-row kernel, prefix read from the group's first row)
    (8, 8, 28, 4, 1000, 1, 0, -8), (8, 4, 28, 4, 515, 130, 7, -5), (6, 2, 28, 4, 97, 300, 40, -3), (6, 6, 28, 4, 4490, 511, 0, -40)])
def test_attn_decode_group(dev, B, G, Hq, Hkv, P, own, pad, nsp):
    """Shared-prefix group decode attention: rows of a group read the prefix K/V of the group's FIRST row only (the other
    rows' prefix slots are poisoned here), their own keys from their own row; checked against fp32 softmax attention."""
    import kernel_ops as ops
    D = 128
    ctx = P + own
    Tmax = ctx + 5
    g = torch.Generator().manual_seed(P + own)
    q = torch.randn(B, Hq, D, generator=g).to(BF)
    kp = torch.randn(B // G, Hkv, P, D, generator=g).to(BF)
    vp = torch.randn(B // G, Hkv, P, D, generator=g).to(BF)
    k = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    v = torch.randn(B, Hkv, Tmax, D, generator=g).to(BF)
    k_true, v_true = k.clone(), v.clone()
    k_true[:, :, :P] = kp.repeat_interleave(G, dim=0)
    v_true[:, :, :P] = vp.repeat_interleave(G, dim=0)
    k[:, :, :P] = 100.0                                     # poison: must never be read for non-leader rows
    v[:, :, :P] = -100.0
    k[::G, :, :P] = kp
    v[::G, :, :P] = vp
    pads = torch.full((B,), pad, dtype=torch.int32)
    out = ops.attn_decode_group(q.to(dev), k.to(dev), v.to(dev), pads.to(dev), G, P, ctx, nsp, D ** -0.5).cpu()
    rep = Hq // Hkv
    for b in range(B):
        kk = k_true[b, :, pad:ctx].repeat_interleave(rep, dim=0)
        vv = v_true[b, :, pad:ctx].repeat_interleave(rep, dim=0)
        ref = _attn_ref(q[b][:, None, :], kk, vv, D ** -0.5)[:, 0]
        close_bf16(out[b], ref, ulps=3, atol=4e-3)
    # the ungrouped kernel on the fanned-out cache gives the same answer to within the split-order rounding
    std = ops.attn_decode(q.to(dev), k_true.to(dev), v_true.to(dev), pads.to(dev), ctx, max(1, min(16, ctx // 128)), D ** -0.5).cpu()
    close_bf16(out, std, ulps=3, atol=4e-3)
    # shared prompt entries: the prompts' K/V kept ONCE ([B/G, Hkv, P + 3, D]), the rows' caches hold only their own keys from
    # slot 0 -- bit-identical to the per-row-copy layout (same kernels, other addresses)
    cap = P + 3
    kpre = torch.full((B // G, Hkv, cap, D), 50.0).to(BF)
    vpre = torch.full((B // G, Hkv, cap, D), -50.0).to(BF)
    kpre[:, :, :P], vpre[:, :, :P] = kp, vp
    k_own = k_true[:, :, P:].contiguous()
    v_own = v_true[:, :, P:].contiguous()
    out2 = ops.attn_decode_group_prefix(q.to(dev), k_own.to(dev), v_own.to(dev), kpre.to(dev), vpre.to(dev), G, pads.to(dev), G, P, ctx,
                                        nsp, D ** -0.5).cpu()
    assert torch.equal(out2, out)
    if G % 2 == 0 and nsp > 0:      # sub-groups of G/2 rows inside a prompt's G rows
        sub = ops.attn_decode_group_prefix(q.to(dev), k_own.to(dev), v_own.to(dev), kpre.to(dev), vpre.to(dev), G, pads.to(dev), G // 2, P,
                                           ctx, nsp, D ** -0.5).cpu() if G // 2 > 1 else out2
        close_bf16(sub, out, ulps=3, atol=4e-3)


@pytest.mark.parametrize("D,Hq,Hkv,G,P,L,pad,rpt", [(128, 8, 2, 3, 200, 70, 0, 128), (128, 4, 4, 2, 129, 300, 5, 128), (32, 4, 2, 4, 77, 40, 0, 128),
                                                    (64, 4, 2, 2, 64, 64, 3, 64)])
def test_attn_tiles_prefix_equals_copied_prompt(dev, D, Hq, Hkv, G, P, L, pad, rpt):
    """o3v_attn_tiles_prefix (the G completions' L tokens behind ONE copy of the prompt's P keys) == o3v_attn_tiles on caches that
    each hold a copy of the prompt, bit for bit: causal tiles with past = P, left padding inside the prompt."""
    import kernel_ops as ops
    from open_o3_video_amd import indexing
    g = torch.Generator().manual_seed(D + P + L)
    q = torch.randn(G * L, Hq, D, generator=g).to(BF).to(dev)
    kp = torch.randn(1, Hkv, P + 2, D, generator=g).to(BF)
    vp = torch.randn(1, Hkv, P + 2, D, generator=g).to(BF)
    ko = torch.randn(G, Hkv, L, D, generator=g).to(BF)
    vo = torch.randn(G, Hkv, L, D, generator=g).to(BF)
    kfull = torch.cat([kp[:, :, :P].expand(G, -1, -1, -1), ko], dim=2).contiguous()
    vfull = torch.cat([vp[:, :, :P].expand(G, -1, -1, -1), vo], dim=2).contiguous()
    tiles = torch.from_numpy(indexing.prefill_tiles(G, L, [pad] * G, tile=rpt, past=P)).to(dev)
    QD = Hq * D
    T1, T2 = P + L, L
    a = torch.zeros(G * L, QD, dtype=BF, device=dev)
    b = torch.zeros_like(a)
    ops.attn_tiles(q, kfull.to(dev), vfull.to(dev), tiles, Hq, Hq // Hkv, D, QD, D, T1 * D, Hkv * T1 * D, D, T1 * D, Hkv * T1 * D, a, QD,
                   D ** -0.5, rows_per_tile=rpt)
    ops.attn_tiles_prefix(q, ko.to(dev), vo.to(dev), kp.to(dev), vp.to(dev), (P + 2) * D, Hkv * (P + 2) * D, P, G, tiles, Hq, Hq // Hkv, D,
                          QD, D, T2 * D, Hkv * T2 * D, D, T2 * D, Hkv * T2 * D, b, QD, D ** -0.5, rows_per_tile=rpt)
    assert torch.equal(a, b)
    assert a.float().abs().sum().item() > 0


def test_patchify_matches_hf_processor(dev, golden_dir):
    """uint8 frames -> pixel rows: bit-exact against Qwen2VLImageProcessor output (golden G3), after the bf16 cast."""
    import ctypes as C
    from open_o3_video_amd import _lib
    g = np.load(os.path.join(golden_dir, "g3_patchify.npz"))
    mean = (C.c_float * 3)(*g["mean"].tolist())   # host arrays
    std = (C.c_float * 3)(*g["std"].tolist())
    P = lambda t: t if isinstance(t, C.Array) else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tag in "abc":
        fr = torch.from_numpy(g[f"{tag}_frames"])
        T, _, H, W = fr.shape
        ref = torch.from_numpy(g[f"{tag}_pixel_values"]).to(BF)
        for is_u8, src in ((1, fr.to(dev)), (0, fr.float().to(dev))):
            out = torch.full((ref.shape[0], 1216), 7.0, dtype=BF, device=dev)
            _lib.call("o3v_patchify", P(src), is_u8, P(out), T, H, W, 1216, P(mean), P(std), st)
            assert torch.equal(out[:, :1176].cpu(), ref)
            assert (out[:, 1176:] == 0).all()
        # processor-output path: f32 pixel_values -> bf16 padded
        pv = torch.from_numpy(g[f"{tag}_pixel_values"]).to(dev)
        out = torch.empty((pv.shape[0], 1216), dtype=BF, device=dev)
        _lib.call("o3v_cast_pad_f32_bf16", P(pv), P(out), pv.shape[0], 1176, 1216, st)
        assert torch.equal(out[:, :1176].cpu(), ref) and (out[:, 1176:] == 0).all()


def test_gather_and_embed(dev):
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(2)
    src = torch.randn(50, 256, generator=g).to(BF).to(dev)
    idx = torch.randperm(50, generator=g).to(torch.int32).to(dev)
    assert torch.equal(ops.gather_rows(src, idx), src[idx.long()])
    table = torch.randn(300, 128, generator=g).to(BF).to(dev)
    vis = torch.randn(6, 128, generator=g).to(BF).to(dev)
    rows = torch.tensor([5, -1, -2, 299, 0, -3, -4, -5, -6, 17], dtype=torch.int32, device=dev)
    out = torch.empty(10, 128, dtype=BF, device=dev)
    _lib.call("o3v_embed_scatter", C.c_void_p(table.data_ptr()), C.c_void_p(vis.data_ptr()), C.c_void_p(rows.data_ptr()),
              C.c_void_p(out.data_ptr()), 10, 128, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ref = torch.stack([table[r] if r >= 0 else vis[-r - 1] for r in rows.tolist()])
    assert torch.equal(out, ref)


def test_sample_greedy_and_logprob(dev):
    import ctypes as C
    from open_o3_video_amd import _lib
    from oracle import model_ref
    g = torch.Generator().manual_seed(9)
    B, V, T = 3, 152064, 4
    logits = (torch.randn(B, V, generator=g) * 4).to(BF)
    ids = torch.randint(0, V, (B, 50), generator=g)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld = logits.to(dev)
    for pen in (1.0, 1.05):
        seen = torch.zeros(B, V, dtype=torch.uint8, device=dev)
        idd = ids.to(torch.int32).to(dev)
        _lib.call("o3v_mark_seen", P(idd), P(seen), B, 50, V, st)
        cur = torch.zeros(B, dtype=torch.int32, device=dev)
        fin = torch.tensor([0, 1, 0], dtype=torch.int32, device=dev)
        out = torch.full((B, T), -1, dtype=torch.int32, device=dev)
        mar = torch.zeros(B, T, device=dev)
        eos = torch.tensor([V + 5], dtype=torch.int32, device=dev)
        scr = torch.empty(B, 256, device=dev)
        _lib.call("o3v_sample_greedy", P(ld), P(seen), P(cur), P(fin), P(out), P(mar), P(eos), 1, 777, B, V, V, pen, 2, T,
                  P(scr), st)
        sc = model_ref.repetition_penalty(logits.float(), ids, pen) if pen != 1.0 else logits.float()
        top2 = sc.topk(2, dim=-1)
        exp = top2.indices[:, 0].clone()
        exp[1] = 777  # finished row -> pad
        assert out[:, 2].cpu().tolist() == exp.tolist()
        np.testing.assert_allclose(mar[:, 2].cpu().numpy(), (top2.values[:, 0] - top2.values[:, 1]).numpy(), rtol=0, atol=1e-6)
        assert seen[0, exp[0]].item() == 1
    # log-prob gather
    tgt = torch.randint(0, V, (B,), generator=g).to(torch.int32)
    out = torch.empty(B, device=dev)
    _lib.call("o3v_logprob_gather", P(ld), P(tgt.to(dev)), P(out), B, V, V, st)
    ref = torch.log_softmax(logits.float(), dim=-1).gather(1, tgt.long()[:, None])[:, 0]
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)


def test_sample_top_p_support_and_frequencies(dev):
    """Sampled ids must lie inside the reference top-p set (TF TopPLogitsWarper), and frequencies must follow the
    renormalised distribution (chi-square-ish bound on a small vocabulary)."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from oracle import model_ref
    g = torch.Generator().manual_seed(4)
    V, B, N = 64, 8, 400
    logits = (torch.randn(1, V, generator=g) * 2).to(BF).repeat(B, 1)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld = logits.to(dev)
    for top_p, temp in [(0.9, 1.0), (0.5, 0.7), (1.0, 1.0)]:
        sc = model_ref.temperature_warp(logits[:1].float(), temp)
        kept = torch.isfinite(model_ref.top_p_warp(sc, top_p)[0]) if top_p < 1 else torch.ones(V, dtype=torch.bool)
        probs = torch.softmax(sc[0].masked_fill(~kept, float("-inf")), dim=-1)
        counts = torch.zeros(V)
        out = torch.zeros(B, N, dtype=torch.int32, device=dev)
        lp = torch.zeros(B, N, device=dev)
        seen = torch.zeros(B, V, dtype=torch.uint8, device=dev)
        cur = torch.zeros(B, dtype=torch.int32, device=dev)
        fin = torch.zeros(B, dtype=torch.int32, device=dev)
        eos = torch.tensor([V + 1], dtype=torch.int32, device=dev)
        scratch = torch.empty(B, _lib.SAMPLE_SCRATCH_FLOATS, device=dev)
        rid = torch.arange(B, dtype=torch.int32, device=dev)
        for step in range(N):
            _lib.call("o3v_sample_top_p", P(ld), P(seen), P(cur), P(fin), P(out), P(lp), P(eos), 1, 0, B, V, V, 1.0, temp,
                      top_p, 1234, P(rid), step, N, P(scratch), st)
        o = out.cpu().long().view(-1)
        assert kept[o].all(), "sampled a token outside the top-p set"
        counts = torch.bincount(o, minlength=V).float()
        freq = counts / counts.sum()
        assert (freq - probs).abs().max().item() < 0.03
        # reproducible per (seed,row,step); different rows differ
        out2 = torch.zeros_like(out)
        _lib.call("o3v_sample_top_p", P(ld), P(seen), P(cur), P(fin), P(out2), P(lp), P(eos), 1, 0, B, V, V, 1.0, temp,
                  top_p, 1234, P(rid), 5, N, P(scratch), st)
        assert torch.equal(out2[:, 5], out[:, 5])


@pytest.mark.parametrize("top_p,temp,rep", [(0.95, 1.0, 1.0), (0.9, 0.8, 1.1), (0.9999, 1.0, 1.05), (1.0, 1.3, 1.0), (0.05, 1.0, 1.0)])
def test_sample_top_p_full_vocab(dev, top_p, temp, rep):
    """7B vocabulary (152064): every draw lies in the oracle's top-p set (TF TopPLogitsWarper after penalty and
    temperature), the reported log-prob is the softmax log-prob of the processed scores, draws are reproducible, rows with
    different completion ids differ.  top_p=0.9999 puts the threshold deep in the tail; top_p=0.05 keeps only the head."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from oracle import model_ref
    V, B, N = 152064, 4, 48
    g = torch.Generator().manual_seed(11)
    base = torch.randn(V, generator=g) * 2.5
    base[torch.randint(0, V, (12,), generator=g)] += 9.0
    logits = base.to(BF)[None].repeat(B, 1).contiguous()
    seen0 = torch.zeros(B, V, dtype=torch.uint8)
    seen0[:, torch.randint(0, V, (500,), generator=g)] = 1
    sc = logits[:1].float().clone()
    if rep != 1.0:
        m = seen0[:1].bool()
        sc = torch.where(m, torch.where(sc < 0, sc * rep, sc / rep), sc)       # TF:logits_process.py:404-414
    sc = model_ref.temperature_warp(sc, temp)
    kept = torch.isfinite(model_ref.top_p_warp(sc, top_p)[0]) if top_p < 1 else torch.ones(V, dtype=torch.bool)
    lsm = torch.log_softmax(sc[0], dim=-1)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld = logits.to(dev)
    eos = torch.tensor([V + 1], dtype=torch.int32, device=dev)
    rid = torch.tensor([0, 1, 2, 0], dtype=torch.int32, device=dev)           # rows 0 and 3 are the same completion
    scratch = torch.empty(B, _lib.SAMPLE_SCRATCH_FLOATS, device=dev)
    outs = []
    for rep_run in range(2):
        out = torch.zeros(B, N, dtype=torch.int32, device=dev)
        lp = torch.zeros(B, N, device=dev)
        cur = torch.zeros(B, dtype=torch.int32, device=dev)
        fin = torch.zeros(B, dtype=torch.int32, device=dev)
        for step in range(N):
            seen = seen0.to(dev)                                                # same processed scores at every step
            _lib.call("o3v_sample_top_p", P(ld), P(seen), P(cur), P(fin), P(out), P(lp), P(eos), 1, 0, B, V, V, rep, temp,
                      top_p, 77, P(rid), step, N, P(scratch), st)
            assert bool(seen.cpu()[0, out[0, step].item()] == 1)               # the drawn token is marked seen
        outs.append((out.cpu().long(), lp.cpu()))
    o, lp = outs[0]
    assert torch.equal(o, outs[1][0])
    assert torch.equal(o[0], o[3]) and not torch.equal(o[0], o[1])
    assert kept[o.view(-1)].all(), "sampled a token outside the top-p set"
    np.testing.assert_allclose(lp.view(-1).numpy(), lsm[o.view(-1)].numpy(), rtol=0, atol=2e-4)
    if top_p >= 0.9:
        assert len(set(o.view(-1).tolist())) > 3


@pytest.mark.parametrize("T,H,W,h,w", [(4, 360, 640, 224, 420), (2, 360, 640, 364, 644), (3, 50, 70, 112, 84), (1, 28, 28, 28, 28)])
def test_resize_bicubic_antialias(dev, T, H, W, h, w):
    """GPU frame resize against torch's CPU antialiased bicubic (what torchvision's resize runs in fetch_video,
    R:vision_process.py:310-315).  Float frames: fp32 tolerance; uint8 frames: rounded like torchvision's uint8 path, equal
    except where the float result sits within rounding error of a .5 tie."""
    import torch.nn.functional as F
    from open_o3_video_amd import vision_process as vp
    g = torch.Generator().manual_seed(H + w)
    fr = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.uint8)
    ref_f = F.interpolate(fr.float(), size=[h, w], mode="bicubic", antialias=True, align_corners=False)
    out_f = vp.resize_frames_device(fr.float(), (h, w)).cpu()
    assert out_f.shape == ref_f.shape and out_f.dtype == torch.float32
    assert (out_f - ref_f).abs().max().item() < 1e-3
    ref_u = vp.resize_frames(fr, (h, w))                      # CPU form: round + clamp
    out_u = vp.resize_frames_device(fr, (h, w)).cpu()
    d = (out_u - ref_u).abs()
    assert d.max().item() <= 1.0 and (d > 0).float().mean().item() < 1e-4
    assert out_u.min().item() >= 0 and out_u.max().item() <= 255 and torch.equal(out_u, out_u.round())
    if (h, w) == (H, W):
        assert torch.equal(out_u, fr.float())


@pytest.mark.parametrize("Hq,Hkv,H,ctx,Tmax,nsplit,pad", [
    (28, 4, 3584, 4491, 5002, 40, 0),      # 7B, first decode step of the bench prompt
    (28, 4, 3584, 5002, 5002, 40, 37),     # last slot of the cache, left-padded prompt
    (16, 2, 2048, 1500, 1732, 14, 0),      # 3B dims (n_rep 8)
    (32, 8, 4096, 300, 512, 4, 0),         # 8 kv heads (Qwen3-VL-8B text dims), short context: empty waves / splits
    (28, 4, 3584, 97, 20480, 64, 3),       # few keys against 64 splits: most workgroups hold no key at all
    (14, 2, 896, 1, 64, 1, 0),             # one key (the token itself)
])
def test_decode_attn_block_fused_equals_three_launches(dev, Hq, Hkv, H, ctx, Tmax, nsplit, pad):
    """o3v_decode_attn_block (one launch, roles + in-launch hand-offs) == o3v_gemv_norm_qkv_rope + o3v_attn_decode +
    o3v_linear_decode(o_proj, RESIDUAL) BIT FOR BIT: residual stream, q, attention output, the appended K/V row.
    Run three times on the same buffers (L1/L2-warm consumers, epochs 1..3 on one sync buffer) with new inputs each time."""
    import ctypes as C
    from open_o3_video_amd import _lib
    D, Tnew, step = 128, 7, 3
    slot = ctx - 1
    g = torch.Generator().manual_seed(Hq * 1000 + ctx)
    N, QD = (Hq + 2 * Hkv) * D, Hq * D
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
    wqkv = (torch.randn(N, H, generator=g) / math.sqrt(H)).to(BF).to(dev)
    bqkv = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    wo = (torch.randn(H, QD, generator=g) / math.sqrt(QD)).to(BF).to(dev)
    ang = torch.rand(1, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    kc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    vc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    kc0[:, :, slot] = float("nan")   # the slot of the new token holds garbage until the q/k/v role fills it
    vc0[:, :, slot] = float("nan")
    k_lo = torch.tensor([pad], dtype=torch.int32, device=dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)   # zeroed once; epochs 1, 2, 3 below
    part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
    part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
    q2, att2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    k2, v2 = kc0.clone(), vc0.clone()
    for rep in range(3):
        x0 = (torch.randn(1, H, generator=g) * 2).to(BF).to(dev)
        # ---- three stand-alone launches
        x1 = x0.clone()
        q1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        k1, v1 = kc0.clone(), vc0.clone()
        att1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        po1, pm1 = torch.empty_like(part_o), torch.empty_like(part_ml)
        _lib.call("o3v_gemv_norm_qkv_rope", P(x1), P(nw), 1e-6, P(wqkv), None, P(bqkv), 1, H, H, P(cos), P(sin), P(q1), P(k1), P(v1),
                  slot, Hq, Hkv, D, Tmax, Tnew, step, st)
        _lib.call("o3v_attn_decode", P(q1), P(k1), P(v1), P(att1), P(po1), P(pm1), P(k_lo), 1, Hq, Hkv, D, ctx, Tmax, nsplit,
                  scale, st)
        _lib.call("o3v_linear_decode", P(att1), None, 0.0, P(wo), None, None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
        # ---- one launch (same buffers every repetition: consumers find the previous repetition's lines in their caches)
        x2 = x0.clone()
        k2[:, :, slot] = float("nan")
        v2[:, :, slot] = float("nan")
        rc = lib.o3v_decode_attn_block(P(x2), P(nw), 1e-6, P(wqkv), P(bqkv), P(wo), P(cos), P(sin), P(q2), P(att2), P(k2), P(v2),
                                       P(part_o), P(part_ml), P(k_lo), H, Hq, Hkv, D, slot, Tmax, Tnew, step, nsplit, scale,
                                       P(sync), rep + 1, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        tmo = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
        assert tmo == 0, f"an in-launch wait timed out (code {tmo:#x})"
        assert not torch.isnan(x1.float()).any()
        assert torch.equal(q2.view(torch.int16), q1.view(torch.int16))
        assert torch.equal(k2.view(torch.int16), k1.view(torch.int16)) and torch.equal(v2.view(torch.int16), v1.view(torch.int16))
        assert torch.equal(att2.view(torch.int16), att1.view(torch.int16))
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16))



@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("Hq,Hkv,H,ctx,Tmax,nsplit,pad", [
    (32, 8, 4096, 3563, 3820, 20, 0),      # Qwen3-VL-8B text dims, one 32-frame prompt
    (32, 8, 4096, 300, 512, 4, 5),         # short context: the newest key's split is not the last one
    (8, 2, 1024, 61, 80, 1, 0),            # fixture dims (tests/fixture_models_q3.py medium), one split
    (8, 2, 1024, 700, 1024, 9, 0),
    (32, 8, 4096, 1, 64, 1, 0),            # one key (the token itself)
])
def test_decode_attn_block_qknorm_equals_stand_alone_chain(dev, Hq, Hkv, H, ctx, Tmax, nsplit, pad, fp8):
    """o3v_decode_attn_block_qknorm (Qwen3-VL: q/k RMSNorm between projection and rotation, done inside the attention role) ==
    o3v_linear_decode[_fp8](q/k/v) + o3v_qkv_norm_rope_cache + o3v_attn_decode + o3v_linear_decode[_fp8](o_proj, RESIDUAL) BIT FOR
    BIT: residual stream, attention output, the appended K/V row.  Three epochs on one sync buffer."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import quantize_rows_fp8
    D, Tnew, step = 128, 7, 3
    slot = ctx - 1
    g = torch.Generator().manual_seed(Hq * 1000 + ctx)
    N, QD = (Hq + 2 * Hkv) * D, Hq * D
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
    qn = (1 + 0.2 * torch.randn(D, generator=g)).to(BF).to(dev)
    kn = (1 + 0.2 * torch.randn(D, generator=g)).to(BF).to(dev)
    wqkv = (torch.randn(N, H, generator=g) / math.sqrt(H)).to(BF).to(dev)
    wo = (torch.randn(H, QD, generator=g) / math.sqrt(QD)).to(BF).to(dev)
    w8 = s8 = o8 = so8 = None
    if fp8:
        w8, s8 = quantize_rows_fp8(wqkv)
        o8, so8 = quantize_rows_fp8(wo)
    ang = torch.rand(1, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    kc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    vc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    kc0[:, :, slot] = float("nan")
    vc0[:, :, slot] = float("nan")
    k_lo = torch.tensor([pad], dtype=torch.int32, device=dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
    part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
    q2, att2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    raw2 = torch.zeros((Hq + 2 * Hkv) * D, dtype=BF, device=dev)
    k2, v2 = kc0.clone(), vc0.clone()
    for rep in range(3):
        x0 = (torch.randn(1, H, generator=g) * 2).to(BF).to(dev)
        x1 = x0.clone()
        qkv1 = torch.zeros(1, N, dtype=BF, device=dev)
        q1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        k1, v1 = kc0.clone(), vc0.clone()
        att1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        po1, pm1 = torch.empty_like(part_o), torch.empty_like(part_ml)
        if fp8:
            _lib.call("o3v_linear_decode_fp8", P(x1), P(nw), 1e-6, P(w8), P(s8), None, None, P(qkv1), 1, N, H, H, N, 0, _lib.EPI_NONE, st)
        else:
            _lib.call("o3v_linear_decode", P(x1), P(nw), 1e-6, P(wqkv), None, None, None, P(qkv1), 1, N, H, H, N, 0, _lib.EPI_NONE, st)
        _lib.call("o3v_qkv_norm_rope_cache", P(qkv1), P(qn), P(kn), 1e-6, P(cos), P(sin), P(q1), P(k1), P(v1), slot, 1, 1, Hq, Hkv, D, Tmax,
                  Tnew, step, st)
        _lib.call("o3v_attn_decode", P(q1), P(k1), P(v1), P(att1), P(po1), P(pm1), P(k_lo), 1, Hq, Hkv, D, ctx, Tmax, nsplit, scale, st)
        if fp8:
            _lib.call("o3v_linear_decode_fp8", P(att1), None, 0.0, P(o8), P(so8), None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
        else:
            _lib.call("o3v_linear_decode", P(att1), None, 0.0, P(wo), None, None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
        x2 = x0.clone()
        k2[:, :, slot] = float("nan")
        v2[:, :, slot] = float("nan")
        rc = lib.o3v_decode_attn_block_qknorm(P(x2), P(nw), 1e-6, P(w8 if fp8 else wqkv), P(s8), P(o8 if fp8 else wo), P(so8), P(qn), P(kn),
                                              P(raw2), P(cos), P(sin), P(q2), P(att2), P(k2), P(v2), P(part_o), P(part_ml), P(k_lo), H, Hq,
                                              Hkv, D, slot, Tmax, Tnew, step, nsplit, scale, P(sync), rep + 1, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        tmo = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
        assert tmo == 0, f"an in-launch wait timed out (code {tmo:#x})"
        assert not torch.isnan(x1.float()).any()
        assert torch.equal(k2.view(torch.int16), k1.view(torch.int16)) and torch.equal(v2.view(torch.int16), v1.view(torch.int16))
        assert torch.equal(att2.view(torch.int16), att1.view(torch.int16))
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16))


@pytest.mark.parametrize("V,top_k,top_p,temp,rep", [(152064, 50, 0.95, 1.0, 1.0), (152064, 1, 1.0, 1.0, 1.0), (152064, 7, 0.9, 0.8, 1.1),
                                                     (152064, 200000, 0.95, 1.0, 1.0), (997, 20, 0.5, 1.0, 1.0), (997, 3, 1.0, 1.3, 1.0)])
def test_sample_top_k_top_p(dev, V, top_k, top_p, temp, rep):
    """o3v_sample_top_k_top_p: every draw lies in the set TF keeps (penalty -> temperature -> TopKLogitsWarper ->
    TopPLogitsWarper, restated by the oracle and pinned by golden G8b), bf16 ties at the k-th value included (the plateau
    keeps MORE than k tokens, as TF does; a tie group at the top-p boundary is kept whole); top_k = 1 is greedy; k >= vocabulary removes nothing; the frequencies over many draws
    follow the renormalised probabilities; the reported log-prob is the softmax over what top-k kept (before top-p)."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from oracle import model_ref
    B, N = 4, 64 if V > 1000 else 400
    g = torch.Generator().manual_seed(13 + top_k % 97)
    base = torch.randn(V, generator=g) * 2.5
    hot = torch.randint(0, V, (30,), generator=g)
    base[hot] += 8.0
    base[hot[10:18]] = float(base[hot[10]])   # a plateau among the largest scores: ties around the k-th place
    logits = base.to(BF)[None].repeat(B, 1).contiguous()
    seen0 = torch.zeros(B, V, dtype=torch.uint8)
    seen0[:, torch.randint(0, V, (200,), generator=g)] = 1
    sc = logits[:1].float().clone()
    if rep != 1.0:
        m = seen0[:1].bool()
        sc = torch.where(m, torch.where(sc < 0, sc * rep, sc / rep), sc)
    sc = model_ref.temperature_warp(sc, temp)
    filt = model_ref.top_k_warp(sc, top_k)
    if top_p < 1.0:
        filt = model_ref.top_p_warp(filt, top_p)
    kept_tf = torch.isfinite(filt[0])
    # Scores tied with the smallest score TF keeps: TF's ascending sort puts some of them inside the removed prefix and some
    # outside, by the (unspecified) order of equal elements; the kernel keeps the whole tie group.  No tie: identical sets.
    topk_only = model_ref.top_k_warp(sc, top_k)[0]
    kept = topk_only >= filt[0][kept_tf].min()
    assert (kept | ~kept_tf).all() and (topk_only[kept & ~kept_tf] == filt[0][kept_tf].min()).all()
    probs = torch.softmax(topk_only.masked_fill(~kept, float("-inf")), dim=-1)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld = logits.to(dev)
    out = torch.zeros(B, N, dtype=torch.int32, device=dev)
    lp = torch.zeros(B, N, device=dev)
    cur = torch.zeros(B, dtype=torch.int32, device=dev)
    fin = torch.zeros(B, dtype=torch.int32, device=dev)
    eos = torch.tensor([V + 1], dtype=torch.int32, device=dev)
    scratch = torch.empty(B, _lib.SAMPLE_SCRATCH_FLOATS, device=dev)
    rid = torch.arange(B, dtype=torch.int32, device=dev)
    for step in range(N):
        seen = seen0.clone().to(dev)       # the sampler marks what it draws: keep the penalty set fixed for the comparison
        _lib.call("o3v_sample_top_k_top_p", P(ld), P(seen), P(cur), P(fin), P(out), P(lp), P(eos), 1, 0, B, V, V, rep, temp, top_k,
                  top_p, 99, P(rid), step, N, P(scratch), st)
    o = out.cpu().long()
    assert kept[o.view(-1)].all(), f"sampled outside the top-k/top-p set ({int(kept.sum())} kept, TF {int(kept_tf.sum())})"
    if top_k == 1:
        assert (o == int(sc[0].argmax())).all()
    freq = torch.bincount(o.view(-1), minlength=V).float() / o.numel()
    tol = 0.12 if V > 1000 else 0.04
    assert (freq - probs).abs().max().item() < tol
    ref_lp = torch.log_softmax(model_ref.top_k_warp(sc, top_k)[0], dim=-1)[o.view(-1)].view(B, N)   # softmax over what top-k kept
    assert (lp.cpu() - ref_lp).abs().max().item() < 2e-3
    assert len({tuple(r.tolist()) for r in o}) > 1 or int(kept.sum()) == 1


@pytest.mark.parametrize("M", [1, 2, 3])
@pytest.mark.parametrize("N,K,epi_name,norm", [(4608, 3584, "none", True), (37888, 3584, "swiglu", True), (3584, 18944, "res", False),
                                               (3584, 3584, "res", False), (152064, 3584, "none", True), (512, 128, "gelu", False),
                                               (2304, 896, "swiglu", True), (6144, 4096, "none", True)])
def test_linear_decode_fp8_weights(dev, M, N, K, epi_name, norm):
    """o3v_linear_decode_fp8 (OCP e4m3fn weight rows + one fp32 scale per row) against an fp32 matmul over the SAME quantised
    weights (weights.dequantize_rows_fp8: exactly what the kernel multiplies) with the bf16 path's rounding points, and the
    quantiser itself against torch's float8_e4m3fn (bit-exact codes)."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import dequantize_rows_fp8, pack_gate_up, quantize_rows_fp8
    g = torch.Generator().manual_seed(M * 5 + N)
    x = (torch.randn(M, K, generator=g) * 2).to(BF).to(dev)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    w[:, : K // 3] *= 4.0      # uneven magnitudes along k and across rows: the per-row scale has work to do
    w[::7] *= 0.05
    if epi_name == "swiglu":
        w = pack_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous(), N // 2)
    w = w.to(dev)
    q8, sc = quantize_rows_fp8(w)
    assert q8.dtype == torch.uint8 and sc.dtype == torch.float32 and sc.shape == (N,)
    wd = dequantize_rows_fp8(q8, sc)
    assert ((wd - w.float()).abs() <= sc[:, None] * 16.0).all()   # within half an e4m3 step of the top binade (32 scale units)
    assert (torch.log2(sc) == torch.log2(sc).round()).all() and (w.float().abs().amax(dim=1) <= sc * 448).all()
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    epi = {"none": ops.EPI_NONE, "swiglu": ops.EPI_SWIGLU, "res": ops.EPI_RESIDUAL, "gelu": ops.EPI_GELU}[epi_name]
    No = N // 2 if epi == ops.EPI_SWIGLU else N
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty(M, No, dtype=BF, device=dev)
    use_bias = epi != ops.EPI_SWIGLU
    _lib.call("o3v_linear_decode_fp8", P(x), P(nw) if norm else None, 1e-6, P(q8), P(sc), P(bias) if use_bias else None,
              P(res) if epi == ops.EPI_RESIDUAL else None, P(out), M, N, K, K, No, N, epi, st)
    from oracle import model_ref
    xn = model_ref.rmsnorm(x.cpu(), nw.cpu(), 1e-6).to(dev) if norm else x
    acc = xn.float() @ wd.t()
    if epi == ops.EPI_SWIGLU:
        a3 = acc.view(M, N // 32, 2, 16)
        gate, up = rb(a3[:, :, 0].reshape(M, -1)), rb(a3[:, :, 1].reshape(M, -1))
        ref = rb(torch.nn.functional.silu(gate)) * up
    else:
        ref = _epi_ref(acc, bias, res if epi == ops.EPI_RESIDUAL else None, epi)
    close_bf16(out, ref, ulps=1, atol=2e-3, frac=0.999)


@pytest.mark.parametrize("Hq,Hkv,H,ctx,Tmax,nsplit,pad", [(28, 4, 3584, 4491, 5002, 40, 0), (16, 2, 2048, 1500, 1732, 14, 5),
                                                           (32, 8, 4096, 300, 512, 4, 0), (14, 2, 896, 33, 64, 1, 0)])
def test_decode_attn_block_fp8_equals_three_launches(dev, Hq, Hkv, H, ctx, Tmax, nsplit, pad):
    """o3v_decode_attn_block_fp8 == o3v_gemv_norm_qkv_rope_fp8 + o3v_attn_decode + o3v_linear_decode_fp8 BIT FOR BIT (the roles
    instantiate the same device code with one-byte weights), two epochs on one sync buffer."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import quantize_rows_fp8
    D, Tnew, step = 128, 7, 3
    slot = ctx - 1
    g = torch.Generator().manual_seed(Hq * 1000 + ctx + 1)
    N, QD = (Hq + 2 * Hkv) * D, Hq * D
    nw = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
    q8, qs = quantize_rows_fp8((torch.randn(N, H, generator=g) / math.sqrt(H)).to(BF).to(dev))
    o8, os_ = quantize_rows_fp8((torch.randn(H, QD, generator=g) / math.sqrt(QD)).to(BF).to(dev))
    bqkv = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    ang = torch.rand(1, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    kc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    vc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    k_lo = torch.tensor([pad], dtype=torch.int32, device=dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
    part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
    q2, att2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    k2, v2 = kc0.clone(), vc0.clone()
    for rep in range(2):
        x0 = (torch.randn(1, H, generator=g) * 2).to(BF).to(dev)
        x1, q1, k1, v1 = x0.clone(), torch.zeros(1, Hq, D, dtype=BF, device=dev), kc0.clone(), vc0.clone()
        att1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        po1, pm1 = torch.empty_like(part_o), torch.empty_like(part_ml)
        _lib.call("o3v_gemv_norm_qkv_rope_fp8", P(x1), P(nw), 1e-6, P(q8), P(qs), P(bqkv), 1, H, H, P(cos), P(sin), P(q1), P(k1), P(v1),
                  slot, Hq, Hkv, D, Tmax, Tnew, step, st)
        _lib.call("o3v_attn_decode", P(q1), P(k1), P(v1), P(att1), P(po1), P(pm1), P(k_lo), 1, Hq, Hkv, D, ctx, Tmax, nsplit, scale, st)
        _lib.call("o3v_linear_decode_fp8", P(att1), None, 0.0, P(o8), P(os_), None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
        x2 = x0.clone()
        rc = lib.o3v_decode_attn_block_fp8(P(x2), P(nw), 1e-6, P(q8), P(qs), P(bqkv), P(o8), P(os_), P(cos), P(sin), P(q2), P(att2),
                                           P(k2), P(v2), P(part_o), P(part_ml), P(k_lo), H, Hq, Hkv, D, slot, Tmax, Tnew, step, nsplit,
                                           scale, P(sync), rep + 1, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item()) == 0
        assert torch.equal(q2.view(torch.int16), q1.view(torch.int16)) and torch.equal(k2.view(torch.int16), k1.view(torch.int16))
        assert torch.equal(att2.view(torch.int16), att1.view(torch.int16)) and torch.equal(x2.view(torch.int16), x1.view(torch.int16))


@pytest.mark.parametrize("Hq,Hkv,H,I,ctx,Tmax,nsplit,pad", [
    (28, 4, 3584, 18944, 4491, 5002, 37, 0),     # 7B, first decode step of the bench prompt
    (28, 4, 3584, 18944, 5002, 5002, 40, 37),    # last slot of the cache, left-padded prompt
    (16, 2, 2048, 11008, 1500, 1732, 14, 0),     # 3B dims (n_rep 8)
    (16, 2, 2048, 11008, 40, 64, 1, 0),          # 3B dims, one split
    (28, 4, 3584, 18944, 97, 20480, 64, 3),      # few keys against 64 splits: most attention items hold no key at all
    (28, 4, 3584, 18944, 1, 64, 1, 0),           # one key (the token itself)
])
def test_decode_layer_block_equals_stand_alone_launches(dev, Hq, Hkv, H, I, ctx, Tmax, nsplit, pad):
    """o3v_decode_layer_block (ONE persistent launch: q/k/v -> attention -> merge -> o_proj -> RMSNorm -> gate/up + SwiGLU, every wave
    streaming its own weight rows across the in-launch hand-offs) == o3v_gemv_norm_qkv_rope + o3v_attn_decode + o3v_linear_decode(o_proj,
    RESIDUAL) + o3v_linear_decode(gate/up, fused RMSNorm, SWIGLU) BIT FOR BIT: residual stream, q, attention output, the appended K/V
    row, the SwiGLU vector.  Three epochs on one sync buffer with new inputs each time."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_gate_up
    D, Tnew, step = 128, 7, 3
    slot = ctx - 1
    g = torch.Generator().manual_seed(Hq * 1000 + ctx + I)
    N, QD = (Hq + 2 * Hkv) * D, Hq * D
    nw1 = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
    nw2 = (1 + 0.1 * torch.randn(H, generator=g)).to(BF).to(dev)
    wqkv = (torch.randn(N, H, generator=g) / math.sqrt(H)).to(BF).to(dev)
    bqkv = (0.5 * torch.randn(N, generator=g)).to(BF).to(dev)
    wo = (torch.randn(H, QD, generator=g) / math.sqrt(QD)).to(BF).to(dev)
    wgu = pack_gate_up((torch.randn(I, H, generator=g) / math.sqrt(H)).to(BF), (torch.randn(I, H, generator=g) / math.sqrt(H)).to(BF), I).to(dev)
    ang = torch.rand(1, Tnew, D // 2, generator=g) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).to(dev).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).to(dev).contiguous()
    kc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    vc0 = torch.randn(1, Hkv, Tmax, D, generator=g).to(BF).to(dev)
    kc0[:, :, slot] = float("nan")
    vc0[:, :, slot] = float("nan")
    k_lo = torch.tensor([pad], dtype=torch.int32, device=dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
    part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
    q2, att2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    k2, v2 = kc0.clone(), vc0.clone()
    act2 = torch.zeros(1, I, dtype=BF, device=dev)
    for rep in range(3):
        x0 = (torch.randn(1, H, generator=g) * 2).to(BF).to(dev)
        x1 = x0.clone()
        q1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        k1, v1 = kc0.clone(), vc0.clone()
        att1 = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        act1 = torch.zeros(1, I, dtype=BF, device=dev)
        po1, pm1 = torch.empty_like(part_o), torch.empty_like(part_ml)
        _lib.call("o3v_gemv_norm_qkv_rope", P(x1), P(nw1), 1e-6, P(wqkv), None, P(bqkv), 1, H, H, P(cos), P(sin), P(q1), P(k1), P(v1),
                  slot, Hq, Hkv, D, Tmax, Tnew, step, st)
        _lib.call("o3v_attn_decode", P(q1), P(k1), P(v1), P(att1), P(po1), P(pm1), P(k_lo), 1, Hq, Hkv, D, ctx, Tmax, nsplit, scale, st)
        _lib.call("o3v_linear_decode", P(att1), None, 0.0, P(wo), None, None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
        _lib.call("o3v_linear_decode", P(x1), P(nw2), 1e-6, P(wgu), None, None, None, P(act1), 1, 2 * I, H, H, I, 0, _lib.EPI_SWIGLU, st)
        x2 = x0.clone()
        k2[:, :, slot] = float("nan")
        v2[:, :, slot] = float("nan")
        act2.zero_()
        rc = lib.o3v_decode_layer_block(P(x2), P(nw1), 1e-6, P(wqkv), P(bqkv), P(wo), P(nw2), P(wgu), P(act2), P(cos), P(sin), P(q2), P(att2),
                                        P(k2), P(v2), P(part_o), P(part_ml), P(k_lo), H, I, Hq, Hkv, D, slot, Tmax, Tnew, step, nsplit, scale,
                                        P(sync), rep + 1, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        tmo = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
        assert tmo == 0, f"an in-launch wait timed out (code {tmo:#x})"
        assert not torch.isnan(x1.float()).any() and not torch.isnan(act1.float()).any()
        assert torch.equal(q2.view(torch.int16), q1.view(torch.int16))
        assert torch.equal(k2.view(torch.int16), k1.view(torch.int16)) and torch.equal(v2.view(torch.int16), v1.view(torch.int16))
        assert torch.equal(att2.view(torch.int16), att1.view(torch.int16))
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16))
        assert torch.equal(act2.view(torch.int16), act1.view(torch.int16))


def test_fused_rmsnorm_does_not_depend_on_the_decomposition(dev):
    """The RMSNorm prologue of the weight-streaming GEMV adds the squares in one order whatever workgroup size the launcher picks
    (o3v_gemv_body.h: 256 virtual threads, 4 virtual waves): the same weight rows give the same bits through a 6144-row call (3 waves
    per workgroup at K = 4096), a 2048-row call (4 waves) and the gate/up form of a 2-wave workgroup is covered by the layer-block
    test.  x has a wide dynamic range so that the order of the additions matters; before the order was fixed the one-launch attention
    block (4 waves) and the stand-alone q/k/v launch (3 waves) differed about once in 500 layer evaluations."""
    import ctypes as C
    from open_o3_video_amd import _lib
    K, N1, N2 = 4096, 6144, 2048
    g = torch.Generator().manual_seed(77)
    w = (torch.randn(N1, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for trial in range(96):
        x = (torch.randn(1, K, generator=g) * torch.exp(1.5 * torch.randn(1, K, generator=g))).to(BF).to(dev)
        o1 = torch.zeros(1, N1, dtype=BF, device=dev)
        o2 = torch.zeros(1, N2, dtype=BF, device=dev)
        _lib.call("o3v_linear_decode", P(x), P(nw), 1e-6, P(w), None, None, None, P(o1), 1, N1, K, K, N1, 0, _lib.EPI_NONE, st)
        _lib.call("o3v_linear_decode", P(x), P(nw), 1e-6, P(w), None, None, None, P(o2), 1, N2, K, K, N2, 0, _lib.EPI_NONE, st)
        assert torch.equal(o1[:, :N2].view(torch.int16), o2.view(torch.int16)), trial


@pytest.mark.parametrize("M", [8, 11, 16, 24, 32])
@pytest.mark.parametrize("N,K", [(3584, 3584), (3584, 18944), (2048, 11008), (4096, 12288), (896, 4864)])
def test_linear_decode_norm_next_equals_linear_plus_rmsnorm(dev, M, N, K):
    """o3v_linear_decode_norm_next (residual linear whose last M storing waves normalise one row each for the NEXT linear) ==
    o3v_linear_decode(RESIDUAL) + o3v_rmsnorm BIT FOR BIT, residual stream in place, four epochs on one sync buffer, no wait
    gave up.  Shapes: o_proj / down_proj of the 7B, 3B and 8B classes and of the medium fixture."""
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_mfma_fragments
    g = torch.Generator().manual_seed(N + K + M)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wp = pack_mfma_fragments(w)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    lib = _lib.load()
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h2 = torch.zeros(M, N, dtype=BF, device=dev)
    for epoch in range(1, 5):
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        x0 = (torch.randn(M, N, generator=g) * 2).to(BF).to(dev)
        x1, x2 = x0.clone(), x0.clone()
        h1 = torch.zeros(M, N, dtype=BF, device=dev)
        _lib.call("o3v_linear_decode", P(a), None, 0.0, P(w), P(wp), None, P(x1), P(x1), M, N, K, K, N, N, _lib.EPI_RESIDUAL, st)
        _lib.call("o3v_rmsnorm", P(x1), P(nw), P(h1), M, N, N, N, 1e-6, st)
        h2.fill_(float("nan"))
        rc = lib.o3v_linear_decode_norm_next(P(a), P(wp), P(x2), P(x2), M, N, K, K, N, N, P(nw), 1e-6, P(h2), N, P(sync), epoch, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        tmo = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
        assert tmo == 0, f"a wait gave up (code {tmo:#x})"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16))
        assert torch.equal(h2.view(torch.int16), h1.view(torch.int16)), (epoch, (h2.float() - h1.float()).abs().max().item())


def test_linear_decode_norm_next_rejects(dev):
    """Shapes the form does not take return O3V_ERR_SHAPE / O3V_ERR_ARG before any launch (the engine then issues two launches)."""
    import ctypes as C
    from open_o3_video_amd import _lib
    lib = _lib.load()
    t = torch.zeros(32 * 8192, dtype=BF, device=dev)
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    P = lambda x: C.c_void_p(x.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda M, N, K, ep, sy: lib.o3v_linear_decode_norm_next(P(t), P(t), P(t), P(t), M, N, K, K, N, N, P(t), 1e-6, P(t), N, sy, ep, st)
    assert call(4, 3584, 3584, 1, P(sync)) == _lib.ERR_SHAPE          # fewer than 8 rows: the fused-norm linears
    assert call(33, 3584, 3584, 1, P(sync)) == _lib.ERR_SHAPE
    assert call(8, 8192, 1024, 1, P(sync)) == _lib.ERR_SHAPE          # rows longer than a wave normalises from registers
    assert call(16, 128, 1024, 1, P(sync)) == _lib.ERR_SHAPE          # fewer storing waves than rows
    assert call(8, 3584, 3584, 0, P(sync)) == _lib.ERR_ARG            # epochs start at 1
    assert call(8, 3584, 3584, 1, None) == _lib.ERR_ARG
    torch.cuda.synchronize()
    assert int(sync.view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (700, 1280, 384), (515, 3456, 1280), (1030, 520, 256), (300, 4608, 3584),
                                   (4490, 3584, 3584), (2049, 1288, 1152), (4490, 3584, 18944), (9, 256, 512)])
def test_gemm_phased_kernel(dev, M, N, K):
    """The phased 256-tile kernel (csrc/o3v_gemm8p.hip: 16-MFMA phases between raw barriers, half-tile copies 6 deep in flight, the
    two wave rows one barrier apart) is BIT-IDENTICAL to the kernel with one __syncthreads() per K-step for every epilogue, at ragged
    M / N edges, short and long K (4 to 296 K-tiles) -- and stays so over repeated launches with other kernels in between (a misplaced
    wait or a too-early re-stage shows as rare wrong tiles, not as a constant error)."""
    import kernel_ops as ops
    from open_o3_video_amd.weights import pack_gate_up
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    cases = [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, bias, res), (ops.EPI_GELU, None, None)]
    ref = [ops.gemm(a, w, b, r, epi, force="gemm", tile=256) for epi, b, r in cases]
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for rep in range(6):
        for (epi, b, r), want in zip(cases, ref):
            got = ops.gemm(a, w, b, r, epi, force="gemm", tile=257)
            junk.fill_(rep)                                   # other traffic between the launches
            assert torch.equal(got, want), (rep, epi, int((got != want).sum()))
    if N % 32 == 0:
        I = N // 2
        packed = pack_gate_up(w[:I].contiguous(), w[I:].contiguous(), I)
        want = ops.gemm(a, packed, None, None, ops.EPI_SWIGLU, force="gemm", tile=256)
        for rep in range(3):
            assert torch.equal(ops.gemm(a, packed, None, None, ops.EPI_SWIGLU, force="gemm", tile=257), want)
    # an odd number of K-tiles is not taken
    from open_o3_video_amd import _lib
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    o = torch.empty(M, N, dtype=BF, device=dev)
    assert _lib.load().o3v_gemm_bf16_phased(P(a), P(w), None, None, P(o), M, N, 192, K, K, N, 0, 0, st) == _lib.ERR_SHAPE


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("M,N,K", [(17, 32784, 256), (24, 40000, 512), (32, 152064, 896), (32, 32768, 3584), (16, 32784, 256)])
def test_skinny_gemm_many_row_groups(dev, M, N, K, packed):
    """17..32 rows of x against >= 2048 weight row groups (the lm_head of a 32-video decode step): no K split, and a wave takes TWO
    adjacent 16-row weight blocks so that they share the x fragments (gemv_mfma_kernel RX = 2).  An odd number of blocks (N = 32784)
    leaves the last wave one valid block; 16 rows stay on the one-block form.  Row-major and fragment-major weights, every
    single-block epilogue, against the fp32 reference."""
    import ctypes as C
    import kernel_ops as ops
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_mfma_fragments
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF).to(dev)
    wp = pack_mfma_fragments(w) if packed else None
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    acc = a.float() @ w.float().t()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for epi, b, r in [(ops.EPI_NONE, bias, None), (ops.EPI_RESIDUAL, None, res), (ops.EPI_GELU, bias, None)]:
        out = torch.full((M, N), float("nan"), dtype=BF, device=dev)
        _lib.call("o3v_linear_decode", P(a), None, 0.0, P(w), P(wp), P(b), P(r), P(out), M, N, K, K, N, N, epi, st)
        close_bf16(out, _epi_ref(acc, b, r, epi))
