"""GPU tests of the fp8 x fp8 matrix-core path (csrc/o3v_fp8.hip; BASELINE config #5 "fp8 weights on CDNA4 fp8 MFMA"): the per-token
quantiser bit-exact against oracle/quant_ref.py, the W8A8 GEMM against the exact integer-free reference (fp8 products are exact in
fp32; only the accumulation order differs), and the opt-in W8A8 prefill of the engine against the oracle run with the same
quantisation.  Tolerances are written at each assert."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import fixture_models as fm
from oracle import model_ref, quant_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _quant_gpu(x, norm_w=None, eps=1e-6):
    from open_o3_video_amd import _lib
    R, K = x.shape
    q = torch.empty((R, K), dtype=torch.uint8, device="cuda")
    s = torch.empty(R, dtype=torch.float32, device="cuda")
    if norm_w is None:
        _lib.call("o3v_quantize_rows_fp8", _p(x), _p(q), _p(s), R, K, x.stride(0), K, _stream())
    else:
        _lib.call("o3v_rmsnorm_quantize_fp8", _p(x), _p(norm_w), _p(q), _p(s), R, K, x.stride(0), K, eps, _stream())
    return q, s


@pytest.mark.parametrize("K", [128, 896, 2048, 3584, 4096, 18944])
def test_quantize_rows_fp8(need_gpu, K):
    """o3v_quantize_rows_fp8 == quant_ref.quantize_rows bit for bit: power-of-two scale from the row maximum (exact powers of two,
    448 * 2^k boundaries, a zero row, tiny rows), e4m3fn codes with round-to-nearest-even including exact ties."""
    g = torch.Generator().manual_seed(K)
    x = (torch.randn(37, K, generator=g) * torch.logspace(-3, 2, 37)[:, None]).to(torch.bfloat16)
    x[3] = 0
    x[4, 5] = 448.0
    x[5, 7] = 896.0
    x[6, 0] = 447.0
    x[7, :16] = torch.tensor([1.0625, 1.1875, -1.0625, 0.0078125, 17.0, 19.0, 21.0, 23.0, 240.0, 248.0, 0.001953125, -0.0009765625, 3.25, 3.75,
                              1e-6, -448.0]).to(torch.bfloat16)          # ties of the 3-bit mantissa and subnormals
    q, s = _quant_gpu(x.cuda())
    qr, sr = quant_ref.quantize_rows(x)
    assert torch.equal(s.cpu(), sr)
    assert torch.equal(q.cpu(), qr.view(torch.uint8))
    # a strided input (rows of a wider buffer)
    wide = torch.zeros((37, K + 64), dtype=torch.bfloat16)
    wide[:, :K] = x
    q2, s2 = _quant_gpu(wide.cuda()[:, :K])
    assert torch.equal(q2, q) and torch.equal(s2, s)


@pytest.mark.parametrize("K", [128, 896, 3584, 4096])
def test_rmsnorm_quantize_fp8(need_gpu, K):
    """The fused form quantises exactly the bf16 rows o3v_rmsnorm would have written."""
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(K + 1)
    x = (torch.randn(50, K, generator=g) * 2).to(torch.bfloat16).cuda()
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    y = torch.empty_like(x)
    _lib.call("o3v_rmsnorm", _p(x), _p(w), _p(y), 50, K, K, K, 1e-6, _stream())
    q, s = _quant_gpu(x, w, 1e-6)
    qr, sr = quant_ref.quantize_rows(y.cpu())
    assert torch.equal(s.cpu(), sr) and torch.equal(q.cpu(), qr.view(torch.uint8))


def _gemm_fp8(a8, sa, w8, sw, bias, res, M, N, K, epi, n_out):
    from open_o3_video_amd import _lib
    out = torch.zeros((M, n_out), dtype=torch.bfloat16, device="cuda")
    _lib.call("o3v_gemm_fp8", _p(a8), _p(sa), _p(w8), _p(sw), _p(bias), _p(res), _p(out), M, N, K, K, K, n_out, 0 if res is None else res.stride(0),
              epi, _stream())
    return out


def _ulp_bf16(v):
    return torch.maximum(v.abs(), torch.tensor(2.0 ** -126)) * 2.0 ** -7


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 256), (1000, 4608, 896), (77, 1152, 1152), (513, 3584, 3584), (40, 896, 18944)])
def test_gemm_fp8_against_exact_products(need_gpu, M, N, K):
    """o3v_gemm_fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) against fp64 sums of the exact fp8 products: the operand
    lane map, the k order inside the 128-wide instruction, the swizzled LDS image and the tails (M, N not multiples of 256).
    Bound: the fp32 accumulation of K exact products is within K * 2^-24 relative of the true sum of magnitudes; after the single
    bf16 rounding an element may differ by one bf16 ulp.  Asserted: every element within 1 ulp (+ the accumulation bound), and at
    least 97 % identical to the rounded exact value (sums with cancellation sit near a rounding boundary more often)."""
    from open_o3_video_amd.weights import quantize_rows_fp8
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    qx, sx = quant_ref.quantize_rows(x)
    w8, sw = quantize_rows_fp8(w)
    bias = (0.1 * torch.randn(N, generator=g)).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16)
    prod = (qx.double() @ w8.view(torch.float8_e4m3fn).double().t()) * sx.double()[:, None] * sw.double()[None, :]
    mag = (qx.double().abs() @ w8.view(torch.float8_e4m3fn).double().abs().t()) * sx.double()[:, None] * sw.double()[None, :]
    dev = lambda t: t.cuda()
    a8 = dev(qx.view(torch.uint8))
    # plain + bias
    got = _gemm_fp8(a8, dev(sx), dev(w8), dev(sw), dev(bias), None, M, N, K, 0, N).float().cpu()
    want = (prod + bias.double()).float()
    tol = _ulp_bf16(want) + (mag * K * 2.0 ** -24).float()
    assert ((got - want).abs() <= tol).all(), float(((got - want).abs() - tol).max())
    assert (got == want.to(torch.bfloat16).float()).float().mean() > 0.97
    # residual: bf16(acc) + res, one more rounding (TF:692-757)
    got = _gemm_fp8(a8, dev(sx), dev(w8), dev(sw), None, dev(res), M, N, K, 1, N).float().cpu()
    want = (prod.float().to(torch.bfloat16).float() + res.float())
    assert ((got - want).abs() <= 2 * _ulp_bf16(want) + _ulp_bf16(prod.float()) + (mag * K * 2.0 ** -24).float()).all()
    assert (got == want.to(torch.bfloat16).float()).float().mean() > 0.95
    # SwiGLU over the 16-row interleaved gate / up layout (weights.py): out[:, 16 b + j] = silu(row 32 b + j) * row 32 b + 16 + j
    if N % 32 == 0:
        got = _gemm_fp8(a8, dev(sx), dev(w8), dev(sw), None, None, M, N, K, 3, N // 2).float().cpu()
        p = prod.float().to(torch.bfloat16).float().view(M, N // 32, 2, 16)
        want = (torch.nn.functional.silu(p[:, :, 0]).to(torch.bfloat16).float() * p[:, :, 1]).reshape(M, N // 2)
        err = (got - want).abs()
        assert (err <= 4 * _ulp_bf16(want) + 0.02 * want.abs() + 1e-3).all()     # a gate / up value one bf16 ulp off moves the product
        assert (got == want.to(torch.bfloat16).float()).float().mean() > 0.9


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (1000, 4608, 1024), (77, 1152, 1280), (513, 3584, 3584), (4490, 3584, 18944),
                                   (2049, 37888, 3584)])
def test_gemm_fp8_phased_schedule_is_bit_identical(need_gpu, M, N, K):
    """The phased schedule of the fp8 GEMM (o3v_gemm_fp8_sched schedule 2: 8-MFMA phases between raw barriers, half-tile copies six
    deep in flight, wave rows one barrier apart) gives the bits of the kernel with one __syncthreads() per K-tile (schedule 1) for
    every epilogue and at ragged edges, over repeated launches with other traffic in between; the library's own choice (0) is the
    phased kernel on these shapes; an odd number of K-tiles is refused by schedule 2 and served by schedule 0."""
    from open_o3_video_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    a8 = torch.randint(0, 256, (M, K), generator=g, dtype=torch.uint8)
    w8 = torch.randint(0, 256, (N, K), generator=g, dtype=torch.uint8)
    for t in (a8, w8):                       # no NaN encodings (0x7f / 0xff)
        t[(t & 0x7f) == 0x7f] = 0x3c
    a8, w8 = a8.cuda(), w8.cuda()
    sa = torch.exp2(torch.randint(-6, 3, (M,), generator=g).float()).cuda()
    sw = torch.exp2(torch.randint(-9, -3, (N,), generator=g).float()).cuda()
    bias = (0.1 * torch.randn(N, generator=g)).to(torch.bfloat16).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    junk = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")

    def run(sched, epi, b, r, n_out):
        out = torch.zeros((M, n_out), dtype=torch.bfloat16, device="cuda")
        rc = _lib.load().o3v_gemm_fp8_sched(_p(a8), _p(sa), _p(w8), _p(sw), _p(b), _p(r), _p(out), M, N, K, K, K, n_out,
                                            0 if r is None else r.stride(0), epi, sched, _stream())
        assert rc == 0, rc
        return out
    for epi, b, r, n_out in ((0, bias, None, N), (1, None, res, N), (3, None, None, N // 2)):
        want = run(1, epi, b, r, n_out)
        for rep in range(4):
            got = run(2 if rep else 0, epi, b, r, n_out)
            junk.fill_(rep)
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (epi, rep, int((got != want).sum()))
    o = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    lib = _lib.load()
    assert lib.o3v_gemm_fp8_sched(_p(a8), _p(sa), _p(w8), _p(sw), None, None, _p(o), M, N, 384, K, K, N, 0, 0, 2, _stream()) == -2
    assert lib.o3v_gemm_fp8_sched(_p(a8), _p(sa), _p(w8), _p(sw), None, None, _p(o), M, N, 384, K, K, N, 0, 0, 0, _stream()) == 0
    assert lib.o3v_gemm_fp8_sched(_p(a8), _p(sa), _p(w8), _p(sw), None, None, _p(o), M, N, K, K, K, N, 0, 0, 3, _stream()) == -1


def test_gemm_fp8_argument_errors(need_gpu):
    from open_o3_video_amd import _lib
    lib = _lib.load()
    a = torch.zeros((16, 128), dtype=torch.uint8, device="cuda")
    s = torch.ones(16, dtype=torch.float32, device="cuda")
    o = torch.zeros((16, 16), dtype=torch.bfloat16, device="cuda")
    assert lib.o3v_gemm_fp8(_p(a), _p(s), _p(a), _p(s), None, None, _p(o), 16, 16, 64, 128, 128, 16, 0, 0, _stream()) == -2      # K % 128
    assert lib.o3v_gemm_fp8(None, _p(s), _p(a), _p(s), None, None, _p(o), 16, 16, 128, 128, 128, 16, 0, 0, _stream()) == -1
    assert lib.o3v_gemm_fp8(_p(a), _p(s), _p(a), _p(s), None, None, _p(o), 16, 16, 128, 128, 128, 16, 0, 1, _stream()) == -1     # residual without res


def _engine(cfg, W, **kw):
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, getter_from_dict
    c = O3VConfig.from_dict(cfg)
    return O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda", **kw))


@pytest.mark.parametrize("fname,cfgf,wseed", [("g6_tiny.npz", fm.tiny_config, 0), ("g7_medium.npz", fm.medium_config, 2)])
def test_w8a8_prefill_against_oracle(need_gpu, golden_dir, fname, cfgf, wseed):
    """engine.fp8_prefill: the LLM linears of the prefill as W8A8, against oracle/model_ref.py with quant="w8a8" (the same
    quantisation in fp32 torch).  An fp8 quantiser is discontinuous: wherever the engine and the oracle differ by a bf16 ulp in an
    activation (after attention that is most elements -- flash softmax against eager), some fp8 codes flip and each flip is a 6-12 %
    error of that element, so two correct implementations of the quantised function differ by a fraction of the quantisation
    noise itself (the pieces are pinned exactly: quantiser bit for bit, GEMM to one bf16 ulp, above).  Bars, with E_q = the
    oracle's own quantisation shift max|oracle(w8a8) - oracle(bf16)|: the engine is within E_q of the oracle's quantised logits
    (max) and within E_q / 4 on average, its shift against its own bf16 path is of E_q's size (between E_q / 3 and 3 E_q), and the
    arg-max agrees with the oracle's wherever the oracle's margin exceeds 2 E_q.  Off by default."""
    g = np.load(os.path.join(golden_dir, fname))
    cfg = cfgf()
    W = fm.make_weights(cfg, wseed)
    eng = _engine(cfg, W, fp8_decode=True)
    assert eng.fp8_prefill is False
    pv = torch.from_numpy(g["pixel_values"])
    base = eng.forward_logits(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"]).float().cpu()
    eng.fp8_prefill = True
    try:
        q = eng.forward_logits(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"]).float().cpu()
    finally:
        eng.fp8_prefill = False
    ref = model_ref.full_logits(W, cfg, g["input_ids"], None, pv, g["grid"], dtype=torch.bfloat16, quant="w8a8").float()
    ref16 = model_ref.full_logits(W, cfg, g["input_ids"], None, pv, g["grid"], dtype=torch.bfloat16).float()
    e_q, e_q_mean = (ref - ref16).abs().max().item(), (ref - ref16).abs().mean().item()
    err, err_mean = (q - ref).abs().max().item(), (q - ref).abs().mean().item()
    shift = (q - base).abs().max().item()
    print(f"{fname}: oracle quantisation shift E_q max {e_q:.3f} mean {e_q_mean:.4f}; engine W8A8 vs oracle(w8a8) max {err:.3f} mean {err_mean:.4f}; "
          f"engine shift vs its bf16 path {shift:.3f}; |logit| max {ref.abs().max():.1f}")
    assert err <= e_q and err_mean <= max(0.25 * e_q, e_q_mean)
    assert e_q / 3 < shift < 3 * e_q
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * e_q
    assert torch.equal(q.argmax(-1)[safe], ref.argmax(-1)[safe])
    # without the fp8 rows the switch refuses loudly
    eng2 = _engine(cfg, W)
    eng2.fp8_prefill = True
    from open_o3_video_amd import _lib
    with pytest.raises(_lib.O3VError):
        eng2.forward_logits(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"])


def test_w8a8_generate_and_logps(need_gpu, golden_dir):
    """generate() and completion_logps() run end to end with the W8A8 prefill (decode on the fp8 rows): the greedy tokens follow the
    bf16-prefill run of the same fp8-row engine wherever that run's margin is safe, and the log-probs of the completion agree
    with the W8A8 forward's own logits."""
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    eng = _engine(cfg, fm.make_weights(cfg, 2), fp8_decode=True)
    kw = dict(pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=g["grid"])
    a = eng.generate(g["input_ids"], None, max_new_tokens=12, pad_token_id=cfg["pad_token_id"], **kw)
    eng.fp8_prefill = True
    b = eng.generate(g["input_ids"], None, max_new_tokens=12, pad_token_id=cfg["pad_token_id"], **kw)
    S = g["input_ids"].shape[1]
    k = 0
    while k < 12 and a.sequences[0, S + k] == b.sequences[0, S + k]:
        k += 1
    assert k == 12 or a.margins[0, k].item() < 1.0
    comp = b.sequences[:, S:]
    lp = eng.completion_logps(g["input_ids"], comp, **kw)
    lg = eng.forward_logits(b.sequences.cpu().numpy(), None, **kw)
    ref = torch.log_softmax(lg[0, S - 1:-1].float(), dim=-1).gather(1, comp[0].to(lg.device)[:, None])[:, 0]
    assert (lp[0] - ref).abs().max().item() < 0.08
