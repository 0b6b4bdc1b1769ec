"""Oracle: the W8A8 arithmetic of the opt-in fp8 prefill (BASELINE config #5: "fp8 weights on CDNA4 fp8 MFMA"), restated in fp32
torch.  The reference itself holds no fp8 code: it serves the checkpoint through vLLM (R:eval/models/model_vllm.py:18-26), whose fp8
linear method quantises weights per output channel and activations per token on the fly; this module is the definition the HIP
kernels (csrc/o3v_fp8.hip) are checked against -- parity unpinned against a real vLLM run (vllm is not installed offline).

  weights      per output row: scale = smallest power of two s with max|w| / s <= 448, q = e4m3fn(w / s) (round to nearest even)
  activations  per token (row of the linear's input): the same rule on the bf16 values the bf16 path would feed the linear
  linear       y = (sum_k q_x[k] * q_w[n, k]) * s_x * s_w[n] (+ bias), fp32 accumulation, one rounding to the model dtype

Test infrastructure only (see oracle/__init__.py)."""
from __future__ import annotations

import math

import torch

FP8_MAX = 448.0


def pow2_scale(amax: float) -> float:
    """Smallest power of two s with amax / s <= 448 (1.0 for amax == 0): exact, from the mantissa / exponent of amax."""
    if not amax > 0.0:
        return 1.0
    m, e = math.frexp(float(amax))          # amax = m * 2^e, m in [0.5, 1); 448 = 0.875 * 2^9
    return math.ldexp(1.0, e - 9 if m <= 0.875 else e - 8)


def quantize_rows(x: torch.Tensor):
    """[R, K] float -> (float8_e4m3fn [R, K], f32 [R] power-of-two scales)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    s = torch.tensor([pow2_scale(a) for a in amax.tolist()], dtype=torch.float32)
    return (xf / s[:, None]).to(torch.float8_e4m3fn), s


def dequantize_rows(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    return q.float() * s[:, None]


def linear_w8a8(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """x [..., K] in the model dtype, w [N, K] (bf16-representable): both quantised as above, fp32 product, result in x.dtype."""
    shp = x.shape
    qx, sx = quantize_rows(x.reshape(-1, shp[-1]))
    qw, sw = quantize_rows(w)
    y = (qx.float() @ qw.float().t()) * sx[:, None] * sw[None, :]
    if bias is not None:
        y = y + bias.float()
    return y.to(x.dtype).reshape(*shp[:-1], w.shape[0])
