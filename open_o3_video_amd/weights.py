"""Checkpoint -> device weight images for libo3v_hip.so.

Packing done once at load (all bf16, resident in HBM):
  * fused q|k|v weight and bias of every LLM layer (one GEMM / GEMV instead of three);
  * gate/up fused and interleaved in 16-row groups so the SwiGLU epilogue finds gate_j and up_j in one lane
    (csrc/o3v_gemm.hip EPI_SWIGLU); intermediate width zero-padded to a multiple of 64 (ViT 3420 -> 3456);
  * Conv3d patch-embed weight flattened to [hidden, 1176] and zero-padded to K = 1216.
Zero padding adds exact zeros to fp32 sums, so results are unchanged.

Accepts HF 5.x names (model.visual.*, model.language_model.*) and the hub names of Qwen2.5-VL checkpoints
(visual.*, model.layers.*); loads a local safetensors directory (never downloads).
"""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
from typing import Callable, Dict

import torch

from . import _lib
from .config import O3VConfig


def pack_gate_up(gate: torch.Tensor, up: torch.Tensor, ipad: int) -> torch.Tensor:
    """[I,K],[I,K] -> [2*ipad,K]: rows 32g..32g+15 = gate[16g..], rows 32g+16..32g+31 = up[16g..] (zero padded)."""
    I, K = gate.shape[0], (gate.shape[1] if gate.dim() == 2 else 1)
    g = torch.zeros((ipad,) + tuple(gate.shape[1:]), dtype=gate.dtype, device=gate.device)
    u = torch.zeros_like(g)
    g[:I] = gate
    u[:I] = up
    tail = tuple(gate.shape[1:])
    g = g.reshape((ipad // 16, 16) + tail)
    u = u.reshape((ipad // 16, 16) + tail)
    return torch.stack([g, u], dim=1).reshape((2 * ipad,) + tail).contiguous()


def pack_mfma_fragments(w: torch.Tensor) -> torch.Tensor:
    """[N,K] row-major -> MFMA-fragment-major [N/16][K/32][64 lanes][8]: lane l of the A fragment of (row block nb,
    k-step ks) holds W[nb*16 + (l & 15)][ks*32 + (l >> 4)*8 : +8], so one wave-instruction reads one contiguous KiB
    (csrc/o3v_gemm.hip gemv_mfma_kernel<PACKED>).  Requires N % 16 == 0 and K % 32 == 0."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def pack_mfma_fragments_fp8(w8: torch.Tensor) -> torch.Tensor:
    """uint8 [N,K] (fp8 codes, row-major) -> fragment-major [N/16][K/64][64 lanes][16]: lane l of (row block nb, 64-wide double
    step t) holds W[nb*16 + (l & 15)][t*64 + (l >> 4)*16 : +16] -- the A fragments of two MFMAs, one contiguous KiB per
    wave-instruction (csrc/o3v_gemm.hip gemv_mfma_fp8_kernel).  Requires N % 16 == 0 and K % 64 == 0."""
    N, K = w8.shape
    assert N % 16 == 0 and K % 64 == 0
    return w8.view(N // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


FP8_MAX = 448.0   # largest finite OCP e4m3fn value


def quantize_rows_fp8(w: torch.Tensor):
    """[N, K] bf16 -> (uint8 [N, K] holding OCP e4m3fn codes, f32 [N] scales): per output row the scale is the smallest POWER
    OF TWO s with max|w| / s <= 448, q = rne(w / s) (torch's float8_e4m3fn conversion; no overflow); an all-zero row gets
    s = 1.  A power-of-two scale costs no precision (fp8 is floating point: it only shifts the exponent) and makes the
    dequantised weight fp8(q) * s exactly representable in bf16, so the fp8 path can be checked against the bf16 path on the
    very same weight values.  dequantize_rows_fp8 restates what the kernel multiplies."""
    wf = w.float()
    amax = wf.abs().amax(dim=1)
    s = torch.exp2(torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / FP8_MAX)))
    s = torch.where(amax > 0, s, torch.ones_like(s))
    s = torch.where(amax / s > FP8_MAX, s * 2, s)          # guard the log2 rounding at exact powers of two
    q = (wf / s[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), s.contiguous()


def dequantize_rows_fp8(q8: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    return q8.view(torch.float8_e4m3fn).float() * s[:, None].float()


def pad_cols(w: torch.Tensor, kpad: int) -> torch.Tensor:
    if w.shape[1] == kpad:
        return w.contiguous()
    out = torch.zeros((w.shape[0], kpad), dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


class _Getter:
    def __init__(self, fn: Callable[[str], torch.Tensor], device):
        self.fn, self.device = fn, device

    def __call__(self, *names):
        for n in names:
            try:
                t = self.fn(n)
            except KeyError:
                continue
            if t is not None:
                return t.to(device=self.device, dtype=torch.bfloat16)
        raise KeyError(f"weight not found under any of {names}")


class DeviceWeights:
    """Owns the packed tensors and the ctypes descriptors that point at them."""

    def __init__(self, cfg: O3VConfig, get: Callable[[str], torch.Tensor], device="cuda", batched_decode: bool = True,
                 fp8_decode: bool = False):
        """batched_decode: also keep MFMA-fragment-major copies of the LLM matrices (+1x their size in HBM) so that
        decoding 2..8 sequences together (group rollout) streams weights at the same rate as batch 1.
        fp8_decode: also keep fp8 (OCP e4m3fn) copies of the LLM matrices and the head with one fp32 scale per output row
        (quantize_rows_fp8); decode at batch <= 3 then streams those -- half the bytes per step (BASELINE config #5)."""
        self.cfg = cfg
        self.device = torch.device(device)
        g = _Getter(get, self.device)
        self.t: Dict[str, torch.Tensor] = {}
        vc, tc = cfg.vision, cfg.text

        def v(name):
            return g("model.visual." + name, "visual." + name)

        def l(name):
            return g("model.language_model." + name, "model." + name, "language_model.model." + name)

        keep = self.t
        self.vit = self.vit3 = None
        if cfg.arch == "qwen3_vl":
            self._pack_vision_q3(v)
        else:
            self._pack_vision(v)
        # ---- text
        keep["l.embed"] = l("embed_tokens.weight").contiguous()
        self.llm_layers = (_lib.LlmLayerW * tc.num_hidden_layers)()
        for i in range(tc.num_hidden_layers):
            b = f"layers.{i}."
            keep[f"l{i}.ln1"] = l(b + "input_layernorm.weight").contiguous()
            keep[f"l{i}.ln2"] = l(b + "post_attention_layernorm.weight").contiguous()
            keep[f"l{i}.qkv_w"] = torch.cat([l(b + "self_attn.q_proj.weight"), l(b + "self_attn.k_proj.weight"),
                                             l(b + "self_attn.v_proj.weight")], dim=0).contiguous()
            fields = ["ln1", "ln2", "qkv_w", "o_w", "gu_w", "down_w"]
            if tc.attention_bias:
                keep[f"l{i}.qkv_b"] = torch.cat([l(b + "self_attn.q_proj.bias"), l(b + "self_attn.k_proj.bias"),
                                                 l(b + "self_attn.v_proj.bias")], dim=0).contiguous()
                fields.append("qkv_b")
            if tc.qk_norm:
                keep[f"l{i}.q_norm"] = l(b + "self_attn.q_norm.weight").contiguous()
                keep[f"l{i}.k_norm"] = l(b + "self_attn.k_norm.weight").contiguous()
                fields += ["q_norm", "k_norm"]
            keep[f"l{i}.o_w"] = l(b + "self_attn.o_proj.weight").contiguous()
            keep[f"l{i}.gu_w"] = pack_gate_up(l(b + "mlp.gate_proj.weight"), l(b + "mlp.up_proj.weight"), tc.inter_pad)
            keep[f"l{i}.down_w"] = pad_cols(l(b + "mlp.down_proj.weight"), tc.inter_pad)
            for f in fields:
                setattr(self.llm_layers[i], f, keep[f"l{i}.{f}"].data_ptr())
            for f in ("qkv_w", "o_w", "gu_w", "down_w"):
                w = keep[f"l{i}.{f}"]
                if batched_decode and w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0:
                    keep[f"l{i}.{f}p"] = pack_mfma_fragments(w)
                    setattr(self.llm_layers[i], f + "p", keep[f"l{i}.{f}p"].data_ptr())
            if fp8_decode:
                for f in ("qkv_w", "o_w", "gu_w", "down_w"):
                    q8, sc = quantize_rows_fp8(keep[f"l{i}.{f}"])
                    keep[f"l{i}.{f}8"], keep[f"l{i}.{f}8s"] = q8, sc
                    setattr(self.llm_layers[i], f[:-1] + "w8", q8.data_ptr())
                    setattr(self.llm_layers[i], f[:-2] + "_s", sc.data_ptr())
                    if batched_decode and q8.shape[0] % 16 == 0 and q8.shape[1] % 64 == 0:
                        keep[f"l{i}.{f}8p"] = pack_mfma_fragments_fp8(q8)       # 4..32 decode rows (config #5's N = 16 chains)
                        setattr(self.llm_layers[i], f[:-1] + "w8p", keep[f"l{i}.{f}8p"].data_ptr())
        keep["l.norm"] = l("norm.weight").contiguous()
        if tc.tie_word_embeddings:
            keep["l.head"] = keep["l.embed"]
        else:
            try:
                keep["l.head"] = g("lm_head.weight").contiguous()
            except KeyError as e:   # an incomplete shard / index or a renamed key must not silently turn into a tied head
                raise KeyError("lm_head.weight is missing from the checkpoint although tie_word_embeddings is false") from e
        head_p = 0
        if batched_decode and keep["l.head"].shape[0] % 16 == 0 and keep["l.head"].shape[1] % 32 == 0:
            keep["l.headp"] = pack_mfma_fragments(keep["l.head"])
            head_p = keep["l.headp"].data_ptr()
        self.llm = _lib.LlmDesc(hidden=tc.hidden_size, layers=tc.num_hidden_layers, heads=tc.num_attention_heads,
                                kv_heads=tc.num_key_value_heads, head_dim=tc.head_dim, inter=tc.inter_pad,
                                vocab=tc.vocab_size, rms_eps=tc.rms_norm_eps, embed=keep["l.embed"].data_ptr(),
                                layer=self.llm_layers, final_norm=keep["l.norm"].data_ptr(),
                                lm_head=keep["l.head"].data_ptr(), lm_head_p=head_p)
        self.fp8_decode = bool(fp8_decode)
        if fp8_decode:
            keep["l.head8"], keep["l.head8s"] = quantize_rows_fp8(keep["l.head"])
            self.llm.lm_head8, self.llm.lm_head_s = keep["l.head8"].data_ptr(), keep["l.head8s"].data_ptr()
            if batched_decode and keep["l.head8"].shape[0] % 16 == 0 and keep["l.head8"].shape[1] % 64 == 0:
                keep["l.head8p"] = pack_mfma_fragments_fp8(keep["l.head8"])
                self.llm.lm_head8p = keep["l.head8p"].data_ptr()
        self._check_shapes()

    def _pack_vision(self, v):
        """Qwen2.5-VL tower (TF:408-471)."""
        keep, vc = self.t, self.cfg.vision
        pw = v("patch_embed.proj.weight").reshape(vc.hidden_size, -1)
        keep["v.patch_w"] = pad_cols(pw, vc.patch_k_pad)
        self.vit_blocks = (_lib.VitBlockW * vc.depth)()
        for i in range(vc.depth):
            b = f"blocks.{i}."
            keep[f"v{i}.norm1"] = v(b + "norm1.weight").contiguous()
            keep[f"v{i}.norm2"] = v(b + "norm2.weight").contiguous()
            keep[f"v{i}.qkv_w"] = v(b + "attn.qkv.weight").contiguous()
            keep[f"v{i}.qkv_b"] = v(b + "attn.qkv.bias").contiguous()
            keep[f"v{i}.proj_w"] = v(b + "attn.proj.weight").contiguous()
            keep[f"v{i}.proj_b"] = v(b + "attn.proj.bias").contiguous()
            keep[f"v{i}.gu_w"] = pack_gate_up(v(b + "mlp.gate_proj.weight"), v(b + "mlp.up_proj.weight"), vc.inter_pad)
            keep[f"v{i}.gu_b"] = pack_gate_up(v(b + "mlp.gate_proj.bias"), v(b + "mlp.up_proj.bias"), vc.inter_pad)
            keep[f"v{i}.down_w"] = pad_cols(v(b + "mlp.down_proj.weight"), vc.inter_pad)
            keep[f"v{i}.down_b"] = v(b + "mlp.down_proj.bias").contiguous()
            for f in ("norm1", "norm2", "qkv_w", "qkv_b", "proj_w", "proj_b", "gu_w", "gu_b", "down_w", "down_b"):
                setattr(self.vit_blocks[i], f, keep[f"v{i}.{f}"].data_ptr())
        keep["v.ln_q"] = v("merger.ln_q.weight").contiguous()
        keep["v.m0_w"] = v("merger.mlp.0.weight").contiguous()
        keep["v.m0_b"] = v("merger.mlp.0.bias").contiguous()
        keep["v.m2_w"] = v("merger.mlp.2.weight").contiguous()
        keep["v.m2_b"] = v("merger.mlp.2.bias").contiguous()
        mask = 0
        for i in vc.fullatt_block_indexes:
            mask |= 1 << int(i)
        self.vit = _lib.VitDesc(depth=vc.depth, hidden=vc.hidden_size, heads=vc.num_heads, inter_pad=vc.inter_pad,
                                out_hidden=vc.out_hidden_size, patch_k_pad=vc.patch_k_pad, merge_unit=vc.merge_unit,
                                fullatt_mask=mask, patch_w=keep["v.patch_w"].data_ptr(), blocks=self.vit_blocks,
                                ln_q=keep["v.ln_q"].data_ptr(), m0_w=keep["v.m0_w"].data_ptr(),
                                m0_b=keep["v.m0_b"].data_ptr(), m2_w=keep["v.m2_w"].data_ptr(),
                                m2_b=keep["v.m2_b"].data_ptr())

    def _pack_vision_q3(self, v):
        """Qwen3-VL tower (TF3:606-737): LayerNorm + bias, fc1/fc2 MLP, learned position table, main + DeepStack mergers.
        Heads of head_dim 72 are stored head_dim_pad = 80 wide, each rotary half followed by its share of zero rows
        ([36 | 4 zeros | 36 | 4 zeros]): the pairs (j, j + 40) the rotation kernel couples are the model's (j, j + 36), the
        zero dims add nothing to q.k, and the matching zero columns of proj drop the padded outputs."""
        keep, vc = self.t, self.cfg.vision
        hid, heads, hd, Dp, hp, ip = vc.hidden_size, vc.num_heads, vc.head_dim, vc.head_dim_pad, vc.hidden_pad, vc.inter_pad
        dev = self.device
        half, halfp = hd // 2, Dp // 2
        j = torch.arange(Dp, device=dev)
        inner = torch.where(j < halfp, j, j - halfp)
        src_in_head = torch.where(j < halfp, inner, inner + half)
        valid = inner < half
        src = (torch.arange(heads, device=dev)[:, None] * hd + src_in_head[None, :]).reshape(-1)      # [heads*Dp]
        valid = valid[None, :].expand(heads, Dp).reshape(-1)
        src = torch.where(valid, src, torch.zeros_like(src))

        def head_rows(w):      # [heads*hd, ...] -> [heads*Dp, ...]
            out = w[src]
            out[~valid] = 0
            return out

        keep["v.patch_w"] = pad_cols(v("patch_embed.proj.weight").reshape(hid, -1), vc.patch_k_pad)
        keep["v.patch_b"] = v("patch_embed.proj.bias").contiguous()
        keep["v.pos_embed"] = v("pos_embed.weight").contiguous()
        self.vit3_blocks = (_lib.Vit3BlockW * vc.depth)()
        for i in range(vc.depth):
            b = f"blocks.{i}."
            for n in ("norm1", "norm2"):
                keep[f"v{i}.{n}_w"] = v(b + n + ".weight").contiguous()
                keep[f"v{i}.{n}_b"] = v(b + n + ".bias").contiguous()
            qw, qb = v(b + "attn.qkv.weight"), v(b + "attn.qkv.bias")
            keep[f"v{i}.qkv_w"] = pad_cols(torch.cat([head_rows(t) for t in qw.view(3, hid, hid)], dim=0), hp)
            keep[f"v{i}.qkv_b"] = torch.cat([head_rows(t) for t in qb.view(3, hid)], dim=0).contiguous()
            keep[f"v{i}.proj_w"] = head_rows(v(b + "attn.proj.weight").t().contiguous()).t().contiguous()
            keep[f"v{i}.proj_b"] = v(b + "attn.proj.bias").contiguous()
            f1 = torch.zeros((ip, hp), dtype=torch.bfloat16, device=dev)
            f1[: vc.intermediate_size, :hid] = v(b + "mlp.linear_fc1.weight")
            f1b = torch.zeros(ip, dtype=torch.bfloat16, device=dev)
            f1b[: vc.intermediate_size] = v(b + "mlp.linear_fc1.bias")
            keep[f"v{i}.fc1_w"], keep[f"v{i}.fc1_b"] = f1, f1b
            keep[f"v{i}.fc2_w"] = pad_cols(v(b + "mlp.linear_fc2.weight"), ip)
            keep[f"v{i}.fc2_b"] = v(b + "mlp.linear_fc2.bias").contiguous()
            for f in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"):
                setattr(self.vit3_blocks[i], f, keep[f"v{i}.{f}"].data_ptr())

        def merger(prefix, key, post):
            m = _lib.Vit3MergerW(postshuffle=int(post))
            for f, name in (("norm_w", "norm.weight"), ("norm_b", "norm.bias"), ("fc1_w", "linear_fc1.weight"),
                            ("fc1_b", "linear_fc1.bias"), ("fc2_w", "linear_fc2.weight"), ("fc2_b", "linear_fc2.bias")):
                keep[f"{key}.{f}"] = v(prefix + name).contiguous()
                setattr(m, f, keep[f"{key}.{f}"].data_ptr())
            return m

        deep_idx = [int(i) for i in vc.deepstack_visual_indexes]
        if len(deep_idx) > _lib.MAX_DEEPSTACK or sorted(set(deep_idx)) != deep_idx or any(i >= vc.depth for i in deep_idx):
            raise _lib.O3VError(f"deepstack_visual_indexes {deep_idx}: need <= {_lib.MAX_DEEPSTACK} ascending block indexes")
        self.vit3 = _lib.Vit3Desc(depth=vc.depth, hidden=hid, heads=heads, head_dim=hd, head_dim_pad=Dp, inter_pad=ip,
                                  out_hidden=vc.out_hidden_size, patch_k_pad=vc.patch_k_pad, merge_unit=vc.merge_unit,
                                  n_deep=len(deep_idx), patch_w=keep["v.patch_w"].data_ptr(), patch_b=keep["v.patch_b"].data_ptr(),
                                  blocks=self.vit3_blocks, merger=merger("merger.", "v.m", False))
        for k, idx in enumerate(deep_idx):
            self.vit3.deep_index[k] = idx
            self.vit3.deep[k] = merger(f"deepstack_merger_list.{k}.", f"v.ds{k}", True)

    def _check_shapes(self):
        vc, tc = self.cfg.vision, self.cfg.text
        q3 = self.cfg.arch == "qwen3_vl"
        if (vc.hidden_size % 64 and not q3) or tc.hidden_size % 64 or (tc.num_attention_heads * tc.head_dim) % 64:
            raise _lib.O3VError("hidden sizes must be multiples of 64 for the MFMA GEMM K loop")
        if q3 and ((vc.num_heads * vc.head_dim_pad) % 64 or (vc.hidden_size * vc.merge_unit) % 64 or vc.hidden_size % 8 or vc.head_dim % 4):
            raise _lib.O3VError("qwen3_vl vision widths: heads*pad16(head_dim) and 4*hidden must be multiples of 64")
        if vc.head_dim_pad not in (32, 64, 80, 128) or tc.head_dim not in (32, 64, 128):
            raise _lib.O3VError(f"unsupported head_dim (vision {vc.head_dim}, text {tc.head_dim})")
        if tc.num_attention_heads // tc.num_key_value_heads > 8:
            raise _lib.O3VError("GQA group size > 8 not supported by the decode attention kernel")

    def nbytes(self):
        seen, n = set(), 0
        for t in self.t.values():
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                n += t.numel() * t.element_size()
        return n


# -------------------------------------------------------------------------------------------- sources of weights
def getter_from_dict(sd: Dict[str, torch.Tensor]):
    def get(name):
        return sd[name]
    return get


def getter_from_safetensors_dir(path: str):
    """Local HF checkpoint directory (model.safetensors or sharded + index).  Never touches the network."""
    from safetensors import safe_open

    index = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            wm = json.load(f)["weight_map"]
    else:
        wm = {}
        for fn in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
            with safe_open(fn, framework="pt") as f:
                for k in f.keys():
                    wm[k] = os.path.basename(fn)
    if not wm:
        raise FileNotFoundError(f"no safetensors weights under {path}")
    handles = {}

    def get(name):
        fn = wm[name]  # KeyError -> next alias
        if fn not in handles:
            handles[fn] = safe_open(os.path.join(path, fn), framework="pt")
        return handles[fn].get_tensor(name)
    return get


def random_getter(cfg: O3VConfig, seed=1234, device="cuda", std=0.02, head_std=None):
    """Seeded random bf16 weights at the model's true dimensions, generated on the device (benchmarks only:
    there are no checkpoints offline; throughput does not depend on weight values)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    vc, tc = cfg.vision, cfg.text
    shapes = {}
    vh, vi = vc.hidden_size, vc.intermediate_size
    shapes["model.visual.patch_embed.proj.weight"] = (vh, vc.patch_k)
    q3 = cfg.arch == "qwen3_vl"
    mh = vh * vc.merge_unit
    if q3:
        shapes["model.visual.patch_embed.proj.bias"] = (vh,)
        shapes["model.visual.pos_embed.weight"] = (vc.num_position_embeddings, vh)
        for i in range(vc.depth):
            b = f"model.visual.blocks.{i}."
            shapes.update({b + "norm1.weight": (vh,), b + "norm1.bias": (vh,), b + "norm2.weight": (vh,), b + "norm2.bias": (vh,),
                           b + "attn.qkv.weight": (3 * vh, vh), b + "attn.qkv.bias": (3 * vh,), b + "attn.proj.weight": (vh, vh),
                           b + "attn.proj.bias": (vh,), b + "mlp.linear_fc1.weight": (vi, vh), b + "mlp.linear_fc1.bias": (vi,),
                           b + "mlp.linear_fc2.weight": (vh, vi), b + "mlp.linear_fc2.bias": (vh,)})
        for m, post in [("model.visual.merger.", False)] + [(f"model.visual.deepstack_merger_list.{k}.", True)
                                                            for k in range(len(vc.deepstack_visual_indexes))]:
            shapes.update({m + "norm.weight": (mh if post else vh,), m + "norm.bias": (mh if post else vh,),
                           m + "linear_fc1.weight": (mh, mh), m + "linear_fc1.bias": (mh,),
                           m + "linear_fc2.weight": (vc.out_hidden_size, mh), m + "linear_fc2.bias": (vc.out_hidden_size,)})
    for i in range(0 if q3 else vc.depth):
        b = f"model.visual.blocks.{i}."
        shapes.update({b + "norm1.weight": (vh,), b + "norm2.weight": (vh,), b + "attn.qkv.weight": (3 * vh, vh),
                       b + "attn.qkv.bias": (3 * vh,), b + "attn.proj.weight": (vh, vh), b + "attn.proj.bias": (vh,),
                       b + "mlp.gate_proj.weight": (vi, vh), b + "mlp.gate_proj.bias": (vi,), b + "mlp.up_proj.weight": (vi, vh),
                       b + "mlp.up_proj.bias": (vi,), b + "mlp.down_proj.weight": (vh, vi), b + "mlp.down_proj.bias": (vh,)})
    m = "model.visual.merger."
    if not q3:
        shapes.update({m + "ln_q.weight": (vh,), m + "mlp.0.weight": (mh, mh), m + "mlp.0.bias": (mh,),
                       m + "mlp.2.weight": (vc.out_hidden_size, mh), m + "mlp.2.bias": (vc.out_hidden_size,)})
    H, nh, nkv, D, I, V = (tc.hidden_size, tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim,
                           tc.intermediate_size, tc.vocab_size)
    shapes["model.language_model.embed_tokens.weight"] = (V, H)
    for i in range(tc.num_hidden_layers):
        b = f"model.language_model.layers.{i}."
        shapes.update({b + "input_layernorm.weight": (H,), b + "post_attention_layernorm.weight": (H,),
                       b + "self_attn.q_proj.weight": (nh * D, H), b + "self_attn.q_proj.bias": (nh * D,),
                       b + "self_attn.k_proj.weight": (nkv * D, H), b + "self_attn.k_proj.bias": (nkv * D,),
                       b + "self_attn.v_proj.weight": (nkv * D, H), b + "self_attn.v_proj.bias": (nkv * D,),
                       b + "self_attn.o_proj.weight": (H, nh * D), b + "mlp.gate_proj.weight": (I, H),
                       b + "mlp.up_proj.weight": (I, H), b + "mlp.down_proj.weight": (H, I)})
        if tc.qk_norm:
            shapes.update({b + "self_attn.q_norm.weight": (D,), b + "self_attn.k_norm.weight": (D,)})
    shapes["model.language_model.norm.weight"] = (H,)
    shapes["lm_head.weight"] = (V, H)

    def get(name):
        shape = shapes[name]
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("layernorm.weight") \
                or name.endswith("norm.weight") or name.endswith("ln_q.weight"):
            return torch.ones(shape, device=device, dtype=torch.bfloat16)
        if name.endswith("norm1.bias") or name.endswith("norm2.bias") or name.endswith("norm.bias"):
            return torch.zeros(shape, device=device, dtype=torch.bfloat16)
        t = torch.empty(shape, device=device, dtype=torch.bfloat16)
        t.normal_(0.0, head_std if (head_std is not None and name == "lm_head.weight") else std, generator=gen)
        return t
    return get
