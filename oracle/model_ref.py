"""Oracle: Qwen2.5-VL forward + generate, restated op-by-op on torch-CPU.

No ``transformers`` import.  Each function cites the TF (transformers 5.15.0) line it
restates; the op order and every rounding point (where a tensor is materialised in the
model dtype) follow TF exactly, so running this in bf16 reproduces the HF CPU bf16 path
and running it in fp32 gives the tolerance reference for the HIP kernels.
Test infrastructure only (see oracle/__init__.py).

Weights: dict name -> tensor with HF 5.x names (``model.visual.*``,
``model.language_model.*``, ``lm_head.weight``).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import index_ref


# ----------------------------------------------------------------------------- helpers
def rmsnorm(x, w, eps):
    # TF:modeling_qwen2_5_vl.py:74-79
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rotate_half(x):
    # TF:152-156
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _sdpa_eager(q, k, v, scale, mask=None):
    # TF:186-208 eager_attention_forward (q,k,v: [B,H,L,D]; kv already repeated)
    w = torch.matmul(q, k.transpose(2, 3)) * scale
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(w, v)


# ----------------------------------------------------------------------------- vision
def vit_forward(W, cfg, pixel_values, grid_thw, dtype=torch.float32, taps=None):
    """TF:modeling_qwen2_5_vl.py:408-471.  pixel_values [P, C*2*14*14] f32; returns
    (last_hidden [P,hid] in window order, merged [P/4,out] in original order)."""
    vc = cfg["vision_config"]
    hid, heads = vc["hidden_size"], vc["num_heads"]
    hd = hid // heads
    merge = vc["spatial_merge_size"]
    unit = merge * merge
    p = "model.visual."
    grid = [tuple(int(v) for v in g) for g in np.asarray(grid_thw)]

    pos_ids = index_ref.vision_position_ids(grid, merge)
    cu_full = index_ref.vision_cu_seqlens(grid)
    win_idx, cu_win = index_ref.vision_window_index(grid, merge, vc["window_size"], vc["patch_size"])

    # patch embed: Conv3d(k=s) == matmul over the flattened patch (TF:116-122)
    wpe = W[p + "patch_embed.proj.weight"].to(dtype)
    ks = tuple(wpe.shape[2:])
    x = pixel_values.to(dtype).view(-1, wpe.shape[1], *ks)
    x = F.conv3d(x, wpe, stride=ks).view(-1, hid)  # stride == kernel: one dot product per patch
    if taps is not None:
        taps["patch_embed"] = x.clone()
    P = x.shape[0]
    wi = torch.from_numpy(win_idx)
    x = x.reshape(P // unit, unit, -1)[wi].reshape(P, -1)  # TF:436-439

    # rotary table (TF:125-134, :441-446): dim = hd//2, 2 axes -> hd/2 freqs, cat twice
    rdim = hd // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rdim, 2, dtype=torch.float) / rdim))
    pid = torch.from_numpy(pos_ids)
    rot = (pid.unsqueeze(-1) * inv_freq).flatten(1)  # [P, hd/2]
    rot = rot.reshape(P // unit, unit, -1)[wi].reshape(P, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos(), emb.sin()

    full = set(vc["fullatt_block_indexes"])
    for li in range(vc["depth"]):
        b = f"{p}blocks.{li}."
        cu = cu_full if li in full else cu_win
        h = rmsnorm(x, W[b + "norm1.weight"].to(dtype), 1e-6)
        qkv = F.linear(h, W[b + "attn.qkv.weight"].to(dtype), W[b + "attn.qkv.bias"].to(dtype))
        q, k, v = qkv.reshape(P, 3, heads, hd).permute(1, 0, 2, 3).unbind(0)
        # TF:160-171 rope in fp32, cast back
        qf, kf = q.float(), k.float()
        c, s = cos.unsqueeze(-2).float(), sin.unsqueeze(-2).float()
        q = ((qf * c) + (rotate_half(qf) * s)).to(dtype)
        k = ((kf * c) + (rotate_half(kf) * s)).to(dtype)
        q, k, v = (t.transpose(0, 1).unsqueeze(0) for t in (q, k, v))  # [1,H,P,D]
        outs = []
        for a, e in zip(cu[:-1], cu[1:]):  # TF:268-287 per-segment eager attention
            o = _sdpa_eager(q[:, :, a:e], k[:, :, a:e], v[:, :, a:e], hd ** -0.5)
            outs.append(o.transpose(1, 2))
        a_out = torch.cat(outs, dim=1).reshape(P, -1)
        a_out = F.linear(a_out, W[b + "attn.proj.weight"].to(dtype), W[b + "attn.proj.bias"].to(dtype))
        x = x + a_out
        h = rmsnorm(x, W[b + "norm2.weight"].to(dtype), 1e-6)
        g = F.linear(h, W[b + "mlp.gate_proj.weight"].to(dtype), W[b + "mlp.gate_proj.bias"].to(dtype))
        u = F.linear(h, W[b + "mlp.up_proj.weight"].to(dtype), W[b + "mlp.up_proj.bias"].to(dtype))
        m = F.linear(F.silu(g) * u, W[b + "mlp.down_proj.weight"].to(dtype), W[b + "mlp.down_proj.bias"].to(dtype))
        x = x + m
        if taps is not None:
            taps[f"vit_block_{li}"] = x.clone()

    # merger TF:137-150 then un-permute TF:464-466
    h = rmsnorm(x, W[p + "merger.ln_q.weight"].to(dtype), 1e-6).view(-1, hid * unit)
    h = F.linear(h, W[p + "merger.mlp.0.weight"].to(dtype), W[p + "merger.mlp.0.bias"].to(dtype))
    h = F.gelu(h)
    h = F.linear(h, W[p + "merger.mlp.2.weight"].to(dtype), W[p + "merger.mlp.2.bias"].to(dtype))
    rev = torch.argsort(wi)
    return x, h[rev]


# ----------------------------------------------------------------------------- text
def mrope_cos_sin(cfg, position_ids, dtype):
    """TF:525-538 + section select of TF:590-596.  position_ids [3,B,S] -> cos,sin [B,S,D]."""
    tc = cfg["text_config"]
    hd = tc["hidden_size"] // tc["num_attention_heads"]
    theta = tc["rope_theta"]
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    pos = position_ids.float()  # [3,B,S]
    inv = inv_freq[None, None, :, None].expand(3, pos.shape[1], -1, 1)
    freqs = (inv @ pos[:, :, None, :]).transpose(2, 3)  # [3,B,S,hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(dtype), emb.sin().to(dtype)
    sec = list(tc["mrope_section"]) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


class KVCache:
    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def update(self, li, k, v):
        if self.k[li] is None:
            self.k[li], self.v[li] = k, v
        else:
            self.k[li] = torch.cat([self.k[li], k], dim=2)
            self.v[li] = torch.cat([self.v[li], v], dim=2)
        return self.k[li], self.v[li]

    def length(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def text_forward(W, cfg, inputs_embeds, position_ids, attention_mask_2d, cache, dtype, taps=None, quant=None):
    """TF:790-872 language model + TF:602-757 layers.  inputs_embeds [B,L,H];
    position_ids [3,B,L]; attention_mask_2d [B, past+L] (1 = real token).
    quant="w8a8": the four linears of every layer through oracle/quant_ref.linear_w8a8 (the opt-in fp8 prefill)."""
    if quant == "w8a8":
        from . import quant_ref

        def lin(x, w, b=None):
            return quant_ref.linear_w8a8(x, w, b)
    else:
        def lin(x, w, b=None):
            return F.linear(x, w, b)
    tc = cfg["text_config"]
    H, nh, nkv = tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"]
    hd = H // nh
    rep = nh // nkv
    eps = tc["rms_norm_eps"]
    B, L, _ = inputs_embeds.shape
    past = cache.length()
    T = past + L
    cos, sin = mrope_cos_sin(cfg, position_ids, dtype)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    # causal + padding mask (TF:836-851), additive, min-value style
    neg = torch.finfo(dtype).min
    qpos = torch.arange(past, T).view(L, 1)
    kpos = torch.arange(T).view(1, T)
    allowed = (kpos <= qpos).unsqueeze(0).expand(B, L, T)
    if attention_mask_2d is not None:
        allowed = allowed & attention_mask_2d.bool()[:, None, :T]
    mask = torch.zeros(B, 1, L, T, dtype=dtype).masked_fill(~allowed.unsqueeze(1), neg)
    x = inputs_embeds
    p = "model.language_model."
    for li in range(tc["num_hidden_layers"]):
        b = f"{p}layers.{li}."
        res = x
        h = rmsnorm(x, W[b + "input_layernorm.weight"].to(dtype), eps)
        q = lin(h, W[b + "self_attn.q_proj.weight"].to(dtype), W[b + "self_attn.q_proj.bias"].to(dtype))
        k = lin(h, W[b + "self_attn.k_proj.weight"].to(dtype), W[b + "self_attn.k_proj.bias"].to(dtype))
        v = lin(h, W[b + "self_attn.v_proj.weight"].to(dtype), W[b + "self_attn.v_proj.bias"].to(dtype))
        q = q.view(B, L, nh, hd).transpose(1, 2)
        k = k.view(B, L, nkv, hd).transpose(1, 2)
        v = v.view(B, L, nkv, hd).transpose(1, 2)
        q = (q * cos) + (rotate_half(q) * sin)  # TF:598-599 in model dtype
        k = (k * cos) + (rotate_half(k) * sin)
        k, v = cache.update(li, k, v)
        kr = k[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)  # TF:174-183
        vr = v[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        a = _sdpa_eager(q, kr, vr, hd ** -0.5, mask)
        a = a.transpose(1, 2).reshape(B, L, -1)
        a = lin(a, W[b + "self_attn.o_proj.weight"].to(dtype))
        x = res + a
        res = x
        h = rmsnorm(x, W[b + "post_attention_layernorm.weight"].to(dtype), eps)
        g = lin(h, W[b + "mlp.gate_proj.weight"].to(dtype))
        u = lin(h, W[b + "mlp.up_proj.weight"].to(dtype))
        m = lin(F.silu(g) * u, W[b + "mlp.down_proj.weight"].to(dtype))
        x = res + m
        if taps is not None:
            taps[f"llm_layer_{li}_p{past}"] = x.clone()
    return rmsnorm(x, W[p + "norm.weight"].to(dtype), eps)


def lm_head_weight(W, cfg):
    if cfg["text_config"].get("tie_word_embeddings", cfg.get("tie_word_embeddings", False)) or "lm_head.weight" not in W:
        return W["model.language_model.embed_tokens.weight"]
    return W["lm_head.weight"]


def embed_with_vision(W, cfg, input_ids, pixel_values, image_grid_thw, dtype, taps=None, pixel_values_videos=None,
                      video_grid_thw=None):
    """TF:1206-1215: embedding gather, then masked_scatter of the merged image tokens at the image placeholders and of the
    merged video tokens at the video placeholders (each modality runs the tower on its own rows, TF:1060-1092)."""
    emb = W["model.language_model.embed_tokens.weight"].to(dtype)
    x = emb[input_ids]
    for pv, grid, tok, tag in ((pixel_values, image_grid_thw, cfg["image_token_id"], "vit_merged"),
                               (pixel_values_videos, video_grid_thw, cfg["video_token_id"], "vit_merged_video")):
        if pv is None:
            continue
        _, vis = vit_forward(W, cfg, pv, grid, dtype, taps)
        if taps is not None:
            taps[tag] = vis.clone()
        m = input_ids == tok
        assert int(m.sum()) == vis.shape[0], "visual features and placeholder tokens do not match"
        x = x.clone()
        x[m] = vis.to(dtype)
    return x


def _positions(cfg, input_ids, attention_mask, image_grid_thw, video_grid_thw, second_per_grid_ts):
    """3-D positions + rope deltas of a prompt (TF:1135-1181 compute_3d_position_ids over get_rope_index)."""
    B = input_ids.shape[0]
    if image_grid_thw is None and video_grid_thw is None:
        p1 = (attention_mask.cumsum(-1) - 1).masked_fill(attention_mask == 0, 0)
        return p1.unsqueeze(0).expand(3, -1, -1).contiguous(), torch.zeros(B, 1, dtype=torch.long)
    types = index_ref.token_types(input_ids.numpy(), cfg["image_token_id"], cfg["video_token_id"])
    pos, deltas = index_ref.rope_index(input_ids.numpy(), types, image_grid_thw, attention_mask.numpy(),
                                       cfg["vision_config"]["spatial_merge_size"], video_grid_thw=video_grid_thw,
                                       second_per_grid_ts=second_per_grid_ts,
                                       tokens_per_second=cfg["vision_config"].get("tokens_per_second", 1))
    return torch.from_numpy(pos), torch.from_numpy(deltas)


# ----------------------------------------------------------------------------- logits processors
def repetition_penalty(scores, seen_ids, penalty):
    # TF:generation/logits_process.py:404-414 ; seen_ids [B, n] (prompt + generated)
    g = torch.gather(scores, 1, seen_ids)
    g = torch.where(g < 0, g * penalty, g / penalty)
    return scores.scatter(1, seen_ids, g)


def temperature_warp(scores, t):
    # TF:generation/logits_process.py:301-303
    return scores / t


def top_k_warp(scores, top_k, filter_value=-float("inf")):
    # TF:generation/logits_process.py:590-594 (TopKLogitsWarper): everything below the k-th largest score goes; ties stay
    top_k = min(int(top_k), scores.shape[-1])
    kth = torch.topk(scores, top_k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, filter_value)


def top_p_warp(scores, top_p, min_keep=1, filter_value=-float("inf")):
    # TF:generation/logits_process.py:527-539
    sl, si = torch.sort(scores, descending=False)
    cp = sl.softmax(dim=-1).cumsum(dim=-1)
    rm = cp <= (1 - top_p)
    rm[..., -min_keep:] = 0
    rm = rm.scatter(1, si, rm)
    return scores.masked_fill(rm, filter_value)


# ----------------------------------------------------------------------------- generate
def generate(W, cfg, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens,
             dtype=torch.float32, eos_token_ids=(), pad_token_id=0, rep_penalty=1.0,
             do_sample=False, temperature=1.0, top_p=1.0, generator=None, taps=None,
             return_logits=False, top_k=0, pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None):
    """Greedy / sampled decode, TF:generation/utils.py:2783-2942 (_sample) over
    TF:modeling_qwen2_5_vl.py:1185-1253,1308-1402.  Returns ids [B, S+T] (and the fp32
    last-token logits per step when return_logits)."""
    input_ids = torch.as_tensor(input_ids, dtype=torch.long)
    B, S = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    attention_mask = torch.as_tensor(attention_mask, dtype=torch.long)
    pos, deltas = _positions(cfg, input_ids, attention_mask, None if pixel_values is None else image_grid_thw,
                             None if pixel_values_videos is None else video_grid_thw, second_per_grid_ts)
    tc = cfg["text_config"]
    cache = KVCache(tc["num_hidden_layers"])
    x = embed_with_vision(W, cfg, input_ids, pixel_values, image_grid_thw, dtype, taps, pixel_values_videos, video_grid_thw)
    if taps is not None:
        taps["inputs_embeds"] = x.clone()
        taps["position_ids"] = pos.clone()
        taps["rope_deltas"] = deltas.clone()
    h = text_forward(W, cfg, x, pos, attention_mask, cache, dtype, taps)
    head = lm_head_weight(W, cfg).to(dtype)
    ids = input_ids.clone()
    mask = attention_mask.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    eos = torch.tensor(list(eos_token_ids), dtype=torch.long)
    step_logits = []
    for step in range(max_new_tokens):
        logits = F.linear(h[:, -1:, :], head)[:, -1, :].to(torch.float32)  # TF:utils.py:2894
        if return_logits:
            step_logits.append(logits.clone())
        scores = logits
        if rep_penalty != 1.0:
            scores = repetition_penalty(scores, ids, rep_penalty)
        if do_sample:
            if temperature != 1.0:
                scores = temperature_warp(scores, temperature)
            if top_k:
                scores = top_k_warp(scores, top_k)
            if top_p < 1.0:
                scores = top_p_warp(scores, top_p)
            probs = F.softmax(scores, dim=-1)
            nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
        else:
            nxt = torch.argmax(scores, dim=-1)
        if len(eos):
            nxt = nxt * unfinished + pad_token_id * (1 - unfinished)  # TF:utils.py:2929
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], dim=-1)
        if len(eos):
            unfinished = unfinished & ~torch.isin(nxt, eos)
            if unfinished.max() == 0:
                break
        if step == max_new_tokens - 1:
            break
        # decode positions: TF:1164-1174  (cumsum(mask)-1 at the new column) + delta
        p1 = (mask.cumsum(-1) - 1)[:, -1:]
        dpos = (p1 + deltas).unsqueeze(0).expand(3, -1, -1)
        xe = W["model.language_model.embed_tokens.weight"].to(dtype)[nxt][:, None, :]
        h = text_forward(W, cfg, xe, dpos, mask, cache, dtype, None)
    if return_logits:
        return ids, torch.stack(step_logits, dim=1)
    return ids


def full_logits(W, cfg, input_ids, attention_mask, pixel_values, image_grid_thw, dtype=torch.float32, pixel_values_videos=None,
                video_grid_thw=None, second_per_grid_ts=None, quant=None):
    """model(input_ids, ...).logits [B,L,V] as R:grpo_trainer.py:375 calls it."""
    input_ids = torch.as_tensor(input_ids, dtype=torch.long)
    B, S = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    attention_mask = torch.as_tensor(attention_mask, dtype=torch.long)
    pos, _ = _positions(cfg, input_ids, attention_mask, None if pixel_values is None else image_grid_thw,
                        None if pixel_values_videos is None else video_grid_thw, second_per_grid_ts)
    cache = KVCache(cfg["text_config"]["num_hidden_layers"])
    x = embed_with_vision(W, cfg, input_ids, pixel_values, image_grid_thw, dtype, None, pixel_values_videos, video_grid_thw)
    h = text_forward(W, cfg, x, pos, attention_mask, cache, dtype, quant=quant)
    return F.linear(h, lm_head_weight(W, cfg).to(dtype))


def per_token_logps(logits, input_ids):
    """R:grpo_trainer.py:371-384."""
    logits = logits[:, :-1, :]
    ids = torch.as_tensor(input_ids)[:, 1:]
    out = []
    for lr, ir in zip(logits, ids):
        lp = lr.log_softmax(dim=-1)
        out.append(torch.gather(lp, 1, ir.unsqueeze(1)).squeeze(1))
    return torch.stack(out)
