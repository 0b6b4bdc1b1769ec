"""Oracle for the Qwen3-VL family (BASELINE config #5's base model): forward + generate restated op-by-op on torch-CPU.

No ``transformers`` import.  Cites TF3 = transformers 5.15.0 ``models/qwen3_vl/modeling_qwen3_vl.py`` (third-party; the
reference uses Qwen3-VL-8B through vLLM, R:README.md:29,37, R:eval/test/test_videomme.py:129-226).  What differs from the
Qwen2.5-VL oracle (model_ref.py): patch 16 with a Conv3d bias, a learned position table resampled bilinearly (align_corners)
to every image grid, LayerNorm + GELU-tanh ViT blocks with a plain two-layer MLP and NO window attention (one segment per
temporal patch), DeepStack mergers on selected ViT blocks whose outputs are added to the LLM hidden states after the first
decoder layers at the visual positions, interleaved M-RoPE, RMSNorm on every q / k head before the rotation, no q/k/v bias.
Test infrastructure only (see oracle/__init__.py)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import index_ref
from .model_ref import KVCache, _sdpa_eager, repetition_penalty, rmsnorm, rotate_half  # noqa: F401


# ----------------------------------------------------------------------------- indices
def pos_embed_taps(grid, side, merge):
    """TF:vision_utils.get_vision_interpolation_indices_and_weights (bilinear, align_corners=True) in merge-block order:
    per patch the 4 table rows and weights.  grid: [(t,h,w)] ; returns (idx int64 [P,4], w float32 [P,4])."""
    idx_all, w_all = [], []
    for (t, h, w) in grid:
        def axis(n):
            # TF:vision_utils._interpolation_axis_taps_weights, bilinear, align_corners: the same fp32 operation order
            i = torch.arange(n, dtype=torch.float32)
            src = i * (side - 1) / max(n - 1, 1)
            fl = torch.floor(src)
            lo = fl.long().clamp(0, side - 1)
            hi = (fl.long() + 1).clamp(0, side - 1)
            w_lo = (1 - (src - fl).abs()).clamp(min=0)
            w_hi = (1 - (src - fl - 1).abs()).clamp(min=0)
            return lo, hi, (w_lo, w_hi)
        rlo, rhi, rf = axis(h)
        clo, chi, cf = axis(w)
        # merge-block order of the (row, col) pairs of one frame
        rows = torch.arange(h).view(h // merge, merge, 1, 1).expand(h // merge, merge, w // merge, merge)
        cols = torch.arange(w).view(1, 1, w // merge, merge).expand(h // merge, merge, w // merge, merge)
        rows = rows.permute(0, 2, 1, 3).reshape(-1)
        cols = cols.permute(0, 2, 1, 3).reshape(-1)
        idx = torch.stack([rlo[rows] * side + clo[cols], rlo[rows] * side + chi[cols],
                           rhi[rows] * side + clo[cols], rhi[rows] * side + chi[cols]], dim=1)
        wt = torch.stack([rf[0][rows] * cf[0][cols], rf[0][rows] * cf[1][cols], rf[1][rows] * cf[0][cols], rf[1][rows] * cf[1][cols]], dim=1)
        idx_all.append(idx.repeat(t, 1))
        w_all.append(wt.repeat(t, 1))
    return torch.cat(idx_all), torch.cat(w_all)


def interleaved_mrope_cos_sin(cfg, position_ids, dtype):
    """TF3:352-387: freqs of the three axes, then frequency j of the H (W) axis replaces T's at j = 1 (2) mod 3 below
    3*section; cos / sin in fp32, cast to the model dtype."""
    tc = cfg["text_config"]
    hd = tc["head_dim"]
    inv_freq = 1.0 / (tc["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    pos = position_ids.float()
    inv = inv_freq[None, None, :, None].expand(3, pos.shape[1], -1, 1)
    freqs = (inv @ pos[:, :, None, :]).transpose(2, 3)          # [3,B,S,hd/2]
    ft = freqs[0].clone()
    sec = tc["mrope_section"]
    for dim, offset in enumerate((1, 2), start=1):
        idx = slice(offset, sec[dim] * 3, 3)
        ft[..., idx] = freqs[dim][..., idx]
    emb = torch.cat((ft, ft), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


# ----------------------------------------------------------------------------- vision
def gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def _merger(W, prefix, x, hid_u, post, dtype):
    """TF3:122-135 Qwen3VLVisionPatchMerger."""
    nw, nb = W[prefix + "norm.weight"].to(dtype), W[prefix + "norm.bias"].to(dtype)
    if post:
        h = F.layer_norm(x.view(-1, hid_u), (hid_u,), nw, nb, 1e-6)
    else:
        h = F.layer_norm(x, (x.shape[-1],), nw, nb, 1e-6).view(-1, hid_u)
    h = F.linear(h, W[prefix + "linear_fc1.weight"].to(dtype), W[prefix + "linear_fc1.bias"].to(dtype))
    h = F.gelu(h)
    return F.linear(h, W[prefix + "linear_fc2.weight"].to(dtype), W[prefix + "linear_fc2.bias"].to(dtype))


def vit_forward(W, cfg, pixel_values, grid_thw, dtype=torch.float32, taps=None):
    """TF3:606-737.  pixel_values [P, C*2*16*16]; returns (last_hidden [P,hid], merged [P/4,out], [deepstack [P/4,out]...])."""
    vc = cfg["vision_config"]
    hid, heads = vc["hidden_size"], vc["num_heads"]
    hd = hid // heads
    merge = vc["spatial_merge_size"]
    unit = merge * merge
    p = "model.visual."
    grid = [tuple(int(v) for v in g) for g in np.asarray(grid_thw)]
    side = int(vc["num_position_embeddings"] ** 0.5)
    wpe = W[p + "patch_embed.proj.weight"].to(dtype)
    ks = tuple(wpe.shape[2:])
    x = pixel_values.to(dtype).view(-1, wpe.shape[1], *ks)
    x = F.conv3d(x, wpe, W[p + "patch_embed.proj.bias"].to(dtype), stride=ks).view(-1, hid)
    idx, wt = pos_embed_taps(grid, side, merge)
    pe = (W[p + "pos_embed.weight"].to(dtype)[idx] * wt[:, :, None]).sum(1)       # TF3:706: table dtype x fp32 weights -> fp32 sum
    x = x + pe.to(x.dtype)
    if taps is not None:
        taps["patch_pos"] = x.clone()
    P = x.shape[0]
    pos_ids = index_ref.vision_position_ids(grid, merge)
    rdim = hd // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rdim, 2, dtype=torch.float) / rdim))
    rot = (torch.from_numpy(pos_ids).unsqueeze(-1) * inv_freq).flatten(1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    cu = index_ref.vision_cu_seqlens(grid)          # one segment per temporal patch (TF:vision_utils.get_vision_attention_seqlens)
    deep, deep_idx = [], list(vc["deepstack_visual_indexes"])
    for li in range(vc["depth"]):
        b = f"{p}blocks.{li}."
        h = F.layer_norm(x, (hid,), W[b + "norm1.weight"].to(dtype), W[b + "norm1.bias"].to(dtype), 1e-6)
        qkv = F.linear(h, W[b + "attn.qkv.weight"].to(dtype), W[b + "attn.qkv.bias"].to(dtype))
        q, k, v = qkv.reshape(P, 3, heads, hd).permute(1, 0, 2, 3).unbind(0)
        qf, kf = q.float(), k.float()                  # TF3:145-156 rope in fp32, cast back
        c, s = cos.unsqueeze(-2).float(), sin.unsqueeze(-2).float()
        q = ((qf * c) + (rotate_half(qf) * s)).to(dtype)
        k = ((kf * c) + (rotate_half(kf) * s)).to(dtype)
        q, k, v = (t.transpose(0, 1).unsqueeze(0) for t in (q, k, v))
        outs = []
        for a, e in zip(cu[:-1], cu[1:]):
            outs.append(_sdpa_eager(q[:, :, a:e], k[:, :, a:e], v[:, :, a:e], hd ** -0.5).transpose(1, 2))
        a_out = torch.cat(outs, dim=1).reshape(P, -1)
        x = x + F.linear(a_out, W[b + "attn.proj.weight"].to(dtype), W[b + "attn.proj.bias"].to(dtype))
        h = F.layer_norm(x, (hid,), W[b + "norm2.weight"].to(dtype), W[b + "norm2.bias"].to(dtype), 1e-6)
        m = F.linear(gelu_tanh(F.linear(h, W[b + "mlp.linear_fc1.weight"].to(dtype), W[b + "mlp.linear_fc1.bias"].to(dtype))),
                     W[b + "mlp.linear_fc2.weight"].to(dtype), W[b + "mlp.linear_fc2.bias"].to(dtype))
        x = x + m
        if li in deep_idx:
            deep.append(_merger(W, f"{p}deepstack_merger_list.{deep_idx.index(li)}.", x, hid * unit, True, dtype))
    merged = _merger(W, p + "merger.", x, hid * unit, False, dtype)
    return x, merged, deep


# ----------------------------------------------------------------------------- text
def text_forward(W, cfg, inputs_embeds, position_ids, attention_mask_2d, cache, dtype, visual_mask=None, deepstack=None):
    """TF3:746-862 + :438-566.  inputs_embeds [B,L,H]; position_ids [3,B,L]; visual_mask bool [B,L] and deepstack
    (list of [n_visual, H]) only on the prefill call."""
    tc = cfg["text_config"]
    H, nh, nkv, hd = tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"], tc["head_dim"]
    rep, eps = nh // nkv, tc["rms_norm_eps"]
    B, L, _ = inputs_embeds.shape
    past = cache.length()
    T = past + L
    cos, sin = interleaved_mrope_cos_sin(cfg, position_ids, dtype)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
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
        q = F.linear(h, W[b + "self_attn.q_proj.weight"].to(dtype)).view(B, L, nh, hd)
        k = F.linear(h, W[b + "self_attn.k_proj.weight"].to(dtype)).view(B, L, nkv, hd)
        v = F.linear(h, W[b + "self_attn.v_proj.weight"].to(dtype)).view(B, L, nkv, hd).transpose(1, 2)
        q = rmsnorm(q, W[b + "self_attn.q_norm.weight"].to(dtype), eps).transpose(1, 2)      # TF3:480-481
        k = rmsnorm(k, W[b + "self_attn.k_norm.weight"].to(dtype), eps).transpose(1, 2)
        q = (q * cos) + (rotate_half(q) * sin)
        k = (k * cos) + (rotate_half(k) * sin)
        k, v = cache.update(li, k, v)
        kr = k[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        vr = v[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        a = _sdpa_eager(q, kr, vr, hd ** -0.5, mask).transpose(1, 2).reshape(B, L, -1)
        x = res + F.linear(a, W[b + "self_attn.o_proj.weight"].to(dtype))
        res = x
        h = rmsnorm(x, W[b + "post_attention_layernorm.weight"].to(dtype), eps)
        m = F.linear(F.silu(F.linear(h, W[b + "mlp.gate_proj.weight"].to(dtype))) * F.linear(h, W[b + "mlp.up_proj.weight"].to(dtype)),
                     W[b + "mlp.down_proj.weight"].to(dtype))
        x = res + m
        if deepstack is not None and li < len(deepstack):        # TF3:839-862
            x = x.clone()
            x[visual_mask] = x[visual_mask] + deepstack[li].to(x.dtype)
    return rmsnorm(x, W[p + "norm.weight"].to(dtype), eps)


def generate(W, cfg, input_ids, attention_mask, pixel_values, image_grid_thw, max_new_tokens, dtype=torch.float32,
             pad_token_id=0, rep_penalty=1.0, return_logits=False, taps=None, pixel_values_videos=None, video_grid_thw=None):
    """Greedy decode (TF:generation/utils.py _sample over TF3:1260-1580).  Video inputs (TF3:1170-1218): the tower runs on
    the video rows on its own, its tokens go to the <|video_pad|> positions and its DeepStack features to the same
    positions; with images AND videos the DeepStack rows are interleaved in sequence order (TF3:1198-1211).  Positions:
    every frame of a video is its own [1,h,w] group (TF3:966-969)."""
    input_ids = torch.as_tensor(input_ids, dtype=torch.long)
    B, S = input_ids.shape
    attention_mask = torch.ones_like(input_ids) if attention_mask is None else torch.as_tensor(attention_mask, dtype=torch.long)
    merge = cfg["vision_config"]["spatial_merge_size"]
    emb = W["model.language_model.embed_tokens.weight"].to(dtype)
    x = emb[input_ids]
    vmask, deep = None, None
    if pixel_values is not None or pixel_values_videos is not None:
        types = index_ref.token_types(input_ids.numpy(), cfg["image_token_id"], cfg["video_token_id"])
        pos, deltas = index_ref.rope_index(input_ids.numpy(), types, None if pixel_values is None else np.asarray(image_grid_thw),
                                           attention_mask.numpy(), merge,
                                           video_grid_thw=None if pixel_values_videos is None else np.asarray(video_grid_thw),
                                           split_video_frames=True)
        pos, deltas = torch.from_numpy(pos), torch.from_numpy(deltas)
        x = x.clone()
        vmask = torch.zeros_like(input_ids, dtype=torch.bool)
        parts = []
        for pv, grid, tok, tag in ((pixel_values, image_grid_thw, cfg["image_token_id"], ""),
                                   (pixel_values_videos, video_grid_thw, cfg["video_token_id"], "_video")):
            if pv is None:
                continue
            _, vis, dp = vit_forward(W, cfg, pv, grid, dtype, taps)
            if taps is not None:
                taps["vit_merged" + tag] = vis.clone()
                taps["deepstack" + tag] = [d.clone() for d in dp]
            m = input_ids == tok
            assert int(m.sum()) == vis.shape[0]
            x[m] = vis.to(dtype)
            vmask |= m
            parts.append((m, dp))
        if len(parts) == 1:
            deep = parts[0][1]
        else:                                                     # TF3:1198-1211: joint rows in sequence order
            deep = []
            for j in range(len(parts[0][1])):
                joint = parts[0][1][j].new_zeros(int(vmask.sum()), parts[0][1][j].shape[-1])
                for m, dp in parts:
                    joint[m[vmask]] = dp[j]
                deep.append(joint)
    else:
        p1 = (attention_mask.cumsum(-1) - 1).masked_fill(attention_mask == 0, 0)
        pos, deltas = p1.unsqueeze(0).expand(3, -1, -1).contiguous(), torch.zeros(B, 1, dtype=torch.long)
    cache = KVCache(cfg["text_config"]["num_hidden_layers"])
    h = text_forward(W, cfg, x, pos, attention_mask, cache, dtype, vmask, deep)
    head = W["lm_head.weight"].to(dtype)
    ids, mask, step_logits = input_ids.clone(), attention_mask.clone(), []
    for step in range(max_new_tokens):
        logits = F.linear(h[:, -1:, :], head)[:, -1, :].to(torch.float32)
        if return_logits:
            step_logits.append(logits.clone())
        scores = repetition_penalty(logits, ids, rep_penalty) if rep_penalty != 1.0 else logits
        nxt = torch.argmax(scores, dim=-1)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], dim=-1)
        if step == max_new_tokens - 1:
            break
        p1 = (mask.cumsum(-1) - 1)[:, -1:]
        dpos = (p1 + deltas).unsqueeze(0).expand(3, -1, -1)
        h = text_forward(W, cfg, emb[nxt][:, None, :], dpos, mask, cache, dtype)
    if return_logits:
        return ids, torch.stack(step_logits, dim=1)
    return ids
