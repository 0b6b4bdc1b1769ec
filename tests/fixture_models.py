"""Seeded fixture models shared by tools/make_golden.py (HF side) and the tests (our side).

The configs are HF ``config.json``-style dicts; the weights are regenerated from a seed
with torch's CPU generator (bit-stable for a given torch build, and the GPU box runs the
same image), so only inputs / expected outputs are committed under tests/golden/.
"""
from __future__ import annotations

import torch

SPECIAL = dict(image_token_id=500, video_token_id=501, vision_start_token_id=498,
               vision_end_token_id=499, eos_token_id=510, pad_token_id=511, bos_token_id=509)


def tiny_config():
    """ViT head_dim 32, LLM head_dim 32 -- smallest shapes the kernels accept."""
    return {
        "model_type": "qwen2_5_vl",
        **SPECIAL,
        "tie_word_embeddings": False,
        "vision_config": {
            "depth": 4, "hidden_size": 64, "num_heads": 2, "intermediate_size": 96,
            "out_hidden_size": 128, "patch_size": 14, "temporal_patch_size": 2,
            "spatial_merge_size": 2, "window_size": 112, "fullatt_block_indexes": [1, 3],
            "in_channels": 3, "hidden_act": "silu", "tokens_per_second": 2,
        },
        "text_config": {
            "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 4,
            "num_key_value_heads": 2, "intermediate_size": 256, "vocab_size": 512,
            "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "mrope_section": [4, 6, 6],
            "hidden_act": "silu", "max_position_embeddings": 4096, "tie_word_embeddings": False,
        },
    }


def medium_config():
    """True head geometry of Qwen2.5-VL-7B (ViT head_dim 80, LLM head_dim 128, GQA 7:1,
    mrope [16,24,24], odd ViT MLP width that needs padding) with few layers."""
    return {
        "model_type": "qwen2_5_vl",
        **SPECIAL,
        "image_token_id": 4000, "video_token_id": 4001, "vision_start_token_id": 3998,
        "vision_end_token_id": 3999, "eos_token_id": 4010, "pad_token_id": 4011, "bos_token_id": 4009,
        "tie_word_embeddings": False,
        "vision_config": {
            "depth": 4, "hidden_size": 320, "num_heads": 4, "intermediate_size": 428,
            "out_hidden_size": 896, "patch_size": 14, "temporal_patch_size": 2,
            "spatial_merge_size": 2, "window_size": 112, "fullatt_block_indexes": [1, 3],
            "in_channels": 3, "hidden_act": "silu", "tokens_per_second": 2,
        },
        "text_config": {
            "hidden_size": 896, "num_hidden_layers": 3, "num_attention_heads": 7,
            "num_key_value_heads": 1, "intermediate_size": 1152, "vocab_size": 4096,
            "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "mrope_section": [16, 24, 24],
            "hidden_act": "silu", "max_position_embeddings": 8192, "tie_word_embeddings": False,
        },
    }


def full7b_config():
    """Qwen2.5-VL-7B at TRUE depth and width (32 ViT blocks of 1280 / 3420, 28 LLM layers of 3584 / 18944, GQA 28:4,
    mrope [16,24,24]) with a small vocabulary (4096) so that the seeded weights (6.5 G parameters) and
    the HF fp32 run fit the build container.  Golden G10 pins the engine at this depth."""
    return {
        "model_type": "qwen2_5_vl",
        **SPECIAL,
        "image_token_id": 4000, "video_token_id": 4001, "vision_start_token_id": 3998,
        "vision_end_token_id": 3999, "eos_token_id": 4010, "pad_token_id": 4011, "bos_token_id": 4009,
        "tie_word_embeddings": False,
        "vision_config": {
            "depth": 32, "hidden_size": 1280, "num_heads": 16, "intermediate_size": 3420,
            "out_hidden_size": 3584, "patch_size": 14, "temporal_patch_size": 2,
            "spatial_merge_size": 2, "window_size": 112, "fullatt_block_indexes": [7, 15, 23, 31],
            "in_channels": 3, "hidden_act": "silu", "tokens_per_second": 2,
        },
        "text_config": {
            "hidden_size": 3584, "num_hidden_layers": 28, "num_attention_heads": 28,
            "num_key_value_heads": 4, "intermediate_size": 18944, "vocab_size": 4096,
            "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "mrope_section": [16, 24, 24],
            "hidden_act": "silu", "max_position_embeddings": 32768, "tie_word_embeddings": False,
        },
    }


def tied3b_config():
    """Qwen2.5-VL-3B's distinguishing features at fixture size: tied word embeddings (lm_head = embed_tokens), GQA 8:1
    (16 query / 2 kv heads of 128), 36-layer structure cut to 3 layers.  Golden G11."""
    c = medium_config()
    c["tie_word_embeddings"] = True
    c["fixture_embed_scale"] = 0.06      # the embedding doubles as the head: keeps the logits O(1..10)
    c["text_config"] = dict(c["text_config"], hidden_size=2048, num_attention_heads=16, num_key_value_heads=2,
                            intermediate_size=2752, tie_word_embeddings=True)
    c["vision_config"] = dict(c["vision_config"], out_hidden_size=2048)
    return c


def weight_specs(cfg):
    """[(name, shape, kind)] in a fixed order; kind in {linear, norm, bias, embed, head}."""
    vc, tc = cfg["vision_config"], cfg["text_config"]
    vh, vi, vo = vc["hidden_size"], vc["intermediate_size"], vc["out_hidden_size"]
    kpe = vc["in_channels"] * vc["temporal_patch_size"] * vc["patch_size"] ** 2
    unit = vc["spatial_merge_size"] ** 2
    specs = [("model.visual.patch_embed.proj.weight",
              (vh, vc["in_channels"], vc["temporal_patch_size"], vc["patch_size"], vc["patch_size"]), "patch")]
    for i in range(vc["depth"]):
        b = f"model.visual.blocks.{i}."
        specs += [(b + "norm1.weight", (vh,), "norm"), (b + "norm2.weight", (vh,), "norm"),
                  (b + "attn.qkv.weight", (3 * vh, vh), "linear"), (b + "attn.qkv.bias", (3 * vh,), "bias"),
                  (b + "attn.proj.weight", (vh, vh), "linear"), (b + "attn.proj.bias", (vh,), "bias"),
                  (b + "mlp.gate_proj.weight", (vi, vh), "linear"), (b + "mlp.gate_proj.bias", (vi,), "bias"),
                  (b + "mlp.up_proj.weight", (vi, vh), "linear"), (b + "mlp.up_proj.bias", (vi,), "bias"),
                  (b + "mlp.down_proj.weight", (vh, vi), "linear"), (b + "mlp.down_proj.bias", (vh,), "bias")]
    m = "model.visual.merger."
    specs += [(m + "ln_q.weight", (vh,), "norm"),
              (m + "mlp.0.weight", (vh * unit, vh * unit), "linear"), (m + "mlp.0.bias", (vh * unit,), "bias"),
              (m + "mlp.2.weight", (vo, vh * unit), "linear"), (m + "mlp.2.bias", (vo,), "bias")]
    H, nh, nkv, I, V = (tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"],
                        tc["intermediate_size"], tc["vocab_size"])
    hd = H // nh
    specs.append(("model.language_model.embed_tokens.weight", (V, H), "embed"))
    for i in range(tc["num_hidden_layers"]):
        b = f"model.language_model.layers.{i}."
        specs += [(b + "input_layernorm.weight", (H,), "norm"),
                  (b + "post_attention_layernorm.weight", (H,), "norm"),
                  (b + "self_attn.q_proj.weight", (nh * hd, H), "linear"), (b + "self_attn.q_proj.bias", (nh * hd,), "bias"),
                  (b + "self_attn.k_proj.weight", (nkv * hd, H), "linear"), (b + "self_attn.k_proj.bias", (nkv * hd,), "bias"),
                  (b + "self_attn.v_proj.weight", (nkv * hd, H), "linear"), (b + "self_attn.v_proj.bias", (nkv * hd,), "bias"),
                  (b + "self_attn.o_proj.weight", (H, nh * hd), "linear"),
                  (b + "mlp.gate_proj.weight", (I, H), "linear"), (b + "mlp.up_proj.weight", (I, H), "linear"),
                  (b + "mlp.down_proj.weight", (H, I), "linear")]
    specs.append(("model.language_model.norm.weight", (H,), "norm"))
    if not cfg.get("tie_word_embeddings", False):
        specs.append(("lm_head.weight", (V, H), "head"))
    return specs


def iter_weights(cfg, seed=0, dtype=torch.float32):
    """(name, tensor) in weight_specs order, one tensor alive at a time (the full-depth fixture is 26 GB in fp32)."""
    g = torch.Generator().manual_seed(seed)
    for name, shape, kind in weight_specs(cfg):
        if kind == "norm":
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "bias":
            w = 0.05 * torch.randn(shape, generator=g)
        elif kind == "embed":
            w = cfg.get("fixture_embed_scale", 0.5) * torch.randn(shape, generator=g)
        elif kind == "head":
            w = (4.0 / shape[1] ** 0.5) * torch.randn(shape, generator=g)
        elif kind == "patch":
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            w = (1.0 / fan_in ** 0.5) * torch.randn(shape, generator=g)
        else:
            w = (1.0 / shape[1] ** 0.5) * torch.randn(shape, generator=g)
        # round through bf16 so the fp32 and bf16 runs see identical parameter values
        yield name, w.to(torch.bfloat16).to(dtype)


def make_weights(cfg, seed=0, dtype=torch.float32):
    """Deterministic weights.  Scales chosen so activations stay O(1) and logits are peaky
    enough for stable argmax (top-1/top-2 margins are recorded with the goldens)."""
    return dict(iter_weights(cfg, seed, dtype))


def make_prompt(cfg, grids, n_text_pre=5, n_text_mid=3, n_text_post=6, seed=0):
    """Synthetic frames-as-images prompt: text, then per frame [text.., <vs>, pad*, <ve>], text."""
    g = torch.Generator().manual_seed(1000 + seed)
    V = cfg["text_config"]["vocab_size"]
    lo, hi = 10, min(V, cfg["vision_start_token_id"]) - 1
    merge = cfg["vision_config"]["spatial_merge_size"]

    def text(n):
        return torch.randint(lo, hi, (n,), generator=g).tolist()

    ids = text(n_text_pre)
    for (t, h, w) in grids:
        ids += text(n_text_mid) + [cfg["vision_start_token_id"]]
        ids += [cfg["image_token_id"]] * (t * h * w // (merge * merge))
        ids += [cfg["vision_end_token_id"]]
    ids += text(n_text_post)
    return ids


def make_prompt_mm(cfg, items, n_text_pre=5, n_text_mid=3, n_text_post=6, seed=0, per_frame_video=False):
    """Synthetic prompt with native video inputs.  items: [("image" | "video", (t, h, w))] in sequence order.  A video is ONE
    run of t*h*w/4 <|video_pad|> tokens between <vs> / <ve> (Qwen2.5-VL, TF:models/qwen2_5_vl/processing_qwen2_5_vl.py:64-67);
    per_frame_video (Qwen3-VL, TF:models/qwen3_vl/processing_qwen3_vl.py:80-106): every temporal patch is its own
    [timestamp text.., <vs>, h*w/4 pads, <ve>] block."""
    g = torch.Generator().manual_seed(1000 + seed)
    V = cfg["text_config"]["vocab_size"]
    lo, hi = 10, min(V, cfg["vision_start_token_id"]) - 1
    unit = cfg["vision_config"]["spatial_merge_size"] ** 2

    def text(n):
        return torch.randint(lo, hi, (n,), generator=g).tolist()

    ids = text(n_text_pre)
    for kind, (t, h, w) in items:
        tok = cfg["image_token_id"] if kind == "image" else cfg["video_token_id"]
        if kind == "video" and per_frame_video:
            for _ in range(t):
                ids += text(n_text_mid) + [cfg["vision_start_token_id"]] + [tok] * (h * w // unit) + [cfg["vision_end_token_id"]]
        else:
            ids += text(n_text_mid) + [cfg["vision_start_token_id"]] + [tok] * (t * h * w // unit) + [cfg["vision_end_token_id"]]
    ids += text(n_text_post)
    return ids


def make_frames(n, H, W, seed=0):
    """uint8 RGB frames [n,3,H,W]."""
    g = torch.Generator().manual_seed(2000 + seed)
    return torch.randint(0, 256, (n, 3, H, W), generator=g, dtype=torch.uint8)
