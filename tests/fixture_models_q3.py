"""Seeded Qwen3-VL fixture models (BASELINE config #5's family) shared by tools/make_golden.py (HF side)
and the tests (our side): same conventions as fixture_models.py (weights regenerated from a seed, rounded through bf16)."""
from __future__ import annotations

import torch

import fixture_models as fm


def tiny_q3_config():
    """Smallest shapes: ViT head_dim 32 (2 heads of 64 hidden), LLM head_dim 32."""
    return {
        "model_type": "qwen3_vl",
        **fm.SPECIAL,
        "tie_word_embeddings": False,
        "vision_config": {
            "depth": 4, "hidden_size": 64, "num_heads": 2, "intermediate_size": 96, "out_hidden_size": 128,
            "patch_size": 16, "temporal_patch_size": 2, "spatial_merge_size": 2, "in_channels": 3,
            "hidden_act": "gelu_pytorch_tanh", "num_position_embeddings": 64, "deepstack_visual_indexes": [0, 2],
        },
        "text_config": {
            "hidden_size": 128, "num_hidden_layers": 3, "num_attention_heads": 4, "num_key_value_heads": 2, "head_dim": 32,
            "intermediate_size": 256, "vocab_size": 512, "rms_norm_eps": 1e-6, "rope_theta": 5000000.0,
            "mrope_section": [6, 5, 5], "hidden_act": "silu", "max_position_embeddings": 4096, "attention_bias": False,
            "tie_word_embeddings": False,
        },
    }


def medium_q3_config():
    """True head geometry of Qwen3-VL-8B (ViT head_dim 72 = 1152/16, LLM head_dim 128, GQA 4:1, interleaved mrope
    [24,20,20], three DeepStack taps) with few layers and narrow widths."""
    return {
        "model_type": "qwen3_vl",
        **fm.SPECIAL,
        "image_token_id": 4000, "video_token_id": 4001, "vision_start_token_id": 3998,
        "vision_end_token_id": 3999, "eos_token_id": 4010, "pad_token_id": 4011, "bos_token_id": 4009,
        "tie_word_embeddings": False,
        "vision_config": {
            "depth": 5, "hidden_size": 288, "num_heads": 4, "intermediate_size": 428, "out_hidden_size": 1024,
            "patch_size": 16, "temporal_patch_size": 2, "spatial_merge_size": 2, "in_channels": 3,
            "hidden_act": "gelu_pytorch_tanh", "num_position_embeddings": 144, "deepstack_visual_indexes": [1, 2, 4],
        },
        "text_config": {
            "hidden_size": 1024, "num_hidden_layers": 4, "num_attention_heads": 8, "num_key_value_heads": 2, "head_dim": 128,
            "intermediate_size": 1536, "vocab_size": 4096, "rms_norm_eps": 1e-6, "rope_theta": 5000000.0,
            "mrope_section": [24, 20, 20], "hidden_act": "silu", "max_position_embeddings": 8192, "attention_bias": False,
            "tie_word_embeddings": False,
        },
    }


def weight_specs(cfg):
    vc, tc = cfg["vision_config"], cfg["text_config"]
    vh, vi, vo = vc["hidden_size"], vc["intermediate_size"], vc["out_hidden_size"]
    unit = vc["spatial_merge_size"] ** 2
    p = "model.visual."
    specs = [(p + "patch_embed.proj.weight", (vh, vc["in_channels"], vc["temporal_patch_size"], vc["patch_size"], vc["patch_size"]), "patch"),
             (p + "patch_embed.proj.bias", (vh,), "bias"),
             (p + "pos_embed.weight", (vc["num_position_embeddings"], vh), "pos")]
    for i in range(vc["depth"]):
        b = f"{p}blocks.{i}."
        specs += [(b + "norm1.weight", (vh,), "norm"), (b + "norm1.bias", (vh,), "bias"),
                  (b + "norm2.weight", (vh,), "norm"), (b + "norm2.bias", (vh,), "bias"),
                  (b + "attn.qkv.weight", (3 * vh, vh), "linear"), (b + "attn.qkv.bias", (3 * vh,), "bias"),
                  (b + "attn.proj.weight", (vh, vh), "linear"), (b + "attn.proj.bias", (vh,), "bias"),
                  (b + "mlp.linear_fc1.weight", (vi, vh), "linear"), (b + "mlp.linear_fc1.bias", (vi,), "bias"),
                  (b + "mlp.linear_fc2.weight", (vh, vi), "linear"), (b + "mlp.linear_fc2.bias", (vh,), "bias")]

    def merger(prefix, post):
        return [(prefix + "norm.weight", (vh * unit if post else vh,), "norm"), (prefix + "norm.bias", (vh * unit if post else vh,), "bias"),
                (prefix + "linear_fc1.weight", (vh * unit, vh * unit), "linear"), (prefix + "linear_fc1.bias", (vh * unit,), "bias"),
                (prefix + "linear_fc2.weight", (vo, vh * unit), "linear"), (prefix + "linear_fc2.bias", (vo,), "bias")]
    specs += merger(p + "merger.", False)
    for j in range(len(vc["deepstack_visual_indexes"])):
        specs += merger(f"{p}deepstack_merger_list.{j}.", True)
    H, nh, nkv, hd, I, V = (tc["hidden_size"], tc["num_attention_heads"], tc["num_key_value_heads"], tc["head_dim"],
                            tc["intermediate_size"], tc["vocab_size"])
    specs.append(("model.language_model.embed_tokens.weight", (V, H), "embed"))
    for i in range(tc["num_hidden_layers"]):
        b = f"model.language_model.layers.{i}."
        specs += [(b + "input_layernorm.weight", (H,), "norm"), (b + "post_attention_layernorm.weight", (H,), "norm"),
                  (b + "self_attn.q_proj.weight", (nh * hd, H), "linear"), (b + "self_attn.k_proj.weight", (nkv * hd, H), "linear"),
                  (b + "self_attn.v_proj.weight", (nkv * hd, H), "linear"), (b + "self_attn.o_proj.weight", (H, nh * hd), "linear"),
                  (b + "self_attn.q_norm.weight", (hd,), "norm"), (b + "self_attn.k_norm.weight", (hd,), "norm"),
                  (b + "mlp.gate_proj.weight", (I, H), "linear"), (b + "mlp.up_proj.weight", (I, H), "linear"),
                  (b + "mlp.down_proj.weight", (H, I), "linear")]
    specs.append(("model.language_model.norm.weight", (H,), "norm"))
    specs.append(("lm_head.weight", (V, H), "head"))
    return specs


def make_weights(cfg, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape, kind in weight_specs(cfg):
        if kind == "norm":
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "bias":
            w = 0.05 * torch.randn(shape, generator=g)
        elif kind == "embed":
            w = 0.5 * torch.randn(shape, generator=g)
        elif kind == "pos":
            w = 0.3 * torch.randn(shape, generator=g)
        elif kind == "head":
            w = (4.0 / shape[1] ** 0.5) * torch.randn(shape, generator=g)
        elif kind == "patch":
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            w = (1.0 / fan_in ** 0.5) * torch.randn(shape, generator=g)
        else:
            w = (1.0 / shape[1] ** 0.5) * torch.randn(shape, generator=g)
        W[name] = w.to(torch.bfloat16).to(dtype)
    return W
