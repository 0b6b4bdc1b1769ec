"""Model configuration read from an HF ``config.json`` (dict) of a Qwen2.5-VL or Qwen3-VL checkpoint (`arch`)."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List


def _round_up(n, m):
    return (n + m - 1) // m * m


@dataclass
class VisionCfg:
    depth: int
    hidden_size: int
    num_heads: int
    intermediate_size: int
    out_hidden_size: int
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: List[int] = field(default_factory=lambda: [7, 15, 23, 31])
    in_channels: int = 3
    # Qwen3-VL only (model_type qwen3_vl): learned position table, DeepStack taps, fc1/GELU(tanh)/fc2 MLP
    num_position_embeddings: int = 0
    deepstack_visual_indexes: List[int] = field(default_factory=list)
    hidden_act: str = "silu"
    # rope positions per second of video (TF:models/qwen2_5_vl/configuration_qwen2_5_vl.py:60; the 7B / 3B checkpoints say 2)
    tokens_per_second: int = 4

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def head_dim_pad(self):
        """Stored width of a vision head: 72 (Qwen3-VL, 1152/16) is kept 80 wide so that the 80-wide attention tiles and the
        16-byte rotary chunks apply (weights.py packs [36 | 4 zeros | 36 | 4 zeros])."""
        return _round_up(self.head_dim, 16)

    @property
    def hidden_pad(self):
        return _round_up(self.hidden_size, 64)

    @property
    def patch_k(self):
        return self.in_channels * self.temporal_patch_size * self.patch_size ** 2

    @property
    def patch_k_pad(self):  # GEMM K must be a multiple of 64
        return _round_up(self.patch_k, 64)

    @property
    def inter_pad(self):
        return _round_up(self.intermediate_size, 64)

    @property
    def merge_unit(self):
        return self.spatial_merge_size ** 2


@dataclass
class TextCfg:
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    tie_word_embeddings: bool = False
    explicit_head_dim: int = 0          # Qwen3-VL states head_dim in config.json
    mrope_interleaved: bool = False     # Qwen3-VL: t/h/w frequencies interleaved (TF3:368-390) instead of in sections
    qk_norm: bool = False               # Qwen3-VL: RMSNorm on the q and k heads
    attention_bias: bool = True         # Qwen2.5-VL has q/k/v biases, Qwen3-VL none

    @property
    def head_dim(self):
        return self.explicit_head_dim or self.hidden_size // self.num_attention_heads

    @property
    def inter_pad(self):
        return _round_up(self.intermediate_size, 64)


@dataclass
class O3VConfig:
    vision: VisionCfg
    text: TextCfg
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_id: int = 151645
    pad_token_id: int = 151643
    name_or_path: str = ""
    arch: str = "qwen2_5_vl"

    @staticmethod
    def from_dict(d: dict, name_or_path: str = "") -> "O3VConfig":
        vc = d["vision_config"]
        tc = dict(d.get("text_config") or {})
        q3 = str(d.get("model_type", "")).startswith("qwen3_vl")
        # older checkpoints keep the text fields at the top level of config.json
        for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "intermediate_size",
                  "vocab_size", "rms_norm_eps", "rope_theta", "tie_word_embeddings"):
            if k not in tc and k in d:
                tc[k] = d[k]
        rope = tc.get("rope_parameters") or tc.get("rope_scaling") or d.get("rope_scaling") or {}
        mrope = tc.get("mrope_section") or rope.get("mrope_section") or ([24, 20, 20] if q3 else [16, 24, 24])
        theta = tc.get("rope_theta") or rope.get("rope_theta") or (5000000.0 if q3 else 1000000.0)
        vision = VisionCfg(
            depth=vc["depth"], hidden_size=vc["hidden_size"], num_heads=vc["num_heads"],
            intermediate_size=vc["intermediate_size"], out_hidden_size=vc["out_hidden_size"],
            patch_size=vc.get("patch_size", 16 if q3 else 14),
            num_position_embeddings=int(vc.get("num_position_embeddings", 2304 if q3 else 0)),
            deepstack_visual_indexes=list(vc.get("deepstack_visual_indexes", [8, 16, 24] if q3 else [])),
            hidden_act=vc.get("hidden_act", "gelu_pytorch_tanh" if q3 else "silu"), temporal_patch_size=vc.get("temporal_patch_size", 2),
            spatial_merge_size=vc.get("spatial_merge_size", 2), window_size=vc.get("window_size", 112),
            fullatt_block_indexes=list(vc.get("fullatt_block_indexes", [7, 15, 23, 31])), in_channels=vc.get("in_channels", 3),
            tokens_per_second=int(vc.get("tokens_per_second", 4)))
        text = TextCfg(
            hidden_size=tc["hidden_size"], num_hidden_layers=tc["num_hidden_layers"],
            num_attention_heads=tc["num_attention_heads"], num_key_value_heads=tc["num_key_value_heads"],
            intermediate_size=tc["intermediate_size"], vocab_size=tc["vocab_size"],
            rms_norm_eps=tc.get("rms_norm_eps", 1e-6), rope_theta=float(theta), mrope_section=list(mrope),
            tie_word_embeddings=bool(tc.get("tie_word_embeddings", d.get("tie_word_embeddings", False))),
            explicit_head_dim=int(tc.get("head_dim") or 0),
            mrope_interleaved=bool(rope.get("mrope_interleaved", q3)), qk_norm=q3,
            attention_bias=bool(tc.get("attention_bias", not q3)))
        if q3 and (text.attention_bias or vision.hidden_act != "gelu_pytorch_tanh"):
            raise ValueError("qwen3_vl config with attention_bias or a vision activation other than gelu_pytorch_tanh is not supported")
        eos = d.get("eos_token_id", tc.get("eos_token_id", 151645))
        if isinstance(eos, (list, tuple)):
            eos = eos[0]
        return O3VConfig(vision=vision, text=text,
                         image_token_id=d.get("image_token_id", 151655), video_token_id=d.get("video_token_id", 151656),
                         vision_start_token_id=d.get("vision_start_token_id", 151652),
                         vision_end_token_id=d.get("vision_end_token_id", 151653),
                         eos_token_id=eos if eos is not None else 151645,
                         pad_token_id=d.get("pad_token_id", tc.get("pad_token_id", 151643)) or 151643,
                         name_or_path=name_or_path, arch="qwen3_vl" if q3 else "qwen2_5_vl")

    @staticmethod
    def from_pretrained(path: str) -> "O3VConfig":
        with open(os.path.join(path, "config.json")) as f:
            return O3VConfig.from_dict(json.load(f), name_or_path=path)


def qwen25vl_7b_dict():
    """Public config.json values of Qwen/Qwen2.5-VL-7B-Instruct (the Open-o3-Video base), SURVEY.md section 8."""
    return {
        "model_type": "qwen2_5_vl", "image_token_id": 151655, "video_token_id": 151656, "vision_start_token_id": 151652,
        "vision_end_token_id": 151653, "eos_token_id": 151645, "pad_token_id": 151643, "tie_word_embeddings": False,
        "vision_config": {"depth": 32, "hidden_size": 1280, "num_heads": 16, "intermediate_size": 3420,
                          "out_hidden_size": 3584, "patch_size": 14, "temporal_patch_size": 2, "spatial_merge_size": 2,
                          "window_size": 112, "fullatt_block_indexes": [7, 15, 23, 31], "in_channels": 3, "tokens_per_second": 2},
        "text_config": {"hidden_size": 3584, "num_hidden_layers": 28, "num_attention_heads": 28, "num_key_value_heads": 4,
                        "intermediate_size": 18944, "vocab_size": 152064, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                        "mrope_section": [16, 24, 24], "tie_word_embeddings": False},
    }


def qwen25vl_3b_dict():
    d = qwen25vl_7b_dict()
    d["tie_word_embeddings"] = True
    d["vision_config"] = dict(d["vision_config"], out_hidden_size=2048)
    d["text_config"] = {"hidden_size": 2048, "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2,
                        "intermediate_size": 11008, "vocab_size": 151936, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0,
                        "mrope_section": [16, 24, 24], "tie_word_embeddings": True}
    return d


def qwen3vl_8b_dict():
    """Public config.json values of Qwen/Qwen3-VL-8B-Instruct (BASELINE config #5, R:eval/tts.py:27-32)."""
    return {
        "model_type": "qwen3_vl", "image_token_id": 151655, "video_token_id": 151656, "vision_start_token_id": 151652,
        "vision_end_token_id": 151653, "eos_token_id": 151645, "pad_token_id": 151643, "tie_word_embeddings": False,
        "vision_config": {"depth": 27, "hidden_size": 1152, "num_heads": 16, "intermediate_size": 4304, "out_hidden_size": 4096,
                          "patch_size": 16, "temporal_patch_size": 2, "spatial_merge_size": 2, "in_channels": 3,
                          "hidden_act": "gelu_pytorch_tanh", "num_position_embeddings": 2304,
                          "deepstack_visual_indexes": [8, 16, 24]},
        "text_config": {"hidden_size": 4096, "num_hidden_layers": 36, "num_attention_heads": 32, "num_key_value_heads": 8,
                        "head_dim": 128, "intermediate_size": 12288, "vocab_size": 151936, "rms_norm_eps": 1e-6,
                        "rope_theta": 5000000.0, "mrope_section": [24, 20, 20], "attention_bias": False,
                        "tie_word_embeddings": False},
    }
