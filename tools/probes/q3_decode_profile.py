#!/usr/bin/env python3
"""Decode at Qwen3-VL-8B dims for rocprofv3 --kernel-trace --stats (O3V_FUSED_DECODE=0/1 selects the attention-half form)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen3vl_8b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen3vl_8b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False, fp8_decode="--fp8" in sys.argv))
ids = build_prompt(cfg, 32, 91, 32 * (91 + 15) + 170)
frames = torch.randint(0, 256, (32, 3, 224, 416), dtype=torch.uint8, device=dev)
for _ in range(2):
    out = eng.generate([ids], None, frames=frames, max_new_tokens=128, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                       sync_timings=True)
print({k: round(v, 2) for k, v in out.timings.items()}, "ms/step", round(out.timings["decode_ms"] / 128, 4), flush=True)
