#!/usr/bin/env python3
"""Decode with fp8 weight rows at 7B dims for rocprofv3 --kernel-trace --stats (per-kernel times of the fp8 path)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402
from open_o3_video_amd import _lib  # noqa: E402
if os.environ.get("O3V_LIB"):
    _lib.LIB_PATH = os.environ["O3V_LIB"]

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False, fp8_decode=True))
ids = build_prompt(cfg, 32, 120, 4490)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
for _ in range(2):
    out = eng.generate([ids], None, frames=frames, max_new_tokens=256, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                       sync_timings=True)
print({k: round(v, 2) for k, v in out.timings.items()}, "ms/step", round(out.timings["decode_ms"] / 256, 4))
