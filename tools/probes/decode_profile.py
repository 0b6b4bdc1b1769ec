#!/usr/bin/env python3
"""Batch-1 decode timing at 7B (default) or Qwen3-VL-8B (--q3) dims, bf16 or --fp8 rows; O3V_LIB selects another build of the
library (A/B of kernel variants in one gpurun call), O3V_FUSED_DECODE=0 the stand-alone attention-half kernels."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_o3_video_amd import _lib  # noqa: E402
if os.environ.get("O3V_LIB"):
    _lib.LIB_PATH = os.environ["O3V_LIB"]
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict, qwen3vl_8b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

q3 = "--q3" in sys.argv
cfg = O3VConfig.from_dict(qwen3vl_8b_dict() if q3 else qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False, fp8_decode="--fp8" in sys.argv))
H, W, tpf = (224, 416, 91) if q3 else (224, 420, 120)
ids = build_prompt(cfg, 32, tpf, 32 * (tpf + 15) + 170)
frames = torch.randint(0, 256, (32, 3, H, W), dtype=torch.uint8, device=dev)
T = 256
best = None
for _ in range(4):
    out = eng.generate([ids], None, frames=frames, max_new_tokens=T, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                       sync_timings=True)
    ms = out.timings["decode_ms"] / T
    best = ms if best is None else min(best, ms)
print(os.environ.get("O3V_LIB", "default lib"), "q3" if q3 else "7b", "fp8" if "--fp8" in sys.argv else "bf16", "ms/step best of 4:",
      round(best, 4), flush=True)
