#!/usr/bin/env python3
"""A/B of library builds on the isolated decode gate/up GEMV (bench.py kernel_roofline) and on the batch-1 decode step:
`O3V_LIB=path python tools/probes/gateup_lib_ab.py` per build (one process per library)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_o3_video_amd import _lib  # noqa: E402
if os.environ.get("O3V_LIB"):
    _lib.LIB_PATH = os.environ["O3V_LIB"]
import bench  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False))
r = [bench.kernel_roofline(eng, reps=5)["avg_launch_us"] for _ in range(5)]
ids = bench.build_prompt(cfg, 32, 120, 4490)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
ms = []
for _ in range(3):
    out = eng.generate([ids], None, frames=frames, max_new_tokens=256, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                       sync_timings=True)
    ms.append(out.timings["decode_ms"] / 256)
print(os.path.basename(os.environ.get("O3V_LIB", "default")), "gate/up us:", r, " decode ms/step:", [round(m, 4) for m in ms], flush=True)
