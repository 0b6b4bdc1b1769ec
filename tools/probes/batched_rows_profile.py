#!/usr/bin/env python3
"""Decode with R independent rows (R videos decoding together, the batched_videos leg of bench.py) for kernel-level profiling:
`rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rows -- python3 tools/probes/batched_rows_profile.py 32 48`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 48
cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
ids = build_prompt(cfg, 32, 120, 32 * (120 + 15) + 170)
frames = torch.randint(0, 256, (R * 32, 3, 224, 420), dtype=torch.uint8, device=dev)
for _ in range(2):
    out = eng.generate([ids] * R, None, frames=frames, max_new_tokens=T, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                       sync_timings=True)
print(f"rows {R}: decode {out.timings['decode_ms'] / T:.3f} ms/step, vit {out.timings['vit_ms']:.1f} ms, prefill {out.timings['prefill_ms']:.1f} ms,"
      f" launches/layer {out.timings['launches_per_layer']}", flush=True)
