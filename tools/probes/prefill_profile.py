#!/usr/bin/env python3
"""ViT + prefill of ONE 32-frame video (S = 4490, the bench's prompt) for kernel-level profiling:
`rocprofv3 --kernel-trace --stats -d gpurun_out/prof_prefill -- python3 tools/probes/prefill_profile.py [n_iter]`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
ids = build_prompt(cfg, 32, 120, 4490)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
for _ in range(n):
    out = eng.generate([ids], None, frames=frames, max_new_tokens=1, eos_token_ids=(), return_margins=False, sync_timings=True)
    print(f"vit {out.timings['vit_ms']:.2f} ms  prefill {out.timings['prefill_ms']:.2f} ms", flush=True)
