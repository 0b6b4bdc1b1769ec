#!/usr/bin/env python3
"""Group rollout timing at 7B dims (G rows, 256 sampled tokens), bf16 or --fp8 rows; O3V_LIB selects another build of the library."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_o3_video_amd import _lib  # noqa: E402
if os.environ.get("O3V_LIB"):
    _lib.LIB_PATH = os.environ["O3V_LIB"]
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, fp8_decode="--fp8" in sys.argv))
ids = build_prompt(cfg, 32, 120, 4490)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
for G in [int(a[1:]) for a in sys.argv[1:] if a.startswith("G")] or [8, 16]:
    kw = dict(max_new_tokens=256, num_return_sequences=G, do_sample=True, top_p=0.95, temperature=1.0, seed=1, return_margins=False)
    best = None
    for _ in range(3):
        out = eng.generate([ids], None, frames=frames, sync_timings=True, **kw)
        ms = out.timings["decode_ms"] / 256
        best = ms if best is None else min(best, ms)
    print(json.dumps({"lib": os.path.basename(os.environ.get("O3V_LIB", "default")), "rows": "fp8" if "--fp8" in sys.argv else "bf16", "G": G,
                      "decode_ms_per_step_best_of_3": round(best, 3)}), flush=True)
