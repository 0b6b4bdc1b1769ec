#!/usr/bin/env python3
"""A/B of the prefetch role of the one-launch decode block (o3v_decode_prefetch_config): while the dependent attention chain of a
layer leaves HBM idle, extra workgroups touch the head of the gate/up weight stream so that the next launch finds it in the
memory-side Infinity Cache.  Batch-1 decode at 7B dims, ms per step (best of 3 runs of 256 tokens) per (bytes, policy, workgroups)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_o3_video_amd import _lib  # noqa: E402
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False))
ids = build_prompt(cfg, 32, 120, 32 * (120 + 15) + 170)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
T = 256
lib = _lib.load()


def run(mb, policy, wgs):
    lib.o3v_decode_prefetch_config(C.c_size_t(mb << 20), policy, wgs)
    best = None
    for _ in range(3):
        out = eng.generate([ids], None, frames=frames, max_new_tokens=T, eos_token_ids=(), repetition_penalty=1.05, return_margins=False,
                           sync_timings=True)
        ms = out.timings["decode_ms"] / T
        best = ms if best is None else min(best, ms)
    print(f"prefetch {mb:4d} MB  policy {policy}  wgs {wgs:4d}   {best:.4f} ms/step   launches/layer {out.timings['launches_per_layer']}", flush=True)
    return out.sequences


ref = run(0, 0, 160)
cases = [(16, 0, 160), (32, 0, 160), (64, 0, 160), (96, 0, 160), (128, 0, 160), (64, 1, 160), (64, 0, 80), (64, 0, 320), (32, 0, 80), (0, 0, 160)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for mb, pol, wgs in cases:
    seq = run(mb, pol, wgs)
    assert torch.equal(seq, ref), "the prefetch role changed the tokens"
