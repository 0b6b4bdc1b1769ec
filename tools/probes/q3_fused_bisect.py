#!/usr/bin/env python3
"""Qwen3-VL-8B dims, batch-1 greedy decode: the one-launch attention block against the stand-alone kernels, with the decode
steps enqueued as one call or as 16 + 8; prints the generated ids of every variant and the first step at which two differ."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open_o3_video_amd.config import O3VConfig, qwen3vl_8b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen3vl_8b_dict())
fp8 = "--bf16-only" not in sys.argv
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 21, "cuda", std=0.02, head_std=0.08), "cuda", batched_decode=False,
                                   fp8_decode=fp8))
F, H, W, T = 8, 224, 416, 24
tpf = (H // 32) * (W // 32)
g = np.random.default_rng(4)
ids = g.integers(1000, 150000, 150).tolist()
for _ in range(F):
    ids += g.integers(1000, 150000, 12).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * tpf + \
        [cfg.vision_end_token_id] + g.integers(1000, 150000, 1).tolist()
ids += g.integers(1000, 150000, 20).tolist()
gen = torch.Generator(device="cuda").manual_seed(5)
frames = torch.randint(0, 256, (F, 3, H, W), generator=gen, dtype=torch.uint8, device="cuda")
if fp8:
    eng.w.llm.layer[0].qkv_w8 = 0
S = len(ids)
runs = {}
for name, fused, first in (("fused 16+8", True, 16), ("fused 24", True, 1 << 30), ("alone 24", False, 1 << 30),
                           ("fused 16+8 again", True, 16), ("alone 24 again", False, 1 << 30), ("fused 8+16", True, 8)):
    eng.fused_decode = fused
    eng._first_chunk = first
    o = eng.generate([ids], None, frames=frames, max_new_tokens=T)
    runs[name] = (o.sequences[0, S:].tolist(), o.margins[0].tolist())
    print(f"{name:18s} {o.timings.get('launches_per_layer')} {runs[name][0]}", flush=True)
ref = runs["alone 24"]
for name, (seq, mg) in runs.items():
    k = next((i for i in range(T) if seq[i] != ref[0][i]), T)
    km = next((i for i in range(T) if mg[i] != ref[1][i]), T)
    print(f"{name:18s} first differing id at step {k}, first differing margin at step {km}"
          + (f" (margins there: {mg[km]:.4f} vs {ref[1][km]:.4f})" if km < T else ""))

# finer than the margins: K/V caches, last hidden row and logits after T steps, one-launch block against the stand-alone kernels
eng._debug_keep = True
eng._first_chunk = 16
for T2 in (4, 16):
    st = {}
    for fused in (True, False):
        eng.fused_decode = fused
        eng.generate([ids], None, frames=frames, max_new_tokens=T2)
        st[fused] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng._debug_last.items()}
    a, b = st[True], st[False]
    print(f"T={T2}: cache shape {tuple(a['kc'].shape)}  x equal {torch.equal(a['x'], b['x'])}  logits equal {torch.equal(a['logits'], b['logits'])}")
    L = cfg.text.num_hidden_layers
    for nm in ("kc", "vc"):
        ka = a[nm].view(L, -1, a[nm].shape[-2], a[nm].shape[-1])[:, :, S:S + T2 - 1].view(torch.int16)
        kb = b[nm].view(L, -1, b[nm].shape[-2], b[nm].shape[-1])[:, :, S:S + T2 - 1].view(torch.int16)
        ne = (ka != kb)
        if not ne.any():
            print(f"  {nm}: equal")
            continue
        per = ne.sum(dim=(1, 3))          # [layer, slot]
        first = [(int(s_), int(l)) for s_ in range(per.shape[1]) for l in range(L) if per[l, s_] > 0][:6]
        print(f"  {nm}: {int(ne.sum())} differing elements; first (step, layer): {first}")
        s0, l0 = first[0]
        idx = ne[l0, :, s0].nonzero()[:8].tolist()
        print(f"    at step {s0} layer {l0}: {int(per[l0, s0])} elements, (kv head, dim) {idx}")
        for h, dd in idx[:4]:
            print(f"      fused {a[nm].view(L, -1, a[nm].shape[-2], a[nm].shape[-1])[l0, h, S + s0, dd].item():.6f}  "
                  f"alone {b[nm].view(L, -1, b[nm].shape[-2], b[nm].shape[-1])[l0, h, S + s0, dd].item():.6f}")
