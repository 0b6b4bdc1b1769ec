#!/usr/bin/env python3
"""Qwen3-VL-8B dims: decode step 13 of the full-size test (context 1032) replayed layer by layer through the C-ABI -- the attention
half as the stand-alone launches and as the one-launch block on the same inputs, every output compared bit for bit."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open_o3_video_amd import _lib  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen3vl_8b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

BF = torch.bfloat16
cfg = O3VConfig.from_dict(qwen3vl_8b_dict())
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 21, "cuda", std=0.02, head_std=0.08), "cuda", batched_decode=False,
                                   fp8_decode=False))
F, Hh, W = 8, 224, 416
tpf = (Hh // 32) * (W // 32)
g = np.random.default_rng(4)
ids = g.integers(1000, 150000, 150).tolist()
for _ in range(F):
    ids += g.integers(1000, 150000, 12).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * tpf + \
        [cfg.vision_end_token_id] + g.integers(1000, 150000, 1).tolist()
ids += g.integers(1000, 150000, 20).tolist()
gen = torch.Generator(device="cuda").manual_seed(5)
frames = torch.randint(0, 256, (F, 3, Hh, W), generator=gen, dtype=torch.uint8, device="cuda")
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 13
T2 = STEP + 1                       # forwards for steps 0..STEP-1, token STEP sampled and embedded, its forward skipped
eng._debug_keep = True
eng.fused_decode = False
out = eng.generate([ids], None, frames=frames, max_new_tokens=T2)
d = eng._debug_last
S, Tmax, nsplit = d["S"], d["Tmax"], d["nsplit"]
print(f"S={S} Tmax={Tmax} nsplit={nsplit} tokens {out.sequences[0, S:].tolist()}")
tc = cfg.text
H, I, Hq, Hkv, D, L = tc.hidden_size, tc.intermediate_size, tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim, tc.num_hidden_layers
N, QD = (Hq + 2 * Hkv) * D, Hq * D
dev = "cuda"
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
scale = 1.0 / math.sqrt(D)
slot, ctx = S + STEP, S + STEP + 1
sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
po1, pm1 = torch.empty_like(part_o), torch.empty_like(part_ml)
x = d["x"].clone()
eps = tc.rms_norm_eps
vp = C.c_void_p
for l in range(L):
    lw = eng.w.llm.layer[l]
    k1, v1 = d["kc"][l].clone(), d["vc"][l].clone()
    k2, v2 = d["kc"][l].clone(), d["vc"][l].clone()
    x1, x2 = x.clone(), x.clone()
    qkv1 = torch.zeros(1, N, dtype=BF, device=dev)
    raw2 = torch.zeros(N, dtype=BF, device=dev)
    q1, q2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    att1, att2 = torch.zeros(1, Hq, D, dtype=BF, device=dev), torch.zeros(1, Hq, D, dtype=BF, device=dev)
    _lib.call("o3v_linear_decode", P(x1), vp(lw.ln1), eps, vp(lw.qkv_w), vp(lw.qkv_wp), None, None, P(qkv1), 1, N, H, H, N, 0, _lib.EPI_NONE, st)
    _lib.call("o3v_qkv_norm_rope_cache", P(qkv1), vp(lw.q_norm), vp(lw.k_norm), eps, P(d["cos"]), P(d["sin"]), P(q1), P(k1), P(v1), slot, 1, 1,
              Hq, Hkv, D, Tmax, T2, STEP, st)
    _lib.call("o3v_attn_decode", P(q1), P(k1), P(v1), P(att1), P(po1), P(pm1), P(d["k_lo"]), 1, Hq, Hkv, D, ctx, Tmax, nsplit, scale, st)
    _lib.call("o3v_linear_decode", P(att1), None, 0.0, vp(lw.o_w), vp(lw.o_wp), None, P(x1), P(x1), 1, H, QD, QD, H, H, _lib.EPI_RESIDUAL, st)
    rc = lib.o3v_decode_attn_block_qknorm(P(x2), vp(lw.ln1), eps, vp(lw.qkv_w), None, vp(lw.o_w), None, vp(lw.q_norm), vp(lw.k_norm), P(raw2),
                                          P(d["cos"]), P(d["sin"]), P(q2), P(att2), P(k2), P(v2), P(part_o), P(part_ml), P(d["k_lo"]), H, Hq,
                                          Hkv, D, slot, Tmax, T2, STEP, nsplit, scale, P(sync), l + 1, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    tmo = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
    eq = lambda a, b: bool(torch.equal(a.view(torch.int16), b.view(torch.int16)))
    flags = {"raw": eq(raw2, qkv1.view(-1)), "q": eq(q2, q1), "k": eq(k2, k1), "v": eq(v2, v1), "att": eq(att2, att1), "x": eq(x2, x1)}
    if not all(flags.values()) or tmo:
        print(f"layer {l}: {flags} tmo={tmo:#x}")
        if not flags["att"]:
            ne = (att2.view(torch.int16) != att1.view(torch.int16))[0]
            heads = ne.any(dim=1).nonzero().flatten().tolist()
            print(f"   att differs in heads {heads}; elements per head {[int(ne[h].sum()) for h in heads]}")
            h = heads[0]
            dd = ne[h].nonzero().flatten().tolist()[:6]
            print(f"   head {h} dims {dd}: fused {att2[0, h, dd].tolist()} alone {att1[0, h, dd].tolist()}")
            # the splits' partial results of that head, alone vs fused
            ns = abs(nsplit)
            po_a, po_f = po1[:Hq * ns * D].view(Hq, ns, D), part_o[:Hq * ns * D].view(Hq, ns, D)
            pm_a, pm_f = pm1[:Hq * ns * 2].view(Hq, ns, 2), part_ml[:Hq * ns * 2].view(Hq, ns, 2)
            for sp in range(abs(nsplit)):
                print(f"   split {sp}: m,l alone {pm_a[h, sp].tolist()} fused {pm_f[h, sp].tolist()}  partial o equal "
                      f"{bool(torch.equal(po_a[h, sp], po_f[h, sp]))}")
    # the MLP half on the stand-alone result
    mlp = torch.empty(1, I, dtype=BF, device=dev)
    _lib.call("o3v_linear_decode", P(x1), vp(lw.ln2), eps, vp(lw.gu_w), vp(lw.gu_wp), None, None, P(mlp), 1, 2 * I, H, H, I, 0, _lib.EPI_SWIGLU, st)
    _lib.call("o3v_linear_decode", P(mlp), None, 0.0, vp(lw.down_w), vp(lw.down_wp), None, P(x1), P(x1), 1, H, I, I, H, H, _lib.EPI_RESIDUAL, st)
    x = x1
print("done")
