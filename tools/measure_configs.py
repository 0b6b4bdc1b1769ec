#!/usr/bin/env python3
"""Measure the secondary BASELINE.json configs on one GPU (7B dims, random weights): EVAL-RES, G=8 group rollout,
256-frame long video.  Prints one JSON line per config."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_prompt  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402
from open_o3_video_amd.engine import O3VEngine  # noqa: E402
from open_o3_video_amd.weights import DeviceWeights, random_getter  # noqa: E402

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
which = [a for a in sys.argv[1:] if not a.startswith("G")] or ["eval", "rollout", "long"]


def run(tag, frames_n, H, W, S_target, T, **kw):
    tpf = (H // 28) * (W // 28)
    ids = build_prompt(cfg, frames_n, tpf, S_target)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (frames_n, 3, H, W), generator=g, dtype=torch.uint8, device=dev)
    eng.generate([ids], None, frames=frames, max_new_tokens=T, **kw)  # warm-up with the same shapes (allocator, code paths)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate([ids], None, frames=frames, max_new_tokens=T, return_margins=False, sync_timings=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = out.sequences.shape[0]
    print(json.dumps({"config": tag, "S": len(ids), "rows": rows, "new_tokens": T, "wall_s": round(dt, 3),
                      "tokens_per_s": round(rows * T / dt, 1), "stage_ms": {k: round(v, 1) for k, v in out.timings.items() if k.endswith("_ms")}, "kv_cache_GB": round(out.timings.get("kv_cache_bytes", 0) / 1e9, 3),
                      "decode_ms_per_step": round(out.timings["decode_ms"] / T, 3)}), flush=True)


if "eval" in which:
    run("EVAL-RES 32x364x644 greedy B=1", 32, 364, 644, 10218, 256, repetition_penalty=1.05)
if "rollout" in which:
    for G in ([int(a[1:]) for a in sys.argv[1:] if a.startswith('G')] or (2, 4, 8, 16)):
        run(f"rollout G={G} sampled top_p=0.95 (TRAIN-RES)", 32, 224, 420, 4490, 256, num_return_sequences=G, do_sample=True,
            top_p=0.95, temperature=1.0, seed=1)
if "rollout_eval" in which:
    run("rollout G=8 sampled top_p=0.95 (EVAL-RES, S=10218)", 32, 364, 644, 10218, 128, num_return_sequences=8, do_sample=True,
        top_p=0.95, temperature=1.0, seed=1)
if "long" in which:
    run("LONG 256x224x224 greedy B=1", 256, 224, 224, 256 * (64 + 15) + 170, 128, repetition_penalty=1.05)

if "vstar5" in which:
    # the V-STAR harness asks 5 questions per video (R:eval/test/test_vstar_multi_images.py:511-544); frame
    # block first, then a ~64-token question; 128 greedy tokens each.  With / without visual + prefix-K/V reuse.
    tpf = (224 // 28) * (420 // 28)
    base = build_prompt(cfg, 32, tpf, 4490 - 64)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (32, 3, 224, 420), generator=g, dtype=torch.uint8, device=dev)
    rng = __import__("numpy").random.default_rng(5)
    qs = []
    for q in range(5):
        qs.append(list(base) + [int(t) for t in rng.integers(1000, 150000, 64)])   # frame block, then the question
    for reuse in (False, True):
        eng.drop_prefix_cache()
        eng.generate([qs[0]], None, frames=frames, max_new_tokens=4)  # warm-up (code paths, allocator)
        eng.drop_prefix_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vis, reused = None, 0
        for ids in qs:
            if reuse and vis is None:
                px, grid = eng.pixels_from_frames(frames)
                vis = eng.vit_forward(px, grid)
            if reuse:
                out = eng.generate([ids], None, vis_embeds=vis, image_grid_thw=grid, max_new_tokens=128, repetition_penalty=1.05,
                                   prefix_key="video", return_margins=False, sync_timings=bool(os.environ.get("O3V_STAGE_TIMES")))
                if os.environ.get("O3V_STAGE_TIMES"):
                    print({k: round(v, 2) for k, v in out.timings.items()}, flush=True)
            else:
                out = eng.generate([ids], None, frames=frames, max_new_tokens=128, repetition_penalty=1.05, return_margins=False)
            reused += out.timings["prefix_tokens_reused"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": f"V-STAR 5 questions/video, 128 tok each, reuse={reuse}", "S": len(base), "wall_s": round(dt, 3),
                          "videos_per_min": round(60 / dt, 1), "questions_per_s": round(5 / dt, 2),
                          "prefix_tokens_reused": int(reused)}), flush=True)

if "frames" in which:
    # host-buffer boundary: raw decoded frames uint8 [32,3,360,640] in host memory -> H2D -> antialiased bicubic resize to
    # TRAIN-RES 224x420 -> rescale/normalise/patchify -> ViT, vs the same with the frames already resident and resized
    from open_o3_video_amd import vision_process as vp
    raw = torch.randint(0, 256, (32, 3, 360, 640), dtype=torch.uint8).pin_memory()
    t_cpu0 = time.perf_counter()
    ref = vp.resize_frames(raw, (224, 420))
    t_cpu = time.perf_counter() - t_cpu0
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = raw.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rs = vp.resize_frames_device(d, (224, 420))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        px, grid = eng.pixels_from_frames(rs)
        vis = eng.vit_forward(px, grid)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(json.dumps({"config": "frame pipeline 32x3x360x640 u8 host -> ViT tokens", "h2d_ms": round((t1 - t0) * 1e3, 2),
                      "resize_ms": round((t2 - t1) * 1e3, 3), "patchify_vit_ms": round((t3 - t2) * 1e3, 2),
                      "cpu_resize_ms": round(t_cpu * 1e3, 1), "max_abs_diff_vs_cpu_resize": float((rs.cpu() - ref).abs().max())}),
          flush=True)

if "logps" in which:
    # log-prob pass of the GSPO step (R:grpo_trainer.py:601-632) for G=8 completions of 256 tokens behind the 4490-token
    # prompt: shared-prompt formulation vs the reference's row-by-row full-sequence formulation on the same engine
    tpf = (224 // 28) * (420 // 28)
    ids = build_prompt(cfg, 32, tpf, 4490)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (32, 3, 224, 420), generator=g, dtype=torch.uint8, device=dev)
    comp = torch.randint(1000, 150000, (8, 256), generator=torch.Generator().manual_seed(2))
    for name in ("shared", "rowwise"):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "shared":
                lp = eng.completion_logps(ids, comp, frames=frames)
            else:
                rows = []
                for r in range(8):
                    seq = torch.cat([torch.tensor(ids), comp[r]])[None]
                    rows.append(eng.per_token_logps(eng.forward_logits(seq.numpy(), None, frames=frames), seq)[:, len(ids) - 1:])
                lp2 = torch.cat(rows)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(json.dumps({"config": f"completion log-probs G=8 x 256 tok, S=4490, {name}", "wall_ms": round(dt * 1e3, 1)}), flush=True)
    print(json.dumps({"max_abs_diff_between_formulations": round((lp - lp2).abs().max().item(), 4)}), flush=True)

if "tts16" in which:
    # BASELINE config #5 (test-time scaling, N=16 chains of one 32-frame question, R:eval/test/test_videomme.py:129-226):
    # one group of 16 sampled rows behind one ViT pass and one prefill.  256 tokens per chain.  The reference runs 16
    # sequential generate calls, each with its own ViT pass and prefill.
    from open_o3_video_amd import tts
    tpf = (224 // 28) * (420 // 28)
    ids = build_prompt(cfg, 32, tpf, 4490)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (32, 3, 224, 420), generator=g, dtype=torch.uint8, device=dev)
    kw = dict(max_new_tokens=256, do_sample=True, top_p=0.95, temperature=1.0, repetition_penalty=1.05, seed=3,
              return_margins=False)
    for rep in range(2):
        eng.drop_prefix_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        px, grid = eng.pixels_from_frames(frames)
        vis = eng.vit_forward(px, grid)
        reused = 0
        out = eng.generate([ids], None, vis_embeds=vis, image_grid_thw=grid, num_return_sequences=16, prefix_key="q", **kw)
        reused += out.timings["prefix_tokens_reused"]
        claims = [{"obj": "o", "box_xyxy": [20 + 3 * i, 10 + 2 * i, 200 + 5 * i, 150 + 3 * i], "t_sec": float(i)} for i in range(10)]
        crops = tts.extract_and_crop(frames, 1.0, claims)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"config": "TTS N=16 chains x 256 tok, 32 frames (config #5 sampling side)", "S": len(ids), "wall_s": round(dt, 3),
                      "tokens_per_s": round(16 * 256 / dt, 1), "questions_per_min": round(60 / dt, 1),
                      "prefix_tokens_reused": int(reused), "crops": list(crops.shape)}), flush=True)

if "3b" in which:
    # BASELINE config #1 shapes: Qwen2.5-VL-3B dims (tied embeddings, GQA 8:1), 4 frames at EVAL-RES, 256 new tokens
    from open_o3_video_amd.config import qwen25vl_3b_dict
    cfg = O3VConfig.from_dict(qwen25vl_3b_dict())
    eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
    run("config#1 3B dims, 4x364x644 greedy", 4, 364, 644, 4 * (299 + 15) + 170, 256, repetition_penalty=1.05)

if "q3" in which:
    # BASELINE config #5's model family: Qwen3-VL-8B dims (random weights), 32 frames of 224x416 (7x13 = 91 tokens each),
    # greedy B=1 with bf16 rows and with fp8 rows, then the N=16 self-consistency group (one ViT pass + one prefill).
    from open_o3_video_amd.config import qwen3vl_8b_dict
    del eng
    torch.cuda.empty_cache()
    cfg = O3VConfig.from_dict(qwen3vl_8b_dict())
    H, W, NF, T = 224, 416, 32, 256
    tpf = (H // 32) * (W // 32)
    ids = build_prompt(cfg, NF, tpf, NF * (tpf + 15) + 170)
    g = torch.Generator(device=dev).manual_seed(1)
    frames = torch.randint(0, 256, (NF, 3, H, W), generator=g, dtype=torch.uint8, device=dev)
    for fp8 in (False, True):
        eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, fp8_decode=fp8))
        for tag, kw in (("greedy B=1", dict(repetition_penalty=1.05)),
                        ("N=16 sampled group", dict(num_return_sequences=16, do_sample=True, top_p=0.95, temperature=1.0,
                                                    repetition_penalty=1.05, seed=3))):
            eng.generate([ids], None, frames=frames, max_new_tokens=T, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.generate([ids], None, frames=frames, max_new_tokens=T, return_margins=False, sync_timings=True, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rows = out.sequences.shape[0]
            print(json.dumps({"config": f"Qwen3-VL-8B dims, {NF}x{H}x{W}, {tag}, {'fp8' if fp8 else 'bf16'} decode rows", "S": len(ids),
                              "rows": rows, "new_tokens": T, "wall_s": round(dt, 3), "tokens_per_s": round(rows * T / dt, 1),
                              "stage_ms": {k: round(v, 1) for k, v in out.timings.items() if k.endswith("_ms")}, "kv_cache_GB": round(out.timings.get("kv_cache_bytes", 0) / 1e9, 3),
                              "decode_ms_per_step": round(out.timings["decode_ms"] / T, 3),
                              "weights_GB": round(eng.w.nbytes() / 1e9, 2)}), flush=True)
        del eng
        torch.cuda.empty_cache()

if "fp8prefill" in which:
    # VERDICT r2 item 6 / BASELINE config #5 ("fp8 weights on CDNA4 fp8 MFMA"): the LLM prefill with its linears as fp8 x fp8 on the
    # matrix cores (engine.fp8_prefill, W8A8) against the bf16 prefill, at 7B and at Qwen3-VL-8B dims; then the GEMM rates of
    # the four linear shapes of a layer (TFLOP/s; roofline: dense fp8 peak 5 PFLOP/s, bf16 2.5, MI355X_MICROARCH.md Matrix cores).
    import ctypes as C
    from open_o3_video_amd import _lib
    from open_o3_video_amd.config import qwen3vl_8b_dict
    del eng
    torch.cuda.empty_cache()
    for name, cd, H, W, fac in (("7B", qwen25vl_7b_dict(), 224, 420, 28), ("Qwen3-VL-8B", qwen3vl_8b_dict(), 224, 416, 32)):
        c = O3VConfig.from_dict(cd)
        e = O3VEngine(c, DeviceWeights(c, random_getter(c, 1234, dev), dev, batched_decode=False, fp8_decode=True))
        tpf = (H // fac) * (W // fac)
        ids = build_prompt(c, 32, tpf, 32 * (tpf + 15) + 170)
        g = torch.Generator(device=dev).manual_seed(1)
        frames = torch.randint(0, 256, (32, 3, H, W), generator=g, dtype=torch.uint8, device=dev)
        res = {}
        for flag in (False, True):
            e.fp8_prefill = flag
            best = None
            for _ in range(3):
                out = e.generate([ids], None, frames=frames, max_new_tokens=2, return_margins=False, sync_timings=True)
                best = out.timings["prefill_ms"] if best is None else min(best, out.timings["prefill_ms"])
            res["w8a8" if flag else "bf16"] = round(best, 2)
        tc = c.text
        S = len(ids)
        lin_flop = 2.0 * S * tc.num_hidden_layers * (tc.hidden_size * (tc.num_attention_heads + 2 * tc.num_key_value_heads) * tc.head_dim +
                                                      tc.num_attention_heads * tc.head_dim * tc.hidden_size + 3 * tc.hidden_size * tc.intermediate_size)
        print(json.dumps({"config": f"prefill {name} dims, S={S}", "prefill_ms": res, "speedup": round(res["bf16"] / res["w8a8"], 3),
                          "linear_TFLOP": round(lin_flop / 1e12, 2)}), flush=True)
        # per-shape GEMM rates on this model's widths
        lib = _lib.load()
        M = S
        Hd, QD, I = tc.hidden_size, tc.num_attention_heads * tc.head_dim, tc.intermediate_size
        NQ = (tc.num_attention_heads + 2 * tc.num_key_value_heads) * tc.head_dim
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())
        for tag, N, K in (("qkv", NQ, Hd), ("o_proj", Hd, QD), ("gate/up", 2 * I, Hd), ("down", Hd, I)):
            a16 = torch.randn((M, K), device=dev).to(torch.bfloat16)
            w16 = (torch.randn((N, K), device=dev) / K ** 0.5).to(torch.bfloat16)
            a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
            w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
            sa, sw = torch.ones(M, device=dev), torch.ones(N, device=dev)
            n_out = N // 2 if tag == "gate/up" else N
            epi = 3 if tag == "gate/up" else 0
            o = torch.empty((M, n_out), dtype=torch.bfloat16, device=dev)
            rates = {}
            for kind in ("bf16", "fp8"):
                def call():
                    if kind == "bf16":
                        return lib.o3v_gemm_bf16(P(a16), P(w16), None, None, P(o), M, N, K, K, K, n_out, 0, epi, st)
                    return lib.o3v_gemm_fp8(P(a8), P(sa), P(w8), P(sw), None, None, P(o), M, N, K, K, K, n_out, 0, epi, st)
                assert call() == 0
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(10):
                    call()
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / 10
                rates[kind] = round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1)
            print(json.dumps({"gemm": f"{name} {tag} M={M} N={N} K={K}", "TFLOP_per_s": rates,
                              "frac_of_peak": {"bf16": round(rates["bf16"] / 2500, 3), "fp8": round(rates["fp8"] / 5000, 3)}}), flush=True)
        del e
        torch.cuda.empty_cache()
    sys.exit(0)

if "rollout_long" in which:
    # G = 16 completions behind a 20k-token prompt (256 frames 224x224) -- the prompt's K/V is kept once
    # (kv_cache_GB counts the shared entry + the 16 rows' own tokens; one prompt's K/V at S = 20394 is 1.17 GB)
    cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
    eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
    run("rollout G=16 sampled top_p=0.95 (LONG 256x224x224, S=20394)", 256, 224, 224, 256 * (64 + 15) + 170, 128, num_return_sequences=16,
        do_sample=True, top_p=0.95, temperature=1.0, seed=1)
