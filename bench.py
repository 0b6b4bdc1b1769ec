#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X generate path (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic video: fused frame pipeline (uint8 frames already in HBM
-> normalise/patchify) -> ViT -> merger -> LLM prefill -> greedy grounded-CoT decode of --new-tokens tokens with
EOS suppressed (R:eval/inference_example.py:28 stop_token_ids=[]), Qwen2.5-VL-7B dimensions, 32 frames at the
reference's training resolution 224x420 (SURVEY.md section 8 TRAIN-RES; S = 4490 prompt tokens), repetition
penalty 1.05 (R:eval/models/model_vllm.py:30).  Weights: seeded random bf16 at true 7B dims (no checkpoints
offline; throughput does not depend on weight values).  value = generated tokens / wall time, whole job.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (open_o3_video_amd/launch.py: one child process per GPU,
rendezvous on 127.0.0.1; the parent never touches the GPU); under torchrun (WORLD_SIZE set) it is one of the ranks.
Every rank owns a replica and its own videos (weak scaling, no data-path collective -- the eval path shards by
independent videos); timing is bracketed by barrier + synchronize and reduced with MAX over
ranks.  `n_gpus` in the result line is torch.distributed's world size; a rank-count mismatch exits non-zero.

Second measured leg (`rollout` in the line; BASELINE config #3, R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:402-742):
rank r takes prompt r, samples G = 8 completions (top_p 0.95, T 1) behind one ViT pass and one prefill, computes their
per-token log-probs and the rewards, and the ranks exchange the packed [G, 9+] record with ONE all_gather_into_tensor
per step (RCCL over xGMI when the backend is nccl) -- the collective is inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def build_prompt(cfg, n_frames, tok_per_frame, S_target, seed=1234):
    """Synthetic frames-as-images prompt with the real special ids."""
    g = np.random.default_rng(seed)
    per = 12 + 1 + tok_per_frame + 1 + 1
    fixed = n_frames * per
    pre = max(8, (S_target - fixed) * 7 // 8)
    post = max(4, S_target - fixed - pre)

    def text(n):
        return g.integers(1000, 150000 if cfg.text.vocab_size > 150000 else cfg.text.vocab_size // 2, n).tolist()

    ids = text(pre)
    for _ in range(n_frames):
        ids += text(12) + [cfg.vision_start_token_id] + [cfg.image_token_id] * tok_per_frame + [cfg.vision_end_token_id] + text(1)
    ids += text(post)
    return ids


def cpu_baseline(cfg_dict, n_frames, H, W, S, new_tokens, budget_layers=2):
    """The CPU oracle (oracle/model_ref.py, same torch-CPU ops as the HF bf16 path) timed on this box's host cores on
    a bounded sample: true 7B widths, `budget_layers` of the 32 ViT blocks / 28 LLM layers, 4 frames for the ViT,
    S/8 prompt tokens for prefill, 4 decode steps; per-layer times are scaled to the full depth/length."""
    import copy
    from oracle import model_ref, index_ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_models as fm
    cd = copy.deepcopy(cfg_dict)
    full_v, full_l = cd["vision_config"]["depth"], cd["text_config"]["num_hidden_layers"]
    cd["vision_config"]["depth"] = budget_layers
    cd["vision_config"]["fullatt_block_indexes"] = [budget_layers - 1]
    cd["text_config"]["num_hidden_layers"] = budget_layers
    g = torch.Generator().manual_seed(0)
    W_ = {}
    for name, shape, kind in fm.weight_specs(cd):
        t = torch.empty(shape, dtype=torch.bfloat16)
        if kind == "norm":
            t.fill_(1.0)
        else:
            t.normal_(0, 0.02, generator=g)
        W_[name] = t
    dt = torch.bfloat16
    nf = 4
    frames = torch.randint(0, 256, (nf, 3, H, W), generator=g, dtype=torch.uint8)
    mean = np.asarray(index_ref.CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(index_ref.CLIP_STD, dtype=np.float32)[None, :, None, None]
    xf = ((frames.numpy().astype(np.float64) / 255.0).astype(np.float32) - mean) / std
    pv, grid = index_ref.patchify_frames(xf.astype(np.float32))
    t0 = time.perf_counter()
    with torch.no_grad():
        model_ref.vit_forward(W_, cd, torch.from_numpy(pv), grid, dt)
    t_vit = (time.perf_counter() - t0) * (n_frames / nf) * (full_v / budget_layers)
    Ss = max(64, S // 8)
    tc = cd["text_config"]
    x = torch.randn(1, Ss, tc["hidden_size"], generator=g).to(dt)
    pos = torch.arange(Ss).view(1, 1, Ss).expand(3, 1, Ss).contiguous()
    cache = model_ref.KVCache(budget_layers)
    t0 = time.perf_counter()
    with torch.no_grad():
        model_ref.text_forward(W_, cd, x, pos, torch.ones(1, Ss, dtype=torch.long), cache, dt)
    t_pre_sample = time.perf_counter() - t0
    # linear part scales with S, attention part with S^2: bound from below by linear scaling (favours the CPU)
    t_prefill = t_pre_sample * (S / Ss) * (full_l / budget_layers)
    head = model_ref.lm_head_weight(W_, cd).to(dt)
    steps = 4
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(steps):
            xe = torch.randn(1, 1, tc["hidden_size"], generator=g).to(dt)
            p = torch.full((3, 1, 1), Ss + i)
            h = model_ref.text_forward(W_, cd, xe, p, torch.ones(1, Ss + i + 1, dtype=torch.long), cache, dt)
            torch.nn.functional.linear(h[:, -1], head).float().argmax(-1)
    t_step_sample = (time.perf_counter() - t0) / steps
    # per-step: layers scale with depth, lm_head does not; estimate head share by timing it alone
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(2):
            torch.nn.functional.linear(h[:, -1], head)
    t_head = (time.perf_counter() - t0) / 2
    t_step = (t_step_sample - t_head) * (full_l / budget_layers) + t_head
    total = t_vit + t_prefill + new_tokens * t_step
    return {
        "value": round(new_tokens / total, 4), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"oracle/model_ref.py (torch-CPU bf16, same ops as HF transformers Qwen2_5_VL) at true 7B widths: "
                   f"{budget_layers}/{full_v} ViT blocks on {nf}/{n_frames} frames, {budget_layers}/{full_l} LLM layers on "
                   f"{Ss}/{S} prompt tokens, {steps} decode steps + lm_head; scaled linearly to full depth/length "
                   f"(est. vit {t_vit:.1f}s prefill {t_prefill:.1f}s step {t_step * 1e3:.0f}ms)"),
    }


def kernel_roofline(eng, reps=3):
    """Dominant kernel of the decode loop = the weight-streaming GEMV (csrc/o3v_gemm.hip gemv_bf16_kernel); its
    largest instance is the fused gate/up projection: algorithmic bytes per launch = 2*I*H*2 B of weights (+x, out).
    Timed with HIP events on the launch stream, rotating over all layers so every launch streams cold weights."""
    import ctypes as C
    from open_o3_video_amd import _lib
    tc = eng.cfg.text
    H, I, L = tc.hidden_size, tc.inter_pad, tc.num_hidden_layers
    x = torch.randn(1, H, device=eng.dev).to(torch.bfloat16)
    out = torch.empty(1, I, dtype=torch.bfloat16, device=eng.dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        for l in range(L):
            _lib.call("o3v_gemv_norm_bf16", C.c_void_p(x.data_ptr()), C.c_void_p(eng.w.t[f"l{l}.ln2"].data_ptr()),
                      float(tc.rms_norm_eps), C.c_void_p(eng.w.t[f"l{l}.gu_w"].data_ptr()), None, None,
                      C.c_void_p(out.data_ptr()), 1, 2 * I, H, H, H, I, 0, _lib.EPI_SWIGLU, st)
    for _ in range(3):   # the first passes after other work run 5-10 % slower (46 -> 42 us; clocks / caches settle): not timed
        run()
    torch.cuda.synchronize()
    per_pass = []
    for _ in range(max(3, reps)):      # every pass = L launches, each streaming its own layer's (cold) weights
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        per_pass.append(e0.elapsed_time(e1) / L)
    avg_ms = sorted(per_pass)[len(per_pass) // 2]   # median pass
    bytes_per_launch = 2 * I * H * 2 + H * 2 + I * 2
    ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = None  # HBM bytes per launch from the PMC passes (profiles/r03_gemv_pmc.json; FETCH_SIZE x2 on gfx950)
    try:
        with open(os.path.join(ROOT, "profiles", "r03_gemv_pmc.json")) as f:
            pm = json.load(f)
        if pm.get("algorithmic_bytes_per_launch") == bytes_per_launch:
            traffic = pm["hbm_bytes_per_launch"]
    except OSError:
        pass
    return {"bound": "hbm", "kernel": "gemv_bf16_kernel<M=1,R=2,KS=1,SWIGLU,NORM> (RMSNorm + LLM gate/up projection, decode)",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": "profiles/r03_gemv_pmc.json (rocprofv3 --pmc passes of this kernel on this shape; "
                                                  "not a counter of this run)" if traffic is not None else None,
            "bytes_per_launch": bytes_per_launch, "avg_launch_us": round(avg_ms * 1e3, 2)}


def fp8_gemm_roofline(tc, S, dev):
    """o3v_gemm_fp8 on the gate/up shape of a layer (M = S prompt tokens): 10 launches between HIP events on the launch stream."""
    import ctypes as C
    from open_o3_video_amd import _lib
    lib = _lib.load()
    M, N, K = int(S), 2 * tc.intermediate_size, tc.hidden_size
    a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
    w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
    sa, sw = torch.ones(M, device=dev), torch.ones(N, device=dev)
    o = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    call = lambda: lib.o3v_gemm_fp8(P(a8), P(sa), P(w8), P(sw), None, None, P(o), M, N, K, K, K, N // 2, 0, 3, st)
    assert call() == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(10):
        call()
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 10 * 1e3
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "gemm256_fp8_kernel<SWIGLU> (v_mfma_scale_f32_16x16x128_f8f6f4; LLM gate/up projection of the W8A8 prefill)",
            "achieved": round(tf, 1), "peak": 5000.0, "unit": "TFLOP/s", "frac": round(tf / 5000.0, 4), "flop_per_launch": 2.0 * M * N * K,
            "avg_launch_us": round(us, 1), "traffic": None}


def cpu_baseline_config1(new_tokens=32):
    """BASELINE config #1 timed IN FULL on the host cores: Qwen2.5-VL-3B dims (32 ViT blocks, 36 LLM layers, tied
    embeddings), 4 frames 364x644 (4784 patches -> 1196 visual tokens), the whole prompt, `new_tokens` greedy decode steps,
    through oracle/model_ref.py (the same torch-CPU bf16 ops as the HF path).  Nothing is extrapolated."""
    from oracle import model_ref, index_ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_models as fm
    from open_o3_video_amd.config import O3VConfig, qwen25vl_3b_dict
    cd = qwen25vl_3b_dict()
    cfg = O3VConfig.from_dict(cd)
    g = torch.Generator().manual_seed(0)
    W_ = {}
    for name, shape, kind in fm.weight_specs(cd):
        t = torch.empty(shape, dtype=torch.bfloat16)
        if kind == "norm":
            t.fill_(1.0)
        else:
            t.normal_(0, 0.02, generator=g)
        W_[name] = t
    nf, H, W = 4, 364, 644
    tpf = (H // 28) * (W // 28)
    ids = np.asarray([build_prompt(cfg, nf, tpf, nf * (tpf + 15) + 170)], dtype=np.int64)
    frames = torch.randint(0, 256, (nf, 3, H, W), generator=g, dtype=torch.uint8)
    mean = np.asarray(index_ref.CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(index_ref.CLIP_STD, dtype=np.float32)[None, :, None, None]
    xf = ((frames.numpy().astype(np.float64) / 255.0).astype(np.float32) - mean) / std
    pv, grid = index_ref.patchify_frames(xf.astype(np.float32))
    t0 = time.perf_counter()
    with torch.no_grad():
        out = model_ref.generate(W_, cd, ids, None, torch.from_numpy(pv), grid, new_tokens, dtype=torch.bfloat16,
                                 pad_token_id=cd["pad_token_id"], eos_token_ids=())
    dt = time.perf_counter() - t0
    assert out.shape[1] == ids.shape[1] + new_tokens
    return {"value": round(new_tokens / dt, 4), "unit": "tokens/s (end to end: ViT + prefill + decode)", "wall_s": round(dt, 2),
            "sample": f"config #1 in full: 3B dims, {nf} frames {H}x{W}, S={ids.shape[1]}, {new_tokens} greedy tokens, nothing scaled"}


# reward functions of the rollout leg: the product's own seven (R:src/r1-v/src/open_r1/grpo.py:58-66) on the decoded text
def _rollout_rewards():
    from open_o3_video_amd import rewards
    rewards.ALLOW_APPROX_ROUGE = True   # rouge_score is not installable offline; the values do not enter the timing
    return [rewards.REWARD_FUNCS[k] for k in rewards.REWARD_FUNCS]


def rollout_leg(cfg, eng, ids, pixel_values, grid, dist, dev, G=8, T=768, steps=2, warmup=1):
    """GSPO rollout steps (BASELINE config #3): rank r's prompt, G sampled completions, log-probs, rewards, ONE all_gather
    of the packed per-sample record per step -- all inside the timed region."""
    from open_o3_video_amd.hf_api import Qwen2_5_VLForConditionalGeneration
    from open_o3_video_amd.rollout import GroupRollout
    model = Qwen2_5_VLForConditionalGeneration(cfg, eng)

    def decode(comp):   # no tokenizer offline: a deterministic stand-in text per completion
        return ["<think>" + " ".join(str(int(t)) for t in row[:32]) + "</think><answer>A</answer>" for row in comp.cpu().tolist()]

    ro = GroupRollout(model, _rollout_rewards(), decode, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                      num_generations=G, max_completion_length=T)
    inputs = dict(input_ids=torch.tensor([ids]), attention_mask=torch.ones(1, len(ids), dtype=torch.long),
                  pixel_values=pixel_values, image_grid_thw=torch.from_numpy(grid))
    # one dataset row with every column the seven rewards read (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:455-469, :648-656)
    example = {"prompt": "p", "task": "temporal-spatial free-form QA", "answer": "<answer>A</answer>", "step_percent": 0.5,
               "image_size": (420, 224), "image_size_refine": (420, 224), "video_sample_fps": 1.0,
               "key_frames": [{"idx": 3, "time": 3.0}], "key_items": {"3": {"person": [[10, 10, 100, 100]]}}}
    n_tok = 0
    for i in range(warmup):
        ro.step(inputs, example)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        res = ro.step(inputs, example)
        n_tok += int(res.completion_mask.sum().item())
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tok = torch.tensor([float(n_tok), dt], device=dev, dtype=torch.float64)
    world = 1
    if dist:
        world = dist.get_world_size()
        tmax = tok[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tok[:1].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        n_all, dt = float(tsum.item()), float(tmax.item())
    else:
        n_all = float(n_tok)
    return {"G": G, "max_completion_length": T, "steps": steps, "prompts_per_step": world, "tokens_per_s": round(n_all / dt, 1),
            "ms_per_step": round(dt / steps * 1e3, 1), "record_columns": int(res.rewards_per_func.shape[1]) + 4,
            "collective": "one all_gather_into_tensor of the packed [G, n_rewards+4] f32 record per step"
                          + ("" if dist else " (skipped: no process group)"),
            "backend": dist.get_backend() if dist else None}


def launch_check():
    """CPU-only check of the launch path (tests/test_bench_launch_cpu.py): join the world over gloo, run the rollout's
    metrics collective on a dummy record, let rank 0 print the world size."""
    import torch.distributed as dist
    from open_o3_video_amd import dist as o3v_dist
    rank, world = o3v_dist.init("gloo")
    rec = torch.full((8, 11), float(rank))
    allr = o3v_dist.all_gather_records(rec)
    assert allr.shape == (8 * world, 11) and float(allr[-1, 0]) == world - 1
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size() if world > 1 else 1, "backend": "gloo"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--res", default="train", choices=["train", "eval"])
    ap.add_argument("--model", default="7b", choices=["7b", "3b"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra 16-videos-per-step throughput measurement")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the GSPO rollout leg (G=8 sampled + log-probs + all_gather)")
    ap.add_argument("--rollout-tokens", type=int, default=768)
    ap.add_argument("--launch-check", action="store_true", help="CPU-only: join the world over gloo and print its size")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra fp8-weight decode measurement")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # started as `python bench.py --gpus N`: become the launcher (no GPU call in this process)
        from open_o3_video_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, [os.path.abspath(__file__), *sys.argv[1:]]))
    if args.launch_check:
        launch_check()
        return

    # stdout carries exactly ONE line, the result: libraries that write to file descriptor 1 (RCCL prints a version banner there
    # when a communicator is created) are sent to stderr for the whole run; the result line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal on a one-GPU box: O3V_BENCH_ONE_DEVICE=1 puts every rank on cuda:0; RCCL refuses two ranks on one device,
    # so the exchange then runs over gloo and the line says so
    one_device = os.environ.get("O3V_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    backend = os.environ.get("O3V_DIST_BACKEND", "gloo" if one_device else "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL on ROCm
        else:
            dist_.init_process_group(backend)
        dist = dist_
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"world size {dist.get_world_size()} != --gpus {args.gpus}")
    elif backend == "nccl" and os.environ.get("O3V_BENCH_SOLO_GROUP", "1") != "0":
        # N = 1: a one-rank RCCL group, so that the rollout leg's all_gather / barriers run through the library the N = 8 run uses
        try:
            import torch.distributed as dist_
            from open_o3_video_amd.launch import free_port
            dist_.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                     device_id=torch.device("cuda", local_rank))
            dist = dist_
        except Exception as e:   # the headline does not depend on it
            print(f"[bench] one-rank nccl group not available ({type(e).__name__}: {e}); the rollout leg skips its collective", file=sys.stderr)
            dist = None

    from open_o3_video_amd.config import O3VConfig, qwen25vl_3b_dict, qwen25vl_7b_dict
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, random_getter

    cfg_dict = qwen25vl_7b_dict() if args.model == "7b" else qwen25vl_3b_dict()
    cfg = O3VConfig.from_dict(cfg_dict)
    dev = torch.device("cuda", local_rank)
    eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=not (args.no_batched and args.no_rollout)))
    Hres, Wres = (224, 420) if args.res == "train" else (364, 644)
    tpf = (Hres // 28) * (Wres // 28)
    S = 4490 if args.res == "train" else 10218
    if args.frames != 32:
        S = args.frames * (tpf + 15) + 170
    ids = build_prompt(cfg, args.frames, tpf, S)
    S = len(ids)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    videos = [torch.randint(0, 256, (args.frames, 3, Hres, Wres), generator=gen, dtype=torch.uint8, device=dev)
              for _ in range(2)]  # resident in HBM before the timed region

    def step(i, timings=False):
        return eng.generate([ids], None, frames=videos[i % len(videos)], max_new_tokens=args.new_tokens, eos_token_ids=(),
                            repetition_penalty=1.05, return_margins=False, sync_timings=timings)

    def reduce_max(x):
        if not dist:
            return x
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = reduce_max(time.perf_counter() - t0)
    assert out.sequences.shape[1] == S + args.new_tokens

    # per-stage breakdown (one extra, untimed, synchronised step)
    stages = step(0, timings=True).timings

    # second measured leg: GSPO rollout with the metrics all_gather inside the timed region
    roll = None
    if not args.no_rollout:
        pv = torch.rand((args.frames * (Hres // 14) * (Wres // 14), 1176), generator=gen, device=dev) * 4 - 2   # processor-shaped pixel_values
        grid = np.asarray([[1, Hres // 14, Wres // 14]] * args.frames, dtype=np.int64)
        roll = rollout_leg(cfg, eng, ids, pv, grid, dist, dev, T=args.rollout_tokens)
        roll["backend"] = backend if dist else None
        del pv

    # extra (not `value`): throughput with 16 videos decoding together per GPU -- the eval path of the reference runs a
    # vLLM engine with max_num_seqs=5 (R:eval/models/model_vllm.py:23), i.e. it batches concurrent requests too.  Decode is
    # weight-bandwidth-bound, so the 16 sequences share every streamed weight byte (MFMA skinny-GEMM path).
    # The rank-0-only extras below (batched videos, fp8 rows, the CPU baseline) run only at N = 1: under
    # --gpus N the other ranks would sit in the final barrier for tens of seconds while rank 0 measures them.
    world_n = dist.get_world_size() if dist else 1
    solo = world_n == 1
    batched = None
    if not args.no_batched and rank == 0 and solo:
        NB = O3VEngine.MAX_ROWS
        vids8 = torch.randint(0, 256, (NB * args.frames, 3, Hres, Wres), generator=gen, dtype=torch.uint8, device=dev)

        def step8():
            return eng.generate([ids] * NB, None, frames=vids8, max_new_tokens=args.new_tokens, eos_token_ids=(),
                                repetition_penalty=1.05, return_margins=False)
        step8()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        o8 = step8()
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        assert o8.sequences.shape == (NB, S + args.new_tokens)
        batched = {"videos_per_step": NB, "tokens_per_s_per_gpu": round(NB * args.new_tokens / tb, 1),
                   "videos_per_min_per_gpu": round(NB / tb * 60.0, 1), "ms_per_step": round(tb * 1e3, 1)}
    roof = None if (args.no_roofline or rank != 0) else kernel_roofline(eng)   # a fraction of a second: kept at every N
    fused_flag = eng.fused_decode
    wbytes = sum(eng.w.t[f"l{l}.{k}"].numel() * 2 for l in range(cfg.text.num_hidden_layers) for k in ("qkv_w", "o_w", "gu_w", "down_w"))
    wbytes += eng.w.t["l.head"].numel() * 2

    # extra (not `value`, which stays bf16): the same decode on fp8 (OCP e4m3fn) weight rows with per-row scales -- BASELINE
    # config #5's weight format; half the weight bytes per step
    fp8 = None
    if not args.no_fp8 and rank == 0 and solo:
        del eng
        torch.cuda.empty_cache()
        eng8 = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=True, fp8_decode=True))
        kw8 = dict(max_new_tokens=args.new_tokens, eos_token_ids=(), repetition_penalty=1.05, return_margins=False)
        eng8.generate([ids], None, frames=videos[0], **kw8)
        t8 = eng8.generate([ids], None, frames=videos[1], sync_timings=True, **kw8).timings
        tc8 = cfg.text
        wb8 = sum(eng8.w.t[f"l{l}.{k}8"].numel() for l in range(tc8.num_hidden_layers) for k in ("qkv_w", "o_w", "gu_w", "down_w"))
        wb8 += eng8.w.t["l.head8"].numel()
        kvb = 2 * tc8.num_hidden_layers * tc8.num_key_value_heads * tc8.head_dim * 2 * (S + args.new_tokens / 2)
        ms8 = t8["decode_ms"] / args.new_tokens
        fp8 = {"weights": "fp8 e4m3fn rows + f32 power-of-two scale per output row, activations / KV / accumulation as in bf16",
               "decode_ms_per_step": round(ms8, 4), "decode_tokens_per_s": round(1e3 / ms8, 1),
               "algorithmic_bytes": int(wb8 + kvb), "achieved_GBps": round((wb8 + kvb) / (ms8 * 1e-3) / 1e9, 1),
               "frac_of_8TBps": round((wb8 + kvb) / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # the same rows under a G = 8 completion group (4..32 decode rows stream fragment-major fp8 images on the matrix cores)
        kwg = dict(max_new_tokens=256, num_return_sequences=8, do_sample=True, top_p=0.95, temperature=1.0, seed=1, eos_token_ids=(),
                   return_margins=False)
        eng8.generate([ids], None, frames=videos[0], **kwg)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        og = eng8.generate([ids], None, frames=videos[1], sync_timings=True, **kwg)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg
        fp8["group_g8"] = {"new_tokens": 256, "tokens_per_s": round(8 * 256 / tg, 1),
                           "decode_ms_per_step": round(og.timings["decode_ms"] / 256, 3)}
        # opt-in W8A8 prefill (fp8 x fp8 on the matrix cores, per-token activation scales): prefill time against the bf16 prefill of
        # the same engine, and the roofline of its GEMM on the layer's gate/up shape against the dense fp8 peak
        pre = {}
        for flag in (False, True):
            eng8.fp8_prefill = flag
            best = None
            for _ in range(2):
                tt = eng8.generate([ids], None, frames=videos[0], max_new_tokens=2, return_margins=False, sync_timings=True).timings
                best = tt["prefill_ms"] if best is None else min(best, tt["prefill_ms"])
            pre["w8a8" if flag else "bf16"] = round(best, 2)
        eng8.fp8_prefill = False
        fp8["prefill_ms"] = pre
        fp8["prefill_speedup"] = round(pre["bf16"] / pre["w8a8"], 3)
        fp8["roofline_fp8_gemm"] = fp8_gemm_roofline(tc8, S, dev)
        del eng8

    if rank == 0:
        n_gpus = dist.get_world_size() if dist else 1
        total_tokens = n_gpus * args.steps * args.new_tokens
        ms_per_step = dt / args.steps * 1e3
        tc = cfg.text
        kv_bytes = 2 * tc.num_hidden_layers * tc.num_key_value_heads * tc.head_dim * 2 * (S + args.new_tokens / 2)
        dec_ms = stages.get("decode_ms", 0.0) / max(1, args.new_tokens)
        par = f"dp{n_gpus} (replica per GPU, independent videos, no data-path collective in the headline leg)"
        if dist:
            par += f"; torch.distributed backend {backend}" + (" -- REHEARSAL: every rank on cuda:0" if one_device else " (RCCL)")
        rec = {
            "metric": "grounded_cot_tokens_per_sec", "value": round(total_tokens / dt, 2), "unit": "tokens/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Qwen2.5-VL-{args.model.upper()} dims, {args.frames} frames {Hres}x{Wres} -> "
                                   f"{args.frames * tpf} visual tokens, prompt S={S}, greedy decode {args.new_tokens} new tokens "
                                   f"(EOS suppressed, repetition_penalty 1.05), batch 1 per GPU, random-init bf16 weights",
                       "parallelism": par},
            "videos_per_min": round(n_gpus * args.steps / dt * 60.0, 2),
            "stage_ms": {k: round(v, 2) for k, v in stages.items() if k.endswith("_ms") or k == "prefix_tokens_reused"},
            "decode_tokens_per_sec_per_gpu": round(1e3 / dec_ms, 1) if dec_ms else None,
            "decode_step_hbm": {"algorithmic_bytes": int(wbytes + kv_bytes), "ms": round(dec_ms, 4),
                                "achieved_GBps": round((wbytes + kv_bytes) / (dec_ms * 1e-3) / 1e9, 1) if dec_ms else None,
                                "frac_of_8TBps": round((wbytes + kv_bytes) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms else None,
                                "launches_per_layer": stages.get("launches_per_layer"),
                                "fused_attention_layers": stages.get("fused_attention_layers"),
                                "standalone_attention_layers": stages.get("standalone_attention_layers")},
        }
        if roll:
            rec["rollout"] = roll
        if batched:
            rec["batched_videos"] = batched
        if roof:
            rec["roofline"] = roof
        if fp8:
            rec["fp8_decode"] = fp8
        if not args.no_cpu_baseline and n_gpus == 1:
            rec["cpu_baseline"] = cpu_baseline(cfg_dict, args.frames, Hres, Wres, S, args.new_tokens)
            rec["cpu_baseline"]["config1_full"] = cpu_baseline_config1()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(rec) + "\n").encode())
    os.close(result_fd)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
