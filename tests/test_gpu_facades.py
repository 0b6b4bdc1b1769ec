"""GPU tests of the reference-shaped call surfaces: HF-style generate()/__call__ (R:grpo_trainer.py:581-586,:375),
vLLM-style LLM.generate (R:eval/inference_example.py:15-29,81), local safetensors loading with hub weight names,
and one GSPO group-rollout step."""
import json
import os

import numpy as np
import pytest
import torch

import fixture_models as fm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


class StubTokenizer:
    """Whitespace tokenizer over 'w<ID>' words plus the special tags (no real tokenizer files exist offline)."""
    specials = {"<|vision_start|>": "vision_start_token_id", "<|image_pad|>": "image_token_id", "<|vision_end|>": "vision_end_token_id"}

    def __init__(self, cfg):
        self.cfg = cfg

    def encode(self, text, add_special_tokens=False):
        for s in self.specials:
            text = text.replace(s, f" {s} ")
        out = []
        for w in text.split():
            out.append(self.cfg[self.specials[w]] if w in self.specials else int(w[1:]))
        return out

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(f"w{int(i)}" for i in ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        return [self.decode(r) for r in ids.tolist()]


def test_hf_facade_generate_and_logits(need_gpu, golden_dir):
    from open_o3_video_amd.hf_api import GenerationConfigLike, Qwen2_5_VLForConditionalGeneration, Qwen2VLForConditionalGeneration
    assert Qwen2VLForConditionalGeneration is Qwen2_5_VLForConditionalGeneration
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    model = Qwen2_5_VLForConditionalGeneration.from_state_dict(cfg, fm.make_weights(cfg, 2))
    assert model.eval() is model and hasattr(model, "warnings_issued") and hasattr(model.config, "_name_or_path")
    prompt_inputs = dict(input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.ones_like(torch.from_numpy(g["input_ids"])),
                         pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=torch.from_numpy(g["grid"]))
    gc = GenerationConfigLike(max_new_tokens=16, do_sample=False, num_return_sequences=1, pad_token_id=cfg["pad_token_id"],
                              eos_token_id=None)
    out = model.generate(**prompt_inputs, generation_config=gc)
    assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), g["bf16_ids"])
    # two prompts (the same one twice) x G=2 -> rows b*G+g, all equal under greedy decoding
    two = {k: (torch.cat([v, v]) if k != "image_grid_thw" else torch.cat([v, v])) for k, v in prompt_inputs.items()}
    gc2 = GenerationConfigLike(max_new_tokens=6, do_sample=False, num_return_sequences=2, pad_token_id=cfg["pad_token_id"], eos_token_id=None)
    out2 = model.generate(**two, generation_config=gc2)
    assert out2.shape[0] == 4 and all(torch.equal(out2[i], out[0, : out2.shape[1]]) for i in range(4))
    # forward logits + the reference's log-prob recipe on them
    lg = model(out, attention_mask=torch.ones_like(out), pixel_values=prompt_inputs["pixel_values"],
               image_grid_thw=prompt_inputs["image_grid_thw"]).logits
    assert lg.shape == (1, out.shape[1], cfg["text_config"]["vocab_size"])
    S = g["input_ids"].shape[1]
    assert torch.equal(lg[0, S - 1:-1].float().argmax(-1).cpu(), out[0, S:].cpu())  # teacher-forced argmax == greedy ids
    ref_lp = torch.log_softmax(lg[:, :-1].float(), -1).gather(2, out[:, 1:, None].to(lg.device))[..., 0]
    lp = model.per_token_logps(out, torch.ones_like(out), prompt_inputs["pixel_values"], prompt_inputs["image_grid_thw"])
    assert torch.allclose(lp, ref_lp, atol=2e-4, rtol=0)
    with pytest.raises(ValueError):
        model.generate(input_ids=prompt_inputs["input_ids"], pixel_values=prompt_inputs["pixel_values"][:-4],
                       image_grid_thw=prompt_inputs["image_grid_thw"], generation_config=gc)


def test_from_pretrained_local_dir_with_hub_names(need_gpu, tmp_path, golden_dir):
    """config.json + model.safetensors with the hub's weight names (visual.*, model.layers.*) load identically."""
    from safetensors.torch import save_file
    from open_o3_video_amd.hf_api import Qwen2_5_VLForConditionalGeneration
    cfg = fm.tiny_config()
    W = fm.make_weights(cfg, 0)
    hub = {}
    for k, v in W.items():
        k2 = k.replace("model.visual.", "visual.").replace("model.language_model.", "model.")
        hub[k2] = v.to(torch.bfloat16).contiguous()
    flat = dict(cfg["text_config"], **{k: v for k, v in cfg.items() if k != "text_config"})
    flat["rope_scaling"] = {"type": "mrope", "mrope_section": flat.pop("mrope_section")}
    (tmp_path / "config.json").write_text(json.dumps(flat))
    save_file(hub, str(tmp_path / "model.safetensors"))
    model = Qwen2_5_VLForConditionalGeneration.from_pretrained(str(tmp_path), torch_dtype=torch.bfloat16,
                                                               attn_implementation="flash_attention_2", use_cache=False)
    g = np.load(os.path.join(golden_dir, "g6_tiny.npz"))
    out = model.generate(input_ids=torch.from_numpy(g["input_ids"]), pixel_values=torch.from_numpy(g["pixel_values"]),
                         image_grid_thw=torch.from_numpy(g["grid"]), max_new_tokens=16, do_sample=False, eos_token_id=None,
                         pad_token_id=cfg["pad_token_id"])
    assert np.array_equal(out.cpu().numpy(), g["bf16_ids"])
    with pytest.raises(OSError):
        Qwen2_5_VLForConditionalGeneration.from_pretrained("Qwen/Qwen2.5-VL-7B-Instruct")  # never downloads
    # vLLM's quantization="fp8": fp8 decode rows are built on load from the bf16 checkpoint
    from open_o3_video_amd.vllm_api import LLM
    (tmp_path / "preprocessor_config.json").write_text(json.dumps({"size": {"shortest_edge": 3136, "longest_edge": 200704}}))
    llm = LLM(model=str(tmp_path), tokenizer=StubTokenizer(cfg), quantization="fp8")
    assert llm.engine.w.fp8_decode and llm.engine.w.llm.lm_head8
    assert (llm.min_pixels, llm.max_pixels) == (3136, 200704)       # transformers 5.x layout of the processor's pixel budget
    with pytest.raises(ValueError):
        LLM(model=str(tmp_path), tokenizer=StubTokenizer(cfg), quantization="awq")


def test_vllm_facade(need_gpu, golden_dir):
    from open_o3_video_amd.vllm_api import LLM, SamplingParams
    from test_gpu_model import build_engine
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    llm = LLM(engine=eng, tokenizer=StubTokenizer(cfg), limit_mm_per_prompt={"image": 32}, max_model_len=4096)
    # rebuild the prompt text with ONE <|image_pad|> per frame, as the eval scripts do
    words, ids = [], g["input_ids"][0].tolist()
    i = 0
    while i < len(ids):
        if ids[i] == cfg["image_token_id"]:
            words.append("<|image_pad|>")
            while i < len(ids) and ids[i] == cfg["image_token_id"]:
                i += 1
            continue
        words.append({cfg["vision_start_token_id"]: "<|vision_start|>", cfg["vision_end_token_id"]: "<|vision_end|>"}.get(ids[i], f"w{ids[i]}"))
        i += 1
    prompt = " ".join(words)
    frames = torch.from_numpy(g["frames"])
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=16, stop_token_ids=[])
    outs = llm.generate([{"prompt": prompt, "multi_modal_data": {"image": frames}}], sampling_params=sp)
    exp = g["bf16_ids_rp105"][0, g["input_ids"].shape[1]:].tolist()
    assert outs[0].prompt_token_ids == ids
    assert outs[0].outputs[0].token_ids == exp and outs[0].outputs[0].text == " ".join(f"w{t}" for t in exp)
    # float frames (what process_vision_info returns for a video) and a list of HWC arrays give the same tokens
    o2 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": frames.float()}}, sp)
    o3 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": [f.permute(1, 2, 0).numpy() for f in frames]}}, sp)
    assert o2[0].outputs[0].token_ids == exp and o3[0].outputs[0].token_ids == exp
    assert llm.vis_cache_hits >= 1  # identical uint8 frames were encoded once (cross-prompt visual reuse)
    # stop token: generation ends there and the stop token is not part of the text
    assert exp[3] not in exp[:3]
    sp2 = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=16, stop_token_ids=[exp[3]])
    o4 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": frames}}, sp2)[0].outputs[0]
    assert o4.token_ids == exp[:3] and o4.finish_reason == "stop"
    with pytest.raises(ValueError):
        llm.generate({"prompt": prompt.replace("<|image_pad|>", "", 1), "multi_modal_data": {"image": frames}}, sp)
    # sampled request with n completions
    o5 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": frames}},
                      SamplingParams(temperature=0.7, top_p=0.9, repetition_penalty=1.05, max_tokens=8, n=3, seed=3))
    assert len(o5[0].outputs) == 3
    # prompts over identical frames reused their common prefix K/V (o3, o4 and o5 repeat the first prompt)
    assert llm.prefix_tokens_reused >= 3 * (len(ids) - 1)
    # n > 8 samples run as groups of <= 8 rows; sample i depends on (seed, i) only
    sp10 = SamplingParams(temperature=1.0, top_p=0.95, repetition_penalty=1.05, max_tokens=6, n=10, seed=11)
    o6 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": frames}}, sp10)[0].outputs
    sp4 = SamplingParams(temperature=1.0, top_p=0.95, repetition_penalty=1.05, max_tokens=6, n=4, seed=11)
    o7 = llm.generate({"prompt": prompt, "multi_modal_data": {"image": frames}}, sp4)[0].outputs
    assert [o.index for o in o6] == list(range(10)) and len({tuple(o.token_ids) for o in o6}) > 1
    assert all(o6[i].token_ids == o7[i].token_ids for i in range(4))
    # several requests in one call are decoded together (left padded, one weight stream for all rows): same greedy tokens as
    # one at a time; a second, shorter prompt over other frames rides along
    fr2 = fm.make_frames(2, 56, 84, seed=21)
    prompt2 = " ".join(["w7", "<|vision_start|>", "<|image_pad|>", "<|vision_end|>"] * 2 + ["w33", "w34"])
    solo2 = llm.generate({"prompt": prompt2, "multi_modal_data": {"image": fr2}}, sp)[0].outputs[0].token_ids
    both = llm.generate([{"prompt": prompt, "multi_modal_data": {"image": frames}},
                         {"prompt": prompt2, "multi_modal_data": {"image": fr2}},
                         {"prompt": prompt, "multi_modal_data": {"image": frames}}], sp)
    assert [len(b.outputs) for b in both] == [1, 1, 1] and both[0].prompt_token_ids == ids
    assert both[0].outputs[0].token_ids == exp and both[2].outputs[0].token_ids == exp and both[1].outputs[0].token_ids == solo2
    assert [b.request_id for b in both] == [str(int(both[0].request_id) + i) for i in range(3)]
    # switching the reuse off gives the same greedy tokens
    llm2 = LLM(engine=eng, tokenizer=StubTokenizer(cfg), limit_mm_per_prompt={"image": 32}, max_model_len=4096,
               enable_prefix_caching=False)
    o8 = llm2.generate([{"prompt": prompt, "multi_modal_data": {"image": frames}}] * 2, sampling_params=sp)
    assert o8[0].outputs[0].token_ids == exp and o8[1].outputs[0].token_ids == exp and llm2.prefix_tokens_reused == 0


def test_group_rollout_step(need_gpu, golden_dir):
    from open_o3_video_amd import rewards, rollout
    from open_o3_video_amd.hf_api import Qwen2_5_VLForConditionalGeneration
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    model = Qwen2_5_VLForConditionalGeneration.from_state_dict(cfg, fm.make_weights(cfg, 2))
    ref = Qwen2_5_VLForConditionalGeneration.from_state_dict(cfg, fm.make_weights(cfg, 9))
    tok = StubTokenizer(cfg)
    G, T = 4, 10
    gr = rollout.GroupRollout(model, [rewards.format_reward, lambda completions, **kw: [float(len(c[0]["content"]) % 3) for c in completions]],
                              tok.batch_decode, eos_token_id=cfg["eos_token_id"], pad_token_id=cfg["pad_token_id"], ref_model=ref,
                              num_generations=G, max_completion_length=T)
    ids = torch.from_numpy(g["input_ids"])
    res = gr.step(dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=torch.from_numpy(g["pixel_values"]),
                       image_grid_thw=torch.from_numpy(g["grid"])), {"prompt": "p", "task": "temporal-spatial free-form QA"})
    S = ids.shape[1]
    assert res.prompt_completion_ids.shape[0] == G and res.completion_ids.shape[1] <= T
    assert torch.equal(res.prompt_completion_ids[:, :S].cpu(), ids.repeat(G, 1))
    assert res.per_token_logps.shape == res.completion_ids.shape == res.ref_per_token_logps.shape
    assert (res.per_token_logps <= 0).all() and (res.per_token_kl >= 0).all()
    assert res.rewards_per_func.shape == (G, 2) and torch.equal(res.rewards, res.rewards_per_func.sum(1))
    assert torch.isfinite(res.loss) and set(res.metrics) >= {"completion_length", "reward", "reward_std", "kl", "all_wrong", "all_correct"}
    # policy log-probs of the sampled tokens agree with a full forward through the logits facade
    lg = model(res.prompt_completion_ids, pixel_values=torch.from_numpy(g["pixel_values"]).repeat(G, 1),
               image_grid_thw=torch.from_numpy(g["grid"]).repeat(G, 1)).logits
    ref_lp = torch.log_softmax(lg[:, :-1].float(), -1).gather(2, res.prompt_completion_ids[:, 1:, None].to(lg.device))[..., 0][:, S - 1:]
    assert torch.allclose(res.per_token_logps, ref_lp, atol=2e-4, rtol=0)


def _gp_step(golden_dir, group_parallel, G=8, T=8):
    from open_o3_video_amd import rewards, rollout
    from open_o3_video_amd.hf_api import Qwen2_5_VLForConditionalGeneration
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    model = Qwen2_5_VLForConditionalGeneration.from_state_dict(cfg, fm.make_weights(cfg, 2))
    tok = StubTokenizer(cfg)
    gr = rollout.GroupRollout(model, [rewards.format_reward, lambda completions, **kw: [float(sum(map(ord, c[0]["content"])) % 7) for c in completions]],
                              tok.batch_decode, eos_token_id=cfg["eos_token_id"], pad_token_id=cfg["pad_token_id"],
                              num_generations=G, max_completion_length=T, group_parallel=group_parallel)
    ids = torch.from_numpy(g["input_ids"])
    res = gr.step(dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=torch.from_numpy(g["pixel_values"]),
                       image_grid_thw=torch.from_numpy(g["grid"])), {"prompt": "p", "task": "temporal-spatial free-form QA"})
    return dict(pc=res.prompt_completion_ids.cpu(), mask=res.completion_mask.cpu(), lp=res.per_token_logps.cpu(),
                rewards=res.rewards.cpu(), adv=res.advantages.cpu(), loss=float(res.loss))


def _gp_worker(rank, world, port, golden_dir, q):
    import torch.distributed as td
    from open_o3_video_amd import dist as od
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    od.init("gloo")                      # both ranks share the one GPU of the test box; the exchange runs over gloo
    out = _gp_step(golden_dir, True)
    q.put((rank, {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in out.items()}))
    td.destroy_process_group()


def test_group_parallel_rollout_two_ranks(need_gpu, golden_dir):
    """BASELINE config #3 / SURVEY 8e partitioning B on the real engine: the G=8 completions of one prompt decoded as 2 x 4
    rows by two ranks and all-gathered equal the group decoded by one rank -- same sampled ids (the sampler is keyed by the
    global completion index), same rewards, advantages and loss."""
    import socket
    import torch.multiprocessing as mp
    ref = _gp_step(golden_dir, False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gp_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in ps], key=lambda o: o[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    m = ref["mask"].float()
    for rank, out in outs:
        assert out["pc"] == ref["pc"].tolist(), rank
        assert out["mask"] == ref["mask"].tolist() and out["rewards"] == ref["rewards"].tolist()
        assert torch.allclose(torch.tensor(out["lp"]) * m, ref["lp"] * m, atol=1e-5)
        assert torch.allclose(torch.tensor(out["adv"]), ref["adv"], atol=1e-5) and abs(out["loss"] - ref["loss"]) < 1e-5


def _nccl_one_rank(port, q):
    import torch.distributed as td
    from open_o3_video_amd import dist as od
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    td.init_process_group("nccl", device_id=torch.device("cuda", 0))   # RCCL communicator on the one GPU of the box
    rec = torch.arange(8 * 11, dtype=torch.float32, device="cuda").view(8, 11)
    out = torch.empty_like(rec)
    td.all_gather_into_tensor(out, rec)            # the rollout's metrics collective, through RCCL
    t = torch.tensor([3.5], device="cuda", dtype=torch.float64)
    td.all_reduce(t, op=td.ReduceOp.MAX)           # the bench's MAX-over-ranks reduction
    td.barrier()
    q.put((td.get_backend(), bool(torch.equal(out, rec)), float(t.item())))
    td.destroy_process_group()


def test_rccl_backend_collectives_one_rank(need_gpu):
    """backend "nccl" (= RCCL on ROCm) with the device_id init bench.py and dist.init use: communicator creation,
    all_gather_into_tensor, all_reduce(MAX) and barrier on the box's one GPU (RCCL refuses two ranks on one device, so the
    multi-rank RCCL run is the driver's 8-GPU job)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank, args=(port, q))
    p.start()
    backend, same, mx = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert backend == "nccl" and same and mx == 3.5
