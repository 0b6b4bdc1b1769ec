"""GPU parity of the NATIVE video input (multi_modal_data["video"] / pixel_values_videos + <|video_pad|>): what QwenVL_VLLM.__call__
(R:eval/models/model_vllm.py:39-106), the MCQ / test-time-scaling drivers (R:eval/test/test_videomme.py:137-145) and the
trainer's non-multi-image branch (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:555-564,604-606) send.  Goldens G14 (Qwen2.5-VL) and
G15 (Qwen3-VL) come from the in-container transformers 5.15 models (tools/make_golden.py g14): the engine's greedy ids must be
bit-identical through the engine, the HF facade and the vLLM facade, logits within LOGIT_ATOL of HF-fp32.  The facades default to
the rope arithmetic of the libraries the reference pins (position_mode="pinned"); the goldens pin position_mode="tf5"."""
import os

import numpy as np
import pytest
import torch

import fixture_models as fm
import fixture_models_q3 as fq
from test_gpu_model import LOGIT_ATOL, VIT_RTOL, build_engine

pytestmark = pytest.mark.gpu

Q25 = [("g14_video_tiny.npz", fm.tiny_config, 0), ("g14_video_medium.npz", fm.medium_config, 2)]
Q3 = [("g15_q3_video_tiny.npz", fq.tiny_q3_config, 0), ("g15_q3_video_medium.npz", fq.medium_q3_config, 2)]


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _engine(fname, cfgf, wseed):
    cfg = cfgf()
    W = (fq if "q3" in fname else fm).make_weights(cfg, wseed)
    eng = build_engine(cfg, W)
    eng.position_mode = "tf5"
    return cfg, W, eng


def _inputs(g):
    kw = dict(pixel_values_videos=torch.from_numpy(g["pixel_values_videos"]), video_grid_thw=g["video_grid"],
              second_per_grid_ts=list(g["second_per_grid_ts"]))
    if "pixel_values" in g.files:
        kw.update(pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=g["image_grid"])
    return kw


@pytest.mark.parametrize("fname,cfgf,wseed", Q25 + Q3)
def test_patchify_video_kernel(need_gpu, golden_dir, fname, cfgf, wseed):
    """o3v_patchify_video == Qwen2VLVideoProcessor.patchify over per-frame rescale / normalise (the goldens' pixel_values_videos
    cast to bf16, bit for bit): temporal pairs of distinct frames, the odd-count video repeats its last frame; uint8 and f32 in."""
    g = np.load(os.path.join(golden_dir, fname))
    cfg, _, eng = _engine(fname, cfgf, wseed)
    fr = torch.from_numpy(g["video_frames"])
    for frames in (fr, fr.float()):
        px, grid = eng.pixels_from_video(frames)
        assert np.array_equal(grid, g["video_grid"])
        K = g["pixel_values_videos"].shape[1]
        want = torch.from_numpy(g["pixel_values_videos"]).to(torch.bfloat16)
        assert torch.equal(px[:, :K].cpu(), want) and not px[:, K:].any()
    with pytest.raises(ValueError):
        eng.pixels_from_video(fr[:, :, :-1])
    with pytest.raises(ValueError):
        eng.pixels_from_video(fr[:0])


@pytest.mark.parametrize("fname,cfgf,wseed", Q25 + Q3)
def test_video_greedy_ids_and_logits(need_gpu, golden_dir, fname, cfgf, wseed):
    """Engine over the native video input: merged video tokens, rope deltas (through the ids), prefill logits, greedy ids ==
    HF (fp32 and bf16 agree, margins > 0.2), with the eval path's repetition penalty, from processor rows and from raw frames."""
    g = np.load(os.path.join(golden_dir, fname))
    cfg, W, eng = _engine(fname, cfgf, wseed)
    kw = _inputs(g)
    n_new = g["f32_step_logits"].shape[1]
    vis = eng.vit_forward(eng.pixels_from_processor(kw["pixel_values_videos"]), g["video_grid"])
    v0 = (vis[0] if vis.dim() == 3 else vis).float().cpu()
    ref = torch.from_numpy(g["f32_vit_merged_video"])
    rel = ((v0 - ref).norm() / ref.norm()).item()
    rel_hf = ((torch.from_numpy(g["bf16_vit_merged_video"]) - ref).norm() / ref.norm()).item()
    print(f"{fname}: merged video tokens rel-L2 vs HF-fp32: ours {rel:.4f}, HF-bf16 {rel_hf:.4f}")
    assert rel < VIT_RTOL and rel < 2.0 * rel_hf + 1e-3
    lg = eng.forward_logits(g["input_ids"], None, **kw)[:, -1].float().cpu()
    lref = torch.from_numpy(g["f32_prefill_last_logits"])
    err, err_hf = (lg - lref).abs().max().item(), (torch.from_numpy(g["bf16_prefill_last_logits"]) - lref).abs().max().item()
    print(f"{fname}: prefill logits max|err| vs HF-fp32: ours {err:.4f}, HF-bf16 {err_hf:.4f}")
    assert err < LOGIT_ATOL and err < 2.0 * err_hf + 0.02
    out = eng.generate(g["input_ids"], None, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"], **kw)
    got = out.sequences.cpu().numpy()
    assert np.array_equal(got, g["bf16_ids"]), (got[0, -n_new:], g["bf16_ids"][0, -n_new:])
    assert np.array_equal(got, g["f32_ids"])
    out = eng.generate(g["input_ids"], None, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"], repetition_penalty=1.05, **kw)
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids_rp105"])
    # raw frames in (GPU rescale / normalise / temporal-pair patchify)
    kf = dict(video_frames=torch.from_numpy(g["video_frames"]), second_per_grid_ts=list(g["second_per_grid_ts"]))
    if "image_frames" in g.files:
        kf["frames"] = torch.from_numpy(g["image_frames"])
    out = eng.generate(g["input_ids"], None, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"], **kf)
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids"])
    # a group of completions shares the video's prefill; greedy rows repeat the golden
    out = eng.generate(g["input_ids"], None, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"], num_return_sequences=3, **kw)
    assert all(np.array_equal(out.sequences[i].cpu().numpy(), g["bf16_ids"][0]) for i in range(3))
    # mismatched placeholders / features
    bad = dict(kw, video_grid_thw=g["video_grid"] + np.asarray([[1, 0, 0]]))
    with pytest.raises(ValueError):
        eng.generate(g["input_ids"], None, max_new_tokens=2, pad_token_id=cfg["pad_token_id"], **bad)


def test_video_positions_modes_differ_only_in_positions(need_gpu, golden_dir):
    """position_mode="pinned" (transformers @336dc69d / vllm 0.7.2 arithmetic) runs the same engine over other integer positions:
    on G14 tiny (3 temporal patches, 2 tokens/s, 1 s per patch: time 0,2,4 against a 3-wide grid) the text after the video starts
    at 5 instead of 3, so the logits differ from the tf5 golden -- and equal the oracle fed the same positions."""
    from oracle import model_ref
    from open_o3_video_amd import indexing
    fname, cfgf, wseed = Q25[0]
    g = np.load(os.path.join(golden_dir, fname))
    cfg, W, eng = _engine(fname, cfgf, wseed)
    ids = g["input_ids"]
    pa, da = indexing.rope_index(ids, None, None, cfg["image_token_id"], video_grid_thw=g["video_grid"], video_token_id=cfg["video_token_id"],
                                 second_per_grid_ts=[1.0], tokens_per_second=2, mode="tf5")
    pb, db = indexing.rope_index(ids, None, None, cfg["image_token_id"], video_grid_thw=g["video_grid"], video_token_id=cfg["video_token_id"],
                                 second_per_grid_ts=[1.0], tokens_per_second=2, mode="pinned")
    assert np.array_equal(pa, g["position_ids"]) and int(db[0]) == int(da[0]) + 2 and not np.array_equal(pa, pb)
    eng.position_mode = "pinned"
    lg = eng.forward_logits(ids, None, **_inputs(g))[:, -1].float().cpu()
    # oracle with the pinned positions: the text model on the oracle's own embeddings
    x = model_ref.embed_with_vision(W, cfg, torch.from_numpy(ids), None, None, torch.float32, None, torch.from_numpy(g["pixel_values_videos"]),
                                    g["video_grid"])
    h = model_ref.text_forward(W, cfg, x, torch.from_numpy(pb), torch.ones_like(torch.from_numpy(ids)), model_ref.KVCache(
        cfg["text_config"]["num_hidden_layers"]), torch.float32)
    ref = torch.nn.functional.linear(h[:, -1], model_ref.lm_head_weight(W, cfg).float())
    assert (lg - ref).abs().max().item() < LOGIT_ATOL
    assert (lg - torch.from_numpy(g["f32_prefill_last_logits"])).abs().max().item() > 1e-3


@pytest.mark.parametrize("fname,cfgf,wseed", [Q25[1], Q3[1]])
def test_video_through_the_hf_facade(need_gpu, golden_dir, fname, cfgf, wseed):
    """generate(**processor_output) / model(...).logits / completion_logps with pixel_values_videos + video_grid_thw +
    second_per_grid_ts and <|video_pad|> prompts (the trainer's video branch), G14b / G15b: video followed by an image."""
    from open_o3_video_amd.hf_api import GenerationConfigLike, Qwen2_5_VLForConditionalGeneration
    from oracle import model_ref
    g = np.load(os.path.join(golden_dir, fname))
    cfg, W, eng = _engine(fname, cfgf, wseed)
    from open_o3_video_amd.config import O3VConfig
    model = Qwen2_5_VLForConditionalGeneration(O3VConfig.from_dict(cfg), eng)
    assert model.position_mode == "pinned"
    model.position_mode = "tf5"
    n_new = g["f32_step_logits"].shape[1]
    ids = torch.from_numpy(g["input_ids"])
    pin = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in _inputs(g).items()}
    gc = GenerationConfigLike(max_new_tokens=n_new, do_sample=False, num_return_sequences=1, pad_token_id=cfg["pad_token_id"], eos_token_id=None)
    out = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), generation_config=gc, **pin)
    assert np.array_equal(out.cpu().numpy(), g["bf16_ids"])
    # two prompts in one call: pixel rows / grids / seconds are split per prompt by their placeholders
    two = {k: torch.cat([v, v]) for k, v in pin.items()}
    out2 = model.generate(input_ids=torch.cat([ids, ids]), attention_mask=torch.ones(2, ids.shape[1], dtype=torch.int64),
                          generation_config=GenerationConfigLike(max_new_tokens=5, do_sample=False, num_return_sequences=2,
                                                                 pad_token_id=cfg["pad_token_id"], eos_token_id=None), **two)
    assert out2.shape[0] == 4 and all(torch.equal(out2[i], out[0, : out2.shape[1]]) for i in range(4))
    lg = model(out, attention_mask=torch.ones_like(out), **pin).logits
    S = ids.shape[1]
    assert torch.equal(lg[0, S - 1:-1].float().argmax(-1).cpu(), out[0, S:].cpu())
    # log-probs of the completion: the fused pass == the reference recipe on the logits; the trainer drops second_per_grid_ts
    # before its log-prob passes (R:grpo_trainer.py:608-609) -> positions with 1 s per temporal patch
    no_spg = {k: v for k, v in pin.items() if k != "second_per_grid_ts"}
    lg1 = model(out, attention_mask=torch.ones_like(out), **no_spg).logits
    ref_lp = torch.log_softmax(lg1[:, :-1].float(), -1).gather(2, out[:, 1:, None].to(lg1.device))[..., 0][:, S - 1:]
    lp = model.completion_logps(ids, torch.ones_like(ids), out[:, S:], **no_spg)
    assert torch.allclose(lp, ref_lp, atol=0.08, rtol=0)
    with pytest.raises(ValueError):      # <|video_pad|> placeholders without video tensors
        model.generate(input_ids=ids, generation_config=gc, **{k: v for k, v in pin.items() if "video" not in k and k != "second_per_grid_ts"})


class VideoStubTokenizer:
    """Whitespace tokenizer over 'w<ID>' words, the vision tags, and Qwen3-VL's '<x.y seconds>' stamps (three ids each)."""
    specials = {"<|vision_start|>": "vision_start_token_id", "<|image_pad|>": "image_token_id", "<|vision_end|>": "vision_end_token_id",
                "<|video_pad|>": "video_token_id"}

    def __init__(self, cfg):
        self.cfg = cfg

    def encode(self, text, add_special_tokens=False):
        import re
        text = re.sub(r"<(\d+)\.(\d) seconds>", lambda m: f" w{20 + int(m.group(1)) % 50} w{80 + int(m.group(2))} w19 ", text)
        for s in self.specials:
            text = text.replace(s, f" {s} ")
        return [self.cfg[self.specials[w]] if w in self.specials else int(w[1:]) for w in text.split()]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(f"w{int(i)}" for i in ids)


def _prompt_text(cfg, ids, collapse_video_runs):
    """The golden's ids as prompt text with ONE placeholder tag per image / per video (per <vs>..<ve> block when not collapsing)."""
    tag = {cfg["vision_start_token_id"]: "<|vision_start|>", cfg["vision_end_token_id"]: "<|vision_end|>"}
    pads = {cfg["image_token_id"]: "<|image_pad|>", cfg["video_token_id"]: "<|video_pad|>"}
    words, i = [], 0
    while i < len(ids):
        if ids[i] in pads:
            words.append(pads[ids[i]])
            t = ids[i]
            while i < len(ids) and ids[i] == t:
                i += 1
            continue
        words.append(tag.get(ids[i], f"w{ids[i]}"))
        i += 1
    return " ".join(words)


@pytest.mark.parametrize("fname,cfgf,wseed", Q25)
def test_video_through_the_vllm_facade(need_gpu, golden_dir, fname, cfgf, wseed):
    """LLM.generate([{"prompt": ..<|video_pad|>.., "multi_modal_data": {"video": frames[, "image": ...]}}]) as QwenVL_VLLM.__call__
    builds it (R:eval/models/model_vllm.py:72-88): f32 0..255 frames in, one <|video_pad|> expanded, second_per_grid_ts = 2 / fps."""
    from open_o3_video_amd.vllm_api import LLM, SamplingParams
    g = np.load(os.path.join(golden_dir, fname))
    cfg, W, eng = _engine(fname, cfgf, wseed)
    spg = float(g["second_per_grid_ts"][0])
    llm = LLM(engine=eng, tokenizer=VideoStubTokenizer(cfg), limit_mm_per_prompt={"image": 32, "video": 10}, max_model_len=4096,
              position_mode="tf5", mm_processor_kwargs={"fps": 2.0 / spg})
    ids = g["input_ids"][0].tolist()
    prompt = _prompt_text(cfg, ids, True)
    assert prompt.count("<|video_pad|>") == 1
    mm = {"video": g["video_frames"].astype(np.float32)}                  # v_input.numpy(): float frames 0..255
    if "image_frames" in g.files:
        mm["image"] = g["image_frames"][0].transpose(1, 2, 0)             # np.array(PIL image): HWC uint8
    n_new = g["f32_step_logits"].shape[1]
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=n_new, stop_token_ids=[])
    outs = llm.generate([{"prompt": prompt, "multi_modal_data": mm}], sampling_params=sp)
    assert outs[0].prompt_token_ids == ids
    exp = g["bf16_ids_rp105"][0, len(ids):].tolist()
    assert outs[0].outputs[0].token_ids == exp
    # uint8 frames / a list of one video give the same tokens, and the video was encoded once
    o2 = llm.generate({"prompt": prompt, "multi_modal_data": dict(mm, video=[torch.from_numpy(g["video_frames"]).float()])}, sp)
    assert o2[0].outputs[0].token_ids == exp and llm.vis_cache_hits >= 1 and llm.prefix_tokens_reused >= len(ids) - 1
    # two requests decoded together; N sampled chains of one video prompt (the test-time-scaling chain call, R:eval/test/test_videomme.py:137-145)
    both = llm.generate([{"prompt": prompt, "multi_modal_data": mm}] * 2, sp)
    assert [b.outputs[0].token_ids for b in both] == [exp, exp]
    o5 = llm.generate({"prompt": prompt, "multi_modal_data": mm}, SamplingParams(temperature=0.7, top_p=0.9, max_tokens=6, n=4, seed=5))
    assert len(o5[0].outputs) == 4
    with pytest.raises(ValueError):
        llm.generate({"prompt": prompt.replace("<|video_pad|>", ""), "multi_modal_data": mm}, sp)
    with pytest.raises(ValueError):
        LLM(engine=eng, tokenizer=VideoStubTokenizer(cfg), limit_mm_per_prompt={"video": 1}).generate(
            {"prompt": prompt, "multi_modal_data": dict(mm, video=[mm["video"], mm["video"]])}, sp)
    # default arithmetic of the facade = vllm 0.7.2's (position_mode "pinned"): runs, same prompt ids
    d = LLM(engine=eng, tokenizer=VideoStubTokenizer(cfg), max_model_len=4096).generate({"prompt": prompt, "multi_modal_data": mm}, sp)
    assert d[0].prompt_token_ids == ids and len(d[0].outputs[0].token_ids) == n_new


def test_q3_video_through_the_vllm_facade(need_gpu, golden_dir):
    """Qwen3-VL: the <|vision_start|><|video_pad|><|vision_end|> of the template becomes one timestamped block per temporal patch;
    the facade's tokens equal the HF facade's on the ids the facade built (the model path itself is pinned by G15 above)."""
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.hf_api import Qwen3VLForConditionalGeneration
    from open_o3_video_amd.vllm_api import LLM, SamplingParams
    fname, cfgf, wseed = Q3[1]
    g = np.load(os.path.join(golden_dir, fname))
    cfg, W, eng = _engine(fname, cfgf, wseed)
    llm = LLM(engine=eng, tokenizer=VideoStubTokenizer(cfg), max_model_len=4096, position_mode="tf5")
    prompt = "w11 w12 <|vision_start|><|video_pad|><|vision_end|> w13 <|vision_start|><|image_pad|><|vision_end|> w14 w15"
    fr = g["video_frames"].astype(np.float32)
    mm = {"video": (fr, {"fps": 2.0, "frames_indices": list(range(0, 2 * fr.shape[0], 2))}), "image": g["image_frames"][0]}
    sp = SamplingParams(temperature=0.0, max_tokens=8, stop_token_ids=[])
    out = llm.generate({"prompt": prompt, "multi_modal_data": mm}, sp)[0]
    ids = out.prompt_token_ids
    gt, gh, gw = (int(v) for v in g["video_grid"][0])
    assert ids.count(cfg["video_token_id"]) == gt * gh * gw // 4 and ids.count(cfg["vision_start_token_id"]) == gt + 1
    model = Qwen3VLForConditionalGeneration(O3VConfig.from_dict(cfg), eng)
    model.position_mode = "tf5"
    ref = model.generate(input_ids=torch.tensor([ids]), pixel_values_videos=torch.from_numpy(g["pixel_values_videos"]),
                         video_grid_thw=torch.from_numpy(g["video_grid"]), pixel_values=torch.from_numpy(g["pixel_values"]),
                         image_grid_thw=torch.from_numpy(g["image_grid"]), max_new_tokens=8, do_sample=False, eos_token_id=None,
                         pad_token_id=cfg["pad_token_id"])
    assert out.outputs[0].token_ids == ref[0, len(ids):].tolist()
