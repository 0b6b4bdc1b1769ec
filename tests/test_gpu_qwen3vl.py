"""GPU parity of the Qwen3-VL path (BASELINE config #5's family) against goldens G12 / G13, which the in-container
transformers Qwen3VLForConditionalGeneration produced (tools/make_golden.py g12_g13_qwen3vl) in fp32 and bf16: merged visual
tokens, every DeepStack feature, step logits along HF's greedy path, greedy ids.  G13 carries the 8B model's head geometry
(ViT heads of 72 stored 80 wide, LLM heads of 128 at GQA 4:1, interleaved M-RoPE [24,20,20], three DeepStack taps).
Bars as in test_gpu_model.py: no further from HF-fp32 than 2x what HF-bf16 itself is, ids bit-identical where HF's two
precisions agree."""
import os

import numpy as np
import pytest
import torch

import fixture_models as fm
import fixture_models_q3 as fq
from test_gpu_model import LOGIT_ATOL, build_engine, rel_l2

pytestmark = pytest.mark.gpu

CASES = [("g12_q3_tiny.npz", "tiny_q3_config", 0), ("g13_q3_medium.npz", "medium_q3_config", 2)]


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _load(golden_dir, fname, cfgname, wseed, **kw):
    g = np.load(os.path.join(golden_dir, fname))
    cfg = getattr(fq, cfgname)()
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, getter_from_dict
    c = O3VConfig.from_dict(cfg)
    assert c.arch == "qwen3_vl"
    eng = O3VEngine(c, DeviceWeights(c, getter_from_dict(fq.make_weights(cfg, wseed)), "cuda", **kw))
    return g, cfg, eng


@pytest.mark.parametrize("fname,cfgname,wseed", CASES)
def test_q3_vit_merged_and_deepstack(need_gpu, golden_dir, fname, cfgname, wseed):
    g, cfg, eng = _load(golden_dir, fname, cfgname, wseed)
    vis = eng.vit_forward(eng.pixels_from_processor(torch.from_numpy(g["pixel_values"])), g["grid"])
    nd = len(cfg["vision_config"]["deepstack_visual_indexes"])
    assert vis.shape == (1 + nd,) + g["f32_vit_merged"].shape
    names = ["vit_merged"] + [f"deepstack_{k}" for k in range(nd)]
    for k, name in enumerate(names):
        f32 = torch.from_numpy(g["f32_" + name])
        e_gpu, e_hf = rel_l2(vis[k], f32), rel_l2(torch.from_numpy(g["bf16_" + name]), f32)
        print(f"{fname} {name}: rel-L2 vs HF-fp32 ours {e_gpu:.5f}, HF-bf16 {e_hf:.5f}")
        assert e_gpu < 2.0 * e_hf + 1e-3, name


@pytest.mark.parametrize("fname,cfgname,wseed", CASES)
def test_q3_frames_path_equals_processor_rows(need_gpu, golden_dir, fname, cfgname, wseed):
    """uint8 frames through o3v_patchify_ps (patch 16, mean = std = 0.5) == the HF processor's pixel_values rows cast to bf16."""
    g, cfg, eng = _load(golden_dir, fname, cfgname, wseed)
    px, grid = eng.pixels_from_frames(torch.from_numpy(g["frames"]))
    assert np.array_equal(grid, g["grid"])
    want = eng.pixels_from_processor(torch.from_numpy(g["pixel_values"]))
    assert torch.equal(px, want)


@pytest.mark.parametrize("fname,cfgname,wseed", CASES)
def test_q3_greedy_ids_and_step_logits(need_gpu, golden_dir, fname, cfgname, wseed):
    g, cfg, eng = _load(golden_dir, fname, cfgname, wseed)
    n_new = g["f32_step_logits"].shape[1]
    S = g["input_ids"].shape[1]
    pv = torch.from_numpy(g["pixel_values"])
    out = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=n_new,
                       pad_token_id=cfg["pad_token_id"])
    got = out.sequences.cpu().numpy()
    # teacher-forced logits along HF-fp32's path
    seq = g["f32_ids"]
    lg = eng.forward_logits(seq[:, :-1], None, pixel_values=pv, image_grid_thw=g["grid"])[:, S - 1:S - 1 + n_new].float().cpu()
    ref = torch.from_numpy(g["f32_step_logits"])
    d_ours = (lg - ref).abs().max().item()
    d_hf = (torch.from_numpy(g["bf16_step_logits"]) - ref).abs().max().item()
    print(f"{fname}: step logits max|err| vs HF-fp32: ours {d_ours:.4f}, HF-bf16 {d_hf:.4f}; ids ours {got[0, S:].tolist()} "
          f"HF-bf16 {g['bf16_ids'][0, S:].tolist()} HF-fp32 {g['f32_ids'][0, S:].tolist()}")
    assert d_ours < LOGIT_ATOL and d_ours < 2.0 * d_hf + 0.02
    # ids: equal to HF wherever the fp32 margin is safe against the measured logit error
    want = g["f32_ids"][0, S:]
    k = 0
    while k < n_new and got[0, S + k] == want[k]:
        k += 1
    assert k == n_new or g["f32_margins"][0][k] <= 4 * max(d_ours, d_hf), f"ids diverge at step {k} at a safe margin"
    if np.array_equal(g["bf16_ids"], g["f32_ids"]):
        assert np.array_equal(got, g["f32_ids"])


def test_q3_group_rollout_and_logps(need_gpu, golden_dir):
    """The rollout entry points on a Qwen3-VL model: G sampled completions sharing one ViT pass + prefill, batch decode without
    the fused attention block (qk-norm), and completion_logps == log-softmax of forward_logits at the completion tokens."""
    g, cfg, eng = _load(golden_dir, *CASES[1])
    pv = torch.from_numpy(g["pixel_values"])
    S = g["input_ids"].shape[1]
    out = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=10, do_sample=True,
                       temperature=1.0, top_p=0.9, num_return_sequences=4, seed=5, pad_token_id=cfg["pad_token_id"])
    seqs = out.sequences
    assert seqs.shape == (4, S + 10) and len({tuple(r.tolist()) for r in seqs[:, S:]}) > 1
    comp = seqs[:, S:]
    lp = eng.completion_logps(g["input_ids"], comp, pixel_values=pv, image_grid_thw=g["grid"])
    for r in range(4):
        lg = eng.forward_logits(seqs[r:r + 1].cpu().numpy(), None, pixel_values=pv, image_grid_thw=g["grid"])
        ref = torch.log_softmax(lg[0, S - 1:-1].float(), dim=-1).gather(1, comp[r].to(lg.device)[:, None])[:, 0]
        assert (lp[r] - ref).abs().max().item() < 0.08, r
    # greedy G=1 and G=2 agree on row 0 (fan-out of the prompt K/V incl. the DeepStack-modified hidden states)
    a = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=8, pad_token_id=cfg["pad_token_id"])
    b = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=8, num_return_sequences=2,
                     pad_token_id=cfg["pad_token_id"])
    assert torch.equal(a.sequences[0], b.sequences[0]) and torch.equal(b.sequences[0], b.sequences[1])
    # 8+ and 17+ rows: q/k/v runs behind a separate norm launch (two MFMA column blocks above 16 rows)
    for G in (8, 20):
        c = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=8, num_return_sequences=G,
                         pad_token_id=cfg["pad_token_id"])
        k = 0
        while k < 8 and c.sequences[G - 1, S + k] == a.sequences[0, S + k]:
            k += 1
        assert k == 8 or a.margins[0, k].item() < 2 * LOGIT_ATOL, (G, k)


def test_q3_prefix_reuse_with_deepstack(need_gpu, golden_dir):
    """prefix_key reuse on Qwen3-VL: the second question prefills only its own tokens; DeepStack rows before the reused prefix
    are already folded into the cached K/V, rows after it are added again -- ids equal the one-shot run."""
    g, cfg, eng = _load(golden_dir, *CASES[1])
    pv = torch.from_numpy(g["pixel_values"])
    ids = g["input_ids"]
    ids2 = ids.copy()
    ids2[0, -3] = (ids2[0, -3] + 7) % 3000 + 10
    kw = dict(pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=8, pad_token_id=cfg["pad_token_id"])
    eng.generate(ids, None, prefix_key="v", **kw)
    b = eng.generate(ids2, None, prefix_key="v", **kw)
    assert b.timings["prefix_tokens_reused"] == ids.shape[1] - 3
    eng.drop_prefix_cache()
    c = eng.generate(ids2, None, **kw)
    assert torch.equal(b.sequences, c.sequences)


def test_q3_fp8_decode_rows(need_gpu, golden_dir):
    """fp8 decode weights on the Qwen3-VL path (q/k/v as a plain fp8 linear + the qk-norm/rotation kernel): runs, and agrees with
    the bf16-row engine on the first tokens when the weights are fp8-representable."""
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, dequantize_rows_fp8, getter_from_dict, quantize_rows_fp8
    g = np.load(os.path.join(golden_dir, CASES[1][0]))
    cfg = fq.medium_q3_config()
    W = fq.make_weights(cfg, 2)
    for k in list(W):
        if k.startswith("model.language_model.layers.") and k.endswith("_proj.weight") or k == "lm_head.weight":
            q8, sc = quantize_rows_fp8(W[k].to(torch.bfloat16))
            W[k] = dequantize_rows_fp8(q8, sc)
    c = O3VConfig.from_dict(cfg)
    e1 = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda"))
    e2 = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda", fp8_decode=True))
    kw = dict(pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=g["grid"], max_new_tokens=10, pad_token_id=cfg["pad_token_id"])
    a, b = e1.generate(g["input_ids"], None, **kw), e2.generate(g["input_ids"], None, **kw)
    ga, gb = a.sequences[0, -10:].tolist(), b.sequences[0, -10:].tolist()
    m = a.margins[0].tolist()
    k = 0
    while k < 10 and ga[k] == gb[k]:
        k += 1
    print(f"q3 fp8 rows: bf16 {ga} fp8 {gb} margins {np.round(m, 3).tolist()}")
    assert k == 10 or m[k] < 2 * LOGIT_ATOL
    # 4..32 rows: the fragment-major fp8 images on the matrix cores (N chains of one question, BASELINE config #5)
    assert e2.w.llm.layer[0].gu_w8p and e2.w.llm.lm_head8p
    for G in (4, 16, 20):
        c1 = e1.generate(g["input_ids"], None, num_return_sequences=G, **kw)
        c2 = e2.generate(g["input_ids"], None, num_return_sequences=G, **kw)
        r1, r2, mm = c1.sequences[G - 1, -10:].tolist(), c2.sequences[G - 1, -10:].tolist(), c1.margins[G - 1].tolist()
        k = 0
        while k < 10 and r1[k] == r2[k]:
            k += 1
        print(f"q3 fp8 rows, {G} rows: follow the bf16 rows for {k}/10 tokens")
        assert k == 10 or mm[k] < 2 * LOGIT_ATOL, (G, k, mm[k])


def test_q3_through_the_hf_facade(need_gpu, golden_dir):
    """hf_api.Qwen3VLForConditionalGeneration over a Qwen3-VL state dict: generate() with the HF call signature returns HF's greedy
    ids (golden G13), the logits surface agrees with the engine's, per-token log-probs are finite."""
    from open_o3_video_amd.hf_api import Qwen3VLForConditionalGeneration
    g = np.load(os.path.join(golden_dir, CASES[1][0]))
    cfg = fq.medium_q3_config()
    model = Qwen3VLForConditionalGeneration.from_state_dict(cfg, fq.make_weights(cfg, CASES[1][2]))
    assert model.o3v_config.arch == "qwen3_vl"
    n_new = g["f32_step_logits"].shape[1]
    pv, grid = torch.from_numpy(g["pixel_values"]), torch.from_numpy(g["grid"])
    out = model.generate(input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.ones_like(torch.from_numpy(g["input_ids"])),
                         pixel_values=pv, image_grid_thw=grid, max_new_tokens=n_new, do_sample=False, pad_token_id=cfg["pad_token_id"],
                         eos_token_id=None)
    assert np.array_equal(out.cpu().numpy(), g["bf16_ids"])
    logits = model(input_ids=torch.from_numpy(g["bf16_ids"]), pixel_values=pv, image_grid_thw=grid).logits
    assert logits.shape == (1, g["bf16_ids"].shape[1], cfg["text_config"]["vocab_size"]) and torch.isfinite(logits.float()).all()


def test_q3_through_the_vllm_facade(need_gpu, golden_dir):
    """vllm_api.LLM over a Qwen3-VL engine: frames under multi_modal_data['image'], one <|image_pad|> per frame, expanded with the
    model's own resize factor (32) and tokens per frame -- greedy tokens equal HF's (golden G13), from uint8 frames."""
    from test_gpu_facades import StubTokenizer
    from open_o3_video_amd.vllm_api import LLM, SamplingParams
    g, cfg, eng = _load(golden_dir, *CASES[1])
    llm = LLM(engine=eng, tokenizer=StubTokenizer(cfg), limit_mm_per_prompt={"image": 32}, max_model_len=4096)
    assert llm.image_factor == 32
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
    n_new = g["f32_step_logits"].shape[1]
    sp = SamplingParams(temperature=0.0, max_tokens=n_new, stop_token_ids=[])
    outs = llm.generate([{"prompt": " ".join(words), "multi_modal_data": {"image": torch.from_numpy(g["frames"])}}], sampling_params=sp)
    assert outs[0].prompt_token_ids == ids
    assert outs[0].outputs[0].token_ids == g["bf16_ids"][0, len(ids):].tolist()
