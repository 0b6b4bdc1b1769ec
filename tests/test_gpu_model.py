"""End-to-end GPU parity of the generate path against (a) the committed HF goldens (tests/golden, produced by
tools/make_golden.py from transformers' Qwen2_5_VLForConditionalGeneration) and (b) the CPU oracle run live on the
same seeded inputs.  Bars (north_star): greedy token ids bit-identical; fp logits within the tolerance written here.

Logit tolerance: the HIP path computes in bf16 with fp32 accumulation like the HF bf16 path, but with different
accumulation order and a flash-style softmax, so it is compared (i) to the HF *fp32* logits with the bf16-level
tolerance LOGIT_ATOL and (ii) required to be no further from fp32-HF than 2x what bf16-HF itself is."""
import os

import numpy as np
import pytest
import torch

import fixture_models as fm

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 0.25   # abs tolerance on logits of magnitude ~8-16 (bf16 keeps 8 mantissa bits: ulp(16) = 0.125); HF's own
                    # bf16 path is 0.12 away from its fp32 path on these fixtures
VIT_RTOL = 0.03     # relative L2 error of the merged visual tokens vs HF fp32

CASES = [("g6_tiny.npz", fm.tiny_config, 0, 16), ("g6_tiny_b.npz", fm.tiny_config, 1, 12),
         ("g7_medium.npz", fm.medium_config, 2, 16)]


def build_engine(cfg_dict, W):
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, getter_from_dict
    cfg = O3VConfig.from_dict(cfg_dict)
    return O3VEngine(cfg, DeviceWeights(cfg, getter_from_dict(W), "cuda"))


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("fname,cfgf,wseed,n_new", CASES)
def test_vit_and_prefill_logits(need_gpu, golden_dir, fname, cfgf, wseed, n_new):
    g = np.load(os.path.join(golden_dir, fname))
    cfg = cfgf()
    eng = build_engine(cfg, fm.make_weights(cfg, wseed))
    pv = torch.from_numpy(g["pixel_values"])
    vis = eng.vit_forward(eng.pixels_from_processor(pv), g["grid"])
    e_gpu = rel_l2(vis, torch.from_numpy(g["f32_vit_merged"]))
    e_hf = rel_l2(torch.from_numpy(g["bf16_vit_merged"]), torch.from_numpy(g["f32_vit_merged"]))
    print(f"{fname}: ViT rel-L2 vs HF-fp32: ours {e_gpu:.4f}, HF-bf16 {e_hf:.4f}")
    assert e_gpu < VIT_RTOL and e_gpu < 2.0 * e_hf + 1e-3
    # same thing from raw uint8 frames through the fused GPU frame pipeline
    px, grid = eng.pixels_from_frames(torch.from_numpy(g["frames"]))
    assert np.array_equal(grid, g["grid"])
    vis2 = eng.vit_forward(px, grid)
    assert torch.equal(vis2, vis)
    # prefill logits at the last prompt position
    logits = eng.forward_logits(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"])[:, -1].float().cpu()
    ref = torch.from_numpy(g["f32_prefill_last_logits"])
    err = (logits - ref).abs().max().item()
    err_hf = (torch.from_numpy(g["bf16_prefill_last_logits"]) - ref).abs().max().item()
    print(f"{fname}: prefill logits max|err| vs HF-fp32: ours {err:.4f}, HF-bf16 {err_hf:.4f}")
    assert err < LOGIT_ATOL and err < 2.0 * err_hf + 0.02


@pytest.mark.parametrize("fname,cfgf,wseed,n_new", CASES)
def test_greedy_ids_bit_identical(need_gpu, golden_dir, fname, cfgf, wseed, n_new):
    """Greedy decode: ids must equal the HF goldens (fp32 and bf16 HF agree on these prompts and every step's
    top-1/top-2 margin exceeds 0.2, see tools/make_golden.py:search_case)."""
    g = np.load(os.path.join(golden_dir, fname))
    cfg = cfgf()
    eng = build_engine(cfg, fm.make_weights(cfg, wseed))
    pv = torch.from_numpy(g["pixel_values"])
    out = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=n_new,
                       pad_token_id=cfg["pad_token_id"])
    got = out.sequences.cpu().numpy()
    print(f"{fname}: margins ours {np.round(out.margins.cpu().numpy()[0], 2)}\n   HF-bf16 {np.round(g['bf16_margins'][0], 2)}")
    assert np.array_equal(got, g["bf16_ids"]), (got[0, -n_new:], g["bf16_ids"][0, -n_new:])
    assert np.array_equal(got, g["f32_ids"])
    # with the eval path's repetition penalty (R:eval/models/model_vllm.py:30)
    out = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=n_new,
                       pad_token_id=cfg["pad_token_id"], repetition_penalty=1.05)
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids_rp105"])
    # frames-in (GPU patchify) must give the same ids
    out = eng.generate(g["input_ids"], None, frames=torch.from_numpy(g["frames"]), max_new_tokens=n_new,
                       pad_token_id=cfg["pad_token_id"])
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids"])


def test_step_logits_vs_oracle_live(need_gpu, golden_dir):
    """Teacher-forced comparison with the CPU oracle on a fresh seeded case (not a stored golden): run the oracle's
    greedy decode, then check our full-sequence logits at every generated position."""
    from oracle import model_ref
    cfg = fm.medium_config()
    W = fm.make_weights(cfg, 5)
    frames = fm.make_frames(3, 84, 112, seed=77)  # grid 6x8
    eng = build_engine(cfg, W)
    px, grid = eng.pixels_from_frames(frames)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=77)
    # oracle needs processor-style f32 pixel values: reuse the goldens' normalisation restated in the oracle
    from oracle import index_ref
    mean = np.asarray(index_ref.CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(index_ref.CLIP_STD, dtype=np.float32)[None, :, None, None]
    xf = ((frames.numpy().astype(np.float64) / 255.0).astype(np.float32) - mean) / std
    pv, grid_o = index_ref.patchify_frames(xf.astype(np.float32))
    assert np.array_equal(grid_o, grid)
    n_new = 10
    seq, step_logits = model_ref.generate(W, cfg, [ids], None, torch.from_numpy(pv), grid, n_new, dtype=torch.float32,
                                          pad_token_id=cfg["pad_token_id"], return_logits=True)
    full = eng.forward_logits(seq.numpy(), None, frames=frames).float().cpu()     # [1, S+n, V]
    S = len(ids)
    ours = full[0, S - 1:S - 1 + n_new]                                           # logits predicting each new token
    err = (ours - step_logits[0]).abs().max().item()
    print(f"teacher-forced step logits max|err| vs oracle fp32: {err:.4f}")
    assert err < LOGIT_ATOL
    # argmax agrees wherever the oracle's margin is comfortably above the tolerance
    top2 = step_logits[0].topk(2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_ATOL
    assert torch.equal(ours.argmax(-1)[safe], seq[0, S:][safe])
    # per-token log-probs (R:grpo_trainer.py:371-384)
    lp = eng.per_token_logps(eng.forward_logits(seq.numpy(), None, frames=frames), seq).cpu()
    lp_ref = model_ref.per_token_logps(model_ref.full_logits(W, cfg, seq, None, torch.from_numpy(pv), grid), seq)
    assert (lp[0, S - 1:] - lp_ref[0, S - 1:]).abs().max().item() < 0.1


def test_left_padding_and_batch(need_gpu):
    """Two prompts of different length, left padded (R:grpo_trainer.py:540-548): each row must reproduce its own
    unpadded single-row generation; EOS must stop a row and pad the rest (TF:utils.py:2927-2929)."""
    cfg = fm.tiny_config()
    W = fm.make_weights(cfg, 3)
    eng = build_engine(cfg, W)
    fa, fb = fm.make_frames(2, 56, 84, seed=1), fm.make_frames(1, 56, 84, seed=2)
    pa, ga = eng.pixels_from_frames(fa)
    pb, gb = eng.pixels_from_frames(fb)
    ia = fm.make_prompt(cfg, [tuple(r) for r in ga.tolist()], seed=1)
    ib = fm.make_prompt(cfg, [tuple(r) for r in gb.tolist()], seed=2)
    n = 8
    oa = eng.generate([ia], None, frames=fa, max_new_tokens=n).sequences.cpu().numpy()[0]
    ob = eng.generate([ib], None, frames=fb, max_new_tokens=n).sequences.cpu().numpy()[0]
    S = max(len(ia), len(ib))
    pad = cfg["pad_token_id"]
    rows = [[pad] * (S - len(x)) + x for x in (ia, ib)]
    mask = [[0] * (S - len(x)) + [1] * len(x) for x in (ia, ib)]
    both = eng.generate(rows, mask, frames=torch.cat([fa, fb]), max_new_tokens=n).sequences.cpu().numpy()
    assert np.array_equal(both[0, S:], oa[len(ia):])
    assert np.array_equal(both[1, S:], ob[len(ib):])
    # EOS handling: declare the 3rd generated token of row 0 an EOS
    eos = int(oa[len(ia) + 2])
    out = eng.generate(rows, mask, frames=torch.cat([fa, fb]), max_new_tokens=n, eos_token_ids=[eos], pad_token_id=pad,
                       steps_per_sync=2).sequences.cpu().numpy()
    first = list(oa[len(ia):]).index(eos)
    assert np.array_equal(out[0, S:S + first + 1], oa[len(ia):len(ia) + first + 1])
    assert (out[0, S + first + 1:] == pad).all()


def test_group_rollout_shares_prefill(need_gpu):
    """num_return_sequences=G (R:grpo_trainer.py:306-313): greedy rows of one group are identical to the G=1 run
    (one ViT + one prefill fanned out), sampled rows are reproducible per (seed, row id) and differ across rows."""
    cfg = fm.tiny_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 4))
    fr = fm.make_frames(2, 56, 84, seed=5)
    _, grid = eng.pixels_from_frames(fr)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=5)
    one = eng.generate([ids], None, frames=fr, max_new_tokens=6).sequences
    four = eng.generate([ids], None, frames=fr, max_new_tokens=6, num_return_sequences=4).sequences
    assert four.shape[0] == 4 and all(torch.equal(four[i], one[0]) for i in range(4))
    s1 = eng.generate([ids], None, frames=fr, max_new_tokens=12, num_return_sequences=4, do_sample=True, top_p=0.95,
                      temperature=1.0, seed=7).sequences
    s2 = eng.generate([ids], None, frames=fr, max_new_tokens=12, num_return_sequences=4, do_sample=True, top_p=0.95,
                      temperature=1.0, seed=7).sequences
    assert torch.equal(s1, s2)
    assert len({tuple(r.tolist()) for r in s1}) > 1
    # row ids key the RNG: generating completions 2..3 alone reproduces rows 2..3 of the group
    s3 = eng.generate([ids], None, frames=fr, max_new_tokens=12, num_return_sequences=2, do_sample=True, top_p=0.95,
                      temperature=1.0, seed=7, row_ids=[2, 3]).sequences
    assert torch.equal(s3, s1[2:4])


def test_prefix_kv_reuse(need_gpu, golden_dir):
    """SURVEY 8f-1 (several questions / samples per video, R:eval/test/test_vstar_multi_images.py:511-544): with a
    `prefix_key` only the tokens after the longest common prefix are prefilled.  Greedy ids must equal the cold run
    and the HF golden wherever the cut falls (GEMV-sized suffix, GEMM-sized suffix, cut before the frame block)."""
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    pv = torch.from_numpy(g["pixel_values"])
    ids = g["input_ids"][0].copy()
    S = len(ids)
    special = {cfg["image_token_id"], cfg["vision_start_token_id"], cfg["vision_end_token_id"]}
    text_pos = [i for i in range(S) if int(ids[i]) not in special]
    kw = dict(pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=16, repetition_penalty=1.05,
              pad_token_id=cfg["pad_token_id"])
    cold = eng.generate([ids], None, **kw)
    exp = g["bf16_ids_rp105"]
    assert np.array_equal(cold.sequences.cpu().numpy(), exp)
    assert cold.timings["prefix_tokens_reused"] == 0
    cuts = [text_pos[-3], max(p for p in text_pos if p <= S - 20), text_pos[1]]
    for n, cut in enumerate(cuts):
        other = ids.copy()
        other[cut] = ids[cut] + 1 if int(ids[cut]) + 1 not in special else ids[cut] + 2
        eng.generate([other], None, prefix_key=("vid", n), **{**kw, "max_new_tokens": 2})       # a first question
        warm = eng.generate([ids], None, prefix_key=("vid", n), **kw)                            # the next one
        assert warm.timings["prefix_tokens_reused"] == cut, (cut, warm.timings)
        assert np.array_equal(warm.sequences.cpu().numpy(), exp), f"cut at {cut} of {S}"
    # same prompt again (a second group of samples): everything but the last token is reused, rows fan out
    again = eng.generate([ids], None, prefix_key=("vid", len(cuts) - 1), num_return_sequences=3, **kw)
    assert again.timings["prefix_tokens_reused"] == S - 1
    assert all(np.array_equal(again.sequences[i].cpu().numpy(), exp[0]) for i in range(3))
    # an unknown key is a cold run and does not disturb the result
    fresh = eng.generate([ids], None, prefix_key="another video", **kw)
    assert fresh.timings["prefix_tokens_reused"] == 0 and np.array_equal(fresh.sequences.cpu().numpy(), exp)
    # logits of the suffix pass against the full pass (same tolerance as the full pass against HF)
    eng.drop_prefix_cache()
    full = eng.forward_logits(ids[None], None, pixel_values=pv, image_grid_thw=g["grid"])[0, -1].float()
    eng.generate([ids], None, prefix_key="v", **{**kw, "max_new_tokens": 1})
    assert eng._prefix_lookup("v", ids) == S - 1
    ref = torch.from_numpy(g["f32_prefill_last_logits"])[0]
    assert (full.cpu() - ref).abs().max().item() < LOGIT_ATOL


def test_group_rollout_shared_prefix_attention(need_gpu, golden_dir):
    """head_dim-128 model (medium fixture, GQA 7:1): with num_return_sequences=G the decode attention reads the prompt K/V
    once per group (o3v_attn_decode_group).  Greedy rows must still equal the HF golden ids, with the group kernel on and
    off; sampled rows must be reproducible and keyed by completion index."""
    g = np.load(os.path.join(golden_dir, "g7_medium.npz"))
    cfg = fm.medium_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    assert eng.cfg.text.head_dim == 128 and eng.group_attention
    pv = torch.from_numpy(g["pixel_values"])
    kw = dict(pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=16, pad_token_id=cfg["pad_token_id"])
    exp = g["bf16_ids_rp105"][0]
    for mode in ("shared_read", "kernel"):
        eng.group_attention_mode = mode
        for G in (2, 5, 8, 11, 16):   # G * 7 heads > 64 MFMA columns (G > 9): the shared-read form serves the group
            out = eng.generate(g["input_ids"], None, num_return_sequences=G, repetition_penalty=1.05, **kw).sequences.cpu().numpy()
            assert out.shape[0] == G and all(np.array_equal(out[i], exp) for i in range(G)), (mode, G)
    s_on = eng.generate(g["input_ids"], None, num_return_sequences=8, do_sample=True, top_p=0.95, seed=5, **kw).sequences
    eng.group_attention = False
    off = eng.generate(g["input_ids"], None, num_return_sequences=8, repetition_penalty=1.05, **kw).sequences.cpu().numpy()
    assert all(np.array_equal(off[i], exp) for i in range(8))
    eng.group_attention = True
    s_again = eng.generate(g["input_ids"], None, num_return_sequences=8, do_sample=True, top_p=0.95, seed=5, **kw).sequences
    assert torch.equal(s_on, s_again) and len({tuple(r.tolist()) for r in s_on}) > 1
    # (4 rows and 8 rows take the same matrix-core linear path, so the logits of a row do not depend on the group size)
    s_tail = eng.generate(g["input_ids"], None, num_return_sequences=4, do_sample=True, top_p=0.95, seed=5, row_ids=[4, 5, 6, 7],
                          **kw).sequences
    assert torch.equal(s_tail, s_on[4:8])


def test_completion_logps_shared_prompt(need_gpu):
    """G completions of one prompt (R:grpo_trainer.py:371-384, :612-613): one ViT pass + one prompt prefill + the
    completion tokens behind the shared prompt K/V + lm_head on the kept positions only, against (a) the CPU oracle run
    row by row on the concatenated sequences and (b) the engine's own full-sequence formulation."""
    from oracle import index_ref, model_ref
    cfg = fm.medium_config()
    W = fm.make_weights(cfg, 6)
    frames = fm.make_frames(2, 56, 84, seed=9)
    eng = build_engine(cfg, W)
    px, grid = eng.pixels_from_frames(frames)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=9)
    S, G, T = len(ids), 3, 7
    gen = torch.Generator().manual_seed(3)
    comp = torch.randint(10, 3000, (G, T), generator=gen)
    comp[1, 4:] = cfg["pad_token_id"]                       # a finished row padded after its EOS: still scored (masked later)
    mean = np.asarray(index_ref.CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(index_ref.CLIP_STD, dtype=np.float32)[None, :, None, None]
    xf = ((frames.numpy().astype(np.float64) / 255.0).astype(np.float32) - mean) / std
    pv, _ = index_ref.patchify_frames(xf.astype(np.float32))
    ours = eng.completion_logps(ids, comp, frames=frames).cpu()
    assert ours.shape == (G, T) and ours.dtype == torch.float32
    for g in range(G):
        seq = torch.cat([torch.tensor(ids), comp[g]])[None]
        ref = model_ref.per_token_logps(model_ref.full_logits(W, cfg, seq, None, torch.from_numpy(pv), grid), seq)[0, S - 1:]
        full = eng.per_token_logps(eng.forward_logits(seq.numpy(), None, frames=frames), seq).cpu()[0, S - 1:]
        assert (ours[g] - ref).abs().max().item() < 0.1, g
        assert (ours[g] - full).abs().max().item() < 0.1, g
    # chunked head: any chunk size gives the same numbers; T == 1 touches only the prompt's last position
    again = eng.completion_logps(ids, comp, frames=frames, rows_per_chunk=16).cpu()
    assert torch.equal(again, ours)
    one = eng.completion_logps(ids, comp[:, :1], frames=frames).cpu()
    assert (one[:, 0] - ours[:, 0]).abs().max().item() < 1e-6


def test_generate_edge_cases(need_gpu):
    """Degenerate shapes of the generate call: zero (refused) and one new token, a one-token text-only prompt, 16 left-padded rows of
    different lengths in one call (each equal to its own single-row run), every row hitting EOS on its first token."""
    cfg = fm.tiny_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 11))
    fr = fm.make_frames(2, 56, 84, seed=4)
    _, grid = eng.pixels_from_frames(fr)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=4)
    S = len(ids)
    with pytest.raises(ValueError):
        eng.generate([ids], None, frames=fr, max_new_tokens=0)               # refused like GenerationConfig.validate does
    one = eng.generate([ids], None, frames=fr, max_new_tokens=1).sequences
    five = eng.generate([ids], None, frames=fr, max_new_tokens=5).sequences
    assert one.shape == (1, S + 1) and torch.equal(one[0], five[0, :S + 1])
    # text only, a single token
    t1 = eng.generate([[7]], None, max_new_tokens=4).sequences
    assert t1.shape == (1, 5) and t1[0, 0].item() == 7
    # 16 rows, text-only prompts of lengths 1..16, left padded
    pad = cfg["pad_token_id"]
    rng = np.random.default_rng(0)
    prompts = [rng.integers(10, 200, n).tolist() for n in range(1, 17)]
    L = 16
    rows = [[pad] * (L - len(p)) + p for p in prompts]
    mask = [[0] * (L - len(p)) + [1] * len(p) for p in prompts]
    both = eng.generate(rows, mask, max_new_tokens=6).sequences.cpu().numpy()
    assert both.shape == (16, L + 6)
    for i in (0, 1, 7, 15):
        solo = eng.generate([prompts[i]], None, max_new_tokens=6).sequences.cpu().numpy()[0]
        assert np.array_equal(both[i, L:], solo[len(prompts[i]):]), i
    # every row ends on its first generated token: one step, EOS kept, nothing after it
    first = both[:, L]
    out = eng.generate(rows, mask, max_new_tokens=6, eos_token_ids=sorted(set(first.tolist())), pad_token_id=pad,
                       steps_per_sync=1)
    assert out.n_steps == 1 and np.array_equal(out.sequences.cpu().numpy()[:, L], first)
    with pytest.raises(ValueError):
        eng.generate([[1, 2]] * 33, None, max_new_tokens=1)                 # more rows than one engine call takes
    # 17..32 rows: the decode linears run two 16-row column blocks per weight fragment
    many = [[3 + i, 40 + 2 * i, 7, 9 + i] for i in range(21)]
    wide = eng.generate(many, None, max_new_tokens=5).sequences.cpu().numpy()
    for i in (0, 15, 16, 20):
        solo = eng.generate([many[i]], None, max_new_tokens=5)
        k = 0
        while k < 5 and wide[i, 4 + k] == solo.sequences[0, 4 + k].item():
            k += 1
        assert k == 5 or solo.margins[0, k].item() < 2 * LOGIT_ATOL, (i, k)
    with pytest.raises(ValueError):
        eng.generate([ids], None, frames=fr[:1], max_new_tokens=1)          # fewer frames than image placeholders


def _teacher_forced_step_logits(eng, g, n_new, dname="f32"):
    """Our logits at the positions where HF produced its step logits, along HF's own greedy path (teacher forcing)."""
    seq = g[f"{dname}_ids"]
    S = g["input_ids"].shape[1]
    lg = eng.forward_logits(seq[:, :-1] if seq.shape[1] > S + n_new - 1 else seq, None, pixel_values=torch.from_numpy(g["pixel_values"]),
                            image_grid_thw=g["grid"])
    return lg[:, S - 1:S - 1 + n_new].float().cpu()


def test_full_depth_true_width_vs_hf(need_gpu, golden_dir):
    """Golden G10: ALL 28 LLM layers and 32 ViT blocks at the true 7B widths (3584 / 18944 / 1280 / 3420, GQA 28:4, vocabulary
    4096), HF run in fp32 and bf16 in the build container.  The engine must be no further from HF-fp32 than 2x what HF's own
    bf16 path is (visual tokens: relative L2; step logits: max and mean abs), and agree with HF's greedy ids wherever the
    fp32 top-1/top-2 margin exceeds 4x the measured logit error.  These measured errors are the envelope the full-size
    property tests (test_gpu_fullsize.py) take their bounds from."""
    g = np.load(os.path.join(golden_dir, "g10_full7b.npz"))
    cfg = fm.full7b_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 3, dtype=torch.bfloat16))
    n_new = g["f32_step_logits"].shape[1]
    pv = torch.from_numpy(g["pixel_values"])
    vis = eng.vit_forward(eng.pixels_from_processor(pv), g["grid"])
    f32v = torch.from_numpy(g["f32_vit_merged"])
    e_gpu, e_hf = rel_l2(vis, f32v), rel_l2(torch.from_numpy(g["bf16_vit_merged"]), f32v)
    print(f"G10 ViT (32 blocks) rel-L2 vs HF-fp32: ours {e_gpu:.4f}, HF-bf16 {e_hf:.4f}")
    assert e_gpu < 2.0 * e_hf + 1e-3
    ref = torch.from_numpy(g["f32_step_logits"])
    ours = _teacher_forced_step_logits(eng, g, n_new)
    d_ours = (ours - ref).abs()
    # HF-bf16 walked its own path; where its ids equal fp32's the inputs are the same and the logits comparable
    same = int((g["bf16_ids"] == g["f32_ids"]).all(axis=0).cumprod()[g["input_ids"].shape[1]:].sum())
    d_hf = (torch.from_numpy(g["bf16_step_logits"]) - ref).abs()[:, :max(1, same + 1)]
    print(f"G10 step logits (28 layers) vs HF-fp32: ours max {d_ours.max():.4f} mean {d_ours.mean():.4f}; "
          f"HF-bf16 max {d_hf.max():.4f} mean {d_hf.mean():.4f} (|logit| max {ref.abs().max():.2f}, HF paths agree for {same} steps)")
    assert d_ours.max().item() < 2.0 * d_hf.max().item() + 0.05 and d_ours.mean().item() < 2.0 * d_hf.mean().item() + 0.01
    # The same rows left-padded by 37: the causal tiles start at other rows, so the online-softmax chunks and the split-K
    # choice differ.  Both tilings are measured against HF-fp32 here (which of the two is closer was an open question of the
    # full-size test P2, where no fp32 reference exists): neither is systematically closer -- both sit inside HF-bf16's own error.
    pad = 37
    seq = g["f32_ids"][:, :-1]
    S = g["input_ids"].shape[1]
    rows = np.concatenate([np.full((1, pad), cfg["pad_token_id"], dtype=np.int64), seq], axis=1)
    mask = np.concatenate([np.zeros((1, pad), dtype=np.int64), np.ones_like(seq)], axis=1)
    lgp = eng.forward_logits(rows, mask, pixel_values=pv, image_grid_thw=g["grid"])[:, pad + S - 1:pad + S - 1 + n_new].float().cpu()
    d_pad = (lgp - ref).abs()
    print(f"G10 left-padded by {pad}: vs HF-fp32 max {d_pad.max():.4f} mean {d_pad.mean():.4f}; padded vs unpadded max "
          f"{(lgp - ours).abs().max():.4f} mean {(lgp - ours).abs().mean():.4f}")
    assert d_pad.max().item() < 2.0 * d_hf.max().item() + 0.05 and d_pad.mean().item() < 2.0 * d_hf.mean().item() + 0.01
    out = eng.generate(g["input_ids"], None, pixel_values=pv, image_grid_thw=g["grid"], max_new_tokens=n_new,
                       pad_token_id=cfg["pad_token_id"])
    got = out.sequences.cpu().numpy()[0, -n_new:]
    want = g["f32_ids"][0, -n_new:]
    safe = g["f32_margins"][0] > 4.0 * d_ours.max().item()
    k = 0
    while k < n_new and got[k] == want[k]:
        k += 1
    print(f"G10 greedy ids ours {got.tolist()} HF-fp32 {want.tolist()} HF-bf16 {g['bf16_ids'][0, -n_new:].tolist()}; fp32 margins "
          f"{np.round(g['f32_margins'][0], 3).tolist()}")
    assert k == n_new or not safe[k], f"ids diverge at step {k} where the fp32 margin {g['f32_margins'][0][k]:.3f} is safe"


def test_tied_embeddings_gqa8(need_gpu, golden_dir):
    """Golden G11: tie_word_embeddings=True (lm_head reads embed_tokens: the Qwen2.5-VL-3B layout) with 16 query / 2 kv heads
    of 128 -- greedy ids equal HF's, step logits within the fixture tolerance."""
    g = np.load(os.path.join(golden_dir, "g11_tied3b.npz"))
    cfg = fm.tied3b_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 4))
    assert eng.w.t["l.head"].data_ptr() == eng.w.t["l.embed"].data_ptr()
    n_new = g["f32_step_logits"].shape[1]
    out = eng.generate(g["input_ids"], None, pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=g["grid"],
                       max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"])
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids"]) and np.array_equal(g["bf16_ids"], g["f32_ids"])
    ref = torch.from_numpy(g["f32_step_logits"])
    d_ours = (_teacher_forced_step_logits(eng, g, n_new) - ref).abs().max().item()
    d_hf = (torch.from_numpy(g["bf16_step_logits"]) - ref).abs().max().item()
    print(f"G11 tied head: step logits max|err| vs HF-fp32: ours {d_ours:.4f}, HF-bf16 {d_hf:.4f}")
    assert d_ours < LOGIT_ATOL and d_ours < 2.0 * d_hf + 0.02


def test_fp8_decode_weights_engine(need_gpu):
    """Decode on fp8 (OCP e4m3fn, power-of-two row scales) copies of the LLM matrices and the head (BASELINE config #5's
    weight format) through the whole engine.  The fixture's LLM linears are first replaced by their own dequantised values,
    which are exact in bf16, so three runs see identical weight values: (a) the CPU oracle in bf16, (b) this engine
    streaming bf16 rows, (c) this engine streaming the fp8 rows + scales.  Greedy ids of (c) must equal (a) and (b) wherever
    the oracle's top-1/top-2 margin is safe; (c)'s decode differs from (b)'s only by summation order."""
    from oracle import model_ref
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, dequantize_rows_fp8, getter_from_dict, quantize_rows_fp8
    cfg = fm.medium_config()
    W = fm.make_weights(cfg, 6)
    for k in list(W):
        if k.startswith("model.language_model.layers.") and k.endswith("_proj.weight") or k == "lm_head.weight":
            q8, sc = quantize_rows_fp8(W[k].to(torch.bfloat16))
            W[k] = dequantize_rows_fp8(q8, sc)
            assert torch.equal(W[k], W[k].to(torch.bfloat16).float())        # exact in bf16
    frames = fm.make_frames(2, 112, 140, seed=3)
    n_new = 12
    c = O3VConfig.from_dict(cfg)
    eng_bf = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda"))
    eng_f8 = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda", fp8_decode=True))
    assert eng_f8.w.fp8_decode and eng_f8.w.llm.lm_head8
    px, grid = eng_bf.pixels_from_frames(frames)
    ids = np.asarray([fm.make_prompt(cfg, [tuple(g) for g in grid.tolist()], seed=3)])
    # oracle (bf16) on the processor-shaped pixel rows
    from oracle import index_ref
    mean = np.asarray(index_ref.CLIP_MEAN, dtype=np.float32)[None, :, None, None]
    std = np.asarray(index_ref.CLIP_STD, dtype=np.float32)[None, :, None, None]
    xf = ((frames.numpy().astype(np.float64) / 255.0).astype(np.float32) - mean) / std
    pv, grid2 = index_ref.patchify_frames(xf.astype(np.float32))
    ref, ref_logits = model_ref.generate(W, cfg, ids, None, torch.from_numpy(pv), grid2, n_new, dtype=torch.bfloat16,
                                         pad_token_id=cfg["pad_token_id"], return_logits=True)
    top2 = ref_logits.float().topk(2, dim=-1).values
    margins = (top2[..., 0] - top2[..., 1])[0]
    a = eng_bf.generate(ids, None, frames=frames, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"])
    b = eng_f8.generate(ids, None, frames=frames, max_new_tokens=n_new, pad_token_id=cfg["pad_token_id"])
    ga, gb, gr = (a.sequences[0, -n_new:].cpu().numpy(), b.sequences[0, -n_new:].cpu().numpy(), ref[0, -n_new:].numpy())
    print(f"fp8 decode: oracle {gr.tolist()}\n   bf16 rows {ga.tolist()}\n   fp8 rows  {gb.tolist()}\n   oracle margins {np.round(margins.numpy(), 3).tolist()}")
    for name, got in (("bf16 rows", ga), ("fp8 rows", gb)):
        k = 0
        while k < n_new and got[k] == gr[k]:
            k += 1
        assert k == n_new or margins[k] < 0.2, f"{name}: ids leave the oracle's at step {k} with a safe margin {margins[k]:.3f}"
    # the two engines' own margins along the common prefix: same weights, different summation order only
    k = int((ga == gb).cumprod().sum())
    d = (a.margins[0, :k] - b.margins[0, :k]).abs().max().item() if k else 0.0
    print(f"   margins bf16-rows vs fp8-rows over {k} common steps: max |diff| {d:.4f}")
    assert d < LOGIT_ATOL


def test_fp8_rows_batched_decode_engine(need_gpu):
    """fp8 rows at 4..32 decode rows through the engine on a Qwen2.5-VL fixture (fragment-major fp8 images on the matrix cores for
    q/k/v with the fused rotation, o_proj, gate/up, down_proj and the lm_head): with fp8-representable weights the greedy rows of a
    G-way group follow the bf16-row engine wherever its margin is safe (same weight values, other summation order)."""
    from open_o3_video_amd.config import O3VConfig
    from open_o3_video_amd.engine import O3VEngine
    from open_o3_video_amd.weights import DeviceWeights, dequantize_rows_fp8, getter_from_dict, quantize_rows_fp8
    cfg = fm.medium_config()
    W = fm.make_weights(cfg, 6)
    for k in list(W):
        if k.startswith("model.language_model.layers.") and k.endswith("_proj.weight") or k == "lm_head.weight":
            q8, sc = quantize_rows_fp8(W[k].to(torch.bfloat16))
            W[k] = dequantize_rows_fp8(q8, sc)
    frames = fm.make_frames(2, 112, 140, seed=3)
    c = O3VConfig.from_dict(cfg)
    e1 = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda"))
    e2 = O3VEngine(c, DeviceWeights(c, getter_from_dict(W), "cuda", fp8_decode=True))
    assert e2.w.llm.layer[0].qkv_w8p and e2.w.llm.layer[0].down_w8p and e2.w.llm.lm_head8p
    px, grid = e1.pixels_from_frames(frames)
    ids = np.asarray([fm.make_prompt(cfg, [tuple(g) for g in grid.tolist()], seed=3)])
    n_new = 10
    for G in (4, 8, 19):
        a = e1.generate(ids, None, frames=frames, max_new_tokens=n_new, num_return_sequences=G, pad_token_id=cfg["pad_token_id"])
        b = e2.generate(ids, None, frames=frames, max_new_tokens=n_new, num_return_sequences=G, pad_token_id=cfg["pad_token_id"])
        ra, rb, m = a.sequences[G - 1, -n_new:].tolist(), b.sequences[G - 1, -n_new:].tolist(), a.margins[G - 1].tolist()
        k = 0
        while k < n_new and ra[k] == rb[k]:
            k += 1
        print(f"fp8 rows, {G} decode rows: follow the bf16 rows for {k}/{n_new} tokens")
        assert k == n_new or m[k] < 2 * LOGIT_ATOL, (G, k, m[k])


def test_in_launch_wait_give_up_reruns_on_stand_alone_kernels(need_gpu, golden_dir):
    """Robustness of the one-launch decode block: when one of its in-launch waits gives up (here forced -- the q/k/v ticket lines of the
    hand-off buffer are poisoned so that no workgroup ever draws the 'last' ticket and every consumer runs into its bounded spin),
    generate() notices after the first chunk of steps, re-runs the SAME call in process on the stand-alone kernels and returns the
    golden ids; the next call uses the one-launch block again.  Nothing hangs: every spin is bounded and the give-up is sticky."""
    g = np.load(os.path.join(golden_dir, "g11_tied3b.npz"))       # 16 query heads of 128: a shape the one-launch block is built for
    cfg = fm.tied3b_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 4))
    kw = dict(pixel_values=torch.from_numpy(g["pixel_values"]), image_grid_thw=g["grid"], max_new_tokens=12, pad_token_id=cfg["pad_token_id"])
    ok = eng.generate(g["input_ids"], None, **kw)
    assert np.array_equal(ok.sequences.cpu().numpy(), g["bf16_ids"])
    assert ok.timings["fused_attention_layers"] > 0 and ok.timings["standalone_attention_layers"] == 0
    assert abs(ok.timings["launches_per_layer"] - 3.0) < 1e-9            # attention block, gate/up, down
    eng._debug_poison_sync = True
    try:
        out = eng.generate(g["input_ids"], None, **kw)
    finally:
        eng._debug_poison_sync = False
    assert eng.fused_fallbacks == 1
    assert np.array_equal(out.sequences.cpu().numpy(), g["bf16_ids"])
    assert out.timings["fused_attention_layers"] == 0 and out.timings["standalone_attention_layers"] > 0
    again = eng.generate(g["input_ids"], None, **kw)
    assert again.timings["fused_attention_layers"] > 0 and eng.fused_fallbacks == 1
    assert torch.equal(again.sequences, ok.sequences)


def test_batched_rows_tail_norm_equals_separate_norm_launches(need_gpu):
    """8..32 decode rows: o_proj / down_proj normalise their result for the next linear inside their own launch
    (o3v_linear_decode_norm_next).  The ids and margins are bit-identical to the run with the separate o3v_rmsnorm launches
    (engine.tail_norm = False), for independent rows and for a sampled group, and the engine reports two launches fewer per layer (all but layer 0's first norm)."""
    cfg = fm.medium_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    eng.tail_norm = True        # opt-in (measured equal to the separate launches, profiles/r03_tail_norm_ab.txt)
    fr = fm.make_frames(2, 56, 84, seed=8)
    _, grid = eng.pixels_from_frames(fr)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=8)
    cases = [dict(prompts=[ids] * 9, frames=torch.cat([fr] * 9), kw=dict(max_new_tokens=20)),
             dict(prompts=[ids] * 8, frames=torch.cat([fr] * 8), kw=dict(max_new_tokens=20)),
             dict(prompts=[ids], frames=fr, kw=dict(max_new_tokens=20, num_return_sequences=16, do_sample=True, top_p=0.95, top_k=50,
                                                    temperature=1.0, seed=11)),
             dict(prompts=[ids], frames=fr, kw=dict(max_new_tokens=12, num_return_sequences=32, do_sample=True, top_p=0.9,
                                                    temperature=1.0, seed=12))]
    for c in cases:
        a = eng.generate(c["prompts"], None, frames=c["frames"], **c["kw"])
        try:
            eng.tail_norm = False
            b = eng.generate(c["prompts"], None, frames=c["frames"], **c["kw"])
        finally:
            eng.tail_norm = True
        assert torch.equal(a.sequences, b.sequences)
        assert a.margins is None or torch.equal(a.margins, b.margins)
        la, lb = a.timings["launches_per_layer"], b.timings["launches_per_layer"]
        L = eng.cfg.text.num_hidden_layers      # only layer 0's first norm of a step stays a launch of its own
        rows = a.sequences.shape[0]             # (at 8 rows q/k/v carries its norm fused in either run: one launch fewer, not two)
        assert abs((lb - la) - (1 if rows <= 8 else 2 - 1 / L)) < 1e-6, (la, lb)


def test_policy_logps_reuse_the_prompt_generate_prefilled(need_gpu):
    """completion_logps right after the group generate of the same prompt takes the prompt's K/V and last hidden row from that call
    (no tower pass, no prompt prefill) and returns the same bits as the pass that recomputes them; a changed pixel, a changed token
    or another mask is a different prompt and recomputes."""
    cfg = fm.medium_config()
    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    fr = fm.make_frames(2, 56, 84, seed=9)
    _, grid = eng.pixels_from_frames(fr)
    ids = fm.make_prompt(cfg, [tuple(r) for r in grid.tolist()], seed=9)
    kw = dict(max_new_tokens=10, num_return_sequences=4, do_sample=True, top_p=0.95, top_k=50, temperature=1.0, seed=3)
    out = eng.generate([ids], None, frames=fr, **kw)
    comp = out.sequences[:, len(ids):]
    a = eng.completion_logps([ids], comp, frames=fr)
    assert eng.timings_last_logps["prompt_kv_reused"]
    eng.reuse_prompt_kv, keep = False, eng._last_prompt
    eng._last_prompt = None
    b = eng.completion_logps([ids], comp, frames=fr)
    assert not eng.timings_last_logps["prompt_kv_reused"]
    assert torch.equal(a, b)
    eng.reuse_prompt_kv, eng._last_prompt = True, keep
    fr2 = fr.clone()
    fr2[1, 2, 17, 33] ^= 1                                       # one bit of one pixel
    c = eng.completion_logps([ids], comp, frames=fr2)
    assert not eng.timings_last_logps["prompt_kv_reused"]
    ids2 = list(ids)
    ids2[-1] = (ids2[-1] + 1) % 1000 + 5
    eng.completion_logps([ids2], comp, frames=fr)
    assert not eng.timings_last_logps["prompt_kv_reused"]
    # a generate without a group keeps nothing
    eng.generate([ids], None, frames=fr, max_new_tokens=4)
    eng.completion_logps([ids], comp, frames=fr)
    assert not eng.timings_last_logps["prompt_kv_reused"]
