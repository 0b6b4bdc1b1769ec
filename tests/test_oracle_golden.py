"""Pin the CPU oracle (oracle/) against golden vectors captured from the real sources
(tools/make_golden.py: the reference's vision_process.py + transformers 5.15.0)."""
import json
import os

import numpy as np
import pytest
import torch

import fixture_models as fm
from oracle import index_ref, model_ref, vision_policy as vp


@pytest.fixture(scope="module")
def g1(golden_dir):
    with open(os.path.join(golden_dir, "g1_policy.json")) as f:
        return json.load(f)


def test_constants(g1):
    for k, v in g1["constants"].items():
        assert getattr(vp, k) == v, k


def test_smart_resize(g1):
    for h, w, f, mn, mx, exp in g1["smart_resize"]:
        if exp == "ValueError":
            with pytest.raises(ValueError):
                vp.smart_resize(h, w, f, mn, mx)
        else:
            assert list(vp.smart_resize(h, w, f, mn, mx)) == exp, (h, w, mn, mx)


def test_by_factor(g1):
    for n, f, r, c, fl in g1["by_factor"]:
        assert (vp.round_by_factor(n, f), vp.ceil_by_factor(n, f), vp.floor_by_factor(n, f)) == (r, c, fl)


def test_smart_nframes(g1):
    for ele, total, vfps, exp in g1["smart_nframes"]:
        if isinstance(exp, str):
            with pytest.raises((ValueError, AssertionError)):
                vp.smart_nframes(dict(ele), total, vfps)
        else:
            assert vp.smart_nframes(dict(ele), total, vfps) == exp, (ele, total, vfps)


def test_linspace_indices(g1):
    for total, n, idx in g1["linspace"]:
        assert vp.sample_frame_indices(total, n) == idx, (total, n)


def test_video_hw(g1):
    for n, h, w, ele, exp in g1["video_hw"]:
        assert list(vp.video_resize_hw(n, h, w, ele)) == exp, (n, h, w, ele)


def test_frame_prompts(g1):
    for c in g1["frame_prompts"]:
        assert vp.frame_prompt_trainer(c["n"], c["fps"]) == c["trainer"]
        assert vp.frame_prompt_demo(c["n"], c["fps"]) == c["demo"]
        assert vp.frame_prompt_vstar(c["vstar_times"]) == c["vstar"]


def test_patchify_layout(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_patchify.npz"))
    mean = g["mean"].astype(np.float32)
    std = g["std"].astype(np.float32)
    for tag in "abc":
        fr = g[f"{tag}_frames"]
        x = (fr.astype(np.float64) * g["rescale"][0]).astype(np.float32)
        x = (x - mean[None, :, None, None]) / std[None, :, None, None]
        pv, grid = index_ref.patchify_frames(x.astype(np.float32))
        assert np.array_equal(grid, g[f"{tag}_grid"])
        assert np.array_equal(pv, g[f"{tag}_pixel_values"])  # bit-exact layout + arithmetic


def test_vit_index(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_vit_index.npz"))
    tags = sorted({k.split("_")[0] for k in g.files})
    assert len(tags) >= 6
    for t in tags:
        grid = g[f"{t}_grid"]
        wi, cu = index_ref.vision_window_index(grid)
        assert np.array_equal(wi, g[f"{t}_window_index"]), t
        assert np.array_equal(cu, g[f"{t}_cu_window"]), t
        assert np.array_equal(index_ref.vision_cu_seqlens(grid), g[f"{t}_cu_full"]), t
        assert np.array_equal(index_ref.vision_position_ids(grid), g[f"{t}_pos"]), t


def test_rope_index(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_rope_index.npz"))
    cfg = fm.tiny_config()
    for t in sorted({k.split("_")[0] for k in g.files}):
        ids, mask = g[f"{t}_ids"], g[f"{t}_mask"]
        types = (ids == cfg["image_token_id"]).astype(np.int64)
        pos, delta = index_ref.rope_index(ids, types, g[f"{t}_grid"], mask)
        assert np.array_equal(pos, g[f"{t}_pos"]), t
        assert np.array_equal(delta, g[f"{t}_delta"]), t


def test_logits_processors(golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_logits_proc.npz"))
    s, ids = torch.from_numpy(g["scores"]), torch.from_numpy(g["ids"])
    assert np.array_equal(model_ref.repetition_penalty(s.clone(), ids, 1.05).numpy(), g["rp_1p05"])
    assert np.array_equal(model_ref.repetition_penalty(s.clone(), ids, 1.3).numpy(), g["rp_1p3"])
    assert np.array_equal(model_ref.temperature_warp(s.clone(), 0.7).numpy(), g["temp_0p7"])
    for p in (0.95, 0.5, 0.001):
        assert np.array_equal(model_ref.top_p_warp(s.clone(), p).numpy(), g[f"top_p_{p}"])


def test_top_k_processor(golden_dir):
    """TopKLogitsWarper (incl. ties at the k-th value, k larger than the vocabulary) and the rollout's whole chain."""
    g = np.load(os.path.join(golden_dir, "g8b_top_k.npz"))
    s, t = torch.from_numpy(g["scores"]), torch.from_numpy(g["tied"])
    for k in (1, 50, 2000):
        assert np.array_equal(model_ref.top_k_warp(s.clone(), k).numpy(), g[f"top_k_{k}"])
        assert np.array_equal(model_ref.top_k_warp(t.clone(), k).numpy(), g[f"tied_top_k_{k}"])
    chain = model_ref.top_p_warp(model_ref.top_k_warp(model_ref.temperature_warp(s.clone(), 0.8), 50), 0.95)
    assert np.array_equal(chain.numpy(), g["chain_t0p8_k50_p0p95"])


CASES = [("g6_tiny.npz", fm.tiny_config, 0, 16), ("g6_tiny_b.npz", fm.tiny_config, 1, 12),
         ("g7_medium.npz", fm.medium_config, 2, 16)]


@pytest.mark.parametrize("fname,cfgf,wseed,n_new", CASES)
@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_model_against_hf(golden_dir, fname, cfgf, wseed, n_new, dname):
    g = np.load(os.path.join(golden_dir, fname))
    cfg = cfgf()
    W = fm.make_weights(cfg, wseed)
    dt = torch.float32 if dname == "f32" else torch.bfloat16
    pv = torch.from_numpy(g["pixel_values"])
    taps = {}
    ids, step_logits = model_ref.generate(W, cfg, g["input_ids"], None, pv, g["grid"], n_new, dtype=dt,
                                          pad_token_id=cfg["pad_token_id"], taps=taps, return_logits=True)
    # same op order as HF on the same CPU kernels -> expect (near) bit equality
    tol = 2e-5 if dname == "f32" else 0.0
    np.testing.assert_allclose(taps["vit_merged"].float().numpy(), g[f"{dname}_vit_merged"], atol=tol * 10, rtol=tol)
    assert np.array_equal(taps["rope_deltas"].numpy(), g[f"{dname}_rope_deltas"])
    k = g[f"{dname}_step_logits"].shape[1]
    np.testing.assert_allclose(step_logits[:, :k].numpy(), g[f"{dname}_step_logits"], atol=max(tol * 50, 0), rtol=tol)
    assert np.array_equal(ids.numpy(), g[f"{dname}_ids"])
    if f"{dname}_hidden_0" in g.files:
        np.testing.assert_allclose(taps["inputs_embeds"].float().numpy(), g[f"{dname}_hidden_0"], atol=tol * 10, rtol=tol)
    ids_rp = model_ref.generate(W, cfg, g["input_ids"], None, pv, g["grid"], n_new, dtype=dt,
                                pad_token_id=cfg["pad_token_id"], rep_penalty=1.05)
    assert np.array_equal(ids_rp.numpy(), g[f"{dname}_ids_rp105"])


@pytest.mark.parametrize("fname,cfgname,wseed", [("g12_q3_tiny.npz", "tiny_q3_config", 0), ("g13_q3_medium.npz", "medium_q3_config", 2)])
@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_qwen3vl_oracle_against_hf(golden_dir, fname, cfgname, wseed, dname):
    """oracle/model_ref_q3.py == transformers' Qwen3VLForConditionalGeneration (goldens G12 / G13: merged visual tokens, the
    DeepStack features, step logits, greedy ids): bit-exact in bf16, 2e-5 in fp32, like the Qwen2.5-VL oracle."""
    import fixture_models_q3 as fq
    from oracle import model_ref_q3
    g = np.load(os.path.join(golden_dir, fname))
    cfg = getattr(fq, cfgname)()
    W = fq.make_weights(cfg, wseed)
    dt = torch.float32 if dname == "f32" else torch.bfloat16
    n_new = g["f32_step_logits"].shape[1]
    taps = {}
    ids, step_logits = model_ref_q3.generate(W, cfg, g["input_ids"], None, torch.from_numpy(g["pixel_values"]), g["grid"], n_new,
                                             dtype=dt, pad_token_id=cfg["pad_token_id"], return_logits=True, taps=taps)
    tol = 2e-5 if dname == "f32" else 0.0
    np.testing.assert_allclose(taps["vit_merged"].float().numpy(), g[f"{dname}_vit_merged"], atol=tol * 10, rtol=tol)
    for j, d in enumerate(taps["deepstack"]):
        np.testing.assert_allclose(d.float().numpy(), g[f"{dname}_deepstack_{j}"], atol=tol * 10, rtol=tol)
    np.testing.assert_allclose(step_logits.numpy(), g[f"{dname}_step_logits"], atol=max(tol * 50, 0), rtol=tol)
    assert np.array_equal(ids.numpy(), g[f"{dname}_ids"])


# ------------------------------------------------------------------------------------------------ native video inputs
def test_rope_index_video(golden_dir):
    """get_rope_index over native video groups (golden G5b from transformers 5.15's Qwen2.5-VL and Qwen3-VL models): temporal
    spacing tokens_per_second * int(second_per_grid_t), several videos, video + image, left padding; Qwen3-VL's per-frame split."""
    g = np.load(os.path.join(golden_dir, "g5b_rope_index_video.npz"))
    import fixture_models_q3 as fq
    tps = int(g["tokens_per_second"][0])
    tags = sorted({k[:-4] for k in g.files if k.endswith("_ids")})
    assert len(tags) == 14
    for t in tags:
        q3 = t.startswith("q3_")
        cfg = fq.tiny_q3_config() if q3 else fm.tiny_config()
        ids, mask = g[f"{t}_ids"], g[f"{t}_mask"]
        types = index_ref.token_types(ids, cfg["image_token_id"], cfg["video_token_id"])
        spg = list(g[f"{t}_spg"]) if bool(g[f"{t}_has_spg"][0]) and not q3 else None
        pos, delta = index_ref.rope_index(ids, types, g[f"{t}_igrid"] if len(g[f"{t}_igrid"]) else None, mask,
                                          video_grid_thw=g[f"{t}_vgrid"], second_per_grid_ts=spg, tokens_per_second=tps,
                                          split_video_frames=q3)
        assert np.array_equal(pos, g[f"{t}_pos"]), t
        assert np.array_equal(delta, g[f"{t}_delta"]), t


def test_video_patchify_layout(golden_dir):
    """index_ref.patchify_video == Qwen2VLVideoProcessor.patchify (temporal pairs of DISTINCT frames, odd count padded with the
    last frame) on the goldens' frames; rescale / normalise as the PIL image processor (G3's arithmetic)."""
    for fname, patch, mean, std in (("g14_video_tiny.npz", 14, index_ref.CLIP_MEAN, index_ref.CLIP_STD),
                                    ("g14_video_medium.npz", 14, index_ref.CLIP_MEAN, index_ref.CLIP_STD),
                                    ("g15_q3_video_tiny.npz", 16, (0.5,) * 3, (0.5,) * 3)):
        g = np.load(os.path.join(golden_dir, fname))
        fr = g["video_frames"]
        x = (fr.astype(np.float64) * (1 / 255)).astype(np.float32)
        x = (x - np.asarray(mean, np.float32)[None, :, None, None]) / np.asarray(std, np.float32)[None, :, None, None]
        pv, grid = index_ref.patchify_video(x.astype(np.float32), patch=patch)
        assert np.array_equal(grid, g["video_grid"]), fname
        assert np.array_equal(pv, g["pixel_values_videos"]), fname
    assert np.load(os.path.join(golden_dir, "g14_video_medium.npz"))["video_frames"].shape[0] == 5      # the odd-count case


@pytest.mark.parametrize("fname,cfgf,wseed", [("g14_video_tiny.npz", fm.tiny_config, 0), ("g14_video_medium.npz", fm.medium_config, 2)])
@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_model_video_against_hf(golden_dir, fname, cfgf, wseed, dname):
    """Native video input through the Qwen2.5-VL oracle (goldens G14: <|video_pad|> prompt, pixel_values_videos, second_per_grid_ts;
    G14b adds an image after the video): positions, merged video tokens, step logits, greedy ids -- bit-exact in bf16."""
    g = np.load(os.path.join(golden_dir, fname))
    cfg = cfgf()
    W = fm.make_weights(cfg, wseed)
    dt = torch.float32 if dname == "f32" else torch.bfloat16
    n_new = g["f32_step_logits"].shape[1]
    has_img = "pixel_values" in g.files
    kw = dict(pixel_values_videos=torch.from_numpy(g["pixel_values_videos"]), video_grid_thw=g["video_grid"],
              second_per_grid_ts=list(g["second_per_grid_ts"]))
    pv = torch.from_numpy(g["pixel_values"]) if has_img else None
    ig = g["image_grid"] if has_img else None
    taps = {}
    ids, step_logits = model_ref.generate(W, cfg, g["input_ids"], None, pv, ig, n_new, dtype=dt, pad_token_id=cfg["pad_token_id"],
                                          taps=taps, return_logits=True, **kw)
    tol = 2e-5 if dname == "f32" else 0.0
    assert np.array_equal(taps["position_ids"].numpy(), g["position_ids"])
    assert np.array_equal(taps["rope_deltas"].numpy(), g[f"{dname}_rope_deltas"])
    np.testing.assert_allclose(taps["vit_merged_video"].float().numpy(), g[f"{dname}_vit_merged_video"], atol=tol * 10, rtol=tol)
    np.testing.assert_allclose(step_logits.numpy(), g[f"{dname}_step_logits"], atol=max(tol * 50, 0), rtol=tol)
    assert np.array_equal(ids.numpy(), g[f"{dname}_ids"])
    ids_rp = model_ref.generate(W, cfg, g["input_ids"], None, pv, ig, n_new, dtype=dt, pad_token_id=cfg["pad_token_id"], rep_penalty=1.05, **kw)
    assert np.array_equal(ids_rp.numpy(), g[f"{dname}_ids_rp105"])
    lg = model_ref.full_logits(W, cfg, g["input_ids"], None, pv, ig, dtype=dt, **kw)
    np.testing.assert_allclose(lg[:, -1].float().numpy(), g[f"{dname}_prefill_last_logits"], atol=max(tol * 50, 0), rtol=tol)


@pytest.mark.parametrize("fname,cfgname,wseed", [("g15_q3_video_tiny.npz", "tiny_q3_config", 0), ("g15_q3_video_medium.npz", "medium_q3_config", 2)])
@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_qwen3vl_video_oracle_against_hf(golden_dir, fname, cfgname, wseed, dname):
    """Native video input through the Qwen3-VL oracle (goldens G15: one <vs> pads <ve> block per temporal patch, per-frame rope
    groups, DeepStack features at the video positions; G15b interleaves video and image DeepStack rows)."""
    import fixture_models_q3 as fq
    from oracle import model_ref_q3
    g = np.load(os.path.join(golden_dir, fname))
    cfg = getattr(fq, cfgname)()
    W = fq.make_weights(cfg, wseed)
    dt = torch.float32 if dname == "f32" else torch.bfloat16
    n_new = g["f32_step_logits"].shape[1]
    has_img = "pixel_values" in g.files
    taps = {}
    ids, step_logits = model_ref_q3.generate(W, cfg, g["input_ids"], None, torch.from_numpy(g["pixel_values"]) if has_img else None,
                                             g["image_grid"] if has_img else None, n_new, dtype=dt, pad_token_id=cfg["pad_token_id"],
                                             return_logits=True, taps=taps, pixel_values_videos=torch.from_numpy(g["pixel_values_videos"]),
                                             video_grid_thw=g["video_grid"])
    tol = 2e-5 if dname == "f32" else 0.0
    np.testing.assert_allclose(taps["vit_merged_video"].float().numpy(), g[f"{dname}_vit_merged_video"], atol=tol * 10, rtol=tol)
    for j, d in enumerate(taps["deepstack_video"]):
        np.testing.assert_allclose(d.float().numpy(), g[f"{dname}_deepstack_video_{j}"], atol=tol * 10, rtol=tol)
    np.testing.assert_allclose(step_logits.numpy(), g[f"{dname}_step_logits"], atol=max(tol * 50, 0), rtol=tol)
    assert np.array_equal(ids.numpy(), g[f"{dname}_ids"])
