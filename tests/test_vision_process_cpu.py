"""CPU tests of the data facade (open_o3_video_amd/vision_process.py) against the reference-generated policy goldens
and the reference's documented behaviour (R:src/r1-v/src/open_r1/vision_process.py)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from open_o3_video_amd import vision_process as vp


@pytest.fixture(scope="module")
def g1(golden_dir):
    with open(os.path.join(golden_dir, "g1_policy.json")) as f:
        return json.load(f)


def test_policy_functions_match_reference(g1):
    for k, v in g1["constants"].items():
        assert getattr(vp, k) == v, k
    for h, w, f, mn, mx, exp in g1["smart_resize"]:
        if exp == "ValueError":
            with pytest.raises(ValueError):
                vp.smart_resize(h, w, f, mn, mx)
        else:
            assert list(vp.smart_resize(h, w, f, mn, mx)) == exp
    for n, f, r, c, fl in g1["by_factor"]:
        assert (vp.round_by_factor(n, f), vp.ceil_by_factor(n, f), vp.floor_by_factor(n, f)) == (r, c, fl)
    for ele, total, vfps, exp in g1["smart_nframes"]:
        if isinstance(exp, str):
            with pytest.raises((ValueError, AssertionError)):
                vp.smart_nframes(dict(ele), total, vfps)
        else:
            assert vp.smart_nframes(dict(ele), total, vfps) == exp
    for total, n, idx in g1["linspace"]:
        assert vp.sample_indices(total, n).tolist() == idx
    for n, h, w, ele, exp in g1["video_hw"]:
        mn, mx = vp.video_pixel_budget(n, ele)
        assert list(vp.smart_resize(h, w, 28, mn, mx)) == exp
    ev = g1["extract_vision_info"]
    assert vp.extract_vision_info(ev["conv"]) == ev["single"]
    assert vp.extract_vision_info([ev["conv"], ev["conv"]]) == ev["nested"]


def test_frame_prompts_match_reference(g1):
    for c in g1["frame_prompts"]:
        p = "sys " + vp.VIDEO_TAG + " question"
        assert vp.frames_as_images_prompt(p, c["n"], c["fps"], "trainer") == "sys " + c["trainer"] + " question"
        assert vp.frames_as_images_prompt(p, c["n"], c["fps"], "demo") == "sys " + c["demo"] + " question"
        assert vp.frames_as_images_prompt(p, 0, 0, "vstar", c["vstar_times"]) == "sys " + c["vstar"] + " question"
        assert vp.frames_as_images_prompt("no tag", 0, 0, "vstar", c["vstar_times"]) == c["vstar"] + "no tag"


def test_process_vision_info_video_tensor_and_images():
    # pre-decoded 640x360 clip, 491 frames @22.29 fps, nframes 32 -> the SURVEY's canonical TRAIN-RES 224x420
    g = torch.Generator().manual_seed(0)
    clip = torch.randint(0, 256, (491, 3, 36, 64), generator=g, dtype=torch.uint8)  # small spatial size for speed
    clip = torch.nn.functional.interpolate(clip.float(), size=(360, 640)).to(torch.uint8)
    conv = [{"role": "system", "content": "s"},
            {"role": "user", "content": [{"type": "video", "video": clip, "nframes": 32, "video_fps": 22.29},
                                         {"type": "text", "text": "q"}]}]
    imgs, vids, kw = vp.process_vision_info(conv, return_video_kwargs=True)
    assert imgs is None and len(vids) == 1
    assert vids[0].shape == (32, 3, 224, 420) and vids[0].dtype == torch.float32
    assert 0 <= vids[0].min() and vids[0].max() <= 255
    assert kw["fps"][0] == pytest.approx(32 / 491 * 22.29)
    two = vp.process_vision_info(conv)
    assert len(two) == 2
    # images: RGBA -> white background, smart_resize to multiples of 28
    rgba = Image.new("RGBA", (100, 60), (255, 0, 0, 0))
    out, vids2 = vp.process_vision_info([{"role": "user", "content": [{"type": "image", "image": rgba}, {"type": "text", "text": "x"}]}])
    assert vids2 is None and out[0].mode == "RGB" and out[0].size == (112, 56)
    assert out[0].getpixel((5, 5)) == (255, 255, 255)
    # list-of-frames video: padded to an even count (R:vision_process.py:319-333)
    frames = [Image.new("RGB", (64, 36), (i, i, i)) for i in range(3)]
    v, fps = vp.fetch_video({"video": frames, "fps": 1.5}, return_video_sample_fps=True)
    assert len(v) == 4 and fps == 1.5 and v[3].getpixel((0, 0)) == v[2].getpixel((0, 0))
    with pytest.raises(ValueError):
        vp.process_vision_info([{"role": "user", "content": [{"type": "video"}]}])


def test_resized_height_width_and_budget():
    clip = torch.zeros(8, 3, 100, 200, dtype=torch.uint8)
    v = vp.fetch_video({"video": clip, "resized_height": 120, "resized_width": 250})
    assert v.shape[-2:] == (112, 252)
    mn, mx = vp.video_pixel_budget(768, {})
    assert mn == 128 * 784 and mx == int(mn * 1.05)   # many frames: floor at 1.05 * min_pixels (R:...:291)
    assert vp.video_pixel_budget(32, {"max_pixels": 50176})[1] == 50176


def test_aa_tables_reproduce_torch_bicubic_antialias():
    """Tap tables handed to the HIP resize kernel against torch's own antialiased bicubic (the ATen kernel torchvision's
    resize calls, R:vision_process.py:310-315) along one axis: shrink (TRAIN-RES 640->420, 360->224), enlarge, identity."""
    import torch.nn.functional as F
    from open_o3_video_amd.vision_process import aa_tables
    g = torch.Generator().manual_seed(0)
    for n_in, n_out in [(640, 420), (360, 224), (100, 250), (37, 37), (1080, 364), (5, 3)]:
        x = torch.rand(1, 1, 1, n_in, generator=g) * 255
        ref = F.interpolate(x, size=(1, n_out), mode="bicubic", antialias=True)[0, 0, 0].numpy()
        lo, n, w = aa_tables(n_in, n_out)
        assert lo.dtype == np.int32 and w.dtype == np.float32 and (lo >= 0).all() and (lo + n <= n_in).all()
        assert np.abs(w.sum(1) - 1).max() < 1e-6
        xs = x[0, 0, 0].numpy()
        out = np.array([sum(np.float32(xs[lo[i] + j] * w[i, j]) for j in range(n[i])) for i in range(n_out)], dtype=np.float32)
        assert np.abs(out - ref).max() < 5e-4, (n_in, n_out)
    lo, n, w = aa_tables(37, 37)
    assert (n <= 5).all() and np.allclose(w.max(1), 1.0)            # same size: the centre tap only


def test_key_frame_interleave_matches_trainer_loop():
    """R:grpo_trainer.py:495-538 (transcribed in oracle/vision_policy.py): order of spliced key frames, the per-frame time
    stamps, the closing duration line, dropped late key frames; plus a hand-worked case."""
    from PIL import Image
    from oracle import vision_policy as vpo
    from open_o3_video_amd import vision_process as vp
    rng = np.random.default_rng(0)
    video = torch.from_numpy(rng.integers(0, 256, (6, 3, 28, 56), dtype=np.uint8)).float()
    kf_img = Image.fromarray(rng.integers(0, 256, (40, 30, 3), dtype=np.uint8))
    # hand-worked: fps 2 -> video frames at 0, 0.5, 1, 1.5, 2, 2.5 s; key frames at 0.6 s (round -> 1) and 2.2 s (-> 2)
    frames, fp = vp.interleave_key_frames(video, 2.0, [{"time": 0.6, "image": kf_img}, {"time": 2.2, "image": kf_img}])
    lines = fp.strip().split("\n")
    stamps = [l.split(" at ")[1].split(":")[0] for l in lines[:-1]]
    assert stamps == ["0.0s", "0.5s", "1s", "1.0s", "1.5s", "2s", "2.0s", "2.5s"]
    assert lines[-1] == "The video is in total 3 seconds." and frames.shape == (8, 3, 28, 56) and frames.dtype == video.dtype
    assert torch.equal(frames[0], video[0]) and torch.equal(frames[3], video[2]) and torch.equal(frames[7], video[5])
    exp_kf = torch.from_numpy(np.array(kf_img.resize((56, 28))).transpose(2, 0, 1)).float()
    assert torch.equal(frames[2], exp_kf) and torch.equal(frames[5], exp_kf)
    # against the transcription on random schedules (incl. key frames past the end, several at one second, none)
    for trial in range(40):
        T = int(rng.integers(1, 20))
        fps = float(rng.choice([0.5, 1.0, 1.4527, 2.0, 3.0]))
        times = sorted(rng.uniform(0, T / fps + 2, int(rng.integers(0, 5))).tolist())
        vid = torch.zeros(T, 3, 28, 28)
        vid[:, 0, 0, 0] = torch.arange(T).float()
        kfs = [{"time": t, "image": Image.new("RGB", (8, 8), (k + 1, 0, 0))} for k, t in enumerate(times)]
        got_frames, got_fp = vp.interleave_key_frames(vid, fps, kfs)
        exp_fp, order = vpo.frame_prompt_trainer_keyframes(T, fps, [round(t) for t in times])
        assert got_fp == exp_fp and got_frames.shape[0] == len(order)
        for f, (kind, idx) in zip(got_frames, order):
            if kind == "video":
                assert f[0, 0, 0].item() == idx and f[1].abs().sum() == 0
            else:
                assert f[0, 0, 0].item() == idx + 1 and f[0].min().item() == idx + 1
    out, text = vp.interleave_key_frames(video, 2.0, [], prompt="a " + vp.VIDEO_TAG + " b")
    assert torch.equal(out, video) and text == "a " + vp.frames_as_images_prompt(vp.VIDEO_TAG, 6, 2.0) + " b"


def test_upstream_profile():
    """set_profile("upstream"): the limits of the pip `qwen_vl_utils` the eval scripts import (R:eval/inference_example.py:5,
    R:eval/models/model_vllm.py:3) instead of the vendored copy's -- restated from the package's published constants, parity unpinned.
    The canonical 640x360 eval video keeps 364x644 (SURVEY section 8 EVAL-RES) instead of the trainer's 224x420, and fps sampling may take
    more than 16 frames; the vendored profile (golden G1) is untouched after switching back."""
    from open_o3_video_amd import vision_process as vp
    assert vp.get_profile() == "vendored"
    base = (vp.smart_resize(360, 640), vp.video_pixel_budget(32, {}), vp.smart_nframes({}, 491, 22.29))
    assert base[0] == (336, 588) and vp.smart_resize(360, 640, 28, *vp.video_pixel_budget(32, {})) == (224, 420) and base[2] == 16
    prev = vp.set_profile("upstream")
    try:
        assert prev == "vendored" and vp.get_profile() == "upstream"
        assert vp.MAX_PIXELS == 16384 * 784 and vp.VIDEO_MAX_PIXELS == 768 * 784 and vp.FPS_MAX_FRAMES == 768
        assert vp.smart_resize(360, 640) == (364, 644)
        assert vp.smart_resize(360, 640, 28, *vp.video_pixel_budget(32, {})) == (364, 644)
        assert vp.smart_nframes({}, 491, 22.29) == 44                      # 22 s at 2 fps, no longer capped at 16
        assert vp.smart_nframes({"nframes": 32}, 491, 22.29) == 32
        # the reference's wrapper passes max_pixels per video (R:eval/models/model_vllm.py:14,45): it still bounds the frame
        assert vp.smart_resize(360, 640, 28, *vp.video_pixel_budget(16, {"max_pixels": 360 * 420})) == (280, 504)
        with pytest.raises(ValueError):
            vp.set_profile("nope")
    finally:
        vp.set_profile("vendored")
    assert (vp.smart_resize(360, 640), vp.video_pixel_budget(32, {}), vp.smart_nframes({}, 491, 22.29)) == base
