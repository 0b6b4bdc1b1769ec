"""Rewards and span parsers (open_o3_video_amd/rewards.py, spans.py) against goldens produced by running the
reference's own reward_func.py / eval/tts.py / test_vstar_multi_images.py functions (tools/make_golden.py g9).
This is the "integer bbox / timestamp" parity surface of the north star.  The free-form ROUGE branch of
ans_acc_reward needs the `rouge_score` package (absent offline): parity unpinned for that one branch."""
import copy
import json
import os
import re

import pytest

from open_o3_video_amd import rewards, spans


@pytest.fixture(scope="module")
def g9(golden_dir):
    with open(os.path.join(golden_dir, "g9_spans_rewards.json")) as f:
        return json.load(f)


def _kwargs(rec, n):
    key_frames = [{"idx": 3, "time": 3.0}, {"idx": 8, "time": 8.0}, {"idx": 15, "time": 15.5}]
    key_items = {"3": {"man": [[0.0, 0.05, 0.2, 0.6]], "ball": [[0.45, 0.55, 0.55, 0.75]]},
                 "8": {"b": [[0.15, 0.14, 0.47, 0.7]]},
                 "15": {"person": [[0.04, 0.02, 0.15, 0.23], [0.5, 0.3, 0.66, 0.84]]}}
    return dict(key_frames=[copy.deepcopy(key_frames) for _ in range(n)], key_items=[copy.deepcopy(key_items) for _ in range(n)],
                image_size=[(640, 360)] * n, image_size_refine=[(420, 224)] * n, task=[rec["task"]] * n,
                answer=[rec["answer"]] * n, step_percent=[rec["step_percent"]] * n)


def test_rewards_match_reference(g9, capsys):
    comps = [[{"role": "assistant", "content": c}] for c in g9["completions"]]
    n = len(comps)
    checked = 0
    for rec in g9["cases"]:
        for fn, expected in rec["rewards"].items():
            kw = _kwargs(rec, n)
            f = getattr(rewards, fn)
            if fn in ("ans_acc_reward", "ans_tiou_reward", "ans_viou_reward"):
                a = kw.pop("answer")
                got = f(copy.deepcopy(comps), a, **kw)
            else:
                got = f(copy.deepcopy(comps), **kw)
            assert len(got) == n
            for i, (g, e) in enumerate(zip(got, expected)):
                assert g == pytest.approx(e, abs=1e-12), (rec["task"], fn, i, g9["completions"][i])
            checked += n
    assert checked > 1500
    capsys.readouterr()


def test_registry_names():
    # R:src/r1-v/src/open_r1/grpo.py:58-66
    assert set(rewards.REWARD_FUNCS) == {"ans_acc", "ans_tiou", "ans_viou", "thk_temporal_point", "thk_temporal_segment",
                                         "thk_spatial", "format"}


def test_claims_and_tts_patterns(g9):
    for text, claims, tts in zip(g9["completions"], g9["claims"], g9["tts"]):
        m = re.search(r"<think>(.*?)</think>", text, re.DOTALL)
        got = rewards.parse_temporal_spatial_reasoning_process(m.group(1)) if m else None
        assert got == claims, text
        assert spans.parse_patterns(text) == tts, text


def test_vstar_postprocessing(g9):
    for text, ts, bb in zip(g9["completions"], g9["vstar_ts"], g9["vstar_bb"]):
        assert spans.extract_timestamps(text) == ts, text
        assert spans.extract_bounding_boxes(text) == bb, text
    # integer denormalisation int(b / input * original), R:eval/test/test_vstar_multi_images.py:387-401
    assert spans.denormalize_bbox([322, 182, 644, 364], 644, 364, 1280, 720) == [640, 360, 1280, 720]
    assert spans.denormalize_bbox([[1, 2, 3, 4]], 2, 2, 10, 10) == [5, 10, 15, 20]
    assert spans.denormalize_bbox("junk", 2, 2, 10, 10) == "junk"


def test_iou_and_misc(g9):
    for a, b, e in g9["iou"]:
        assert rewards.calculate_iou(a, b) == pytest.approx(e, abs=1e-15)
    for t, fps, n, e in g9["tts_frame_idx"]:
        assert spans.frame_index_at(t, fps, n) == e
    for s, e in g9["relevance"]:
        assert spans.relevance_mapping(s) == e
    assert spans.fix_incomplete_json('{"a": [1, 2') == '{"a": [1, 2]}'
    assert spans.fix_incomplete_json('1, 2]}') == '{[1, 2]}'
