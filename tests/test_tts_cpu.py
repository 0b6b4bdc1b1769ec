"""Host logic of the test-time-scaling loop (R:eval/tts.py, R:eval/test/test_videomme.py:129-226) and the CPU oracle of its
crop step.  The crop oracle restates OpenCV's float32 INTER_LINEAR (cv2 is not installable here): parity with a real cv2
build is UNPINNED; what is pinned here are the properties any correct bilinear resize has, and the selection logic."""
import numpy as np
import pytest

from open_o3_video_amd import tts
from open_o3_video_amd.spans import parse_patterns
from oracle import tts_ref


def test_resize_oracle_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 17, 3)).astype(np.float32)
    # same size: identity
    assert np.array_equal(tts_ref.resize_linear_f32(img, 17, 13), img)
    # constant image stays constant at any size
    c = np.full((5, 7, 3), 93, np.float32)
    assert np.array_equal(tts_ref.resize_linear_f32(c, 40, 31), np.full((31, 40, 3), 93, np.float32))
    # exact 2x upscale of a 1-D ramp [0, 100]: pixel-centre mapping gives 0, 25, 75, 100
    ramp = np.array([[0.0, 100.0]], np.float32)[:, :, None]
    assert tts_ref.resize_linear_f32(ramp, 4, 1)[0, :, 0].tolist() == [0.0, 25.0, 75.0, 100.0]
    # a 1x1 crop fills the frame with its value
    one = np.full((1, 1, 3), 7, np.float32)
    assert (tts_ref.resize_linear_f32(one, 9, 6) == 7).all()
    # values stay inside the source range (convex combination)
    out = tts_ref.resize_linear_f32(img, 50, 41)
    assert out.min() >= img.min() and out.max() <= img.max()


def test_crop_box_and_selection():
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (6, 3, 28, 56), dtype=np.uint8)
    full = tts_ref.crop_box(frames[2], [0, 0, 56, 28])
    assert np.array_equal(full, frames[2])                                  # whole frame: unchanged
    assert tts_ref.crop_box(frames[0], [10, 5, 10, 20]) is None            # empty
    assert tts_ref.crop_box(frames[0], [60, 5, 70, 20]) is None            # outside after clipping
    assert np.array_equal(tts_ref.crop_box(frames[0], [-5.5, -3, 99.9, 99]), frames[0])   # clipped to the frame
    claims = [{"obj": "a", "box_xyxy": [3.9, 2.2, 30.5, 20.0], "t_sec": 1.0},   # frame round(1.0*2)=2
              {"obj": "b", "box_xyxy": [0, 0, 10, 10], "t_sec": 9.0},           # frame 18: past the end
              {"obj": "c", "box_xyxy": [7, 7, 7, 9], "t_sec": 0.0},             # empty
              {"obj": "d", "box_xyxy": [1, 1, 55, 27], "t_sec": 1.25}]          # round(2.5) = 2 (banker's)
    crops = tts_ref.extract_and_crop(frames, 2.0, claims)
    boxes = tts.claim_boxes(claims, 2.0, 6, 28, 56)
    assert boxes.tolist() == [[2, 3, 2, 30, 20], [2, 1, 1, 55, 27]] and len(crops) == 2
    assert np.array_equal(crops[0], tts_ref.crop_box(frames[2], [3, 2, 30, 20]))
    # a box entirely left of / above the frame: after the clip its far corner is still negative and numpy slicing counts it
    # from the far edge -- the build keeps that behaviour of frame[y1:y2, x1:x2]
    odd = [{"obj": "e", "box_xyxy": [-9, -8, -3, -2], "t_sec": 0.0}]
    assert tts.claim_boxes(odd, 2.0, 6, 28, 56).tolist() == [[0, 0, 0, 53, 26]]
    assert np.array_equal(tts_ref.extract_and_crop(frames, 2.0, odd)[0], tts_ref.crop_box(frames[0], [0, 0, 53, 26]))
    many = [claims[0]] * 11
    assert tts_ref.extract_and_crop(frames, 2.0, many) == [] and tts.claim_boxes(many, 2.0, 6, 28, 56).shape[0] == 11


def test_scorer_messages_and_chat_rendering():
    msgs = tts.build_image_scorer_msgs(["i0", "i1", "i2"], "What is shown?\nA. x\nB. y")
    assert msgs[0] == {"role": "system", "content": tts.SCORER_SYSTEM}
    assert [c["type"] for c in msgs[1]["content"]] == ["text", "image", "image", "image"]
    assert msgs[1]["content"][0]["text"].endswith("Question: What is shown?\nA. x\nB. y")
    s = tts.render_chat(msgs)
    assert s.startswith("<|im_start|>system\n" + tts.SCORER_SYSTEM + "<|im_end|>\n<|im_start|>user\n")
    assert s.count("<|vision_start|><|image_pad|><|vision_end|>") == 3
    assert s.endswith("<|im_end|>\n<|im_start|>assistant\n")


class _Out:
    def __init__(self, text):
        self.text = text


class _Req:
    def __init__(self, texts):
        self.outputs = [_Out(t) for t in texts]


class StubLLM:
    """Returns canned chains for the sampling request and canned digits for scorer requests."""
    tokenizer = None

    def __init__(self, chains, digits=()):
        self.chains, self.digits, self.calls = chains, list(digits), []

    def generate(self, inputs, sampling_params=None):
        self.calls.append((inputs, sampling_params))
        if "Score how related" in inputs[0]["prompt"]:
            return [_Req([self.digits.pop(0)])]
        assert sampling_params.n == len(self.chains)
        return [_Req(self.chains)]


class _SP:
    n = 1
    temperature = 1.0


def test_majority_vote_loop():
    import torch
    chains = ["<think>x</think><answer>B</answer>", "<think>y</think><answer>B</answer>", "<think>z</think><answer>C</answer>",
              "<answer>C</answer>",                       # no think block: counted as a prediction with score 0
              "<think>q</think><answer>E</answer>",      # not a choice: NA
              "no tags at all"]
    llm = StubLLM(chains)
    ts = tts.TestTimeScaler(llm, N=len(chains), vote="majority_voting")
    r = ts.answer("p", torch.zeros(2, 3, 28, 28, dtype=torch.uint8), 1.0, "Q", ["A. a", "B. b"], _SP())
    assert r.preds == ["B", "B", "C", "C", "NA", "NA"] and r.scores == [1.0, 1.0, 1.0, 0.0, 0.0, 0.0]
    assert r.choice_score == {"A": 0, "B": 2.0, "C": 1.0, "D": 0} and r.pred == "B" and r.n_scorer_calls == 0
    # all chains invalid: the vote falls to the first key, as max() over the reference's dict does
    r2 = tts.TestTimeScaler(StubLLM(["", ""]), N=2, vote="majority_voting").answer(
        "p", torch.zeros(2, 3, 28, 28, dtype=torch.uint8), 1.0, "Q", [], _SP())
    assert r2.pred == "A"
    with pytest.raises(ValueError):
        tts.TestTimeScaler(llm, vote="plurality")


def test_claims_come_from_the_think_block():
    think = "The <obj>dog</obj><box>[10, 12, 40, 44]</box>at<t>2.5</t>s runs, <obj>cat</obj><box>[1,2,3]</box>at<t>1</t>s"
    c = parse_patterns(think)
    assert c == [{"obj": "dog", "box_xyxy": [10.0, 12.0, 40.0, 44.0], "t_sec": 2.5}]
