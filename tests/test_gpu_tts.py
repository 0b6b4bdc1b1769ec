"""GPU side of the test-time-scaling loop: the HIP crop/resize kernel against the CPU oracle (bit-exact uint8), the
confidence-voting loop end to end with canned model replies, and one pass through the real engine."""
import numpy as np
import pytest
import torch

import fixture_models as fm
from open_o3_video_amd import tts
from oracle import tts_ref
from test_tts_cpu import StubLLM, _SP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("T,H,W,seed", [(4, 28, 56, 0), (32, 224, 420, 1), (3, 364, 644, 2), (2, 17, 9, 3)])
def test_crop_kernel_bit_exact(need_gpu, T, H, W, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (T, 3, H, W), dtype=np.uint8)
    fps = 2.0
    claims = []
    for _ in range(9):
        x1, x2 = sorted(rng.uniform(-0.2 * W, 1.2 * W, 2).tolist())
        y1, y2 = sorted(rng.uniform(-0.2 * H, 1.2 * H, 2).tolist())
        claims.append({"obj": "o", "box_xyxy": [x1, y1, x2, y2], "t_sec": float(rng.integers(0, T)) / fps})
    claims[0]["box_xyxy"] = [0, 0, W, H]                  # identity
    claims[1]["box_xyxy"] = [W // 2, H // 2, W // 2 + 1, H // 2 + 1]   # a single pixel
    claims[2]["box_xyxy"] = [0, 0, 2, H]                  # two columns stretched over the frame
    ref = tts_ref.extract_and_crop(frames, fps, claims)
    got = tts.extract_and_crop(torch.from_numpy(frames), fps, claims)
    assert got is not None and got.shape[0] == len(ref) and got.dtype == torch.uint8
    g = got.cpu().numpy()
    for i, r in enumerate(ref):
        assert np.array_equal(g[i], r), f"crop {i}: max diff {np.abs(g[i].astype(int) - r.astype(int)).max()}"
    assert np.array_equal(g[0], frames[round(claims[0]['t_sec'] * fps)])
    # nothing to score: no valid claim, or more than ten
    assert tts.extract_and_crop(torch.from_numpy(frames), fps, []) is None
    assert tts.extract_and_crop(torch.from_numpy(frames), fps, [claims[0]] * 11) is None


def test_confidence_vote_loop(need_gpu):
    frames = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (8, 3, 56, 84), dtype=np.uint8))
    ev = "<obj>cup</obj><box>[5, 5, 40, 40]</box>at<t>1.0</t>s"
    chains = [f"<think>{ev} so</think><answer>A</answer>",       # scorer says 2 -> 1.0
              f"<think>{ev} hmm</think><answer>B</answer>",      # scorer says 1 -> 0.6
              f"<think>{ev} well</think><answer>B</answer>",     # scorer says 0 -> 0.2
              "<think>no evidence cited</think><answer>C</answer>",          # no crops -> 0.2, no scorer call
              f"<think>{ev}</think><answer>B</answer>",          # scorer replies garbage -> -1 -> 0.2
              "<think>t</think><answer>maybe</answer>"]          # NA
    llm = StubLLM(chains, digits=["2", "1", "0", "two"])
    ts = tts.TestTimeScaler(llm, N=len(chains))
    r = ts.answer("prompt", frames, 2.0, "Which?", ["A. a", "B. b", "C. c", "D. d"], _SP())
    assert r.preds == ["A", "B", "B", "C", "B", "NA"]
    assert r.scores == [1.0, 0.6, 0.2, 0.2, 0.2, 0.0] and r.n_scorer_calls == 4
    assert r.pred == "A" and abs(r.choice_score["B"] - 1.0) < 1e-9 and r.choice_score["A"] == 1.0
    # the scorer request carried one crop (the claim's box of frame round(1.0*2)=2) and the question with its options
    scorer_req = llm.calls[1][0][0]
    assert scorer_req["prompt"].count("<|image_pad|>") == 1 and "Question: Which?\nA. a\nB. b\nC. c\nD. d" in scorer_req["prompt"]
    crop = scorer_req["multi_modal_data"]["image"]
    ref = tts_ref.crop_box(frames[2].numpy(), [5, 5, 40, 40])
    assert crop.shape == (1, 3, 56, 84) and np.array_equal(crop[0].cpu().numpy(), ref)


def test_tts_through_the_engine(need_gpu, golden_dir):
    """Random-weight medium model: the chains are gibberish (every vote is NA), but the whole path -- n sampled chains in
    one call sharing the prefill, the scorer request built from GPU crops -- runs on the real engine."""
    from open_o3_video_amd.vllm_api import LLM, SamplingParams
    import zlib
    from test_gpu_facades import StubTokenizer
    from test_gpu_model import build_engine
    cfg = fm.medium_config()

    class HashTokenizer(StubTokenizer):
        """Free text (the scorer prompt) -> filler ids; 'w<ID>' words and the vision tags as in StubTokenizer."""
        def encode(self, text, add_special_tokens=False):
            for sp in self.specials:
                text = text.replace(sp, f" {sp} ")
            out = []
            for w in text.split():
                if w in self.specials:
                    out.append(self.cfg[self.specials[w]])
                elif w[0] == "w" and w[1:].isdigit():
                    out.append(int(w[1:]))
                else:
                    out.append(100 + zlib.crc32(w.encode()) % 1500)
            return out

    eng = build_engine(cfg, fm.make_weights(cfg, 2))
    llm = LLM(engine=eng, tokenizer=HashTokenizer(cfg), limit_mm_per_prompt={"image": 32}, max_model_len=4096)
    frames = fm.make_frames(4, 56, 84, seed=3)
    prompt = " ".join(["w5", "<|vision_start|>", "<|image_pad|>", "<|vision_end|>"] * 4 + ["w9", "w11"])
    ts = tts.TestTimeScaler(llm, N=10)
    r = ts.answer(prompt, frames, 1.0, "Q?", ["A. x", "B. y"],
                  SamplingParams(temperature=1.0, top_p=0.95, repetition_penalty=1.05, max_tokens=8, seed=1))
    assert len(r.preds) == 10 and set(r.preds) == {"NA"} and r.pred == "A"
    crops = tts.extract_and_crop(frames, 1.0, [{"obj": "o", "box_xyxy": [3, 4, 50, 40], "t_sec": 2.0}])
    assert ts.run_images_scorer(crops, "Q?\nA. x") in (-1, 0, 1, 2)
