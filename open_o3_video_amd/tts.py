"""Test-time scaling (BASELINE config #5): N sampled reasoning chains per question, each scored by how relevant the
image evidence it cites is, then a score-weighted vote over the multiple-choice answers.

Restates R:eval/test/test_videomme.py:129-226 (the sampling / voting loop) and R:eval/tts.py:47-123 (evidence crops and
the scorer prompt).  Differences in HOW, not in WHAT:
  * the N chains are ONE `generate` call with `n=N` (one ViT pass, one prefill, prompt K/V shared by the N decode rows)
    instead of N sequential calls that each re-encode the video;
  * the evidence crops are cut and resized on the GPU from the frames that are already resident for the ViT
    (`o3v_crop_resize_bilinear`), not with numpy/cv2 on the host;
  * the scorer runs on the same engine (the reference's vLLM wrapper also reuses its one engine,
    R:eval/models/model_vllm.py:108-122).
There is no CPU path: the crops need the HIP library.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .spans import parse_patterns, relevance_mapping

SCORER_SYSTEM = "You are a helpful assistant. Only reply with a single digit: 0, 1, or 2."
SCORER_USER = (
    "You will be given a video question and a set of cropped images extracted from the video.\n"
    "Score how related these images are to answering the question.\n\n"
    "Scoring rules:\n"
    "2 = clearly relevant to answering the question\n"
    "1 = might be useful but uncertain\n"
    "0 = not relevant at all\n\n"
    "Only output one of: 0, 1, or 2. No other text.\n"
    "Question: {question}"
)
MAX_CROPS = 10   # tts.py:98-99: more evidence images than this and the chain gets the floor score


def build_image_scorer_msgs(images, question):
    """Chat messages of the scorer request (tts.py:102-123): one text item, then one image item per crop."""
    content = [{"type": "text", "text": SCORER_USER.format(question=question)}]
    content += [{"type": "image", "image": im} for im in images]
    return [{"role": "system", "content": SCORER_SYSTEM}, {"role": "user", "content": content}]


def render_chat(messages, add_generation_prompt=True) -> str:
    """Qwen2.5-VL chat template applied to `messages` (what `processor.apply_chat_template(msgs, tokenize=False,
    add_generation_prompt=True)` returns, R:eval/models/model_vllm.py:109): ChatML turns, every image item rendered as
    <|vision_start|><|image_pad|><|vision_end|> in place.  The checkpoint's own template file is not available offline,
    so this is the published template restated; a tokenizer with `apply_chat_template` takes precedence in the scaler."""
    out = []
    for m in messages:
        out.append(f"<|im_start|>{m['role']}\n")
        if isinstance(m["content"], str):
            out.append(m["content"])
        else:
            for item in m["content"]:
                if item.get("type") == "image" or "image" in item:
                    out.append("<|vision_start|><|image_pad|><|vision_end|>")
                elif item.get("type") == "video" or "video" in item:
                    out.append("<|vision_start|><|video_pad|><|vision_end|>")
                elif "text" in item:
                    out.append(item["text"])
        out.append("<|im_end|>\n")
    if add_generation_prompt:
        out.append("<|im_start|>assistant\n")
    return "".join(out)


def claim_boxes(claims, fps: float, T: int, H: int, W: int) -> np.ndarray:
    """Claims [{box_xyxy, t_sec}] -> int32 [n,5] {frame, x1, y1, x2, y2} of the non-empty in-range ones
    (tts.py:47-52 frame pick with Python's round(); :58-70 int() truncation, clip, empty-crop drop)."""
    rows = []
    for c in claims:
        i = round(c["t_sec"] * fps)
        if not i < T:
            continue
        if i < 0:
            i += T            # the reference indexes a Python sequence: negative indices wrap
            if i < 0:
                continue
        x1, y1, x2, y2 = (int(v) for v in c["box_xyxy"])
        # the reference slices a numpy array with the clipped corners, so a corner that is still negative after the clip
        # counts from the far edge (frame[y1:y2, x1:x2] semantics), e.g. x2 = -3 -> W - 3
        x1, x2, _ = slice(max(0, x1), min(W, x2)).indices(W)
        y1, y2, _ = slice(max(0, y1), min(H, y2)).indices(H)
        if x2 <= x1 or y2 <= y1:
            continue
        rows.append((i, x1, y1, x2, y2))
    return np.asarray(rows, dtype=np.int32).reshape(-1, 5)


def extract_and_crop(frames: torch.Tensor, fps: float, claims) -> Optional[torch.Tensor]:
    """uint8 frames [T,3,H,W] (moved to the GPU if they are not there) -> uint8 crops [n,3,H,W] on the GPU, or None when
    there is nothing to score (no usable claim, or more than MAX_CROPS of them: tts.py:88-100)."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[1] != 3:
        raise ValueError("frames must be uint8 [T,3,H,W]")
    if not torch.cuda.is_available():
        raise _lib.O3VError("test-time-scaling crops run on the GPU (no CPU path)")
    T, _, H, W = frames.shape
    boxes = claim_boxes(claims, fps, T, H, W)
    n = boxes.shape[0]
    if n == 0 or n > MAX_CROPS:
        return None
    fr = frames.cuda().contiguous()
    bx = torch.from_numpy(boxes).to(fr.device)
    out = torch.empty((n, 3, H, W), dtype=torch.uint8, device=fr.device)
    _lib.call("o3v_crop_resize_bilinear", C.c_void_p(fr.data_ptr()), C.c_void_p(bx.data_ptr()), C.c_void_p(out.data_ptr()),
              n, T, H, W, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


_ANSWER = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)
_THINK = re.compile(r"<think>(.*?)</think>", re.DOTALL)
CHOICES = ("A", "B", "C", "D")


@dataclass
class ScaledAnswer:
    pred: str                                   # voted choice
    choice_score: dict                          # {"A": .., "B": .., "C": .., "D": ..}
    preds: List[str] = field(default_factory=list)     # per chain: choice or "NA"
    scores: List[float] = field(default_factory=list)  # per chain
    n_scorer_calls: int = 0


class TestTimeScaler:
    """`llm`: an `open_o3_video_amd.vllm_api.LLM` (or anything with its `generate`).  `sampling_params` must sample
    (temperature > 0) for the chains to differ; its `n` is overridden by N."""
    __test__ = False  # not a pytest class

    def __init__(self, llm, N: int = 16, vote: str = "confidence_voting", scorer_sampling_params=None):
        if vote not in ("confidence_voting", "majority_voting"):
            raise ValueError(vote)
        self.llm, self.N, self.vote = llm, int(N), vote
        self.scorer_sp = scorer_sampling_params

    def _scorer_prompt(self, msgs) -> str:
        tok = getattr(self.llm, "tokenizer", None)
        if tok is not None and getattr(tok, "chat_template", None):
            return tok.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
        return render_chat(msgs)

    def run_images_scorer(self, crops: torch.Tensor, question: str) -> int:
        """R:eval/models/model_vllm.py:108-122: 0/1/2 when the reply is exactly that digit, -1 otherwise."""
        from .vllm_api import SamplingParams
        images = [c for c in crops]
        msgs = build_image_scorer_msgs(images, question)
        sp = self.scorer_sp or SamplingParams(temperature=0.0, max_tokens=4, repetition_penalty=1.05)
        out = self.llm.generate([{"prompt": self._scorer_prompt(msgs), "multi_modal_data": {"image": torch.stack(images)}}],
                                sampling_params=sp)
        text = out[0].outputs[0].text
        return int(text) if text in ("0", "1", "2") else -1

    def score_chain(self, text: str, frames: torch.Tensor, fps: float, question_with_options: str):
        """One chain -> (choice or "NA", score), test_videomme.py:146-217 (think mode)."""
        m = _ANSWER.search(text)
        if not m or m.group(1).strip() not in CHOICES:
            return "NA", 0.0, 0
        ans = m.group(1).strip()
        t = _THINK.search(text)
        if not t:
            return ans, 0.0, 0
        if self.vote == "majority_voting":
            return ans, 1.0, 0
        crops = extract_and_crop(frames, fps, parse_patterns(t.group(1).strip()))
        if crops is None:
            return ans, 0.2, 0
        return ans, relevance_mapping(self.run_images_scorer(crops, question_with_options)), 1

    def answer(self, prompt: str, frames: torch.Tensor, fps: float, question: str, options: Sequence[str],
               sampling_params) -> ScaledAnswer:
        """`prompt`: the full text prompt with one <|image_pad|> per frame; `frames`: uint8 [T,3,H,W]."""
        import copy
        sp = copy.copy(sampling_params)
        sp.n = self.N
        req = {"prompt": prompt, "multi_modal_data": {"image": frames}}
        outs = self.llm.generate([req], sampling_params=sp)[0].outputs
        q = question + "\n" + "\n".join(str(o) for o in options)
        res = ScaledAnswer(pred="", choice_score={c: 0 for c in CHOICES})
        for o in outs:
            ans, score, calls = self.score_chain(o.text, frames, fps, q)
            res.preds.append(ans)
            res.scores.append(score)
            res.n_scorer_calls += calls
            if ans != "NA":
                res.choice_score[ans] += score
        res.pred = max(res.choice_score, key=res.choice_score.get)   # ties: first of A, B, C, D like the reference dict
        return res
