"""Grounded-CoT span parsing: the integer / float surface on which "bbox and timestamp parity" is defined.

One tokenizer-free pass turns a completion into a `Completion` record (think / answer segments, tag counts, time
stamps, boxes, `<obj>..</obj><box>[..]</box>at<t>..</t>s` claims); the reward functions (rewards.py), the V-STAR
post-processing and the test-time-scaling crops all read that record.  Behaviour restated from
R:src/r1-v/src/open_r1/reward_func.py:239-335, R:eval/tts.py:12-52 and
R:eval/test/test_vstar_multi_images.py:132-171,375-449 (goldens: tests/golden/g9_spans_rewards.json).
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

_THINK = re.compile(r"<think>(.*?)</think>", re.DOTALL)
_ANSWER = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)
_ANSWER_TRIM = re.compile(r"<answer>\s*(.*?)\s*</answer>", re.DOTALL)
_TIME = re.compile(r"<t>([\d.]+)</t>s")
_BOX = re.compile(r"<box>(\[.*?\])</box>")
_SEGMENT = re.compile(r"<t>(\d+\.?\d*)</t>s to <t>(\d+\.?\d*)</t>s")
_CLAIM_MULTI = re.compile(r"<obj>(.*?)</obj>((?:<box>\[.*?\]</box>)+)at<t>(.*?)</t>s", re.DOTALL)
_CLAIM_ONE = re.compile(r"<obj>(.*?)</obj><box>(\[.*?\])</box>at<t>(.*?)</t>s", re.DOTALL)
_OBJ_BOX = re.compile(r"<obj>(\w+)</obj><box>(\[.*?\])</box>")
_BRACKETED = re.compile(r"\[.*?\]")


@dataclass
class Claim:
    id: int
    object_name: str
    timestamp: float
    bboxes: List[Any]

    def as_dict(self):
        return {"id": self.id, "object_name": self.object_name, "timestamp": self.timestamp, "bboxes": self.bboxes}


@dataclass
class Completion:
    text: str
    think: Optional[str]            # body of the first <think>..</think>, None when absent
    answer: Optional[str]           # body of the first <answer>..</answer> (untrimmed), None when absent
    answer_trimmed: str             # the trimmed variant the answer rewards use ("" when absent)
    counts: Dict[str, int] = field(default_factory=dict)


def split_completion(text: str) -> Completion:
    t = _THINK.search(text)
    a = _ANSWER.search(text)
    at = _ANSWER_TRIM.search(text)
    counts = {tag: text.count(tag) for tag in ("<think>", "</think>", "<answer>", "</answer>")}
    return Completion(text=text, think=t.group(1) if t else None, answer=a.group(1) if a else None,
                      answer_trimmed=at.group(1).strip() if at else "", counts=counts)


def think_times(think: str) -> List[float]:
    """Every `<t>x</t>s` inside the reasoning; a malformed number voids the whole list (reward_func.py:412-416)."""
    try:
        return [float(m) for m in _TIME.findall(think)]
    except ValueError:
        return []


def boxes_in(text: str) -> List[Any]:
    out = []
    for raw in _BOX.findall(text):
        try:
            out.append(json.loads(raw))
        except Exception:
            pass
    return out


def first_box(text: str):
    """First `<box>[..]</box>` parsed as JSON; raises like json.loads when it is malformed, None when absent."""
    m = _BOX.search(text)
    return json.loads(m.group(1)) if m else None


def answer_segment(answer_text: str) -> List[float]:
    """`<t>a</t>s to <t>b</t>s` -> [a, b]; [] when absent or reversed (reward_func.py:119-133)."""
    m = _SEGMENT.search(answer_text)
    if not m:
        return []
    a, b = float(m.group(1)), float(m.group(2))
    return [] if b < a else [a, b]


def parse_claims(think: str) -> List[Claim]:
    """`<obj>name</obj><box>[..]</box>(+)at<t>time</t>s` claims of the reasoning (reward_func.py:308-335)."""
    out: List[Claim] = []
    for m in _CLAIM_MULTI.finditer(think):
        try:
            ts = float(m.group(3).strip())
            boxes = [json.loads(b) for b in _BRACKETED.findall(m.group(2))]
        except (json.JSONDecodeError, ValueError, IndexError):
            continue
        out.append(Claim(len(out), m.group(1).strip(), ts, boxes))
    return out


def has_obj_box_pair(text: str) -> bool:
    return _OBJ_BOX.search(text) is not None


# ---------------------------------------------------------------------------------- test-time scaling (R:eval/tts.py)
def parse_box_xyxy(box_str: str):
    """'[x1, y1, x2, y2]' -> floats, None unless 4 numbers with x2>=x1 and y2>=y1 (tts.py:14-29)."""
    parts = box_str.strip().replace(" ", "").replace("[", "").replace("]", "").split(",")
    try:
        vals = [float(p) for p in parts]
    except Exception:
        return None
    if len(vals) != 4:
        return None
    return vals if (vals[2] >= vals[0] and vals[3] >= vals[1]) else None


def parse_patterns(text: str):
    """[{obj, box_xyxy, t_sec}] for every single-box claim with a valid box and time (tts.py:32-45)."""
    out = []
    for m in _CLAIM_ONE.finditer(text):
        try:
            t_sec = round(float(m.group(3).strip()), 2)
        except Exception:
            continue
        box = parse_box_xyxy(m.group(2))
        if box is not None:
            out.append({"obj": m.group(1).strip(), "box_xyxy": box, "t_sec": t_sec})
    return out


def frame_index_at(t_sec: float, fps: float, n_frames: int):
    """Frame used for a claim at time t (tts.py:47-52): round(t*fps) (banker's rounding) or None past the end."""
    i = round(t_sec * fps)
    return i if i < n_frames else None


def relevance_mapping(score):
    return {2: 1.0, 1: 0.6, 0: 0.2}.get(score, 0.2)  # tts.py:79-86


# ---------------------------------------------------------------------------------- V-STAR post-processing
_ANSWER_BODY = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)


def extract_timestamps(result: str) -> List[float]:
    """Temporal answer -> [start, end] (test_vstar_multi_images.py:132-145): mm:ss rewritten to seconds, then exactly
    two standalone numbers are required."""
    m = _ANSWER_BODY.search(result)
    if m:
        result = m.group(1).strip()
    for ts in re.findall(r"(\d+:\d+)", result):
        minutes, seconds = map(int, ts.split(":"))
        result = result.replace(ts, f"<t>{minutes * 60 + seconds}</t>s")
    nums = re.findall(r"\b\d+(?:\.\d+)?\b", result)
    return [float(nums[0]), float(nums[1])] if len(nums) == 2 else []


def fix_incomplete_json(s: str) -> str:
    """Balance brackets by appending closers / prepending openers (test_vstar_multi_images.py:146-169)."""
    for o, c in (("[", "]"), ("{", "}")):
        no, nc = s.count(o), s.count(c)
        if no > nc:
            s += c * (no - nc)
        elif nc > no:
            s = o * (nc - no) + s
    return s


def extract_bounding_boxes(answer_spatial: str):
    """Spatial answer -> {second: box} (test_vstar_multi_images.py:375-449).  Boxes are returned as the model wrote
    them; `denormalize_bbox` below is the integer rescale the harness applies per box."""
    m = _ANSWER_BODY.search(answer_spatial)
    if m:
        answer_spatial = m.group(1).strip()
    m = re.search(r"```json\s*\n(\[.*?\]|\{.*?\})\s*\n```", answer_spatial, re.DOTALL)
    if not m:
        m = re.search(r"(\[[\s\S]*\]|\{[\s\S]*\})", answer_spatial, re.DOTALL)
    if not m:
        return None
    body = m.group(1).strip().replace("'", '"')
    try:
        obj = json.loads(body)
        if isinstance(obj, list) and all(isinstance(it, dict) for it in obj):
            merged = {}
            for it in obj:
                merged.update(it)
            obj = merged
        if isinstance(obj, list):
            return {str(b[0]): b[1] for b in obj}
        if isinstance(obj, dict):
            return dict(obj)
    except Exception:
        try:
            obj = json.loads(fix_incomplete_json(body))
        except Exception:
            return None
        if isinstance(obj, list):
            return list(obj)
        if isinstance(obj, dict):
            return dict(obj)
    return None


def denormalize_bbox(bbox, input_width, input_height, w, h):
    """int(b / input * original) per coordinate (test_vstar_multi_images.py:387-401)."""
    try:
        if len(bbox) == 1:
            bbox = bbox[0]
        if len(bbox) == 2:
            bbox = bbox[1]
        return [int(bbox[0] / input_width * w), int(bbox[1] / input_height * h),
                int(bbox[2] / input_width * w), int(bbox[3] / input_height * h)]
    except Exception:
        return bbox
