"""The seven rollout rewards of Open-o3-Video (host-side, float64) over the decoded completions.

Same names, signatures and return values as R:src/r1-v/src/open_r1/reward_func.py (registry at
R:src/r1-v/src/open_r1/grpo.py:58-66) so the trainer can import them unchanged; implemented over the shared span
parser (spans.py) instead of per-function regex copies.  `completions` is a list of [{"role","content"}] lists, the
dataset columns arrive as lists in **kwargs (`task`, `answer`, `key_frames`, `key_items`, `image_size`,
`image_size_refine`, `step_percent`), exactly as R:…/grpo_trainer.py:644-658 passes them.
Goldens: tests/golden/g9_spans_rewards.json (generated from the reference module).
"""
from __future__ import annotations

import ast
import math
from typing import Any, Callable, List, Sequence

from . import spans

# task strings (R:src/r1-v/src/open_r1/data_loader.py)
T_VISUAL, T_TEMPORAL, T_TEMPORAL_MCQ = "visual QA", "temporal QA", "temporal QA (MCQ)"
T_GENERAL_MCQ, T_FREEFORM_TS = "General video QA MCQ", "temporal-spatial free-form QA"


def _texts(completions) -> List[str]:
    return [c[0]["content"] for c in completions]


def _guarded(fn: Callable[[int, str], float], completions) -> List[float]:
    """The reference scores each completion inside try/except and falls back to 0.0 (reward_func.py:50-82)."""
    out = []
    for i, text in enumerate(_texts(completions)):
        try:
            out.append(fn(i, text))
        except Exception as e:  # noqa: BLE001 - parity with the reference's blanket handler
            print(f"Error in reward_fn: {e}")
            out.append(0.0)
    return out


# ---------------------------------------------------------------------------------------------- geometry
def calculate_iou(box_gt, box_pred) -> float:
    """IoU of two xyxy boxes; 0.0 unless the prediction is a 4-list (reward_func.py:356-386)."""
    if not (isinstance(box_pred, list) and len(box_pred) == 4):
        return 0.0
    try:
        a = [float(v) for v in box_gt]
        b = [float(v) for v in box_pred]
    except (ValueError, TypeError, IndexError):
        return 0.0
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / union if union > 0 else 0.0


def convert_coord_format(bbox, image_size):
    """normalised xyxy -> pixels, image_size = (W, H) (reward_func.py:337-347)."""
    w, h = image_size
    return [bbox[0] * w, bbox[1] * h, bbox[2] * w, bbox[3] * h]


def convert_coord_format_gqa(bbox, image_size, image_size_refine):
    """Rescale a GQA box from the original to the refined image size, IN PLACE like the reference (:350-355)."""
    for i in range(4):
        bbox[i] = bbox[i] * image_size_refine[i % 2] / image_size[i % 2]
    return bbox


def _temporal_iou(pred: Sequence[float], gt: Sequence[float]) -> float:
    inter = max(0, min(pred[1], gt[1]) - max(pred[0], gt[0]))
    union = max(pred[1], gt[1]) - min(pred[0], gt[0])
    return inter / union if union != 0 else 0


# ---------------------------------------------------------------------------------------------- ROUGE (free-form)
ALLOW_APPROX_ROUGE = False   # opt-in: free-form answers scored by the approximate ROUGE below when rouge_score is absent
_ROUGE_WARNED = False


def _require_rouge():
    """Raises unless rouge_score is importable or the approximate scorer was opted into (then warns once).  Called by
    ans_acc_reward BEFORE its per-completion try/except, so a missing package is an error, not a silent 0.0 reward."""
    global _ROUGE_WARNED
    try:
        import rouge_score  # type: ignore # noqa: F401
        return
    except ImportError:
        pass
    if not ALLOW_APPROX_ROUGE:
        raise ImportError("ans_acc_reward's free-form branch needs the `rouge_score` package (the reference imports it); set "
                          "open_o3_video_amd.rewards.ALLOW_APPROX_ROUGE = True to use the built-in tokenizer-only ROUGE, whose "
                          "values differ from rouge_score's (no stemming): parity of this branch is unpinned")
    if not _ROUGE_WARNED:
        import warnings
        warnings.warn("rouge_score is not installed: free-form ans_acc rewards use an approximate ROUGE (no stemming); they "
                      "differ from the reference's", RuntimeWarning)
        _ROUGE_WARNED = True


def _rouge_avg_f(reference: str, hypothesis: str) -> float:
    """mean F of ROUGE-1/2/L with stemming, via the `rouge_score` package the reference uses (:28-32).  The package
    is not installable offline; without it a plain-token implementation is used -> parity of this ONE branch is
    unpinned."""
    try:
        from rouge_score import rouge_scorer  # type: ignore
        sc = rouge_scorer.RougeScorer(["rouge1", "rouge2", "rougeL"], use_stemmer=True).score(reference, hypothesis)
        return (sc["rouge1"].fmeasure + sc["rouge2"].fmeasure + sc["rougeL"].fmeasure) / 3
    except ImportError:
        pass
    _require_rouge()
    import re
    tok = lambda s: [t for t in re.sub(r"[^a-z0-9]+", " ", s.lower()).split() if t]  # noqa: E731
    r, h = tok(reference), tok(hypothesis)

    def f(match, nr, nh):
        if nr == 0 or nh == 0 or match == 0:
            return 0.0
        p, rc = match / nh, match / nr
        return 2 * p * rc / (p + rc)

    def ngram_f(n):
        from collections import Counter
        cr = Counter(tuple(r[i:i + n]) for i in range(len(r) - n + 1))
        ch = Counter(tuple(h[i:i + n]) for i in range(len(h) - n + 1))
        return f(sum((cr & ch).values()), sum(cr.values()), sum(ch.values()))

    def lcs():
        prev = [0] * (len(h) + 1)
        for x in r:
            cur = [0]
            for j, y in enumerate(h):
                cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
            prev = cur
        return prev[-1]

    return (ngram_f(1) + ngram_f(2) + f(lcs(), len(r), len(h))) / 3


# ---------------------------------------------------------------------------------------------- answer rewards
def _mcq_hit(choice: str, gt: str) -> float:
    gt = gt.strip()
    return 1.0 if choice.strip() in (gt, gt + ".", "(" + gt + ")", "[" + gt + "]") else 0.0


def ans_acc_reward(completions, answer, **kwargs):
    """Answer accuracy: ROUGE for free-form, exact option match for MCQs, 0 for grounding tasks (:17-84)."""
    task = kwargs["task"][0]
    mode = {T_TEMPORAL_MCQ: "TG_MCQ", T_GENERAL_MCQ: "MCQ", T_VISUAL: "none", T_TEMPORAL: "none"}.get(task, "free-form")
    idx = [0]  # the reference advances its own index only on success (:80): keep that quirk
    if mode == "free-form":
        _require_rouge()

    def score(i, text):
        out = spans.split_completion(text).answer_trimmed
        gt = spans.split_completion(f"<answer>{answer[i]}</answer>").answer_trimmed
        if mode == "TG_MCQ":
            gt = answer[idx[0]].split("\n")[0]
            try:
                r = _mcq_hit(out.split("Correct Option:")[1], gt)
            except Exception:  # noqa: BLE001
                r = 0.0
        elif mode == "free-form":
            r = max(0.0, min(1.0, _rouge_avg_f(gt, out)))
        elif mode == "MCQ":
            r = _mcq_hit(out, gt)
        else:
            r = 0.0
        idx[0] += 1
        return r

    return _guarded(score, completions)


def ans_tiou_reward(completions, answer, **kwargs):
    """Temporal IoU of `<t>a</t>s to <t>b</t>s` in the answer against the ground-truth segment (:86-181)."""
    task = kwargs["task"][0]
    idx = [0]

    def score(i, text):
        r = 0.0
        if task in (T_TEMPORAL, T_TEMPORAL_MCQ):
            gt = answer[idx[0]]
            if task == T_TEMPORAL_MCQ:
                gt = gt.split("\n")[1]
            gt = ast.literal_eval(gt)
            seg = spans.answer_segment(spans.split_completion(text).answer_trimmed)
            if len(seg) == 2:
                r = _temporal_iou(seg, gt)
        idx[0] += 1
        return r

    return _guarded(score, completions)


def ans_viou_reward(completions, answer, **kwargs):
    """Box IoU of the first `<box>` in the answer against the (rescaled) ground-truth box, visual QA only (:184-236)."""
    task = kwargs["task"][0]
    idx = [0]

    def score(i, text):
        r = 0.0
        if task == T_VISUAL:
            gt = spans.first_box(f"<answer>{answer[i]}</answer>")
            pred = spans.first_box(spans.split_completion(text).answer_trimmed)
            if gt is not None and pred is not None:
                gt = convert_coord_format_gqa(gt, kwargs["image_size"][idx[0]], kwargs["image_size_refine"][idx[0]])
                r = calculate_iou(gt, pred)
        idx[0] += 1
        return r

    return _guarded(score, completions)


# ---------------------------------------------------------------------------------------------- format
def format_reward(completions, **kwargs):
    """1.0 well-formed grounded reasoning, 0.5 only think+answer, 0.0 malformed (:239-305)."""
    task = kwargs["task"][0]
    out = []
    for text in _texts(completions):
        c = spans.split_completion(text)
        if c.think is None or c.answer is None:
            out.append(0.0)
            continue
        if c.counts["<think>"] != c.counts["</think>"] or c.counts["<answer>"] != c.counts["</answer>"]:
            out.append(0.0)
            continue
        n = {t: (c.think.count(f"<{t}>"), c.think.count(f"</{t}>")) for t in ("obj", "t", "box")}
        if any(a != b for a, b in n.values()):
            out.append(0.0)
            continue
        grounded = n["obj"][0] > 0 and n["t"][0] > 0 and n["box"][0] > 0
        if task in (T_TEMPORAL, T_TEMPORAL_MCQ):
            grounded = n["t"][0] >= 2
        if task == T_VISUAL and spans.has_obj_box_pair(text):
            grounded = True
        out.append(1.0 if (grounded or "General video QA" in task) else 0.5)
    return out


# ---------------------------------------------------------------------------------------------- thinking rewards
def parse_temporal_spatial_reasoning_process(think_content: str):
    return [c.as_dict() for c in spans.parse_claims(think_content)]


def thk_temporal_segment_reward(completions, **kwargs):
    """Fraction of the reasoning's time stamps that fall inside the ground-truth segment (:388-426)."""
    task = kwargs["task"][0]
    out = []
    for i, text in enumerate(_texts(completions)):
        think = spans.split_completion(text).think
        if think is None or task in (T_VISUAL, T_FREEFORM_TS) or "General video QA" in task:
            out.append(0.0)
            continue
        gt = kwargs["answer"][i]
        if task == T_TEMPORAL_MCQ:
            gt = gt.split("\n")[1]
        gt = ast.literal_eval(gt)
        times = spans.think_times(think)
        r = 0.0
        if times:
            r = sum(1.0 for t in times if gt[0] <= t <= gt[1]) / len(times)
        out.append(r)
    return out


def thk_temporal_point_reward(completions, **kwargs):
    """Mean Gaussian proximity of every reasoning time stamp to the nearest annotated key frame; sigma anneals
    4*(1-step_percent) -> 1 at 75 % of training (:429-472)."""
    task = kwargs["task"][0]
    p = kwargs["step_percent"][0]
    sigma = 4 * (1 - p) if p < 3 / 4 else 1
    out = []
    for i, text in enumerate(_texts(completions)):
        think = spans.split_completion(text).think
        if think is None or task in (T_VISUAL, T_TEMPORAL, T_TEMPORAL_MCQ) or "General video QA" in task:
            out.append(0.0)
            continue
        times = spans.think_times(think)
        if not times:
            out.append(0.0)
            continue
        gt_times = [f["time"] for f in kwargs["key_frames"][i]]
        tot = 0.0
        for t in times:
            d = min(abs(t - g) for g in gt_times)
            tot += float(math.exp(-(d ** 2) / (2 * sigma ** 2)))
        out.append(tot / len(times))
    return out


def _claim_iou(claim_boxes, gt_objects, image_size) -> float:
    """Best object IoU of one claim on one key frame: per object the mean over its gt boxes of the best claim box."""
    best = 0.0
    for gt_boxes in gt_objects.values():
        try:
            multi = isinstance(claim_boxes[0], list)
        except Exception:  # noqa: BLE001
            print("Error:", claim_boxes)
            continue
        cb = claim_boxes if multi else [claim_boxes]
        per_gt = []
        for g in gt_boxes:
            gpx = convert_coord_format(g, image_size)
            ious = [calculate_iou(gpx, c) for c in cb]
            per_gt.append(max(ious) if ious else 0.0)
        if per_gt:
            best = max(best, sum(per_gt) / len(per_gt))
    return best


def thk_spatial_reward(completions, **kwargs):
    """Spatial grounding of the reasoning (:475-605): visual QA -> best think-box IoU with the gt box; temporal tasks
    -> 0; otherwise every parsed claim is matched to the key frame nearest in time among those with
    (gt_time - pred_time) < 1.0 (signed, as in the reference) and scored by IoU, averaged over ALL claims."""
    task = kwargs["task"][0]
    out = []
    for i, text in enumerate(_texts(completions)):
        c = spans.split_completion(text)
        if c.think is None or c.answer is None:
            out.append(0.0)
            continue
        if task == T_VISUAL:
            try:
                gt = spans.first_box(kwargs["answer"][i])
            except Exception:  # noqa: BLE001
                gt = None
            preds = spans.boxes_in(c.think)
            if preds and gt is not None:
                gt = convert_coord_format_gqa(gt, kwargs["image_size"][i], kwargs["image_size_refine"][i])
                out.append(max([0.0] + [calculate_iou(gt, p) for p in preds]))
            else:
                out.append(0.0)
            continue
        if task in (T_TEMPORAL, T_TEMPORAL_MCQ) or "General video QA" in task:
            out.append(0.0)
            continue
        claims = spans.parse_claims(c.think)
        if not claims:
            out.append(0.0)
            continue
        frames = kwargs["key_frames"][i]
        items = kwargs["key_items"][i]
        total = 0.0
        for cl in claims:
            closest, best_d = -1, float("inf")
            for f in frames:
                if f["time"] - cl.timestamp < 1.0:
                    d = abs(f["time"] - cl.timestamp)
                    if d < best_d:
                        best_d, closest = d, f["time"]
            if closest == -1:
                continue
            frame = next((f for f in frames if f["time"] == closest), None)
            if cl.bboxes is not None and isinstance(cl.bboxes, list) and frame is not None:
                total += _claim_iou(cl.bboxes, items[str(frame["idx"])], kwargs["image_size"][i])
        out.append(total / len(claims))
    return out


REWARD_FUNCS = {  # R:src/r1-v/src/open_r1/grpo.py:58-66
    "ans_acc": ans_acc_reward, "ans_tiou": ans_tiou_reward, "ans_viou": ans_viou_reward,
    "thk_temporal_point": thk_temporal_point_reward, "thk_temporal_segment": thk_temporal_segment_reward,
    "thk_spatial": thk_spatial_reward, "format": format_reward,
}
