"""Oracle: integer index bookkeeping of the Qwen2.5-VL path (numpy, loop form).

Restates TF:vision_utils.py:42-65 (cu_seqlens), :81-127 (vision position ids), :130-188
(window index), TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:892-942, :944-1058 (3-D rope
index) and TF:models/qwen2_vl/image_processing_qwen2_vl.py:165-198 (patchify layout).
Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np


def vision_cu_seqlens(grid_thw):
    # TF:vision_utils.py:60-65 -- one segment of h*w per temporal slice
    lens = []
    for t, h, w in grid_thw:
        lens += [int(h) * int(w)] * int(t)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def vision_position_ids(grid_thw, merge=2):
    # TF:vision_utils.py:112-127 -- (h, w) index per patch, merge-block-major order
    out = []
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        rows = []
        for bh in range(h // merge):
            for bw in range(w // merge):
                for ih in range(merge):
                    for iw in range(merge):
                        rows.append((bh * merge + ih, bw * merge + iw))
        rows = np.asarray(rows, dtype=np.int64).reshape(-1, 2)
        out.append(np.tile(rows, (t, 1)))
    return np.concatenate(out, axis=0)


def vision_window_index(grid_thw, merge=2, window_size=112, patch_size=14):
    # TF:vision_utils.py:156-188
    win = window_size // merge // patch_size
    unit = merge * merge
    window_index = []
    cu = [0]
    base = 0
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        lh, lw = h // merge, w // merge
        pad_h = win - lh % win  # NB: a full extra window when divisible (TF:166-167)
        pad_w = win - lw % win
        nh, nw = (lh + pad_h) // win, (lw + pad_w) // win
        for ti in range(t):
            for wh in range(nh):
                for ww in range(nw):
                    cnt = 0
                    for ih in range(win):
                        for iw in range(win):
                            r, c = wh * win + ih, ww * win + iw
                            if r < lh and c < lw:
                                window_index.append(base + ti * lh * lw + r * lw + c)
                                cnt += 1
                    cu.append(cu[-1] + cnt * unit)
        base += t * lh * lw
    cu = np.asarray(cu, dtype=np.int32)
    # torch.unique_consecutive (TF:187) drops the empty windows
    keep = np.concatenate([[True], cu[1:] != cu[:-1]])
    return np.asarray(window_index, dtype=np.int64), cu[keep]


def rope_index(input_ids, mm_token_type_ids, image_grid_thw, attention_mask=None, merge=2, video_grid_thw=None,
               second_per_grid_ts=None, tokens_per_second=1, split_video_frames=False):
    """TF:modeling_qwen2_5_vl.py:944-1058 (+ :892-942 get_vision_position_ids).  mm_token_type_ids: 0 text, 1 image, 2 video.
    A run of equal types is ONE group and consumes ONE grid row of its modality (TF:1027-1032).  Video groups space the
    temporal axis by `tokens_per_second * int(second_per_grid_t)` (TF:1046-1047; 1 when second_per_grid_ts is None).
    split_video_frames: Qwen3-VL (TF:models/qwen3_vl/modeling_qwen3_vl.py:966-969) -- every video grid [t,h,w] becomes t rows
    [1,h,w] (timestamps separate the frames in the prompt) and no temporal scaling is applied.

    Returns position_ids [3,B,S] int64 and rope_deltas [B,1] int64."""
    input_ids = np.asarray(input_ids)
    B, S = input_ids.shape
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = []
    img = [] if image_grid_thw is None else [tuple(int(v) for v in g) for g in np.asarray(image_grid_thw).reshape(-1, 3)]
    vid = [] if video_grid_thw is None else [tuple(int(v) for v in g) for g in np.asarray(video_grid_thw).reshape(-1, 3)]
    if split_video_frames:
        vid = [(1, h, w) for (t, h, w) in vid for _ in range(t)]
    iters = {1: iter(img), 2: iter(vid)}
    spg = iter([1] * S if second_per_grid_ts is None else list(second_per_grid_ts))
    for b in range(B):
        types = np.asarray(mm_token_type_ids[b])
        if attention_mask is not None:
            keep = np.asarray(attention_mask[b]).astype(bool)
            types = types[keep]
        n = len(types)
        # group consecutive runs (TF:1027-1032)
        runs = []
        i = 0
        while i < n:
            j = i
            while j < n and types[j] == types[i]:
                j += 1
            runs.append((int(types[i]), i, j))
            i = j
        cur = 0
        cols = []
        for kind, s, e in runs:
            if kind == 0:
                L = e - s
                cols.append(np.tile(np.arange(L, dtype=np.int64) + cur, (3, 1)))
                cur += L
            else:
                t, h, w = next(iters[kind])
                interval = 1
                if kind == 2 and not split_video_frames:
                    interval = int(tokens_per_second) * int(next(spg))       # TF:1047: int() truncates the seconds
                lh, lw = h // merge, w // merge
                tt, hh, ww = np.meshgrid(np.arange(t) * interval, np.arange(lh) + cur, np.arange(lw) + cur, indexing="ij")
                v = np.stack([tt.reshape(-1) + cur, hh.reshape(-1), ww.reshape(-1)]).astype(np.int64)
                cols.append(v)
                cur += max(h, w) // merge
        llm = np.concatenate(cols, axis=1)
        assert llm.shape[1] == n, "placeholder count does not match grid"
        if attention_mask is not None:
            pos[:, b, keep] = llm
        else:
            pos[:, b] = llm
        deltas.append(int(llm.max()) + 1 - n)
    return pos, np.asarray(deltas, dtype=np.int64).reshape(B, 1)


def token_types(input_ids, image_token_id, video_token_id):
    """mm_token_type_ids as the processors build them (TF:models/qwen2_5_vl/processing_qwen2_5_vl.py: 1 at image pads, 2 at
    video pads)."""
    ids = np.asarray(input_ids)
    return (ids == image_token_id).astype(np.int64) + 2 * (ids == video_token_id).astype(np.int64)


def patchify_video(frames_f32, patch=14, merge=2, temporal=2):
    """TF:models/qwen2_vl/video_processing_qwen2_vl.py:236-274 for ONE video: frames_f32 [T,3,H,W] rescaled + normalised.
    An odd frame count repeats the last frame (:247-250); temporal patch k holds frames 2k and 2k+1; rows are
    (grid_t, gh/m, gw/m, m, m), columns (channel, t, ph, pw).  -> ([T/2*gh*gw, 3*2*p*p], [[T/2, gh, gw]])."""
    T, C, H, W = frames_f32.shape
    if T % temporal:
        frames_f32 = np.concatenate([frames_f32, np.repeat(frames_f32[-1:], temporal - T % temporal, axis=0)], axis=0)
        T = frames_f32.shape[0]
    gt, gh, gw = T // temporal, H // patch, W // patch
    x = frames_f32.reshape(gt, temporal, C, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.transpose(0, 3, 6, 4, 7, 2, 1, 5, 8)   # gt, gh/m, gw/m, m, m, C, tp, ph, pw
    flat = x.reshape(gt * gh * gw, C * temporal * patch * patch)
    return np.ascontiguousarray(flat), np.asarray([[gt, gh, gw]], dtype=np.int64)


def patchify_frames(frames_f32, patch=14, merge=2, temporal=2):
    """TF:image_processing_qwen2_vl.py:165-198 for a batch of single frames.

    frames_f32: [T,3,H,W] already rescaled+normalised float32.  Each frame is one image:
    duplicated along time (TF:189-197) -> [T*gh*gw, 3*2*14*14], rows in merge-block-major
    order, columns ordered (channel, t, ph, pw)."""
    T, C, H, W = frames_f32.shape
    gh, gw = H // patch, W // patch
    x = frames_f32.reshape(T, C, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.transpose(0, 2, 5, 3, 6, 1, 4, 7)  # T, gh/m, gw/m, m, m, C, ph, pw
    x = x.reshape(T * gh * gw, C, 1, patch, patch)
    x = np.repeat(x, temporal, axis=2)
    flat = x.reshape(T * gh * gw, C * temporal * patch * patch)
    grid = np.asarray([[1, gh, gw]] * T, dtype=np.int64)
    return np.ascontiguousarray(flat), grid


# OPENAI_CLIP constants, TF:image_utils (used by TF:image_processing_qwen2_vl.py:94-101)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
