"""Host-side integer bookkeeping of the generate path (vectorised numpy; results are tiny and cached per grid).

Mirrors what transformers computes on the host before launching kernels (TF: = transformers 5.15.0):
window permutation and ragged segments of the ViT (TF:vision_utils.py:42-65, :81-127, :130-188), the 3-D M-RoPE
position index (TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:892-942, :944-1058, :1135-1181) and the tile lists
the attention kernel consumes.  Parity with the oracle / HF goldens: tests/test_host_logic.py.
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np

TILE = 64           # query rows per attention tile for the ragged ViT segments (csrc/o3v_attn.hip, RQ = 1)
PREFILL_TILE = 128  # query rows per tile of the causal prefill (RQ = 2: two 16-row blocks per wave)


def _grid_key(grid_thw):
    return tuple(tuple(int(v) for v in g) for g in np.asarray(grid_thw).reshape(-1, 3))


@lru_cache(maxsize=64)
def _vision_plan_cached(key, merge, window_size, patch_size):
    win = window_size // merge // patch_size
    unit = merge * merge
    widx, seg_len_win, pos_rows, seg_len_full = [], [], [], []
    base = 0
    for (t, h, w) in key:
        lh, lw = h // merge, w // merge
        nwh, nww = lh // win + 1, lw // win + 1           # TF:166-169 (a full pad window when divisible)
        r = np.arange(lh)[:, None]
        c = np.arange(lw)[None, :]
        wid = ((r // win) * nww + (c // win)).reshape(-1)  # window id of every merged token, row-major order
        order = np.argsort(wid, kind="stable")             # tokens of one window stay in row-major order
        counts = np.bincount(wid, minlength=nwh * nww)
        counts = counts[counts > 0]                        # TF:187 unique_consecutive drops empty windows
        for ti in range(t):
            widx.append(base + ti * lh * lw + order)
            seg_len_win.append(counts * unit)
            seg_len_full.append(h * w)
        # (h, w) of every patch in merge-block-major order (TF:112-127)
        bh, bw, ih, iw = np.meshgrid(np.arange(lh), np.arange(lw), np.arange(merge), np.arange(merge), indexing="ij")
        hp = (bh * merge + ih).reshape(-1)
        wp = (bw * merge + iw).reshape(-1)
        pos_rows.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
        base += t * lh * lw
    window_index = np.concatenate(widx).astype(np.int64)
    cu_win = np.concatenate([[0], np.cumsum(np.concatenate(seg_len_win))]).astype(np.int32)
    cu_full = np.concatenate([[0], np.cumsum(seg_len_full)]).astype(np.int32)
    pos = np.concatenate(pos_rows, axis=0).astype(np.int64)
    return window_index, cu_win, cu_full, pos


def vision_plan(grid_thw, merge=2, window_size=112, patch_size=14):
    """-> (window_index [P/4] i64, cu_window_seqlens i32, cu_seqlens i32, position_ids [P,2] i64)."""
    return _vision_plan_cached(_grid_key(grid_thw), merge, window_size, patch_size)


def segment_tiles(cu):
    """Non-causal tiles {q_row0,q_rows,k_row0,k_len,-1,0,0,0} for ragged segments given by cu_seqlens."""
    out = []
    for a, e in zip(cu[:-1], cu[1:]):
        a, e = int(a), int(e)
        for q0 in range(a, e, TILE):
            out.append((q0, min(TILE, e - q0), a, e - a, -1, 0, 0, 0))
    return np.asarray(out, dtype=np.int32).reshape(-1, 8)


def prefill_tiles(B, S, pad, tile=PREFILL_TILE, past=0):
    """Causal tiles over B left-padded rows of S new tokens whose K/V sit in the cache at slots past..past+S-1
    (slots 0..past-1 hold an already-computed prompt prefix; `pad` counts the left padding inside that prefix
    when past > 0, inside the new tokens otherwise)."""
    out = []
    for b in range(B):
        pb = int(pad[b])
        first = 0 if past else (pb // tile) * tile
        for q0 in range(first, S, tile):
            out.append((b * S + q0, min(tile, S - q0), 0, past + S, past + q0, pb, b, 0))
    # longest first: a causal tile's work grows with its row offset and workgroups are dispatched in list order, so the heavy
    # tiles start first and the light ones fill the tail (measured: prefill S = 4490 62.5 -> 62.2 ms, S = 20k 398 -> 396 ms:
    # within noise -- two rounds of 28 heads x 36 tiles already mix all sizes -- kept because it cannot hurt)
    out.sort(key=lambda t: -(t[4] + t[1] - t[5]))
    return np.asarray(out, dtype=np.int32).reshape(-1, 8)


def rope_index(input_ids, attention_mask, image_grid_thw, image_token_id, merge=2, video_grid_thw=None, video_token_id=None,
               second_per_grid_ts=None, tokens_per_second=1, split_video_frames=False, mode="tf5"):
    """3-D M-RoPE positions (TF:944-1058 over TF:892-942).  Returns (pos [3,B,S] i64, deltas [B] i64).

    Text run: arange + cur on all three axes.  Image of (t,h,w): t-axis = cur (+frame index), h-axis = cur + row,
    w-axis = cur + col over the merged (h/2, w/2) grid.  Video (a run of `video_token_id`, grids from `video_grid_thw`): the
    t-axis advances per temporal patch by the seconds it spans (`second_per_grid_ts`, 1 when None) times `tokens_per_second`.
    split_video_frames (Qwen3-VL, TF3:966-969): every temporal patch of a video is its own [1,h,w] block (timestamps in text
    separate them), no temporal scaling.

    mode -- the two published forms of the video arithmetic (identical for images and for t = 1):
      "tf5"    transformers 5.15 (goldens G5b / G14 / G15): step = tokens_per_second * int(second_per_grid_t) and the next
               block starts max(h, w)/merge after this one whatever the temporal extent;
      "pinned" the libraries the reference installs (transformers @336dc69d, R:setup.sh:4, and vllm 0.7.2's
               MRotaryEmbedding.get_input_positions, R:setup.sh:7): t = trunc(k * second_per_grid_t * tokens_per_second) and the
               next block starts at max(all three axes) + 1.  Restated from those releases' published algorithm (neither is
               installed): parity unpinned."""
    if mode not in ("tf5", "pinned"):
        raise ValueError(f"mode={mode!r}: 'tf5' or 'pinned'")
    ids = np.asarray(input_ids)
    B, S = ids.shape
    mask = np.ones_like(ids) if attention_mask is None else np.asarray(attention_mask)
    as_grids = lambda g: [] if g is None else [tuple(int(v) for v in r) for r in np.asarray(g).reshape(-1, 3)]
    grids = {1: as_grids(image_grid_thw), 2: as_grids(video_grid_thw)}
    if split_video_frames:
        grids[2] = [(1, h, w) for (t, h, w) in grids[2] for _ in range(t)]
    spg = None if second_per_grid_ts is None else [float(v) for v in np.asarray(second_per_grid_ts, dtype=np.float64).reshape(-1)]
    gi = {1: 0, 2: 0}
    vi = 0                                              # videos seen: indexes second_per_grid_ts
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = np.zeros(B, dtype=np.int64)
    for b in range(B):
        keep = mask[b].astype(bool)
        row = ids[b][keep]
        n = row.shape[0]
        kind = (row == image_token_id).astype(np.int8)
        if video_token_id is not None:
            kind = kind + 2 * (row == video_token_id).astype(np.int8)
        out = np.empty((3, n), dtype=np.int64)
        change = np.flatnonzero(np.diff(kind)) + 1
        starts = np.concatenate([[0], change])
        ends = np.concatenate([change, [n]])
        cur = 0
        for s, e in zip(starts, ends):
            k = int(kind[s]) if n else 0
            if k == 0:
                L = e - s
                out[:, s:e] = np.arange(L, dtype=np.int64) + cur
                cur += L
                continue
            # HF consumes one grid row per run; a run that holds several back-to-back blocks (no text between them: the
            # reference's prompts never do that) is walked block by block here.
            p = s
            while p < e:
                if gi[k] >= len(grids[k]):
                    raise ValueError("more visual placeholder runs than grid rows")
                t, h, w = grids[k][gi[k]]
                gi[k] += 1
                lh, lw = h // merge, w // merge
                cnt = t * lh * lw
                if p + cnt > e:
                    raise ValueError("Image features and image tokens do not match")
                if k == 2 and not split_video_frames:
                    sec = 1.0 if spg is None else spg[vi]
                    vi += 1
                    if mode == "tf5":
                        tsteps = np.arange(t, dtype=np.int64) * (int(tokens_per_second) * int(sec))
                    else:
                        tsteps = (np.arange(t, dtype=np.float64) * sec * tokens_per_second).astype(np.int64)
                else:
                    tsteps = np.arange(t, dtype=np.int64)
                out[0, p:p + cnt] = np.repeat(tsteps, lh * lw) + cur
                out[1, p:p + cnt] = np.tile(np.repeat(np.arange(lh), lw), t) + cur
                out[2, p:p + cnt] = np.tile(np.arange(lw), t * lh) + cur
                if mode == "tf5":
                    cur += max(h, w) // merge
                else:
                    cur = int(out[:, p:p + cnt].max()) + 1
                p += cnt
        pos[:, b, keep] = out
        deltas[b] = (out.max() + 1 - n) if n else 0
    if gi[1] != len(grids[1]) or gi[2] != len(grids[2]):
        raise ValueError("Image features and image tokens do not match")
    return pos, deltas


def decode_positions(attention_mask, deltas, n_new):
    """Positions of the generated tokens (TF:1164-1174): (#real prompt tokens + t) + delta, same on 3 axes."""
    mask = np.asarray(attention_mask)
    n_real = mask.sum(axis=1).astype(np.int64)
    t = np.arange(n_new, dtype=np.int64)[None, :]
    p = n_real[:, None] + t + np.asarray(deltas).reshape(-1, 1)
    return np.broadcast_to(p[None], (3,) + p.shape).copy()  # [3,B,n_new]


def embed_source_rows(input_ids, image_token_id, video_token_id=None, n_image_rows=None):
    """row >= 0: embedding-table row; row < 0: -(k+1) for row k of the visual-token tensor (masked_scatter order, TF:1206-1215).
    With video placeholders the visual tensor is [image tokens ; video tokens] (each modality scattered in its own order):
    the j-th <|video_pad|> takes row n_image_rows + j.  Returns (src i32, n_image_tokens, n_video_tokens)."""
    ids = np.asarray(input_ids).reshape(-1).astype(np.int64)
    is_img = ids == image_token_id
    n_img = int(is_img.sum())
    src = ids.copy()
    src[is_img] = -(np.arange(n_img, dtype=np.int64) + 1)
    n_vid = 0
    if video_token_id is not None:
        is_vid = ids == video_token_id
        n_vid = int(is_vid.sum())
        base = n_img if n_image_rows is None else int(n_image_rows)
        src[is_vid] = -(base + np.arange(n_vid, dtype=np.int64) + 1)
    return src.astype(np.int32), n_img, n_vid


def mrope_axis_table(mrope_section):
    """axis (0=t,1=h,2=w) of every rotary frequency index (TF:590-596: sections repeat over both halves)."""
    ax = []
    for i, n in enumerate(mrope_section):
        ax += [i % 3] * int(n)
    return np.asarray(ax, dtype=np.int32)


# ------------------------------------------------------------------------------------------------ Qwen3-VL
def mrope_axis_table_interleaved(mrope_section, half):
    """Interleaved M-RoPE (Qwen3-VL, TF3:368-390): rotary frequency j takes the H axis when j % 3 == 1 and j < 3*section[1], the
    W axis when j % 3 == 2 and j < 3*section[2], the T axis otherwise."""
    j = np.arange(int(half))
    ax = np.zeros(int(half), dtype=np.int32)
    ax[(j % 3 == 1) & (j < 3 * int(mrope_section[1]))] = 1
    ax[(j % 3 == 2) & (j < 3 * int(mrope_section[2]))] = 2
    return ax


def _interp_axis(n, side):
    # TF:vision_utils._interpolation_axis_taps_weights (bilinear, align_corners=True), the same float32 operation order
    i = np.arange(n, dtype=np.float32)
    src = i * np.float32(side - 1) / np.float32(max(n - 1, 1))
    fl = np.floor(src)
    lo = np.clip(fl.astype(np.int64), 0, side - 1)
    hi = np.clip(fl.astype(np.int64) + 1, 0, side - 1)
    w_lo = np.maximum(np.float32(1) - np.abs(src - fl), np.float32(0))
    w_hi = np.maximum(np.float32(1) - np.abs(src - fl - np.float32(1)), np.float32(0))
    return lo, hi, w_lo.astype(np.float32), w_hi.astype(np.float32)


@lru_cache(maxsize=64)
def _pos_embed_taps_cached(key, side, merge):
    idx_all, w_all = [], []
    for (t, h, w) in key:
        rlo, rhi, rwl, rwh = _interp_axis(h, side)
        clo, chi, cwl, cwh = _interp_axis(w, side)
        bh, bw, ih, iw = np.meshgrid(np.arange(h // merge), np.arange(w // merge), np.arange(merge), np.arange(merge), indexing="ij")
        rows = (bh * merge + ih).reshape(-1)
        cols = (bw * merge + iw).reshape(-1)
        idx = np.stack([rlo[rows] * side + clo[cols], rlo[rows] * side + chi[cols],
                        rhi[rows] * side + clo[cols], rhi[rows] * side + chi[cols]], axis=1)
        wt = np.stack([rwl[rows] * cwl[cols], rwl[rows] * cwh[cols], rwh[rows] * cwl[cols], rwh[rows] * cwh[cols]], axis=1)
        idx_all.append(np.tile(idx, (t, 1)))
        w_all.append(np.tile(wt, (t, 1)))
    return np.concatenate(idx_all).astype(np.int64), np.concatenate(w_all).astype(np.float32)


def pos_embed_taps(grid_thw, side, merge=2):
    """Qwen3-VL learned position table resampled to each image grid (TF3:643-710): per patch, in merge-block order, the four
    table rows and bilinear weights -> (idx i64 [P,4], w f32 [P,4])."""
    return _pos_embed_taps_cached(_grid_key(grid_thw), int(side), int(merge))


def deepstack_rows(input_ids, image_token_id, first=0, video_token_id=None):
    """Rows of the visual tokens among the flattened prompt rows first.. and the row of the visual-token tensor
    ([image tokens ; video tokens], as embed_source_rows) each one takes its DeepStack features from (TF3:1198-1218)."""
    ids = np.asarray(input_ids).reshape(-1)
    src, _, _ = embed_source_rows(ids, image_token_id, video_token_id)
    at = np.flatnonzero(src < 0)
    order = -(src[at].astype(np.int64) + 1)
    keep = at >= first
    return (at[keep] - first).astype(np.int32), order[keep].astype(np.int32)
