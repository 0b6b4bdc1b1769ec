"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the crop step of the test-time-scaling loop.

Follows R:eval/tts.py:47-75 (`read_frame_at_time`, `crop_box`) and :88-100 (`extract_and_crop`).  `crop_box` calls
`cv2.resize(crop.astype(float32), (W, H), interpolation=cv2.INTER_LINEAR).astype(uint8)`; OpenCV is not installed in
this image (and pip is unavailable), so the resize is restated from OpenCV's published algorithm for float32
INTER_LINEAR (imgproc/resize.cpp `resizeGeneric_` with `HResizeLinear` / `VResizeLinear`: pixel-centre mapping
fx = (dx + 0.5) * scale - 0.5 computed in double and narrowed to float, floor, edge clamp with zeroed fraction,
horizontal two-tap pass then vertical two-tap pass in float32).  PARITY UNPINNED against a real cv2 build: the
reference holds no fixture for this function.
"""
import numpy as np


def _coeff(dsize, ssize):
    scale = np.float64(ssize) / np.float64(dsize)
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize_linear_f32(src_hwc: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(src float32 [h,w,c], (W,H), INTER_LINEAR) restated."""
    h, w = src_hwc.shape[:2]
    xa, xb, ax0, ax1 = _coeff(W, w)
    ya, yb, by0, by1 = _coeff(H, h)
    s = src_hwc.astype(np.float32)
    rows = s[:, xa] * ax0[None, :, None] + s[:, xb] * ax1[None, :, None]          # horizontal pass [h,W,c]
    rows = rows.astype(np.float32)
    out = rows[ya] * by0[:, None, None] + rows[yb] * by1[:, None, None]
    return out.astype(np.float32)


def clip_box(box_xyxy, W, H):
    """tts.py:58-61: int() truncation then clip to the frame."""
    x1, y1, x2, y2 = (int(v) for v in box_xyxy)
    return max(0, x1), max(0, y1), min(W, x2), min(H, y2)


def crop_box(frame_chw: np.ndarray, box_xyxy):
    """tts.py:54-75 -> uint8 [3,H,W] or None for an empty crop."""
    hwc = np.transpose(frame_chw, (1, 2, 0))
    H, W, _ = hwc.shape
    x1, y1, x2, y2 = clip_box(box_xyxy, W, H)
    crop = hwc[y1:y2, x1:x2]
    if crop.size == 0:
        return None
    r = resize_linear_f32(crop.astype(np.float32), W, H).astype(np.uint8)
    return np.transpose(r, (2, 0, 1))


def extract_and_crop(frames: np.ndarray, fps: float, claims):
    """tts.py:88-100: crops of every claim whose frame exists and whose box is non-empty; more than 10 -> []."""
    out = []
    for c in claims:
        i = round(c["t_sec"] * fps)
        if not i < len(frames):
            continue
        crop = crop_box(frames[i], c["box_xyxy"])
        if crop is not None:
            out.append(crop)
    return [] if len(out) > 10 else out
