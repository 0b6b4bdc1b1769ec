"""process_vision_info and friends: the data facade of the generate path (facade 3).

Same call surface and policy as the reference's vendored qwen_vl_utils (R:src/r1-v/src/open_r1/vision_process.py):
`process_vision_info(conversations, return_video_kwargs=False) -> (image_inputs|None, video_inputs|None[, {'fps': [...]}])`,
`smart_resize`, `smart_nframes`, `fetch_image`, `fetch_video`, same constants and environment knobs
(VIDEO_MAX_PIXELS, FORCE_QWENVL_VIDEO_READER).  The integer policy is pinned by tests/golden/g1_policy.json.
Video decoding stays on the host exactly as in the reference (decord / torchvision / cv2, whichever is installed);
pre-decoded frames can be passed directly as a uint8 tensor/ndarray [T,3,H,W] in the "video" field.
The bicubic-antialias resize uses torch's `_upsample_bicubic2d_aa` (the op torchvision dispatches to); torchvision
is not installable offline, so the resize numerics are unpinned.
"""
from __future__ import annotations

import base64
import functools
import importlib.util
import logging
import math
import os
from functools import lru_cache
from io import BytesIO
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

logger = logging.getLogger(__name__)

IMAGE_FACTOR = 28
MIN_PIXELS = 4 * 28 * 28
MAX_PIXELS = 256 * 28 * 28
MAX_RATIO = 200
VIDEO_MIN_PIXELS = 128 * 28 * 28
VIDEO_MAX_PIXELS = 128 * 28 * 28
FRAME_FACTOR = 2
FPS = 2.0
FPS_MIN_FRAMES = 4
FPS_MAX_FRAMES = 16
VIDEO_TOTAL_PIXELS = int(float(os.environ.get("VIDEO_MAX_PIXELS", 128000 * 28 * 28 * 0.9)))
FORCE_QWENVL_VIDEO_READER = os.getenv("FORCE_QWENVL_VIDEO_READER", None)

# Two sets of limits exist around the reference.  "vendored" (default; the values above) is the copy of qwen_vl_utils the
# TRAINER imports (R:src/r1-v/src/open_r1/vision_process.py:25-42, pinned by golden G1).  "upstream" is the pip package
# `qwen_vl_utils` that the EVAL scripts import instead (R:eval/inference_example.py:5, R:eval/models/model_vllm.py:3; unpinned in
# R:setup.sh:5): its published constants allow far larger frames (a 640x360 video stays 364x644 instead of 224x420) and up to
# 768 sampled frames.  The package is not installed offline, so the "upstream" values are a restatement of its published
# vision_process.py (0.0.10 / 0.0.11): parity unpinned.  Eval users select it with set_profile("upstream").
PROFILES = {
    "vendored": dict(MIN_PIXELS=4 * 28 * 28, MAX_PIXELS=256 * 28 * 28, VIDEO_MIN_PIXELS=128 * 28 * 28,
                     VIDEO_MAX_PIXELS=128 * 28 * 28, FPS_MIN_FRAMES=4, FPS_MAX_FRAMES=16),
    "upstream": dict(MIN_PIXELS=4 * 28 * 28, MAX_PIXELS=16384 * 28 * 28, VIDEO_MIN_PIXELS=128 * 28 * 28,
                     VIDEO_MAX_PIXELS=768 * 28 * 28, FPS_MIN_FRAMES=4, FPS_MAX_FRAMES=768),
}
_profile = "vendored"


def set_profile(name: str) -> str:
    """Switch the module's limits (MIN/MAX_PIXELS, VIDEO_MIN/MAX_PIXELS, FPS_MIN/MAX_FRAMES) to `name`; returns the previous
    profile's name.  Every function of this module reads them at call time."""
    global _profile
    if name not in PROFILES:
        raise ValueError(f"profile {name!r}: one of {sorted(PROFILES)}")
    prev = _profile
    globals().update(PROFILES[name])
    _profile = name
    return prev


def get_profile() -> str:
    return _profile


def round_by_factor(number, factor):
    return round(number / factor) * factor


def ceil_by_factor(number, factor):
    return math.ceil(number / factor) * factor


def floor_by_factor(number, factor):
    return math.floor(number / factor) * factor


def smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=None, max_pixels=None):
    """(H, W) divisible by `factor`, area within [min_pixels, max_pixels], aspect ratio kept as closely as possible.
    min_pixels / max_pixels default to the active profile's MIN_PIXELS / MAX_PIXELS."""
    min_pixels = MIN_PIXELS if min_pixels is None else min_pixels
    max_pixels = MAX_PIXELS if max_pixels is None else max_pixels
    ratio = max(height, width) / min(height, width)
    if ratio > MAX_RATIO:
        raise ValueError(f"absolute aspect ratio must be smaller than {MAX_RATIO}, got {ratio}")
    h_bar, w_bar = max(factor, round_by_factor(height, factor)), max(factor, round_by_factor(width, factor))
    area = h_bar * w_bar
    if area > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        return floor_by_factor(height / beta, factor), floor_by_factor(width / beta, factor)
    if area < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        return ceil_by_factor(height * beta, factor), ceil_by_factor(width * beta, factor)
    return h_bar, w_bar


def smart_nframes(ele: dict, total_frames: int, video_fps) -> int:
    """Frame count: explicit `nframes` (rounded to even) or duration*fps clamped to [min_frames, max_frames]."""
    assert not ("fps" in ele and "nframes" in ele), "Only accept either `fps` or `nframes`"
    if "nframes" in ele:
        nframes = round_by_factor(ele["nframes"], FRAME_FACTOR)
    else:
        lo = ceil_by_factor(ele.get("min_frames", FPS_MIN_FRAMES), FRAME_FACTOR)
        hi = floor_by_factor(ele.get("max_frames", min(FPS_MAX_FRAMES, total_frames)), FRAME_FACTOR)
        want = total_frames / video_fps * ele.get("fps", FPS)
        if want > total_frames:
            logger.warning(f"smart_nframes: nframes[{want}] > total_frames[{total_frames}]")
        nframes = floor_by_factor(min(min(max(want, lo), hi), total_frames), FRAME_FACTOR)
    if not (FRAME_FACTOR <= nframes <= total_frames):
        raise ValueError(f"nframes should in interval [{FRAME_FACTOR}, {total_frames}], but got {nframes}.")
    return nframes


def sample_indices(total_frames: int, nframes: int):
    return torch.linspace(0, total_frames - 1, nframes).round().long()


def to_rgb(pil_image):
    from PIL import Image
    if pil_image.mode == "RGBA":
        bg = Image.new("RGB", pil_image.size, (255, 255, 255))
        bg.paste(pil_image, mask=pil_image.split()[3])
        return bg
    return pil_image.convert("RGB")


def fetch_image(ele: dict, size_factor: int = IMAGE_FACTOR):
    from PIL import Image
    image = ele["image"] if "image" in ele else ele["image_url"]
    obj = None
    if isinstance(image, Image.Image):
        obj = image
    elif image.startswith("http://") or image.startswith("https://"):
        import requests
        obj = Image.open(BytesIO(requests.get(image, stream=True).content))
    elif image.startswith("file://"):
        obj = Image.open(image[7:])
    elif image.startswith("data:image"):
        if "base64," in image:
            obj = Image.open(BytesIO(base64.b64decode(image.split("base64,", 1)[1])))
    else:
        obj = Image.open(image)
    if obj is None:
        raise ValueError(f"Unrecognized image input, support local path, http url, base64 and PIL.Image, got {image}")
    image = to_rgb(obj)
    if "resized_height" in ele and "resized_width" in ele:
        rh, rw = smart_resize(ele["resized_height"], ele["resized_width"], factor=size_factor)
    else:
        w, h = image.size
        rh, rw = smart_resize(h, w, factor=size_factor, min_pixels=ele.get("min_pixels", MIN_PIXELS),
                              max_pixels=ele.get("max_pixels", MAX_PIXELS))
    return image.resize((rw, rh))


# ------------------------------------------------------------------------------------------------ video readers
def _read_decord(ele):
    import decord
    if "video_start" in ele or "video_end" in ele:
        raise NotImplementedError("not support start_pts and end_pts in decord for now.")
    vr = decord.VideoReader(ele["video"])
    total, vfps = len(vr), vr.get_avg_fps()
    n = smart_nframes(ele, total_frames=total, video_fps=vfps)
    idx = sample_indices(total, n).tolist()
    video = torch.tensor(vr.get_batch(idx).asnumpy()).permute(0, 3, 1, 2)
    return video, n / max(total, 1e-6) * vfps


def _read_torchvision(ele):
    from torchvision import io
    path = ele["video"]
    if path.startswith("file://"):
        path = path[7:]
    video, _, info = io.read_video(path, start_pts=ele.get("video_start", 0.0), end_pts=ele.get("video_end", None),
                                   pts_unit="sec", output_format="TCHW")
    total, vfps = video.size(0), info["video_fps"]
    n = smart_nframes(ele, total_frames=total, video_fps=vfps)
    return video[sample_indices(total, n)], n / max(total, 1e-6) * vfps


def _read_cv2(ele):
    import cv2
    path = ele["video"][7:] if ele["video"].startswith("file://") else ele["video"]
    cap = cv2.VideoCapture(path)
    if not cap.isOpened():
        raise ValueError(f"Could not open video: {path}")
    total, vfps = int(cap.get(cv2.CAP_PROP_FRAME_COUNT)), cap.get(cv2.CAP_PROP_FPS)
    n = smart_nframes(ele, total_frames=total, video_fps=vfps)
    frames = []
    for i in sample_indices(total, n).tolist():
        cap.set(cv2.CAP_PROP_POS_FRAMES, i)
        ok, fr = cap.read()
        if not ok:
            raise ValueError(f"could not read frame {i} of {path}")
        frames.append(torch.from_numpy(cv2.cvtColor(fr, cv2.COLOR_BGR2RGB)).permute(2, 0, 1))
    cap.release()
    return torch.stack(frames), n / max(total, 1e-6) * vfps


VIDEO_READER_BACKENDS = {"decord": _read_decord, "torchvision": _read_torchvision, "cv2": _read_cv2}


@lru_cache(maxsize=1)
def get_video_reader_backend() -> str:
    if FORCE_QWENVL_VIDEO_READER is not None:
        return FORCE_QWENVL_VIDEO_READER
    for name in ("decord", "torchvision", "cv2"):
        if importlib.util.find_spec(name) is not None:
            return name
    raise RuntimeError("no video reader installed (decord / torchvision / cv2): pass pre-decoded frames "
                       "[T,3,H,W] uint8 in the 'video' field instead of a path")


def video_pixel_budget(nframes: int, ele: dict):
    """Per-frame (min_pixels, max_pixels): min(VIDEO_MAX, total/n*2) but at least 1.05*min, capped by `max_pixels`."""
    min_pixels = ele.get("min_pixels", VIDEO_MIN_PIXELS)
    total_pixels = ele.get("total_pixels", VIDEO_TOTAL_PIXELS)
    cap = max(min(VIDEO_MAX_PIXELS, total_pixels / nframes * FRAME_FACTOR), int(min_pixels * 1.05))
    want = ele.get("max_pixels", cap)
    if want > cap:
        logger.warning(f"The given max_pixels[{want}] exceeds limit[{cap}].")
    return min_pixels, min(want, cap)


@functools.lru_cache(maxsize=64)
def aa_tables(in_size: int, out_size: int):
    """Tap tables of ATen's antialiased bicubic resize along one axis (aten/native/cpu/UpSampleKernel.cpp
    `compute_indices_weights_aa` with the a = -0.5 cubic, align_corners=False), evaluated in float32 like ATen does for a
    float32 input: (first tap [out] i32, tap count [out] i32, normalised weights [out, kmax] f32).  Consumed by the HIP
    resize (o3v_resize_bicubic_aa); `F.interpolate(..., mode="bicubic", antialias=True)` is the CPU reference it is
    tested against."""
    f32 = np.float32
    scale = f32(in_size) / f32(out_size)
    support = f32(2.0) * scale if scale >= 1 else f32(2.0)
    invscale = f32(1.0) / scale if scale >= 1 else f32(1.0)
    kmax = int(np.ceil(support)) * 2 + 1
    i = np.arange(out_size, dtype=np.float32)
    center = (scale * (i + f32(0.5))).astype(np.float32)
    lo = np.maximum((center - support + f32(0.5)).astype(np.float32).astype(np.int64), 0)
    hi = np.minimum((center + support + f32(0.5)).astype(np.float32).astype(np.int64), in_size)
    n = (hi - lo).astype(np.int64)
    j = np.arange(kmax, dtype=np.int64)[None, :]
    x = (((j + lo[:, None]).astype(np.float32) - center[:, None] + f32(0.5)).astype(np.float32) * invscale).astype(np.float32)
    ax = np.abs(x)
    a = f32(-0.5)
    near = ((((a + f32(2)) * ax - (a + f32(3))).astype(np.float32) * ax).astype(np.float32) * ax + f32(1)).astype(np.float32)
    far = (((((ax - f32(5)) * ax + f32(8)).astype(np.float32) * ax).astype(np.float32) - f32(4)) * a).astype(np.float32)
    w = np.where(ax < 1, near, np.where(ax < 2, far, f32(0))).astype(np.float32)
    w = np.where(j < n[:, None], w, f32(0)).astype(np.float32)
    tot = np.zeros(out_size, dtype=np.float32)
    for k in range(kmax):                       # ATen sums the taps in order
        tot = (tot + w[:, k]).astype(np.float32)
    w = np.where(tot[:, None] != 0, (w / np.where(tot == 0, f32(1), tot)[:, None]).astype(np.float32), w)
    return lo.astype(np.int32), n.astype(np.int32), np.ascontiguousarray(w, dtype=np.float32)


def resize_frames_device(video: torch.Tensor, size) -> torch.Tensor:
    """`resize_frames` on the GPU (uint8 / float frames [T,3,H,W] on or off the device -> float32 [T,3,h,w] on the
    device): the frames never return to the host between decode and the ViT."""
    import ctypes as C
    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.O3VError("resize_frames_device needs the GPU (resize_frames is the CPU form)")
    is_u8 = video.dtype == torch.uint8
    src = video.cuda().contiguous() if is_u8 else video.cuda().float().contiguous()
    T, Cn, H, W = src.shape
    h, w = int(size[0]), int(size[1])
    xt, yt = aa_tables(W, w), aa_tables(H, h)
    dev = src.device
    tb = [torch.from_numpy(t).to(dev) for t in (*xt, *yt)]
    tmp = torch.empty((T * Cn, H, w), dtype=torch.float32, device=dev)
    out = torch.empty((T, Cn, h, w), dtype=torch.float32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.call("o3v_resize_bicubic_aa", P(src), int(is_u8), P(tmp), P(out), T * Cn, H, W, h, w, P(tb[0]), P(tb[1]), P(tb[2]),
              xt[2].shape[1], P(tb[3]), P(tb[4]), P(tb[5]), yt[2].shape[1], C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


def resize_frames(video: torch.Tensor, size) -> torch.Tensor:
    """[T,3,H,W] uint8/float -> float32 [T,3,h,w], bicubic + antialias; uint8 inputs are rounded and clamped to
    0..255 before the final `.float()` like torchvision's uint8 path."""
    is_u8 = video.dtype == torch.uint8
    out = F.interpolate(video.float(), size=list(size), mode="bicubic", antialias=True, align_corners=False)
    if is_u8:
        out = out.round().clamp_(0, 255)
    return out


def fetch_video(ele: dict, image_factor: int = IMAGE_FACTOR, return_video_sample_fps: bool = False):
    v = ele["video"]
    if isinstance(v, (list, tuple)):  # list of frame images
        info = {k: x for k, x in ele.items() if k not in ("type", "video")}
        images = [fetch_image({"image": f, **info}, size_factor=image_factor) for f in v]
        n = ceil_by_factor(len(images), FRAME_FACTOR)
        images.extend([images[-1]] * (n - len(images)))
        return (images, info.pop("fps", 2.0)) if return_video_sample_fps else images
    if isinstance(v, str):
        backend = get_video_reader_backend()
        try:
            video, sample_fps = VIDEO_READER_BACKENDS[backend](ele)
        except Exception as e:  # noqa: BLE001 - same fallback order as the reference
            if backend == "torchvision" or importlib.util.find_spec("torchvision") is None:
                raise
            logger.warning(f"video_reader_backend {backend} error, use torchvision as default, msg: {e}")
            video, sample_fps = _read_torchvision(ele)
    else:  # pre-decoded frames [T,3,H,W] (+ optional "video_fps" / "sample_fps")
        video = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
        if video.dim() != 4 or video.shape[1] != 3:
            raise ValueError("pre-decoded video must be [T,3,H,W]")
        if "nframes" in ele or "fps" in ele:
            total, vfps = video.shape[0], float(ele.get("video_fps", FPS))
            n = smart_nframes({k: ele[k] for k in ("nframes", "fps", "min_frames", "max_frames") if k in ele}, total, vfps)
            video, sample_fps = video[sample_indices(total, n)], n / max(total, 1e-6) * vfps
        else:
            sample_fps = float(ele.get("sample_fps", FPS))
    nframes, _, height, width = video.shape
    if "resized_height" in ele and "resized_width" in ele:
        rh, rw = smart_resize(ele["resized_height"], ele["resized_width"], factor=image_factor)
    else:
        min_pixels, max_pixels = video_pixel_budget(nframes, ele)
        rh, rw = smart_resize(height, width, factor=image_factor, min_pixels=min_pixels, max_pixels=max_pixels)
    video = resize_frames(video, (rh, rw))
    return (video, sample_fps) if return_video_sample_fps else video


def extract_vision_info(conversations):
    if isinstance(conversations[0], dict):
        conversations = [conversations]
    out = []
    for conv in conversations:
        for msg in conv:
            if isinstance(msg["content"], list):
                for ele in msg["content"]:
                    if "image" in ele or "image_url" in ele or "video" in ele or ele["type"] in ("image", "image_url", "video"):
                        out.append(ele)
    return out


def process_vision_info(conversations, return_video_kwargs: bool = False):
    images, videos, fps = [], [], []
    for info in extract_vision_info(conversations):
        if "image" in info or "image_url" in info:
            images.append(fetch_image(info))
        elif "video" in info:
            v, f = fetch_video(info, return_video_sample_fps=True)
            videos.append(v)
            fps.append(f)
        else:
            raise ValueError("image, image_url or video should in content.")
    images = images or None
    videos = videos or None
    if return_video_kwargs:
        return images, videos, {"fps": fps}
    return images, videos


# ------------------------------------------------------------------------------------------------ frame prompts
IMAGE_TAG = "<|vision_start|><|image_pad|><|vision_end|>"
VIDEO_TAG = "<|vision_start|><|video_pad|><|vision_end|>"


def frames_as_images_prompt(prompt: str, nframes: int, fps: float, style: str = "trainer", times=None) -> str:
    """Replace the single video placeholder by one `Frame i at t: <image>` line per frame -- the reference's
    video-as-N-images trick.  style: 'trainer' (R:grpo_trainer.py:477-487, adds the total-duration line),
    'demo' (R:eval/inference_example.py:69-72) or 'vstar' (R:eval/test/test_vstar_multi_images.py:173-206, explicit
    `times`; prepended when the prompt has no placeholder)."""
    fp = ""
    if style == "vstar":
        for i, t in enumerate(times):
            fp += f"Frame {i + 1} at {round(t, 1)}s: {IMAGE_TAG}\n"
        return prompt.replace(VIDEO_TAG, fp) if VIDEO_TAG in prompt else fp + prompt
    for i in range(nframes):
        t = round(i / fps, 1)
        fp += f"Frame {i + 1} at {t}s: {IMAGE_TAG}\n" if style == "trainer" else f"Frame {i + 1} at {t} second: {IMAGE_TAG}\n"
    if style == "trainer":
        fp += f"The video is in total {int(nframes / fps)} seconds.\n"
    return prompt.replace(VIDEO_TAG, fp)


def interleave_key_frames(video: torch.Tensor, fps: float, key_frames, prompt: Optional[str] = None):
    """The trainer's key-frame branch (R:grpo_trainer.py:495-538): the dataset's key frames (each `{"time": seconds,
    "image": PIL image | path | array}`), resized to the video frames' (W, H) with PIL's default filter like
    `kf.convert('RGB').resize(image_size)` (:505-507), are spliced between the sampled frames -- a key frame goes in as soon
    as the whole second of the next video frame has reached `round(time)` -- and every frame gets its
    `Frame i at ts:` line.  -> (frames [T+K',3,H,W] in the video's dtype, frame prompt or, when `prompt` is given, the
    prompt with the video placeholder replaced).  Key frames later than the last video frame are dropped, as in the
    reference."""
    from PIL import Image
    T, _, H, W = video.shape
    kfs = []
    for kf in key_frames:
        im = kf["image"] if "image" in kf else kf["path"]
        if isinstance(im, str):
            im = Image.open(im)
        if not hasattr(im, "convert"):
            arr = np.asarray(im)
            if arr.ndim == 3 and arr.shape[0] == 3 and arr.shape[-1] != 3:
                arr = arr.transpose(1, 2, 0)
            im = Image.fromarray(arr.astype(np.uint8))
        arr = np.array(im.convert("RGB").resize((W, H)))
        kfs.append((round(kf["time"]), torch.from_numpy(np.transpose(arr, (2, 0, 1))).to(video.dtype)))
    fp, frames = "", []
    kf_idx = ori_idx = 0
    frame_idx = 1
    while ori_idx < T:
        time_now = int(ori_idx / fps)
        if kf_idx < len(kfs) and time_now >= kfs[kf_idx][0]:
            frames.append(kfs[kf_idx][1])
            time_now = round(kfs[kf_idx][0], 1)
            kf_idx += 1
        else:
            frames.append(video[ori_idx])
            time_now = round(ori_idx / fps, 1)
            ori_idx += 1
        fp += f"Frame {frame_idx} at {time_now}s: {IMAGE_TAG}\n"
        frame_idx += 1
    fp += f"The video is in total {int(T / fps)} seconds.\n"
    out = torch.stack(frames)
    return out, (fp if prompt is None else prompt.replace(VIDEO_TAG, fp))
