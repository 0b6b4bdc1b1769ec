"""Oracle: frame-count / resize policy and frame-prompt text (integer + string work).

Follows R:src/r1-v/src/open_r1/vision_process.py:25-87,145-182,185-219,279-333 and the
frame-prompt builders R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:477-489,
R:eval/inference_example.py:69-72, R:eval/test/test_vstar_multi_images.py:173-183.
Test infrastructure only (see oracle/__init__.py).
"""
from __future__ import annotations

import math

import numpy as np

# R:vision_process.py:25-42
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
VIDEO_TOTAL_PIXELS = int(float(128000 * 28 * 28 * 0.9))


def round_by_factor(number, factor):
    # R:vision_process.py:46-48 -- python round() = round-half-even on the float quotient
    return round(number / factor) * factor


def ceil_by_factor(number, factor):
    # R:vision_process.py:51-53
    return math.ceil(number / factor) * factor


def floor_by_factor(number, factor):
    # R:vision_process.py:56-58
    return math.floor(number / factor) * factor


def smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=MIN_PIXELS, max_pixels=MAX_PIXELS):
    # R:vision_process.py:61-87
    if max(height, width) / min(height, width) > MAX_RATIO:
        raise ValueError("absolute aspect ratio must be smaller than %d" % MAX_RATIO)
    h_bar = max(factor, round_by_factor(height, factor))
    w_bar = max(factor, round_by_factor(width, factor))
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = floor_by_factor(height / beta, factor)
        w_bar = floor_by_factor(width / beta, factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = ceil_by_factor(height * beta, factor)
        w_bar = ceil_by_factor(width * beta, factor)
    return h_bar, w_bar


def smart_nframes(ele, total_frames, video_fps):
    # R:vision_process.py:145-182
    assert not ("fps" in ele and "nframes" in ele)
    if "nframes" in ele:
        nframes = round_by_factor(ele["nframes"], FRAME_FACTOR)
    else:
        fps = ele.get("fps", FPS)
        min_frames = ceil_by_factor(ele.get("min_frames", FPS_MIN_FRAMES), FRAME_FACTOR)
        max_frames = floor_by_factor(ele.get("max_frames", min(FPS_MAX_FRAMES, total_frames)), FRAME_FACTOR)
        nframes = total_frames / video_fps * fps
        nframes = min(min(max(nframes, min_frames), max_frames), total_frames)
        nframes = floor_by_factor(nframes, FRAME_FACTOR)
    if not (FRAME_FACTOR <= nframes and nframes <= total_frames):
        raise ValueError("nframes should in interval [%d, %d], but got %s" % (FRAME_FACTOR, total_frames, nframes))
    return nframes


def sample_frame_indices(total_frames, nframes):
    """R:vision_process.py:216 / :251 -- torch.linspace(0,total-1,n).round().long().

    torch.linspace(float32) computes, for i < n/2, start + i*step and for i >= n/2,
    end - (n-1-i)*step with step=(end-start)/(n-1) in float32; round() is half-to-even.
    """
    n = int(nframes)
    if n == 1:
        return [0]
    start = np.float32(0.0)
    end = np.float32(total_frames - 1)
    step = np.float32((end - start) / np.float32(n - 1))
    out = []
    half = n // 2
    for i in range(n):
        if i < half:
            v = np.float32(start + np.float32(step * np.float32(i)))
        else:
            v = np.float32(end - np.float32(step * np.float32(n - 1 - i)))
        out.append(int(np.rint(v)))  # rint = half-to-even like torch.round
    return out


def sample_fps(total_frames, nframes, video_fps):
    # R:vision_process.py:217 / :254
    return nframes / max(total_frames, 1e-6) * video_fps


def video_max_pixels(nframes, ele=None):
    """Per-frame pixel cap, R:vision_process.py:288-295."""
    ele = ele or {}
    min_pixels = ele.get("min_pixels", VIDEO_MIN_PIXELS)
    total_pixels = ele.get("total_pixels", VIDEO_TOTAL_PIXELS)
    max_pixels = max(min(VIDEO_MAX_PIXELS, total_pixels / nframes * FRAME_FACTOR), int(min_pixels * 1.05))
    max_pixels_supposed = ele.get("max_pixels", max_pixels)
    max_pixels = min(max_pixels_supposed, max_pixels)
    return min_pixels, max_pixels


def video_resize_hw(nframes, height, width, ele=None, image_factor=IMAGE_FACTOR):
    """Target (H, W) of fetch_video, R:vision_process.py:286-309."""
    ele = ele or {}
    if "resized_height" in ele and "resized_width" in ele:
        return smart_resize(ele["resized_height"], ele["resized_width"], factor=image_factor)
    min_pixels, max_pixels = video_max_pixels(nframes, ele)
    return smart_resize(height, width, factor=image_factor, min_pixels=min_pixels, max_pixels=max_pixels)


# ---------------------------------------------------------------- frame prompts (strings)
IMG = "<|vision_start|><|image_pad|><|vision_end|>"
VID = "<|vision_start|><|video_pad|><|vision_end|>"


def frame_prompt_trainer(nframes, fps):
    # R:grpo_trainer.py:477-485 (plain branch)
    s = ""
    for i in range(nframes):
        s += f"Frame {i + 1} at {round(i / fps, 1)}s: {IMG}\n"
    s += f"The video is in total {int(nframes / fps)} seconds.\n"
    return s


def frame_prompt_demo(nframes, fps):
    # R:eval/inference_example.py:69-71
    s = ""
    for i in range(nframes):
        s += f"Frame {i+1} at {round(i / fps,1)} second: {IMG}\n"
    return s


def frame_prompt_vstar(timestamps):
    # R:eval/test/test_vstar_multi_images.py:173-183 -- timestamps in seconds per extracted frame
    s = ""
    for i, t in enumerate(timestamps):
        s += f"Frame {i + 1} at {round(t, 1)}s: {IMG}\n"
    return s


def replace_video_pad(prompt, frame_prompt):
    # R:grpo_trainer.py:487 ; R:eval/inference_example.py:72
    return prompt.replace(VID, frame_prompt)


def frame_prompt_trainer_keyframes(n_video_frames, fps, key_frame_times):
    """R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:513-533: sampled video frames with the dataset's key frames spliced
    in.  `key_frame_times` are the already rounded key-frame times (`round(key_frame["time"])`, :509) in dataset order.
    Returns (frame prompt, order) where order lists ("kf", k) / ("video", i) per emitted frame.  A key frame is emitted
    as soon as the integer second of the next video frame has reached its time; key frames still pending when the video
    frames run out are dropped."""
    prompt, order = "", []
    kf_idx = ori_idx = 0
    frame_idx = 1
    while ori_idx < n_video_frames:
        time_now = int(ori_idx / fps)
        if kf_idx < len(key_frame_times) and time_now >= key_frame_times[kf_idx]:
            order.append(("kf", kf_idx))
            time_now = round(key_frame_times[kf_idx], 1)
            kf_idx += 1
        else:
            order.append(("video", ori_idx))
            time_now = round(ori_idx / fps, 1)
            ori_idx += 1
        prompt += f"Frame {frame_idx} at {time_now}s: <|vision_start|><|image_pad|><|vision_end|>\n"
        frame_idx += 1
    prompt += f"The video is in total {int(n_video_frames / fps)} seconds.\n"
    return prompt, order
