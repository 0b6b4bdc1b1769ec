"""vLLM-style facade (facade 2): `LLM(...).generate(inputs, sampling_params)` as
R:eval/inference_example.py:15-29,75-82 and R:eval/models/model_vllm.py:18-33,103,117,125 call it.

    llm = LLM(model=path, tensor_parallel_size=1, max_model_len=81920, gpu_memory_utilization=0.7,
              limit_mm_per_prompt={"image": 32})
    out = llm.generate([{"prompt": text, "multi_modal_data": {"image": frames}}], sampling_params=SamplingParams(...))
    out[0].outputs[0].text

The engine tokenises `prompt` itself, expands each <|image_pad|> to gh*gw/4 placeholders
(TF:models/qwen2_5_vl/processing_qwen2_5_vl.py:59-62) and runs the fused GPU frame pipeline on the images
(list of PIL images, a [T,3,H,W] tensor / ndarray as process_vision_info returns it, or a single image).
`multi_modal_data["video"]` (what QwenVL_VLLM.__call__ sends, R:eval/models/model_vllm.py:72-88,99-103: the [T,3,H,W] array of
process_vision_info under a <|vision_start|><|video_pad|><|vision_end|> prompt) is the native video input: temporal patches of two
frames, one <|video_pad|> expanded to T/2*gh*gw/4 placeholders, rope time advancing per temporal patch.
temperature == 0 -> greedy; otherwise temperature / top-p sampling with repetition penalty over prompt + output.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import numpy as np
import torch

from . import vision_process as vp
from .config import O3VConfig
from .engine import O3VEngine
from .weights import DeviceWeights, getter_from_safetensors_dir


@dataclass
class SamplingParams:
    temperature: float = 1.0
    top_p: float = 1.0
    repetition_penalty: float = 1.0
    max_tokens: int = 16
    stop_token_ids: Optional[Sequence[int]] = None
    n: int = 1
    seed: Optional[int] = None


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: List[int]
    finish_reason: str = "length"


@dataclass
class RequestOutput:
    request_id: str
    prompt: str
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput] = field(default_factory=list)


def _load_tokenizer(path):
    from transformers import AutoTokenizer  # local files only; never downloads
    return AutoTokenizer.from_pretrained(path, local_files_only=True)


class LLM:
    IMAGE_PAD = "<|image_pad|>"
    VIDEO_PAD = "<|video_pad|>"
    VISION_START, VISION_END = "<|vision_start|>", "<|vision_end|>"

    def __init__(self, model: str = None, tensor_parallel_size: int = 1, max_model_len: int = 81920,
                 gpu_memory_utilization: float = 0.9, limit_mm_per_prompt: Optional[dict] = None, dtype: str = "bfloat16",
                 max_num_seqs: int = 8, engine: Optional[O3VEngine] = None, tokenizer: Any = None,
                 min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280, device="cuda",
                 enable_prefix_caching: bool = True, quantization: Optional[str] = None, mm_processor_kwargs: Optional[dict] = None,
                 position_mode: str = "pinned", fp8_prefill: bool = False, **_):
        if tensor_parallel_size != 1:
            raise ValueError("the reference runs tensor_parallel_size=1 (R:eval/models/model_vllm.py:21); data parallelism "
                             "is one engine per GPU (open_o3_video_amd.dist)")
        if dtype not in ("bfloat16", "auto"):
            raise ValueError("bf16 only")
        if engine is None:
            if model is None or not os.path.isdir(model):
                raise OSError(f"{model} is not a local checkpoint directory (this build never downloads)")
            cfg = O3VConfig.from_pretrained(model)
            # vLLM's keyword: quantization="fp8" quantises the bf16 checkpoint's linears on load (here: fp8 e4m3fn decode rows with
            # one power-of-two scale per output row beside the bf16 matrices, which prefill keeps using; BASELINE config #5)
            if quantization not in (None, "fp8"):
                raise ValueError(f"quantization={quantization!r}: only None or 'fp8' (decode rows) is built")
            engine = O3VEngine(cfg, DeviceWeights(cfg, getter_from_safetensors_dir(model), device, fp8_decode=quantization == "fp8"))
            # vLLM's fp8 linears quantise the activations per token as well (W8A8); here that is opt-in for the compute-bound
            # prefill (fp8 x fp8 on the matrix cores), while the decode keeps bf16 activations on the exactly widened fp8 rows
            if fp8_prefill and quantization != "fp8":
                raise ValueError("fp8_prefill=True needs quantization='fp8'")
            engine.fp8_prefill = bool(fp8_prefill)
            tokenizer = tokenizer or _load_tokenizer(model)
            pp = os.path.join(model, "preprocessor_config.json")
            if os.path.exists(pp):
                import json
                with open(pp) as f:
                    pc = json.load(f)
                # either the flat keys or transformers 5.x's size = {shortest_edge: min pixels, longest_edge: max pixels}
                size = pc.get("size") or {}
                min_pixels = pc.get("min_pixels") or size.get("shortest_edge") or min_pixels
                max_pixels = pc.get("max_pixels") or size.get("longest_edge") or max_pixels
        if tokenizer is None:
            raise ValueError("a tokenizer is required (encode / decode / convert_tokens_to_ids)")
        self.generation_eos_ids = []
        if model is not None and os.path.isdir(model):
            from .hf_api import load_generation_config
            ge = load_generation_config(model).get("eos_token_id")
            self.generation_eos_ids = [] if ge is None else ([int(e) for e in ge] if isinstance(ge, (list, tuple)) else [int(ge)])
        self.engine, self.tokenizer = engine, tokenizer
        self.image_factor = engine.cfg.vision.patch_size * engine.cfg.vision.spatial_merge_size   # 28; Qwen3-VL: 32
        self.max_num_seqs = int(max_num_seqs)   # requests of one generate() call decoded together (<= O3VEngine.MAX_ROWS)
        self.cfg = engine.cfg
        self.max_model_len = max_model_len
        self.limit_mm = dict(limit_mm_per_prompt or {})
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self._req = 0
        # cross-prompt visual reuse: the V-STAR harness asks 5 questions per video and the
        # test-time-scaling loop N samples per video (R:eval/test/test_vstar_multi_images.py:511-544), each of which
        # re-encodes identical frames in the reference.  Merged visual tokens are cached per frame-tensor content.
        self._vis_cache = {}
        self.vis_cache_size = 4
        self.vis_cache_hits = 0
        # prompt-prefix K/V reuse per video (same section): only the tokens after the longest common token prefix of two
        # prompts over identical frames are prefilled; vLLM's `enable_prefix_caching` keyword switches it
        self.enable_prefix_caching = bool(enable_prefix_caching)
        self.prefix_tokens_reused = 0
        # native video inputs: HF's Qwen2_5_VLProcessor gives every video second_per_grid_ts = temporal_patch_size / fps with
        # fps = 2.0 unless mm_processor_kwargs={"fps": ...} says otherwise (TF:models/qwen2_5_vl/processing_qwen2_5_vl.py:143-153;
        # the reference passes none, R:eval/models/model_vllm.py:18-26)
        self.mm_processor_kwargs = dict(mm_processor_kwargs or {})
        # video rope arithmetic: "pinned" = vllm 0.7.2's MRotaryEmbedding.get_input_positions (R:setup.sh:7; restated, parity
        # unpinned), "tf5" = transformers 5.15 (goldens G5b / G14 / G15); see indexing.rope_index
        self.position_mode = position_mode

    # ---- multimodal input -> uint8/f32 frames [T,3,H,W] with H,W multiples of 28
    def _frames(self, mm) -> Optional[torch.Tensor]:
        if not mm:
            return None
        data = mm.get("image", None)
        if data is None:
            return None
        if isinstance(data, (list, tuple)):
            lim = self.limit_mm.get("image")
            if lim is not None and len(data) > lim:
                raise ValueError(f"{len(data)} images exceed limit_mm_per_prompt['image']={lim}")
            frames = []
            for im in data:
                arr = np.asarray(im.convert("RGB") if hasattr(im, "convert") else im)
                if arr.ndim == 3 and arr.shape[-1] == 3:
                    arr = arr.transpose(2, 0, 1)
                frames.append(torch.from_numpy(np.ascontiguousarray(arr)))
            sizes = {tuple(f.shape[1:]) for f in frames}
            if len(sizes) != 1:
                raise NotImplementedError("images of different sizes in one request")
            data = torch.stack(frames)
        else:
            data = torch.as_tensor(np.asarray(data)) if not torch.is_tensor(data) else data
            if data.dim() == 3:
                data = data[None]
        if data.shape[1] != 3 and data.shape[-1] == 3:
            data = data.permute(0, 3, 1, 2)
        T, _, H, W = data.shape
        # the HF image processor resizes every image with smart_resize(factor = patch x merge: 28, Qwen3-VL 32; min/max pixels) -- a no-op for frames
        # that process_vision_info already sized
        rh, rw = vp.smart_resize(H, W, self.image_factor, self.min_pixels, self.max_pixels)
        if (rh, rw) != (H, W):
            data = vp.resize_frames_device(data, (rh, rw))   # antialiased bicubic on the GPU; frames stay on the device
        return data

    def _videos(self, mm):
        """multi_modal_data["video"] -> list of (frames [T,3,H,W] sized by smart_resize, metadata dict).  One array / tensor is one
        video (R:eval/models/model_vllm.py:72-88 sends `v_input.numpy()`, f32 0..255 from process_vision_info); a list holds
        several; an (array, metadata) pair carries {"fps", "frames_indices"} for Qwen3-VL's timestamps."""
        data = None if not mm else mm.get("video", None)
        if data is None:
            return []
        if isinstance(data, tuple) and len(data) == 2 and isinstance(data[1], dict):
            data = [data]
        elif not isinstance(data, list):
            data = [data]
        lim = self.limit_mm.get("video")
        if lim is not None and len(data) > lim:
            raise ValueError(f"{len(data)} videos exceed limit_mm_per_prompt['video']={lim}")
        out = []
        for item in data:
            meta = {}
            if isinstance(item, tuple) and len(item) == 2 and isinstance(item[1], dict):
                item, meta = item
            v = item if torch.is_tensor(item) else torch.as_tensor(np.asarray(item))
            if v.dim() != 4:
                raise ValueError("a video is a [T,3,H,W] (or [T,H,W,3]) array of frames")
            if v.shape[1] != 3 and v.shape[-1] == 3:
                v = v.permute(0, 3, 1, 2)
            T, _, H, W = v.shape
            rh, rw = vp.smart_resize(H, W, self.image_factor, self.min_pixels, self.max_pixels)
            if (rh, rw) != (H, W):
                v = vp.resize_frames_device(v, (rh, rw))
            out.append((v, dict(meta)))
        return out

    def _cached_visual(self, fr: torch.Tensor, video: bool):
        """ViT + merger output for image frames / one native video, cached by content: (kind, shape, dtype, 128-bit hash of the
        bytes computed on the device by o3v_content_hash128 -- one pass over the frames, one 16-byte read-back)."""
        import ctypes as C
        from . import _lib
        fr = fr.to(self.engine.dev).contiguous()
        h = torch.zeros(2, dtype=torch.int64, device=fr.device)
        _lib.call("o3v_content_hash128", C.c_void_p(fr.data_ptr()), fr.numel() * fr.element_size(), C.c_void_p(h.data_ptr()),
                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        key = ("video" if video else "image", tuple(fr.shape), str(fr.dtype)) + tuple(int(v) for v in h.tolist())
        hit = self._vis_cache.get(key)
        if hit is not None:
            self.vis_cache_hits += 1
            return hit[0], hit[1], key
        px, grid = self.engine.pixels_from_video(fr) if video else self.engine.pixels_from_frames(fr)
        vis = self.engine.vit_forward(px, grid)
        if len(self._vis_cache) >= self.vis_cache_size:
            self._vis_cache.pop(next(iter(self._vis_cache)))
        self._vis_cache[key] = (vis, grid)
        return vis, grid, key

    def _visual_tokens(self, frames: torch.Tensor):
        vis, _, key = self._cached_visual(frames, video=False)
        return vis, key

    def _video_placeholder(self, grid_row, meta) -> str:
        """What one <|video_pad|> of the prompt becomes.  Qwen2.5-VL (TF:models/qwen2_5_vl/processing_qwen2_5_vl.py:64-67): t*gh*gw/4
        pads.  Qwen3-VL (TF:models/qwen3_vl/processing_qwen3_vl.py:81-107,178-189): per temporal patch "<{t:.1f} seconds>" +
        <|vision_start|> gh*gw/4 pads <|vision_end|>, the time being the mean of the patch's first / last frame index over the
        video fps (fps 24 and indices 0..T-1 when the request carries no metadata, as HF falls back)."""
        t, gh, gw = (int(v) for v in grid_row)
        per = gh * gw // self.cfg.vision.merge_unit
        if self.cfg.arch != "qwen3_vl":
            return self.VIDEO_PAD * (t * per)
        fps = float(meta.get("fps") or 24)
        idx = list(meta.get("frames_indices") if meta.get("frames_indices") is not None else range(int(meta.get("n_frames", 2 * t))))
        idx = [float(i) for i in idx]
        if len(idx) % 2:
            idx.append(idx[-1])
        ts = [i / fps for i in idx]
        ts = [(ts[i] + ts[i + 1]) / 2 for i in range(0, len(ts), 2)]
        return "".join(f"<{ts[k]:.1f} seconds>" + self.VISION_START + self.VIDEO_PAD * per + self.VISION_END for k in range(t))

    def _tokenize(self, prompt: str, n_frames: int, tok_per_frame: int, video_placeholders: Sequence[str] = ()):
        if n_frames:
            n_tags = prompt.count(self.IMAGE_PAD)
            if n_tags != n_frames:
                raise ValueError(f"prompt has {n_tags} image placeholders but {n_frames} images were given")
            prompt = prompt.replace(self.IMAGE_PAD, self.IMAGE_PAD * tok_per_frame)
        n_vtags = prompt.count(self.VIDEO_PAD)
        if n_vtags != len(video_placeholders):
            raise ValueError(f"prompt has {n_vtags} video placeholders but {len(video_placeholders)} videos were given")
        if n_vtags:
            q3 = self.cfg.arch == "qwen3_vl"
            triple = self.VISION_START + self.VIDEO_PAD + self.VISION_END
            parts, rest = [], prompt
            for ph in video_placeholders:
                i = rest.index(self.VIDEO_PAD)
                # Qwen3-VL's expansion carries its own <|vision_start|> / <|vision_end|> per temporal patch: it replaces the whole
                # <|vision_start|><|video_pad|><|vision_end|> of the chat template when the pad stands inside one
                if q3 and rest[max(0, i - len(self.VISION_START)):i + len(self.VIDEO_PAD) + len(self.VISION_END)] == triple:
                    parts.append(rest[:i - len(self.VISION_START)] + ph)
                    rest = rest[i + len(self.VIDEO_PAD) + len(self.VISION_END):]
                else:
                    parts.append(rest[:i] + ph)
                    rest = rest[i + len(self.VIDEO_PAD):]
            prompt = "".join(parts) + rest
        ids = self.tokenizer.encode(prompt, add_special_tokens=False) if hasattr(self.tokenizer, "encode") else self.tokenizer(prompt)
        return list(ids)

    def generate(self, inputs, sampling_params: Optional[SamplingParams] = None, use_tqdm: bool = False):
        sp = sampling_params or SamplingParams()
        if isinstance(inputs, dict):
            inputs = [inputs]
        greedy = sp.temperature == 0.0
        # vLLM stops on the request's stop ids, the model's eos id and every eos id of the checkpoint's generation_config.json
        stop = list(sp.stop_token_ids) if sp.stop_token_ids else []
        eos = stop + [e for e in ([self.cfg.eos_token_id] + list(self.generation_eos_ids)) if e is not None and e not in stop]
        eos = list(dict.fromkeys(int(e) for e in eos))
        common = dict(max_new_tokens=sp.max_tokens, eos_token_ids=eos, pad_token_id=self.cfg.pad_token_id,
                      repetition_penalty=sp.repetition_penalty, do_sample=not greedy,
                      temperature=1.0 if greedy else sp.temperature, top_p=1.0 if greedy else sp.top_p, return_margins=False)
        # ---- per request: frames -> (cached) visual tokens, prompt -> ids
        prepared = []
        tps = self.cfg.vision.temporal_patch_size
        fps = float(self.mm_processor_kwargs.get("fps", 2.0))
        self.engine.position_mode = self.position_mode
        for req in inputs:
            prompt = req["prompt"] if isinstance(req, dict) else str(req)
            mm = req.get("multi_modal_data") if isinstance(req, dict) else None
            frames = self._frames(mm)
            videos = self._videos(mm)
            tpf = 0 if frames is None else (frames.shape[2] // self.image_factor) * (frames.shape[3] // self.image_factor)
            vis_i, grid, keys = None, None, []
            if frames is not None:
                vis_i, k = self._visual_tokens(frames)
                keys.append(k)
                ps = self.cfg.vision.patch_size
                grid = np.asarray([[1, frames.shape[2] // ps, frames.shape[3] // ps]] * frames.shape[0], dtype=np.int64)
            vis_v, vgrids, placeholders = [], [], []
            for v, meta in videos:
                vv, vg, k = self._cached_visual(v, video=True)
                vis_v.append(vv)
                vgrids.append(vg)
                placeholders.append(self._video_placeholder(vg[0], dict(meta, n_frames=v.shape[0])))
                keys.append(k)
            ids = self._tokenize(prompt, 0 if frames is None else frames.shape[0], tpf, placeholders)
            if len(ids) + sp.max_tokens > self.max_model_len:
                raise ValueError(f"prompt ({len(ids)}) + max_tokens ({sp.max_tokens}) exceeds max_model_len {self.max_model_len}")
            vkey = tuple(keys) if keys else ("text-only",)      # identifies the visual content of the prompt (prefix-K/V reuse)
            vgrid = np.concatenate(vgrids) if vgrids else None
            spg = [tps / fps] * len(vgrids) if vgrids else None
            prepared.append((prompt, ids, vis_i, grid, vkey, vis_v, vgrid, spg))

        def cat_visual(img_parts, vid_parts):
            """[image tokens of every row ; video tokens of every row] -- the order engine.embed scatters them in."""
            parts = [p for p in img_parts if p is not None] + [p for p in vid_parts if p is not None]
            if not parts:
                return None
            return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-2)

        def finish(ro, index, row, n_prompt):
            toks = row[n_prompt:].tolist()
            reason = "length"
            for j, t in enumerate(toks):
                if t in eos:
                    toks, reason = toks[:j], "stop"   # vLLM drops the stop token from the text
                    break
            ro.outputs.append(CompletionOutput(index, self.tokenizer.decode(toks, skip_special_tokens=True), toks, reason))

        results = []
        if len(prepared) > 1 and sp.n == 1 and self.max_num_seqs > 1:
            # several requests in one call (vLLM batches them, R:eval/models/model_vllm.py:23 max_num_seqs): decode up to
            # max_num_seqs of them together, left padded, every streamed weight byte shared by the rows of the step
            rows_per_call = min(self.max_num_seqs, O3VEngine.MAX_ROWS)
            for c0 in range(0, len(prepared), rows_per_call):
                chunk = prepared[c0:c0 + rows_per_call]
                L = max(len(p[1]) for p in chunk)
                pad = self.cfg.pad_token_id
                rows = [[pad] * (L - len(p[1])) + p[1] for p in chunk]
                mask = [[0] * (L - len(p[1])) + [1] * len(p[1]) for p in chunk]
                grid_parts = [p[3] for p in chunk if p[3] is not None]
                vgrid_parts = [p[6] for p in chunk if p[6] is not None]
                spg_all = [x for p in chunk if p[7] for x in p[7]]
                out = self.engine.generate(rows, mask, vis_embeds=cat_visual([p[2] for p in chunk], [v for p in chunk for v in p[5]]),
                                           image_grid_thw=np.concatenate(grid_parts) if grid_parts else None,
                                           video_grid_thw=np.concatenate(vgrid_parts) if vgrid_parts else None,
                                           second_per_grid_ts=spg_all or None,
                                           row_ids=[self._req + i for i in range(len(chunk))],
                                           seed=0 if sp.seed is None else sp.seed, **common)
                for i, p in enumerate(chunk):
                    ro = RequestOutput(request_id=str(self._req), prompt=p[0], prompt_token_ids=p[1])
                    finish(ro, 0, out.sequences[i], L)
                    results.append(ro)
                    self._req += 1
            return results
        for prompt, ids, vis_i, grid, vkey, vis_v, vgrid, spg in prepared:
            vis = cat_visual([vis_i], vis_v)
            ro = RequestOutput(request_id=str(self._req), prompt=prompt, prompt_token_ids=ids)
            # n samples of one prompt (self-consistency, R:eval/tts.py:47-123) run in groups of <= 16 decode rows; sample i is
            # keyed by (seed, i) whatever group it lands in, and groups after the first reuse the whole prompt K/V
            for i0 in range(0, sp.n, O3VEngine.MAX_ROWS):
                g = min(O3VEngine.MAX_ROWS, sp.n - i0)
                out = self.engine.generate([ids], None, vis_embeds=vis, image_grid_thw=grid, video_grid_thw=vgrid,
                                           second_per_grid_ts=spg, num_return_sequences=g,
                                           row_ids=list(range(i0, i0 + g)), seed=self._req if sp.seed is None else sp.seed,
                                           prefix_key=vkey if self.enable_prefix_caching else None, **common)
                self.prefix_tokens_reused += int(out.timings.get("prefix_tokens_reused", 0))
                for i in range(g):
                    finish(ro, i0 + i, out.sequences[i], len(ids))
            results.append(ro)
            self._req += 1
        return results
