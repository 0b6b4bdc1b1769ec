"""O3VEngine: the MI355X generate engine (ViT -> merger -> prefill -> decode) over libo3v_hip.so.

Python only plans (index tables, buffers as torch-ROCm tensors, one ctypes call per stage); every FLOP runs
in the HIP library on the current torch stream.  There is no CPU / eager fallback: without a GPU and the
built library this raises.

Reference behaviour restated: GenerationMixin.generate/_sample on Qwen2_5_VLForConditionalGeneration as
the Open-o3-Video trainer and eval drive it (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:581-586,
R:eval/models/model_vllm.py:103-126); TF = transformers 5.15.0 models/qwen2_5_vl/modeling_qwen2_5_vl.py.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, indexing
from .config import O3VConfig
from .weights import DeviceWeights

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # OPENAI_CLIP_MEAN/STD, TF:models/qwen2_vl/image_processing_qwen2_vl.py:94-101
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class GenerateOutput:
    sequences: torch.Tensor            # i64 [B, S+T]
    margins: Optional[torch.Tensor]    # f32 [B, T]: greedy top1-top2 margin / sampled token log-prob
    n_steps: int
    timings: dict


class O3VEngine:
    MAX_ROWS = 32   # decode rows per call: the skinny MFMA GEMM multiplies 16 weight rows against one or two 16-row blocks of x

    def __init__(self, cfg: O3VConfig, weights: DeviceWeights):
        if not torch.cuda.is_available():
            raise _lib.O3VError("O3VEngine needs a ROCm GPU (no CPU path)")
        _lib.load()
        self.cfg, self.w = cfg, weights
        self.dev = weights.device
        tc = cfg.text
        D = tc.head_dim
        # inv_freq exactly as torch computes it on the host (TF:521)
        inv = 1.0 / (tc.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float) / D))
        self.inv_freq = inv.to(self.dev)
        self.q3 = cfg.arch == "qwen3_vl"
        axis = indexing.mrope_axis_table_interleaved(tc.mrope_section, D // 2) if tc.mrope_interleaved \
            else indexing.mrope_axis_table(tc.mrope_section)
        self.axis_of = torch.from_numpy(axis).to(self.dev)
        # host arrays (read by the launcher, passed by value); Qwen3-VL's processor normalises with mean = std = 0.5
        self.clip_mean = (C.c_float * 3)(*((0.5, 0.5, 0.5) if self.q3 else CLIP_MEAN))
        self.clip_std = (C.c_float * 3)(*((0.5, 0.5, 0.5) if self.q3 else CLIP_STD))
        self._vit_plan_cache = {}
        self._prefix = {}   # prefix_key -> {"ids", "k", "v"}: prompt K/V kept for reuse (see generate(prefix_key=...))
        # The prompt of the last group generate (num_return_sequences > 1): its K/V, kept once anyway, and the hidden row that predicts
        # completion token 0.  The policy's log-prob pass over those completions (completion_logps, R:grpo_trainer.py:612-613) finds the
        # SAME prompt -- ids, mask, grids, positions and visual tensors equal bit for bit (_prompt_key) -- and skips the tower and the
        # prompt prefill; anything else (another model, a video whose second_per_grid_ts the trainer deleted) recomputes.
        self.reuse_prompt_kv = os.environ.get("O3V_REUSE_PROMPT_KV", "1") != "0"
        self._last_prompt = None
        self.timings_last_logps = {}
        self.group_attention = os.environ.get("O3V_GROUP_ATTENTION", "1") != "0"   # A/B switch for the shared-prefix kernel
        self.fused_decode = os.environ.get("O3V_FUSED_DECODE", "1") != "0"         # A/B switch for the one-launch attention block
        self.group_attention_mode = os.environ.get("O3V_GROUP_MODE", "auto")
        # the persistent layer block (o3v_decode_layer_block: attention half + gate/up as one launch of resident, row-pipelined
        # workgroups) is bit-identical but measured SLOWER than the role block + gate/up launch (95.3 vs 90.3 us per layer at 7B,
        # profiles/r03_layer_block_timeline_v2.txt): off unless O3V_LAYER_BLOCK=1
        self.layer_block = os.environ.get("O3V_LAYER_BLOCK", "0") == "1"
        # 8..32 decode rows: o_proj / down_proj can also normalise their result for the linear that follows (o3v_linear_decode_norm_next,
        # bit-identical to the separate o3v_rmsnorm launch it replaces).  Measured equal, not faster (G=8 3.765 vs 3.760 ms/step, G=16
        # 4.324 vs 4.321: the in-launch chain store-ack -> ticket -> poll -> row load costs the residual linear +6.4 us, the launch it
        # removes 5.9 us; profiles/r03_tail_norm_ab.txt): off unless O3V_TAIL_NORM=1
        self.tail_norm = os.environ.get("O3V_TAIL_NORM", "0") == "1"
        # video rope arithmetic (indexing.rope_index): "tf5" = transformers 5.15 (goldens G5b / G14 / G15), "pinned" = the
        # libraries the reference installs (transformers @336dc69d, vllm 0.7.2).  The facades choose; images do not depend on it.
        self.position_mode = "tf5"
        # opt-in (BASELINE config #5): the LLM prefill / log-prob linears as fp8 x fp8 on the matrix cores (W8A8: per-token activation
        # scales, the fp8 rows + per-row scales that fp8_decode builds).  Off by default: the bf16 path and its goldens are untouched.
        self.fp8_prefill = False

    # ------------------------------------------------------------------------------------------ vision
    def _vit_plan(self, grid_thw):
        key = indexing._grid_key(grid_thw)
        p = self._vit_plan_cache.get(key)
        if p is not None:
            return p
        vc = self.cfg.vision
        if self.q3:
            return self._vit_plan_q3(key)
        widx, cu_win, cu_full, pos = indexing.vision_plan(key, vc.spatial_merge_size, vc.window_size, vc.patch_size)
        unit = vc.merge_unit
        P = int(cu_full[-1])
        # rotary table in window order (TF:125-134, :441-446): fp32, computed as torch does on the host
        rdim = vc.head_dim // 2
        inv = 1.0 / (10000.0 ** (torch.arange(0, rdim, 2, dtype=torch.float) / rdim))
        rot = (torch.from_numpy(pos).unsqueeze(-1) * inv).flatten(1)            # [P, hd/2]
        wi = torch.from_numpy(widx)
        rot = rot.reshape(P // unit, unit, -1)[wi].reshape(P, -1)
        plan = dict(
            P=P,
            win_idx=wi.to(torch.int32).to(self.dev),
            rev_idx=torch.argsort(wi).to(torch.int32).to(self.dev),
            cos=rot.cos().contiguous().to(self.dev), sin=rot.sin().contiguous().to(self.dev),
            tiles_win=torch.from_numpy(indexing.segment_tiles(cu_win)).to(self.dev),
            tiles_full=torch.from_numpy(indexing.segment_tiles(cu_full)).to(self.dev),
        )
        if len(self._vit_plan_cache) > 16:
            self._vit_plan_cache.clear()
        self._vit_plan_cache[key] = plan
        return plan

    def _vit_plan_q3(self, key):
        """Qwen3-VL (TF3:606-737): no windows -- patches stay in merge-block order, one attention segment per temporal patch;
        the rotary table is stored head_dim_pad/2 wide (the padded frequencies rotate zeros); the learned position table is
        resampled to this grid once (TF3:643-710: four taps, table dtype x fp32 weights, summed in fp32 in tap order)."""
        vc = self.cfg.vision
        _, _, cu_full, pos = indexing.vision_plan(key, vc.spatial_merge_size, vc.patch_size * vc.spatial_merge_size, vc.patch_size)
        P = int(cu_full[-1])
        rdim = vc.head_dim // 2
        inv = 1.0 / (10000.0 ** (torch.arange(0, rdim, 2, dtype=torch.float) / rdim))
        rot = (torch.from_numpy(pos).unsqueeze(-1) * inv).flatten(1)            # [P, hd/2]
        halfp = vc.head_dim_pad // 2
        cos, sin = torch.ones((P, halfp)), torch.zeros((P, halfp))
        cos[:, :rot.shape[1]], sin[:, :rot.shape[1]] = rot.cos(), rot.sin()
        side = int(round(vc.num_position_embeddings ** 0.5))
        idx, wt = indexing.pos_embed_taps(key, side, vc.spatial_merge_size)
        table = self.w.t["v.pos_embed"]
        idx, wt = torch.from_numpy(idx).to(self.dev), torch.from_numpy(wt).to(self.dev)
        taps = [table[idx[:, k]] * wt[:, k, None] for k in range(4)]           # bf16 x fp32 -> fp32
        pe = (((taps[0] + taps[1]) + taps[2]) + taps[3]).to(torch.bfloat16).contiguous()
        plan = dict(P=P, cos=cos.contiguous().to(self.dev), sin=sin.contiguous().to(self.dev), pos_embed=pe,
                    tiles_full=torch.from_numpy(indexing.segment_tiles(cu_full)).to(self.dev))
        if len(self._vit_plan_cache) > 16:
            self._vit_plan_cache.clear()
        self._vit_plan_cache[key] = plan
        return plan

    def pixels_from_processor(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """HF-processor pixel_values f32 [P,1176] -> bf16 [P,Kp] (TF:1090 cast, zero pad)."""
        vc = self.cfg.vision
        pv = pixel_values.to(self.dev, torch.float32).contiguous()
        out = torch.empty((pv.shape[0], vc.patch_k_pad), dtype=torch.bfloat16, device=self.dev)
        _lib.call("o3v_cast_pad_f32_bf16", _ptr(pv), _ptr(out), pv.shape[0], pv.shape[1], vc.patch_k_pad, _stream())
        return out

    def pixels_from_frames(self, frames: torch.Tensor):
        """frames [T,3,H,W] uint8 or f32 (0..255, H,W multiples of 28) -> (bf16 [P,Kp], grid_thw [T,3])."""
        vc = self.cfg.vision
        if frames.dim() != 4 or frames.shape[1] != 3:
            raise ValueError("frames must be [T,3,H,W]")
        T, _, H, W = frames.shape
        ps = vc.patch_size
        if H % (2 * ps) or W % (2 * ps):
            raise ValueError(f"frame size must be a multiple of {2 * ps} (smart_resize output)")
        is_u8 = frames.dtype == torch.uint8
        fr = frames.to(self.dev).contiguous() if is_u8 else frames.to(self.dev, torch.float32).contiguous()
        P = T * (H // ps) * (W // ps)
        out = torch.empty((P, vc.patch_k_pad), dtype=torch.bfloat16, device=self.dev)
        _lib.call("o3v_patchify_ps", _ptr(fr), int(is_u8), _ptr(out), T, H, W, vc.patch_k_pad, ps, self.clip_mean, self.clip_std,
                  _stream())
        grid = np.asarray([[1, H // ps, W // ps]] * T, dtype=np.int64)
        return out, grid

    def pixels_from_video(self, frames: torch.Tensor):
        """ONE native video [T,3,H,W] uint8 or f32 (0..255) -> (bf16 [ceil(T/2)*gh*gw, Kp], grid [[ceil(T/2), gh, gw]]): temporal
        patch k holds frames 2k and 2k+1, an odd count repeats the last frame (TF:models/qwen2_vl/video_processing_qwen2_vl.py:236-274;
        R:src/r1-v/src/open_r1/vision_process.py:319-333 pads frame lists the same way)."""
        vc = self.cfg.vision
        if frames.dim() != 4 or frames.shape[1] != 3 or frames.shape[0] == 0:
            raise ValueError("video frames must be [T,3,H,W] with T >= 1")
        T, _, H, W = frames.shape
        ps = vc.patch_size
        if H % (2 * ps) or W % (2 * ps):
            raise ValueError(f"frame size must be a multiple of {2 * ps} (smart_resize output)")
        is_u8 = frames.dtype == torch.uint8
        fr = frames.to(self.dev).contiguous() if is_u8 else frames.to(self.dev, torch.float32).contiguous()
        gt = (T + 1) // 2
        out = torch.empty((gt * (H // ps) * (W // ps), vc.patch_k_pad), dtype=torch.bfloat16, device=self.dev)
        _lib.call("o3v_patchify_video", _ptr(fr), int(is_u8), _ptr(out), T, H, W, vc.patch_k_pad, ps, self.clip_mean, self.clip_std,
                  _stream())
        return out, np.asarray([[gt, H // ps, W // ps]], dtype=np.int64)

    def vit_forward(self, pixels_bf16: torch.Tensor, grid_thw) -> torch.Tensor:
        """TF:408-471 -> merged visual tokens bf16 [P/4, out_hidden] in original order."""
        vc = self.cfg.vision
        plan = self._vit_plan(grid_thw)
        P = plan["P"]
        if pixels_bf16.shape[0] != P or pixels_bf16.shape[1] != vc.patch_k_pad:
            raise ValueError(f"pixel rows {tuple(pixels_bf16.shape)} do not match grid ({P} patches)")
        if self.q3:
            # [1 + n_deep, P/4, out_hidden]: row block 0 = the merged visual tokens, 1.. = the DeepStack features (TF3:839-862) that
            # prefill adds to the first decoder layers' outputs at the visual positions; every consumer takes this one tensor
            nd = len(vc.deepstack_visual_indexes)
            nbytes = _lib.load().o3v_vit3_workspace_bytes(C.byref(self.w.vit3), P)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            out = torch.empty((1 + nd, P // vc.merge_unit, vc.out_hidden_size), dtype=torch.bfloat16, device=self.dev)
            _lib.call("o3v_vit3_forward", C.byref(self.w.vit3), _ptr(pixels_bf16), P, _ptr(plan["pos_embed"]), _ptr(plan["cos"]),
                      _ptr(plan["sin"]), _ptr(plan["tiles_full"]), plan["tiles_full"].shape[0], _ptr(ws), nbytes, _ptr(out[0]),
                      _ptr(out[1:]) if nd else None, _stream())
            return out
        nbytes = _lib.load().o3v_vit_workspace_bytes(C.byref(self.w.vit), P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        out = torch.empty((P // vc.merge_unit, vc.out_hidden_size), dtype=torch.bfloat16, device=self.dev)
        _lib.call("o3v_vit_forward", C.byref(self.w.vit), _ptr(pixels_bf16), P, _ptr(plan["win_idx"]), _ptr(plan["rev_idx"]),
                  _ptr(plan["cos"]), _ptr(plan["sin"]), _ptr(plan["tiles_win"]), plan["tiles_win"].shape[0],
                  _ptr(plan["tiles_full"]), plan["tiles_full"].shape[0], _ptr(ws), nbytes, _ptr(out), _stream())
        return out

    # ------------------------------------------------------------------------------------------ text
    def mrope_table(self, pos3: np.ndarray):
        """pos3 int [3, T] -> cos, sin bf16 [T, D] on device."""
        D = self.cfg.text.head_dim
        T = pos3.shape[1]
        p = torch.from_numpy(np.ascontiguousarray(pos3.astype(np.int32))).to(self.dev)
        cos = torch.empty((T, D), dtype=torch.bfloat16, device=self.dev)
        sin = torch.empty_like(cos)
        _lib.call("o3v_mrope_table", _ptr(p), _ptr(self.inv_freq), _ptr(self.axis_of), _ptr(cos), _ptr(sin), T, D, _stream())
        return cos, sin

    def embed(self, input_ids: np.ndarray, vis: Optional[torch.Tensor], first: int = 0, n_image_rows: Optional[int] = None) -> torch.Tensor:
        """inputs_embeds of the (flattened) prompt; `first` > 0 returns only rows first.. (prefix-KV reuse).  `vis` holds the
        image tokens followed by the video tokens (n_image_rows of the former; default: as many as image placeholders)."""
        src, n_img, n_vid = indexing.embed_source_rows(input_ids, self.cfg.image_token_id, self.cfg.video_token_id, n_image_rows)
        if vis is not None and vis.dim() == 3:      # Qwen3-VL: [1 + n_deep, n, H], block 0 = the visual tokens
            vis = vis[0]
        n_rows = 0 if vis is None else vis.shape[0]
        n_img_rows = n_img if n_image_rows is None else int(n_image_rows)
        if n_img != n_img_rows or n_img + n_vid != n_rows:
            what = "Video" if n_img == n_img_rows else "Image"
            raise ValueError(f"{what} features and {what.lower()} tokens do not match, tokens: {n_img} image + {n_vid} video, "
                             f"features: {n_img_rows} image + {n_rows - n_img_rows} video")
        src = src[first:]
        H = self.cfg.text.hidden_size
        T = src.shape[0]
        x = torch.empty((T, H), dtype=torch.bfloat16, device=self.dev)
        s = torch.from_numpy(src).to(self.dev)
        _lib.call("o3v_embed_scatter", _ptr(self.w.t["l.embed"]), _ptr(vis), _ptr(s), _ptr(x), T, H, _stream())
        return x

    def _prompt_key(self, ids, mask, pos, tensors):
        """Identity of a prompt for the generate -> log-prob hand-over: token ids, mask and rope positions as bytes, every visual input
        by shape, dtype and two 64-bit checksums of its bytes (a plain sum and a position-weighted one, computed where the tensor
        lives)."""
        parts = [ids.tobytes(), mask.tobytes(), pos.tobytes(), self.position_mode]
        for name, t in tensors:
            if t is None:
                parts.append((name, None))
                continue
            if isinstance(t, (list, tuple)):
                t = torch.cat([torch.as_tensor(u).reshape(-1).view(torch.uint8) for u in t])
            t = torch.as_tensor(t)
            raw = t.detach().contiguous().reshape(-1).view(torch.uint8)
            pad = (-raw.numel()) % 4
            if pad:
                raw = torch.cat([raw, raw.new_zeros(pad)])
            w = raw.view(torch.int32).to(torch.int64)
            wt = (torch.arange(w.numel(), device=w.device, dtype=torch.int64) & 0xFFFF) + 1
            parts.append((name, tuple(t.shape), str(t.dtype), int(w.sum().item()), int((w * wt).sum().item())))
        return tuple(parts)

    def _visual(self, pixel_values, image_grid_thw, frames, vis_embeds, pixel_values_videos, video_grid_thw, video_frames):
        """The visual side of a call -> (vis, image grid, video grid, n_image_rows): image tokens first, video tokens after them
        (each modality is scattered to its own placeholders in order, TF:1206-1215; the tower treats every temporal patch as its
        own attention segment either way, TF:vision_utils.py:60-65, so the two runs are independent)."""
        as_np = lambda g: None if g is None else np.asarray(g.cpu() if torch.is_tensor(g) else g, dtype=np.int64).reshape(-1, 3)
        grid, vgrid = as_np(image_grid_thw), as_np(video_grid_thw)
        if vis_embeds is not None:
            n_img = None if vgrid is None else (0 if grid is None else int((grid.prod(axis=1) // self.cfg.vision.merge_unit).sum()))
            return vis_embeds, grid, vgrid, n_img
        parts = []
        if frames is not None:
            px, grid = self.pixels_from_frames(frames)
            parts.append(self.vit_forward(px, grid))
        elif pixel_values is not None:
            parts.append(self.vit_forward(self.pixels_from_processor(pixel_values), grid))
        n_img = None
        if video_frames is not None or pixel_values_videos is not None:
            n_img = parts[0].shape[-2] if parts else 0
            if video_frames is not None:
                vids = video_frames if isinstance(video_frames, (list, tuple)) else [video_frames]
                px_g = [self.pixels_from_video(v) for v in vids]
                px, vgrid = torch.cat([p for p, _ in px_g]), np.concatenate([g for _, g in px_g])
            else:
                px = self.pixels_from_processor(pixel_values_videos)
            parts.append(self.vit_forward(px, vgrid))
        if not parts:
            return None, None, None, None
        return (parts[0] if len(parts) == 1 else torch.cat(parts, dim=-2)), grid, vgrid, n_img

    def _positions(self, ids, mask, grid, vgrid, second_per_grid_ts):
        cfg = self.cfg
        if grid is None and vgrid is None:
            p1 = np.where(mask == 0, 0, np.cumsum(mask, axis=1) - 1)
            return np.broadcast_to(p1[None], (3,) + ids.shape).copy(), np.zeros(ids.shape[0], dtype=np.int64)
        return indexing.rope_index(ids, mask, grid, cfg.image_token_id, cfg.vision.spatial_merge_size, video_grid_thw=vgrid,
                                   video_token_id=cfg.video_token_id, second_per_grid_ts=second_per_grid_ts,
                                   tokens_per_second=cfg.vision.tokens_per_second, split_video_frames=self.q3,
                                   mode="tf5" if self.q3 else self.position_mode)

    def alloc_cache(self, B, Tmax):
        tc = self.cfg.text
        shape = (tc.num_hidden_layers, B, tc.num_key_value_heads, Tmax, tc.head_dim)
        return (torch.empty(shape, dtype=torch.bfloat16, device=self.dev),
                torch.empty(shape, dtype=torch.bfloat16, device=self.dev))

    def prefill(self, x: torch.Tensor, pos3: np.ndarray, pad: Sequence[int], B: int, S: int, kc, vc, past: int = 0,
                deepstack=None, prefix=None):
        """Runs the prompt through the LLM (TF:790-872); x [B*S,H] becomes the last layer's residual stream.
        past > 0: `past` tokens per row are already computed (HF `past_key_values` semantics); x / pos3 are the S tokens
        after them.  By default they sit in slots 0..past-1 of kc / vc.  prefix = (kc0, vc0, rows_per_prefix): they sit ONCE
        per prompt in kc0 / vc0 [layers, B/rows_per_prefix, Hkv, >=past, D] instead, and kc / vc hold only the new tokens (the
        G completions of a prompt behind one copy of its K/V).  deepstack = (input_ids, vis [1+n_deep, n, H]) of a Qwen3-VL
        prompt: the DeepStack features are added after the first decoder layers at the visual rows (TF3:839-862)."""
        Tmax = kc.shape[3]
        cos, sin = self.mrope_table(pos3.reshape(3, B * S))
        tiles = torch.from_numpy(indexing.prefill_tiles(B, S, pad, past=past)).to(self.dev)
        nbytes = _lib.load().o3v_llm_workspace_bytes(C.byref(self.w.llm), B * S)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        opts = _lib.PrefillOpts()
        if self.fp8_prefill:
            if not getattr(self.w, "fp8_decode", False):
                raise _lib.O3VError("fp8_prefill needs the fp8 weight rows (DeviceWeights(fp8_decode=True) / quantization='fp8')")
            opts.w8a8 = 1
        keep = []
        if deepstack is not None and deepstack[1] is not None and deepstack[1].dim() == 3 and deepstack[1].shape[0] > 1:
            ids_all, vis = deepstack
            rows, src = indexing.deepstack_rows(ids_all, self.cfg.image_token_id, first=past, video_token_id=self.cfg.video_token_id)
            rows_d, src_d = torch.from_numpy(rows).to(self.dev), torch.from_numpy(src).to(self.dev)
            feat = vis[1:]
            keep += [rows_d, src_d, feat]
            opts.ds_rows, opts.ds_src, opts.n_ds = rows_d.data_ptr(), src_d.data_ptr(), int(rows.shape[0])
            opts.ds_feat, opts.n_deep, opts.ds_stride = feat.data_ptr(), int(feat.shape[0]), int(feat.stride(0))
        if prefix is not None:
            kc0, vc0, rpp = prefix
            if kc0.shape[3] < past or not kc0.is_contiguous() or not vc0.is_contiguous() or B % rpp or kc0.shape[1] != B // rpp:
                raise ValueError("prefix caches do not match the rows they serve")
            opts.kprefix, opts.vprefix = kc0.data_ptr(), vc0.data_ptr()
            opts.prefix_len, opts.prefix_cap, opts.rows_per_prefix = int(past), int(kc0.shape[3]), int(rpp)
        _lib.call("o3v_llm_prefill_ex", C.byref(self.w.llm), _ptr(x), _ptr(cos), _ptr(sin), _ptr(tiles), tiles.shape[0],
                  indexing.PREFILL_TILE, _ptr(kc), _ptr(vc), B, S, int(past), Tmax, C.byref(opts), _ptr(ws), nbytes, _stream())
        del keep
        return x

    def head(self, x_rows: torch.Tensor) -> torch.Tensor:
        """final norm + lm_head: bf16 [R,H] (row stride may exceed H) -> logits bf16 [R,V]."""
        R = x_rows.shape[0]
        H, V = self.cfg.text.hidden_size, self.cfg.text.vocab_size
        normed = torch.empty((R, H), dtype=torch.bfloat16, device=self.dev)
        logits = torch.empty((R, V), dtype=torch.bfloat16, device=self.dev)
        _lib.call("o3v_llm_head", C.byref(self.w.llm), _ptr(x_rows), x_rows.stride(0), R, _ptr(normed), _ptr(logits), _stream())
        return logits

    # ------------------------------------------------------------------------------------------ generate
    class _InLaunchWaitGaveUp(_lib.O3VError):
        pass

    @torch.no_grad()
    def generate(self, *args, **kw) -> "GenerateOutput":
        """`_generate` with one safety net: if an in-launch wait of the one-launch decode block gives up (its waiting workgroups were
        not all resident -- e.g. another process holds part of the GPU -- and the sticky time-out word is set), the SAME call is
        run again in this process on the stand-alone kernels (`fused_decode=False` for that call), logged once.  The word is read
        after the first chunk of steps and at the end, so a starved run is noticed after milliseconds, not after the completion."""
        try:
            return self._generate(*args, **kw)
        except O3VEngine._InLaunchWaitGaveUp as e:
            if not getattr(self, "_warned_fused_fallback", False):
                import logging
                logging.getLogger(__name__).warning("%s; re-running this generate call on the stand-alone decode kernels", e)
                self._warned_fused_fallback = True
            self.fused_fallbacks = getattr(self, "fused_fallbacks", 0) + 1
            return self._generate(*args, **kw, _fused_ok=False)

    @torch.no_grad()
    def _generate(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None, frames=None,
                 max_new_tokens=16, eos_token_ids: Sequence[int] = (), pad_token_id: Optional[int] = None,
                 repetition_penalty: float = 1.0, do_sample: bool = False, temperature: float = 1.0, top_p: float = 1.0,
                 top_k: int = 0, num_return_sequences: int = 1, seed: int = 0, row_ids: Optional[Sequence[int]] = None,
                 vis_embeds: Optional[torch.Tensor] = None, steps_per_sync: int = 16, return_margins: bool = True,
                 sync_timings: bool = False, prefix_key=None, pixel_values_videos=None, video_grid_thw=None, video_frames=None,
                 second_per_grid_ts=None, _fused_ok: bool = True) -> GenerateOutput:
        """HF-semantics generate.  `num_return_sequences=G` shares ONE ViT pass and ONE prefill across the G
        completions of a prompt (the reference recomputes both G times, TF:1493-1579) and fans the KV cache out.

        `prefix_key` (hashable, single un-padded prompt only): identifies the visual content of the prompt.  The
        K/V of the prompt is kept under that key; a later call with the same key prefills only the tokens after
        the longest common token prefix (V-STAR asks 5 questions per video with the frame block first,
        R:eval/test/test_vstar_multi_images.py:205-206,511-544; self-consistency draws N samples of one prompt,
        R:eval/tts.py:47-123).  Causal attention makes the prefix K/V independent of what follows it."""
        cfg, tc = self.cfg, self.cfg.text
        ids = np.asarray(input_ids.cpu() if torch.is_tensor(input_ids) else input_ids, dtype=np.int64)
        if ids.ndim == 1:
            ids = ids[None]
        B0, S = ids.shape
        mask = np.ones_like(ids) if attention_mask is None else np.asarray(
            attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask, dtype=np.int64)
        pad = (mask == 0).sum(axis=1)
        if not all((mask[b, pad[b]:] == 1).all() for b in range(B0)):
            raise ValueError("only left padding is supported (padding_side='left', R:grpo_trainer.py:546)")
        G = int(num_return_sequences)
        B = B0 * G
        if B > self.MAX_ROWS:
            raise ValueError(f"at most {self.MAX_ROWS} sequences per engine call (two 16-column MFMA blocks in the decode linears); "
                             "shard larger batches")
        pad_id = cfg.pad_token_id if pad_token_id is None else int(pad_token_id)
        T = int(max_new_tokens)
        if T <= 0:
            raise ValueError(f"`max_new_tokens` must be greater than 0, but is {T}.")   # GenerationConfig.validate's rule
        tm = {}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if sync_timings else None

        def mark(i):
            if ev is not None:
                ev[i].record()

        mark(0)
        # ---- vision (once per prompt batch): images (frames-as-images, R:…/grpo_trainer.py:540-548) and / or native videos
        # (pixel_values_videos + <|video_pad|>, R:…:555-564; R:eval/models/model_vllm.py:72-88)
        vis, grid, vgrid, n_img_rows = self._visual(pixel_values, image_grid_thw, frames, vis_embeds, pixel_values_videos, video_grid_thw,
                                                    video_frames)
        mark(1)
        # ---- positions + prefill (once per prompt)
        pos, deltas = self._positions(ids, mask, grid, vgrid, second_per_grid_ts)
        # G completions of a prompt share its K/V (head_dim 128, own keys in <= 32 splits): the prompt's K/V is kept ONCE
        # (kc0 / vc0) and every row's cache holds only its generated tokens -- no G-fold copy of the prompt
        own_splits = (T + 127) // 128
        shared_prompt = G > 1 and tc.head_dim == 128 and own_splits <= 32 and self.group_attention
        Tmax = T if shared_prompt else S + T
        kc, vc = self.alloc_cache(B, Tmax)
        kc0 = vc0 = None
        past = 0
        use_prefix = prefix_key is not None and B0 == 1 and int(pad[0]) == 0
        if use_prefix:
            past = self._prefix_lookup(prefix_key, ids[0])
        x = self.embed(ids, vis, first=past, n_image_rows=n_img_rows)
        ds = (ids, vis) if self.q3 else None
        if G == 1 and not use_prefix:
            self.prefill(x, pos, pad, B0, S, kc, vc, deepstack=ds)
        else:
            kc0, vc0 = self.alloc_cache(B0, S)
            if past:
                ent = self._prefix[prefix_key]
                kc0[:, :, :, :past].copy_(ent["k"][:, :, :, :past])
                vc0[:, :, :, :past].copy_(ent["v"][:, :, :, :past])
            self.prefill(x, pos[:, :, past:], pad, B0, S - past, kc0, vc0, past=past, deepstack=ds)
            if not shared_prompt:
                # KV fan-out: completion g of prompt b is row b*G+g (repeat_interleave order, R:grpo_trainer.py:586)
                # (a broadcast copy into the [layers, B0, G, ...] view of the caches: no G-times temporary)
                shp = (kc.shape[0], B0, G) + tuple(kc.shape[2:])
                kc.view(shp)[:, :, :, :, :S].copy_(kc0[:, :, None])
                vc.view(shp)[:, :, :, :, :S].copy_(vc0[:, :, None])
            if use_prefix:
                self._prefix_store(prefix_key, ids[0], kc0, vc0)
            if not shared_prompt:
                kc0 = vc0 = None
        tm["prefix_tokens_reused"] = past
        self._last_prompt = None
        if self.reuse_prompt_kv and shared_prompt and B0 == 1:
            key = self._prompt_key(ids, mask, pos, (("pv", pixel_values), ("grid", grid), ("frames", frames), ("vis", vis_embeds),
                                                    ("pvv", pixel_values_videos), ("vgrid", vgrid), ("vframes", video_frames)))
            self._last_prompt = {"key": key, "kc0": kc0, "vc0": vc0, "x_last": x.view(B0, S - past, -1)[0, -1:, :].clone(),
                                 "deltas": np.array(deltas, copy=True)}
        last = x.view(B0, S - past, -1)[:, -1, :]               # left padding: every row ends at S-1
        logits0 = self.head(last)                               # [B0, V]
        logits = logits0.repeat_interleave(G, dim=0).contiguous() if G > 1 else logits0
        mark(2)
        # ---- decode state
        V, H = tc.vocab_size, tc.hidden_size
        dpos = indexing.decode_positions(mask, deltas, T)       # [3,B0,T]
        if G > 1:
            dpos = np.repeat(dpos, G, axis=1)
        cosd, sind = self.mrope_table(dpos.reshape(3, B * T))
        seen = torch.zeros((B, V), dtype=torch.uint8, device=self.dev)
        ids_rep = np.repeat(ids, G, axis=0) if G > 1 else ids
        pad_rep = np.repeat(pad, G) if G > 1 else pad
        if repetition_penalty != 1.0:
            ids_dev = torch.from_numpy(ids_rep.astype(np.int32)).to(self.dev)
            _lib.call("o3v_mark_seen", _ptr(ids_dev), _ptr(seen), B, S, V, _stream())
        cur_tok = torch.zeros(B, dtype=torch.int32, device=self.dev)
        finished = torch.zeros(B, dtype=torch.int32, device=self.dev)
        out_ids = torch.full((B, T), pad_id, dtype=torch.int32, device=self.dev)
        margins = torch.zeros((B, T), dtype=torch.float32, device=self.dev) if return_margins else None
        eos = torch.tensor(list(eos_token_ids) or [0], dtype=torch.int32, device=self.dev)
        k_lo = torch.from_numpy(pad_rep.astype(np.int32)).to(self.dev)
        rid = torch.tensor(list(row_ids) if row_ids is not None else list(range(B)), dtype=torch.int32, device=self.dev)
        n_rep_total = B * tc.num_attention_heads
        # context splits of the decode attention: one 128-key chunk per block at 1-4 rows; from 8 rows on fewer, longer chunks
        # so that the grid stays near 640 blocks (measured, 7B: 8 rows 40 -> 20 splits 3.85 -> 3.72 ms/step; 16 independent rows
        # 32 -> 8 splits 2020 -> 2128 tok/s; 2 and 4 rows are best at 40)
        nsplit = max(1, min(64, (S + T + 127) // 128, max(1, 640 // max(1, B * tc.num_key_value_heads))))
        if os.environ.get("O3V_NSPLIT"):      # A/B switch (tools/probes)
            nsplit = int(os.environ["O3V_NSPLIT"])
        # G completions of a prompt share its K/V: the group kernel reads the prompt keys once per group (head_dim 128,
        # G * n_rep <= 64 query rows per kv head, prefix splits + own-key splits <= 64)
        n_rep = tc.num_attention_heads // tc.num_key_value_heads
        group = G if shared_prompt else 0
        mode = self.group_attention_mode
        if group:
            # the one-pass kernel holds a sub-group's query rows as MFMA columns (<= 64); sub-groups of 4 rows measured best
            sub = max(d for d in range(1, G + 1) if G % d == 0 and d <= 4 and d * n_rep <= 64)
            if mode == "auto":
                # measured, 7B dims, ms per decode step (tools/measure_configs.py rollout / rollout_eval), per-row kernel reading
                # the leader's prefix through the shared L2 | one-pass kernel in sub-groups of 4 (prefix + own keys, one launch):
                #   S=4.5k: G=4 3.45 | 3.53   G=8 3.71 | 3.79   G=12 4.06 | 4.09   G=16 4.42 | 4.36
                #   S=10k : G=8 4.37 | 3.97
                mode = "kernel" if (S >= 8192 or B >= 16) and sub > 1 else "shared_read"
            if mode == "kernel" and sub > 1:
                group = sub
                # prefix splits: ~320 blocks in all (measured: G=16, S=4.5k 36 -> 18 splits 4.80 -> 4.63 ms/step; G=8, S=10k is
                # best left at its 63)
                nsplit = max(1, min(64 - own_splits, (S + 127) // 128, max(1, 320 // (tc.num_key_value_heads * (B // sub)))))
                if S >= 8192:
                    nsplit = max(1, min(64 - own_splits, (S + 127) // 128))
            else:
                nsplit = -nsplit
        part_o = torch.empty(n_rep_total * 64 * tc.head_dim, dtype=torch.float32, device=self.dev)
        part_ml = torch.empty(n_rep_total * 64 * 2, dtype=torch.float32, device=self.dev)
        scratch = torch.empty((B, _lib.SAMPLE_SCRATCH_FLOATS if do_sample else 256), dtype=torch.float32, device=self.dev)
        xdec = torch.empty((B, H), dtype=torch.bfloat16, device=self.dev)
        nbytes = _lib.load().o3v_llm_workspace_bytes(C.byref(self.w.llm), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        # tickets / mailboxes of the one-launch attention block (batch 1): zeroed once per call, epochs count the launches
        sync = None
        # (8..32 rows: the residual linears of the batched decode normalise for the next linear through the same buffer)
        if self.fused_decode and _fused_ok and (B == 1 or (B >= 8 and self.tail_norm)):
            sync = torch.zeros(_lib.load().o3v_decode_sync_bytes(), dtype=torch.uint8, device=self.dev)
            if getattr(self, "_debug_poison_sync", False):     # test hook: tickets that can never reach "last" -> the waits give up
                sync[:1024].view(torch.int32).fill_(0x40000000)
        st = _lib.DecodeState(B=B, S=S, Tmax=Tmax, Tnew=T, nsplit=nsplit, pad_id=pad_id, n_eos=len(eos_token_ids),
                              do_sample=int(do_sample), rep_penalty=float(repetition_penalty), temperature=float(temperature),
                              top_p=float(top_p), seed=int(seed) & (2 ** 64 - 1), x=xdec.data_ptr(), kcache=kc.data_ptr(),
                              vcache=vc.data_ptr(), cosT=cosd.data_ptr(), sinT=sind.data_ptr(), logits=logits.data_ptr(),
                              seen=seen.data_ptr(), cur_tok=cur_tok.data_ptr(), finished=finished.data_ptr(),
                              out_ids=out_ids.data_ptr(), margins=0 if margins is None else margins.data_ptr(),
                              eos_ids=eos.data_ptr(), k_lo=k_lo.data_ptr(), row_id=rid.data_ptr(),
                              part_o=part_o.data_ptr(), part_ml=part_ml.data_ptr(),
                              sample_scratch=scratch.data_ptr(), workspace=ws.data_ptr(),
                              ws_bytes=nbytes, group=group, sync=0 if sync is None else sync.data_ptr(),
                              top_k=max(0, int(top_k or 0)), kprefix=kc0.data_ptr() if shared_prompt else 0,
                              vprefix=vc0.data_ptr() if shared_prompt else 0, prefix_cap=S if shared_prompt else 0,
                              rows_per_prompt=G if shared_prompt else 0)
        stats = (C.c_longlong * 4)(0, 0, 0, 0)      # decode forwards, launches in their layer loops, fused / stand-alone attention halves
        st.host_stats = C.cast(stats, C.c_void_p)
        st.flags = (1 if self.layer_block else 0) | (0 if self.tail_norm else 2) | (4 if os.environ.get("O3V_QKV_FUSED_NORM", "1") == "0" else 0)
        tm["kv_cache_bytes"] = int((kc.numel() + vc.numel() + (kc0.numel() + vc0.numel() if shared_prompt else 0)) * 2)
        # ---- decode loop: chunks of steps enqueued from C++, one host check per chunk (TF:utils.py:2936-2937)
        done = 0
        use_eos = len(eos_token_ids) > 0
        chunk = max(1, int(steps_per_sync)) if use_eos else T

        def check_waits():
            if sync is None:
                return
            code = int(sync[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
            if code:
                raise O3VEngine._InLaunchWaitGaveUp(f"decode: an in-launch wait gave up (code {code:#x})")

        while done < T:
            # the first chunk is short whenever the one-launch block is in use: its time-out word is read right after it
            n = min(chunk if (done or sync is None) else min(chunk, getattr(self, "_first_chunk", 16)), T - done)
            _lib.call("o3v_llm_decode", C.byref(self.w.llm), C.byref(st), done, n, int(done + n == T), _stream())
            first = done == 0
            done += n
            if first and done < T:
                check_waits()
            if use_eos and done < T and bool(finished.all().item()):
                break
        mark(3)
        layers_run = stats[0] * tc.num_hidden_layers
        tm["decode_forwards"] = int(stats[0])
        tm["launches_per_layer"] = (stats[1] / layers_run) if layers_run else None
        tm["fused_attention_layers"], tm["standalone_attention_layers"] = int(stats[2]), int(stats[3])
        check_waits()
        if getattr(self, "_debug_keep", False):     # probe hook: the decode's state after the last step
            self._debug_last = {"kc": kc, "vc": vc, "logits": logits, "x": xdec, "S": S, "cos": cosd, "sin": sind, "k_lo": k_lo,
                                "nsplit": nsplit, "Tmax": Tmax, "out_ids": out_ids}
        gen = out_ids[:, :done].to(torch.int64)
        if use_eos and done > 0:
            # HF stops at the step where every row has finished: trim trailing all-pad columns generated past it
            fin_col = self._first_all_finished(gen, eos_token_ids, pad_id)
            gen = gen[:, :fin_col]
            done = fin_col
        prompt = torch.from_numpy(ids_rep).to(self.dev)
        seqs = torch.cat([prompt, gen], dim=1)
        if ev is not None:
            torch.cuda.synchronize()
            tm.update({"vit_ms": ev[0].elapsed_time(ev[1]), "prefill_ms": ev[1].elapsed_time(ev[2]),
                       "decode_ms": ev[2].elapsed_time(ev[3])})
        return GenerateOutput(sequences=seqs, margins=None if margins is None else margins[:, :done], n_steps=done,
                              timings=tm)

    # ------------------------------------------------------------------------------------------ prefix K/V
    PREFIX_ENTRIES = 2   # videos whose prompt K/V stay resident (7B, S=4.5k: 0.26 GB each; 288 GB HBM is not the limit)

    def _prefix_lookup(self, key, ids_row: np.ndarray) -> int:
        """Tokens of `ids_row` whose K/V are cached under `key`: the longest common prefix, capped at S-1 so the
        last prompt token is always run (its hidden state feeds the first logits)."""
        ent = self._prefix.get(key)
        if ent is None:
            return 0
        old = ent["ids"]
        n = min(len(old), len(ids_row) - 1)
        neq = np.flatnonzero(old[:n] != ids_row[:n])
        return int(neq[0]) if neq.size else int(n)

    def _prefix_store(self, key, ids_row: np.ndarray, kc0, vc0):
        self._prefix.pop(key, None)
        while len(self._prefix) >= self.PREFIX_ENTRIES:
            self._prefix.pop(next(iter(self._prefix)))
        self._prefix[key] = {"ids": np.array(ids_row, copy=True), "k": kc0, "v": vc0}

    def drop_prefix_cache(self):
        self._prefix = {}

    @staticmethod
    def _first_all_finished(gen: torch.Tensor, eos_ids, pad_id) -> int:
        """Number of columns HF would have produced: up to and including the step at which the last row hit EOS."""
        g = gen.cpu().numpy()
        B, T = g.shape
        fin_at = np.full(B, T, dtype=np.int64)
        for b in range(B):
            hit = np.flatnonzero(np.isin(g[b], list(eos_ids)))
            if hit.size:
                fin_at[b] = hit[0] + 1
        return int(min(T, fin_at.max()))

    # ------------------------------------------------------------------------------------------ logits / logps
    @torch.no_grad()
    def forward_logits(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None, frames=None,
                       vis_embeds=None, pixel_values_videos=None, video_grid_thw=None, video_frames=None,
                       second_per_grid_ts=None) -> torch.Tensor:
        """model(input_ids, ...).logits bf16 [B,L,V] (R:grpo_trainer.py:375)."""
        ids = np.asarray(input_ids.cpu() if torch.is_tensor(input_ids) else input_ids, dtype=np.int64)
        B, S = ids.shape
        mask = np.ones_like(ids) if attention_mask is None else np.asarray(
            attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask, dtype=np.int64)
        pad = (mask == 0).sum(axis=1)
        vis, grid, vgrid, n_img_rows = self._visual(pixel_values, image_grid_thw, frames, vis_embeds, pixel_values_videos, video_grid_thw,
                                                    video_frames)
        pos, _ = self._positions(ids, mask, grid, vgrid, second_per_grid_ts)
        kc, vc = self.alloc_cache(B, S)
        x = self.embed(ids, vis, n_image_rows=n_img_rows)
        self.prefill(x, pos, pad, B, S, kc, vc, deepstack=(ids, vis) if self.q3 else None)
        return self.head(x).view(B, S, -1)

    @torch.no_grad()
    def completion_logps(self, prompt_ids, completion_ids, attention_mask=None, pixel_values=None, image_grid_thw=None,
                         frames=None, vis_embeds=None, rows_per_chunk: int = 2048, pixel_values_videos=None, video_grid_thw=None,
                         video_frames=None, second_per_grid_ts=None) -> torch.Tensor:
        """log p(completion token | everything before it) for the G completions of ONE prompt -> f32 [G, T].

        What R:grpo_trainer.py:371-384 + :612-613 compute (`_get_per_token_logps(model, prompt_completion_ids, ...)
        [:, prompt_length-1:]`) for the policy and for the reference model, restructured around what the G rows share:
        the ViT runs once (the reference re-encodes the frames per row), the prompt is prefilled once and its K/V fanned
        out, the G x T completion tokens run as one pass behind that prefix, and the lm_head + log-softmax touch only the
        G x T positions that are kept, `rows_per_chunk` rows of logits at a time (the reference materialises
        [G, S+T, vocab]: 12.8 GB at G=8, S+T=5.2k)."""
        cfg, tc = self.cfg, self.cfg.text
        ids = np.asarray(prompt_ids.cpu() if torch.is_tensor(prompt_ids) else prompt_ids, dtype=np.int64).reshape(1, -1)
        comp = torch.as_tensor(completion_ids).to(torch.int64)
        if comp.dim() != 2:
            raise ValueError("completion_ids must be [G, T]")
        G, T = comp.shape
        S = ids.shape[1]
        mask = np.ones_like(ids) if attention_mask is None else np.asarray(
            attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask, dtype=np.int64).reshape(1, -1)
        pad = (mask == 0).sum(axis=1)
        if not (mask[0, pad[0]:] == 1).all():
            raise ValueError("only left padding is supported (padding_side='left', R:grpo_trainer.py:546)")
        H = tc.hidden_size
        out = torch.empty((G, T), dtype=torch.float32, device=self.dev)
        if T == 0 or G == 0:
            return out
        as_np = lambda g: None if g is None else np.asarray(g.cpu() if torch.is_tensor(g) else g, dtype=np.int64).reshape(-1, 3)
        grid, vgrid = as_np(image_grid_thw), as_np(video_grid_thw)
        if frames is not None:
            ps = cfg.vision.patch_size
            grid = np.asarray([[1, frames.shape[2] // ps, frames.shape[3] // ps]] * frames.shape[0], dtype=np.int64)
        if video_frames is not None:
            vgrid = None      # derived from the frames by the tower's front end: no hand-over for this input form
        memo, self.timings_last_logps = self._last_prompt, {"prompt_kv_reused": False}
        pos = deltas = None
        if memo is not None and (video_frames is None):
            pos, deltas = self._positions(ids, mask, grid, vgrid, second_per_grid_ts)
            key = self._prompt_key(ids, mask, pos, (("pv", pixel_values), ("grid", grid), ("frames", frames), ("vis", vis_embeds),
                                                    ("pvv", pixel_values_videos), ("vgrid", vgrid), ("vframes", video_frames)))
            if key != memo["key"]:
                memo = None
        else:
            memo = None
        if memo is not None:
            # the prompt generate has just prefilled: same bytes in, same kernels -> the K/V and the last hidden row it kept
            kc0, vc0, x_last, deltas = memo["kc0"], memo["vc0"], memo["x_last"], memo["deltas"]
            self.timings_last_logps["prompt_kv_reused"] = True
        else:
            vis, grid, vgrid, n_img_rows = self._visual(pixel_values, image_grid_thw, frames, vis_embeds, pixel_values_videos,
                                                        video_grid_thw, video_frames)
            pos, deltas = self._positions(ids, mask, grid, vgrid, second_per_grid_ts)
            # prompt once
            kc0, vc0 = self.alloc_cache(1, S)
            x = self.embed(ids, vis, n_image_rows=n_img_rows)
            self.prefill(x, pos, pad, 1, S, kc0, vc0, deepstack=(ids, vis) if self.q3 else None)
            x_last = x[S - 1:S].clone()                               # hidden state that predicts completion token 0
            del x
        rows = torch.empty((G, T, H), dtype=torch.bfloat16, device=self.dev)
        rows[:, 0] = x_last
        if T > 1:
            # the first T-1 completion tokens of every row behind the shared prompt K/V
            # (their caches hold only their own T-1 tokens; the prompt's K/V is read from kc0 / vc0, kept once)
            kc, vc = self.alloc_cache(G, T - 1)
            ctok = comp[:, :T - 1].to(self.dev, torch.int32).contiguous().view(-1)
            xc = torch.empty((G * (T - 1), H), dtype=torch.bfloat16, device=self.dev)
            _lib.call("o3v_embed_tokens", _ptr(self.w.t["l.embed"]), _ptr(ctok), _ptr(xc), G * (T - 1), H, _stream())
            dpos = np.repeat(indexing.decode_positions(mask, deltas, T - 1), G, axis=1)      # [3, G, T-1]
            self.prefill(xc, dpos, np.repeat(pad, G), G, T - 1, kc, vc, past=S, prefix=(kc0, vc0, G))
            rows[:, 1:] = xc.view(G, T - 1, H)
            del kc, vc, xc
        del kc0, vc0
        flat = rows.view(G * T, H)
        tgt = comp.to(self.dev, torch.int32).contiguous().view(-1)
        V = tc.vocab_size
        of = out.view(-1)
        step = max(16, int(rows_per_chunk))
        for r0 in range(0, G * T, step):
            r1 = min(G * T, r0 + step)
            lg = self.head(flat[r0:r1])
            _lib.call("o3v_logprob_gather", _ptr(lg), _ptr(tgt[r0:r1]), _ptr(of[r0:r1]), r1 - r0, V, V, _stream())
            del lg
        return out

    @torch.no_grad()
    def per_token_logps(self, logits: torch.Tensor, input_ids) -> torch.Tensor:
        """R:grpo_trainer.py:371-384: log_softmax(logits[:, :-1]) gathered at input_ids[:, 1:] -> f32 [B, L-1]."""
        B, L, V = logits.shape
        ids = torch.as_tensor(input_ids).to(self.dev)
        tgt = ids[:, 1:].to(torch.int32).contiguous().view(-1)
        lg = logits[:, :-1, :].contiguous().view(-1, V)
        out = torch.empty(lg.shape[0], dtype=torch.float32, device=self.dev)
        _lib.call("o3v_logprob_gather", _ptr(lg), _ptr(tgt), _ptr(out), lg.shape[0], V, V, _stream())
        return out.view(B, L - 1)
