"""HF-style facade (facade 1): an object the Open-o3-Video GRPO trainer can use as `model`.

    model = Qwen2_5_VLForConditionalGeneration.from_pretrained(local_dir, torch_dtype=torch.bfloat16, ...)
    ids   = model.generate(**prompt_inputs, generation_config=GenerationConfig(...))     # R:grpo_trainer.py:581-582
    logit = model(input_ids, attention_mask=..., pixel_values=..., image_grid_thw=...).logits   # R:grpo_trainer.py:375

Same argument names and output conventions as transformers' Qwen2_5_VLForConditionalGeneration.generate:
returns i64 [B*G, S+T], prompt columns preserved, rows padded with pad_token_id after EOS, row b*G+g is completion g
of prompt b (TF:modeling_qwen2_5_vl.py:1493-1579).  Unlike the reference it runs the ViT and the prefill once per
prompt and fans the KV cache out to the G completions.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .config import O3VConfig
from .engine import O3VEngine
from .weights import DeviceWeights, getter_from_dict, getter_from_safetensors_dir

MAX_ROWS = 32  # sequences per engine call (O3VEngine.MAX_ROWS)


class GenerationConfigLike(SimpleNamespace):
    """Duck-typed subset of transformers.GenerationConfig (max_new_tokens, do_sample, temperature, top_k, top_p,
    repetition_penalty, num_return_sequences, pad_token_id, eos_token_id).  A field that is absent or None is "not set by the
    caller", exactly as in transformers >= 5 where GenerationConfig() leaves everything it was not given at None."""


# GenerationConfig._get_default_generation_params() of transformers 5.15 for the fields the path reads
GLOBAL_GENERATION_DEFAULTS = dict(max_new_tokens=20, do_sample=False, temperature=1.0, top_k=50, top_p=1.0,
                                  repetition_penalty=1.0, num_return_sequences=1)
_GEN_FIELDS = tuple(GLOBAL_GENERATION_DEFAULTS) + ("pad_token_id", "eos_token_id", "bos_token_id")


def load_generation_config(path: str) -> dict:
    """<checkpoint>/generation_config.json (what GenerationConfig.from_pretrained reads); {} if the file is absent."""
    f = os.path.join(path, "generation_config.json")
    if not os.path.isfile(f):
        return {}
    import json
    with open(f) as fh:
        d = json.load(fh)
    return {k: v for k, v in d.items() if k in _GEN_FIELDS and v is not None}


_SPECIAL_TOKEN_FIELDS = ("pad_token_id", "eos_token_id", "bos_token_id")


def resolve_generation_config(passed, model_gc, kwargs, mode: str = "pinned") -> dict:
    """GenerationMixin._prepare_generation_config.  Priority: generate() kwargs > the passed generation_config > ... and the
    rest depends on the library version, hence `mode`:

    "pinned" (default) -- the library the reference installs (transformers @336dc69d, R:setup.sh:4, before 4.50): a PASSED
        generation_config is used as it is; only the special tokens it leaves unset (eos / pad / bos) fall back to the model's
        generation config, everything else it does not carry takes GenerationConfig()'s constructor defaults (top_k 50,
        repetition_penalty 1.0, temperature 1.0, top_p 1.0).  So the trainer's config (R:src/r1-v/src/open_r1/trainer/
        grpo_trainer.py:306-313) samples with top_k = 50 and no repetition penalty whatever the checkpoint's
        generation_config.json says.  With NO passed config the model's generation config is the base.  Restated from the
        published behaviour of that release line; the pinned commit is not installed here: parity unpinned.
    "tf5" -- transformers >= 5 (5.15 installed here, golden G8b): GenerationConfig() leaves what it was not given at None
        and every such field is taken from the model's generation config (the checkpoint's generation_config.json) before
        the global defaults."""
    if mode not in ("pinned", "tf5"):
        raise ValueError(f"mode={mode!r}: 'pinned' or 'tf5'")
    out = {}
    for k in _GEN_FIELDS:
        v = kwargs.get(k)
        if v is None and passed is not None:
            v = getattr(passed, k, None)
        from_model = mode == "tf5" or passed is None or k in _SPECIAL_TOKEN_FIELDS
        if v is None and model_gc is not None and from_model:
            v = getattr(model_gc, k, None)
        if v is None:
            v = GLOBAL_GENERATION_DEFAULTS.get(k)
        out[k] = v
    return out


class Qwen2_5_VLForConditionalGeneration:
    def __init__(self, cfg: O3VConfig, engine: O3VEngine):
        self.engine = engine
        self.o3v_config = cfg
        self.config = SimpleNamespace(_name_or_path=cfg.name_or_path, image_token_id=cfg.image_token_id,
                                      video_token_id=cfg.video_token_id, eos_token_id=cfg.eos_token_id,
                                      pad_token_id=cfg.pad_token_id, vocab_size=cfg.text.vocab_size)
        self.warnings_issued = {}          # R:grpo_trainer.py:330 touches it
        # GenerationConfig.from_model_config: only the special tokens come from config.json; from_pretrained replaces this by
        # the checkpoint's generation_config.json when there is one
        self.generation_config = GenerationConfigLike(pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id)
        self.training = False
        self._seed = 0
        # how a passed GenerationConfig is completed from the checkpoint's (resolve_generation_config): "pinned" = the library
        # version the reference installs, "tf5" = transformers >= 5
        self.generation_config_mode = "pinned"
        # video rope arithmetic (indexing.rope_index mode): "pinned" = transformers @336dc69d as the reference installs it
        # (R:setup.sh:4; restated, parity unpinned), "tf5" = transformers 5.15 (goldens G5b / G14 / G15).  Images do not depend on it.
        self.position_mode = "pinned"

    # ---- construction
    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, attn_implementation=None, use_cache=True, device="cuda", fp8_decode=False,
                        fp8_prefill=False, **_):
        """Local checkpoint directory only (config.json + *.safetensors).  `attn_implementation` is accepted and
        ignored: attention always runs in the hand-written HIP kernels."""
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local checkpoint directory (this build never downloads)")
        if torch_dtype not in (None, torch.bfloat16, "bfloat16", "auto"):
            raise ValueError("the MI355X path computes in bf16 (torch_dtype=torch.bfloat16)")
        cfg = O3VConfig.from_pretrained(path)
        w = DeviceWeights(cfg, getter_from_safetensors_dir(path), device, fp8_decode=bool(fp8_decode or fp8_prefill))
        model = cls(cfg, O3VEngine(cfg, w))
        # opt-in W8A8 prefill / log-prob pass on the fp8 matrix cores (BASELINE config #5); the bf16 path is the default
        model.engine.fp8_prefill = bool(fp8_prefill)
        gc = load_generation_config(path)
        if gc:
            model.generation_config = GenerationConfigLike(**gc)
        return model

    @classmethod
    def from_state_dict(cls, cfg_dict: dict, state_dict, device="cuda"):
        cfg = O3VConfig.from_dict(cfg_dict)
        return cls(cfg, O3VEngine(cfg, DeviceWeights(cfg, getter_from_dict(state_dict), device)))

    # ---- nn.Module-ish surface the trainer touches
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training (backward / optimizer) is out of scope of the generate path")
        return self

    def to(self, *_, **__):
        return self

    @property
    def device(self):
        return self.engine.dev

    def manual_seed(self, seed: int):
        self._seed = int(seed)

    # ---- generate
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None,
                 pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None, generation_config=None, **kw):
        """`unwrapped_model.generate(**prompt_inputs, generation_config=...)` (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:581-582)
        over either processor output: frames as images (`pixel_values` / `image_grid_thw`, :540-548) or the native video branch
        (`pixel_values_videos` / `video_grid_thw` / `second_per_grid_ts` with <|video_pad|> prompts, :555-564), or both."""
        ids = torch.as_tensor(input_ids).cpu().numpy()
        pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw = self._route_video(
            ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw)
        r = resolve_generation_config(generation_config, self.generation_config, kw, self.generation_config_mode)
        G, T, do_sample = int(r["num_return_sequences"]), int(r["max_new_tokens"]), bool(r["do_sample"])
        eos = r["eos_token_id"]
        eos = [] if eos is None else ([int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)])
        pad = r["pad_token_id"]
        if pad is None:
            if not eos:
                raise ValueError("generate needs a pad_token_id (or an eos_token_id to fall back on)")
            pad = eos[0]          # TF:generation/utils.py _prepare_special_tokens: pad defaults to the first eos id
        pad = int(pad)
        top_k = int(r["top_k"] or 0) if do_sample else 0
        common = dict(max_new_tokens=T, eos_token_ids=eos, pad_token_id=pad, repetition_penalty=float(r["repetition_penalty"]),
                      do_sample=do_sample, temperature=float(r["temperature"] or 1.0), top_p=float(r["top_p"] or 1.0),
                      top_k=top_k, seed=self._seed)
        self._seed += 1
        B = ids.shape[0]
        mask = None if attention_mask is None else torch.as_tensor(attention_mask).cpu().numpy()
        per_prompt = self._split_inputs(ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        rows = []
        if G > MAX_ROWS:
            raise ValueError(f"num_return_sequences={G} > {MAX_ROWS}")
        step = max(1, MAX_ROWS // G)
        # first global completion index of this call: a rank that decodes rows row0..row0+G-1 of a larger group keys its
        # sampler by those indices, so the group is the same whichever ranks produced its rows (partitioning B)
        row0 = int(kw.get("row_id_offset", getattr(generation_config, "row_id_offset", 0) or 0))
        self.engine.position_mode = self.position_mode
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            out = self.engine.generate(ids[b0:b1], None if mask is None else mask[b0:b1], num_return_sequences=G,
                                       row_ids=list(range(row0 + b0 * G, row0 + b1 * G)), return_margins=False,
                                       **self._cat_inputs(per_prompt[b0:b1]), **common)
            rows.append(out.sequences)
        L = max(r.shape[1] for r in rows)
        rows = [torch.nn.functional.pad(r, (0, L - r.shape[1]), value=pad) for r in rows]
        return torch.cat(rows, dim=0)

    def _route_video(self, ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw):
        """Video tensors behind <|video_pad|> placeholders run the native video path.  Video tensors whose prompt marks the frames
        with IMAGE placeholders (no <|video_pad|> at all, every grid row t = 1) are the frames-as-images layout under another
        argument name and take the image path."""
        if pixel_values_videos is None:
            if (ids == self.o3v_config.video_token_id).any():
                raise ValueError("Video features and video tokens do not match, tokens: "
                                 f"{int((ids == self.o3v_config.video_token_id).sum())}, features: 0")
            return pixel_values, image_grid_thw, None, None
        if video_grid_thw is None:
            raise ValueError("pixel_values_videos needs video_grid_thw")
        if pixel_values is None and not (ids == self.o3v_config.video_token_id).any():
            g = torch.as_tensor(video_grid_thw).cpu().numpy().reshape(-1, 3)
            if (g[:, 0] == 1).all():
                return pixel_values_videos, g, None, None
        return pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw

    def _split_inputs(self, ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts):
        """Per prompt: dict(pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts) -- placeholders
        of each modality consume that modality's grid rows in order."""
        cfg = self.o3v_config
        np_grid = lambda g: None if g is None else torch.as_tensor(g).cpu().numpy().reshape(-1, 3)
        img = self._split_pixels(ids, pixel_values, np_grid(image_grid_thw), cfg.image_token_id)
        vid = self._split_pixels(ids, pixel_values_videos, np_grid(video_grid_thw), cfg.video_token_id)
        spg = None if second_per_grid_ts is None else [float(v) for v in torch.as_tensor(second_per_grid_ts).reshape(-1).tolist()]
        out, vi = [], 0
        for (pv, gr), (pvv, vgr) in zip(img, vid):
            nv = 0 if vgr is None else len(vgr)
            # Qwen3-VL counts one grid row per VIDEO although its prompt holds one placeholder run per temporal patch
            out.append(dict(pixel_values=pv, image_grid_thw=gr, pixel_values_videos=pvv, video_grid_thw=vgr,
                            second_per_grid_ts=None if spg is None or not nv else spg[vi:vi + nv]))
            vi += nv
        return out

    @staticmethod
    def _cat_inputs(parts):
        kw = {}
        for pk, gk in (("pixel_values", "image_grid_thw"), ("pixel_values_videos", "video_grid_thw")):
            pvs = [p[pk] for p in parts if p[pk] is not None and len(p[pk])]
            if pvs:
                kw[pk] = torch.cat(pvs, dim=0)
                kw[gk] = np.concatenate([p[gk] for p in parts if p[pk] is not None and len(p[pk])], axis=0)
        if "pixel_values_videos" in kw:
            sp = [p["second_per_grid_ts"] for p in parts if p["pixel_values_videos"] is not None and len(p["pixel_values_videos"])]
            if all(x is not None for x in sp):
                kw["second_per_grid_ts"] = [v for x in sp for v in x]
        return kw

    def _split_pixels(self, ids, pixel_values, grid, token_id):
        """Pixel rows / grid rows belonging to each prompt (placeholders of `token_id` are consumed in order)."""
        B = ids.shape[0]
        if pixel_values is None or grid is None:
            return [(None, None)] * B
        unit = self.o3v_config.vision.merge_unit
        tok_per_img = (grid[:, 0] * grid[:, 1] * grid[:, 2]) // unit
        n_tok = (ids == token_id).sum(axis=1)
        out, gi, prow = [], 0, 0
        pv = torch.as_tensor(pixel_values)
        for b in range(B):
            need, g0 = int(n_tok[b]), gi
            got = 0
            while got < need:
                if gi >= len(grid):
                    raise ValueError("Image features and image tokens do not match")
                got += int(tok_per_img[gi])
                gi += 1
            if got != need:
                raise ValueError("Image features and image tokens do not match")
            rows_n = int((grid[g0:gi, 0] * grid[g0:gi, 1] * grid[g0:gi, 2]).sum())
            out.append((pv[prow:prow + rows_n], grid[g0:gi]))
            prow += rows_n
        if gi != len(grid):
            raise ValueError("Image features and image tokens do not match")
        return out

    # ---- forward (logits), as _get_per_token_logps calls it
    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, pixel_values=None, image_grid_thw=None, pixel_values_videos=None,
                 video_grid_thw=None, second_per_grid_ts=None, **kw):
        ids = torch.as_tensor(input_ids).cpu().numpy()
        pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw = self._route_video(
            ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw)
        B = ids.shape[0]
        mask = None if attention_mask is None else torch.as_tensor(attention_mask).cpu().numpy()
        parts = self._split_inputs(ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        self.engine.position_mode = self.position_mode
        outs = []
        for b in range(B):  # one prompt at a time keeps the [L, V] logits slab bounded (7B: 1.4 GB per 4.6k tokens)
            outs.append(self.engine.forward_logits(ids[b:b + 1], None if mask is None else mask[b:b + 1], **self._cat_inputs(parts[b:b + 1])))
        return SimpleNamespace(logits=torch.cat(outs, dim=0))

    forward = __call__

    @torch.no_grad()
    def per_token_logps(self, input_ids, attention_mask=None, pixel_values=None, image_grid_thw=None, pixel_values_videos=None,
                        video_grid_thw=None, second_per_grid_ts=None):
        """Fast path for R:grpo_trainer.py:371-384: never keeps more than one row of logits alive."""
        ids = torch.as_tensor(input_ids)
        ids_np = ids.cpu().numpy()
        out = []
        mask = None if attention_mask is None else torch.as_tensor(attention_mask)
        pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw = self._route_video(
            ids_np, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw)
        parts = self._split_inputs(ids_np, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        self.engine.position_mode = self.position_mode
        for b in range(ids.shape[0]):
            lg = self.engine.forward_logits(ids[b:b + 1], None if mask is None else mask[b:b + 1], **self._cat_inputs(parts[b:b + 1]))
            out.append(self.engine.per_token_logps(lg, ids[b:b + 1]))
        return torch.cat(out, dim=0)

    @torch.no_grad()
    def completion_logps(self, prompt_ids, prompt_mask, completion_ids, pixel_values=None, image_grid_thw=None,
                         pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None):
        """`_get_per_token_logps(model, cat([prompt]*G, completions), ...)[:, prompt_length-1:]` (R:grpo_trainer.py:371-384,
        :612-613) for the G completions of one prompt, f32 [G, T]: one ViT pass, one prompt prefill, logits only where kept
        (see O3VEngine.completion_logps).  The trainer drops `second_per_grid_ts` before this pass (R:…:608-609): None = 1 s."""
        ids = torch.as_tensor(prompt_ids).cpu().numpy().reshape(1, -1)
        pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw = self._route_video(
            ids, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw)
        self.engine.position_mode = self.position_mode
        return self.engine.completion_logps(prompt_ids, completion_ids, prompt_mask, pixel_values=pixel_values,
                                            image_grid_thw=image_grid_thw, pixel_values_videos=pixel_values_videos,
                                            video_grid_thw=video_grid_thw, second_per_grid_ts=second_per_grid_ts)


# the north-star text names the Qwen2-VL class; both resolve to the same engine
Qwen2VLForConditionalGeneration = Qwen2_5_VLForConditionalGeneration


class Qwen3VLForConditionalGeneration(Qwen2_5_VLForConditionalGeneration):
    """The same surface over a Qwen3-VL checkpoint (BASELINE config #5's base model, R:README.md:37): `config.json` with
    model_type qwen3_vl selects the Qwen3-VL tower, DeepStack and q/k norm inside the engine; either class name loads either
    family."""
