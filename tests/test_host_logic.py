"""CPU tests of the product's host-side logic (open_o3_video_amd/indexing.py, config, weight packing) against the
HF/reference goldens and the oracle, and of the C-ABI library: it must load and export every symbol that
include/o3v.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import fixture_models as fm
from open_o3_video_amd import _lib, indexing
from open_o3_video_amd.config import O3VConfig, qwen25vl_3b_dict, qwen25vl_7b_dict
from oracle import index_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vision_plan_matches_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_vit_index.npz"))
    for t in sorted({k.split("_")[0] for k in g.files}):
        wi, cu_win, cu_full, pos = indexing.vision_plan(g[f"{t}_grid"])
        assert np.array_equal(wi, g[f"{t}_window_index"]), t
        assert np.array_equal(cu_win, g[f"{t}_cu_window"]), t
        assert np.array_equal(cu_full, g[f"{t}_cu_full"]), t
        assert np.array_equal(pos, g[f"{t}_pos"]), t


def test_canonical_shapes_from_survey():
    # TRAIN-RES 16x30 -> 8 windows (6x64, 2x48); EVAL-RES 26x46 -> 24 windows
    _, cu, _, _ = indexing.vision_plan([[1, 16, 30]])
    assert sorted(np.diff(cu).tolist()) == [48, 48] + [64] * 6
    _, cu, _, _ = indexing.vision_plan([[1, 26, 46]])
    lens = np.diff(cu).tolist()
    assert len(lens) == 24 and sorted(set(lens)) == [12, 16, 48, 64]


def test_rope_index_matches_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_rope_index.npz"))
    cfg = fm.tiny_config()
    for t in sorted({k.split("_")[0] for k in g.files}):
        pos, delta = indexing.rope_index(g[f"{t}_ids"], g[f"{t}_mask"], g[f"{t}_grid"], cfg["image_token_id"])
        assert np.array_equal(pos, g[f"{t}_pos"]), t
        assert np.array_equal(delta.reshape(-1, 1), g[f"{t}_delta"]), t


def test_rope_index_errors():
    cfg = fm.tiny_config()
    ids = fm.make_prompt(cfg, [(1, 4, 6)])
    with pytest.raises(ValueError):
        indexing.rope_index([ids], None, [[1, 4, 8]], cfg["image_token_id"])       # wrong grid
    with pytest.raises(ValueError):
        indexing.rope_index([ids], None, [[1, 4, 6], [1, 4, 6]], cfg["image_token_id"])  # unused grid


def test_decode_positions_follow_hf_rule():
    mask = np.asarray([[0, 0, 1, 1, 1], [1, 1, 1, 1, 1]])
    d = indexing.decode_positions(mask, np.asarray([-2, 0]), 3)
    assert d.shape == (3, 2, 3)
    assert d[0].tolist() == [[1, 2, 3], [5, 6, 7]] and np.array_equal(d[0], d[1]) and np.array_equal(d[0], d[2])


def test_tiles():
    t = indexing.segment_tiles(np.asarray([0, 12, 76, 206]))
    assert t.tolist() == [[0, 12, 0, 12, -1, 0, 0, 0], [12, 64, 12, 64, -1, 0, 0, 0], [76, 64, 76, 130, -1, 0, 0, 0],
                          [140, 64, 76, 130, -1, 0, 0, 0], [204, 2, 76, 130, -1, 0, 0, 0]]
    # causal tiles come longest first (keys a tile walks = causal_off + rows - left padding): same set, heavy ones dispatched first
    p = indexing.prefill_tiles(2, 100, [0, 70], 64)
    assert p.tolist() == [[64, 36, 0, 100, 64, 0, 0, 0], [0, 64, 0, 100, 0, 0, 0, 0], [164, 36, 0, 100, 64, 70, 1, 0]]
    assert indexing.prefill_tiles(1, 300, [130]).tolist() == [[256, 44, 0, 300, 256, 130, 0, 0], [128, 128, 0, 300, 128, 130, 0, 0]]
    assert indexing.segment_tiles(np.asarray([0])).shape == (0, 8)
    # suffix pass behind a cached prompt prefix of 1000 tokens: queries are rows 0..149, keys slots 0..1149
    assert indexing.prefill_tiles(1, 150, [0], past=1000).tolist() == [[128, 22, 0, 1150, 1128, 0, 0, 0], [0, 128, 0, 1150, 1000, 0, 0, 0]]
    big = indexing.prefill_tiles(1, 4490, [0])
    work = big[:, 4] + big[:, 1]
    assert (np.diff(work) <= 0).all() and sorted(big[:, 0].tolist()) == list(range(0, 4490, 128))


def test_embed_source_rows():
    src, n, _ = indexing.embed_source_rows([[5, 500, 500, 7], [500, 1, 2, 500]], 500)
    assert n == 4 and src.tolist() == [5, -1, -2, 7, -3, 1, 2, -4]


def test_config_parsing():
    c = O3VConfig.from_dict(qwen25vl_7b_dict())
    assert (c.vision.head_dim, c.vision.inter_pad, c.vision.patch_k, c.vision.patch_k_pad) == (80, 3456, 1176, 1216)
    assert (c.text.head_dim, c.text.num_attention_heads // c.text.num_key_value_heads) == (128, 7)
    c3 = O3VConfig.from_dict(qwen25vl_3b_dict())
    assert c3.text.tie_word_embeddings and c3.text.head_dim == 128
    # hub-style flat config (text fields at top level, rope_scaling.mrope_section)
    flat = dict(qwen25vl_7b_dict()["text_config"], vision_config=qwen25vl_7b_dict()["vision_config"],
                rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]})
    flat.pop("mrope_section")
    cf = O3VConfig.from_dict(flat)
    assert cf.text.mrope_section == [16, 24, 24] and cf.text.hidden_size == 3584
    t = O3VConfig.from_dict(fm.tiny_config())
    assert t.image_token_id == 500 and t.pad_token_id == 511


def test_pack_gate_up_layout():
    from open_o3_video_amd.weights import pack_gate_up, pad_cols
    g = torch.arange(40 * 3, dtype=torch.float32).reshape(40, 3)
    u = -g
    p = pack_gate_up(g, u, 64)
    assert p.shape == (128, 3)
    for j in range(64):
        grow = (j // 16) * 32 + j % 16
        assert torch.equal(p[grow], g[j] if j < 40 else torch.zeros(3))
        assert torch.equal(p[grow + 16], u[j] if j < 40 else torch.zeros(3))
    b = pack_gate_up(torch.arange(40.0), torch.arange(40.0) + 100, 64)
    assert b.shape == (128,) and b[16].item() == 100 and b[32].item() == 16
    assert pad_cols(torch.ones(2, 5), 8)[:, 5:].abs().sum() == 0


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "o3v.h")).read()
    declared = set(re.findall(r"\b(o3v_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"o3v_stream_t"}
    assert len(declared) >= 24
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/o3v.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert _lib.load().o3v_abi_version() == 6


def test_c_abi_argument_errors_without_gpu():
    """Error behaviour of the boundary: bad arguments are rejected before any launch."""
    lib = _lib.load()
    assert lib.o3v_rmsnorm(None, None, None, 1, 64, 64, 64, 1e-6, None) == _lib.ERR_ARG
    assert lib.o3v_gemm_bf16(None, None, None, None, None, 1, 1, 64, 64, 64, 1, 0, 0, None) == _lib.ERR_ARG
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.o3v_gemm_bf16(p, p, None, None, p, 4, 8, 100, 100, 100, 8, 0, 0, None) == _lib.ERR_SHAPE
    assert lib.o3v_gemv_bf16(p, p, None, None, p, 9, 8, 64, 64, 64, 8, 0, 0, None) == _lib.ERR_SHAPE
    assert lib.o3v_gemm_bf16(p, p, None, None, p, 0, 8, 64, 64, 64, 8, 0, 0, None) == _lib.OK  # empty input
    assert lib.o3v_rmsnorm(p, p, p, 0, 64, 64, 64, 1e-6, None) == _lib.OK
    assert lib.o3v_attn_tiles(p, p, p, p, p, 0, 64, 4, 1, 80, 240, 240, 80, 0, 240, 80, 0, 80, 0.1, None) == _lib.OK
    assert lib.o3v_attn_tiles(p, p, p, p, p, 1, 64, 4, 1, 48, 240, 240, 80, 0, 240, 80, 0, 80, 0.1, None) == _lib.ERR_SHAPE
    assert lib.o3v_attn_tiles(p, p, p, p, p, 1, 96, 4, 1, 80, 240, 240, 80, 0, 240, 80, 0, 80, 0.1, None) == _lib.ERR_ARG
    # this round's entries: shared prompt entries, 17..32 rows, fp8 rows, Qwen3-VL ops, the one-launch block
    assert lib.o3v_attn_tiles_prefix(p, p, p, p, None, 0, 0, 5, 1, p, p, 1, 64, 4, 1, 128, 512, 128, 0, 0, 128, 0, 0, 512, 0.1, None) == _lib.ERR_ARG
    assert lib.o3v_attn_tiles_prefix(p, p, p, p, p, 0, 0, 0, 1, p, p, 1, 64, 4, 1, 128, 512, 128, 0, 0, 128, 0, 0, 512, 0.1, None) == _lib.ERR_ARG
    f32p = ctypes.cast(buf, ctypes.POINTER(ctypes.c_float))
    assert lib.o3v_attn_decode_group_prefix(p, p, p, p, p, 10, 3, p, f32p, f32p, None, 8, 4, 8, 2, 128, 10, 12, 8, 2, 0.1, None) == _lib.ERR_ARG  # 4 rows per group do not divide 3 rows per prompt
    assert lib.o3v_attn_decode_group_prefix(p, p, p, p, p, 10, 4, p, f32p, f32p, None, 8, 4, 8, 2, 64, 10, 12, 8, 2, 0.1, None) == _lib.ERR_SHAPE  # head_dim 64
    assert lib.o3v_linear_decode(p, p, 1e-6, p, p, None, None, p, 17, 64, 64, 64, 64, 0, _lib.EPI_NONE, None) == _lib.ERR_SHAPE  # fused norm above 16 rows
    assert lib.o3v_linear_decode(p, None, 0.0, p, None, None, None, p, 33, 64, 64, 64, 64, 0, _lib.EPI_NONE, None) == _lib.ERR_SHAPE
    assert lib.o3v_linear_decode_fp8_rows(p, p, f32p, None, None, p, 3, 64, 64, 64, 64, 0, _lib.EPI_NONE, None) == _lib.ERR_ARG
    assert lib.o3v_linear_decode_fp8_rows(p, p, f32p, None, None, p, 8, 64, 96, 96, 64, 0, _lib.EPI_NONE, None) == _lib.ERR_SHAPE  # K % 64
    assert lib.o3v_linear_decode_fp8_rows(p, p, f32p, None, None, p, 8, 64, 64, 64, 64, 0, _lib.EPI_RESIDUAL, None) == _lib.ERR_ARG  # no residual
    assert lib.o3v_layernorm(p, p, None, p, 1, 64, 64, 64, 1e-6, None) == _lib.ERR_ARG
    assert lib.o3v_qkv_norm_rope_cache(p, p, p, 1e-6, p, p, p, p, p, 0, 1, 1, 4, 2, 96, 8, 1, 0, None) == _lib.ERR_SHAPE  # 6 lanes per head
    assert lib.o3v_add_rows(p, p, p, p, 0, 64, None) == _lib.OK
    assert lib.o3v_patchify_ps(p, 1, p, 1, 48, 64, 1536, 16, ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)),
                               ctypes.cast(buf, ctypes.POINTER(ctypes.c_float)), None) == _lib.ERR_ARG                # 48 is not a multiple of 32
    assert lib.o3v_decode_attn_block_qknorm(p, p, 1e-6, p, None, p, None, None, p, p, p, p, p, p, p, p, f32p, f32p, None, 64, 2, 1, 128, 0, 4, 1,
                                            0, 1, 0.1, ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint32)), 1, None) == _lib.ERR_ARG  # no q_norm
    assert lib.o3v_vit3_forward(None, p, 4, p, f32p, f32p, p, 1, p, 64, p, None, None) == _lib.ERR_ARG
    opts = _lib.PrefillOpts(kprefix=ctypes.addressof(buf), vprefix=0, prefix_len=4, prefix_cap=4, rows_per_prefix=1)
    d = _lib.LlmDesc(hidden=64, layers=0, heads=2, kv_heads=1, head_dim=32, inter=64, vocab=8, rms_eps=1e-6)
    assert lib.o3v_llm_prefill_ex(ctypes.byref(d), p, p, p, p, 1, 128, p, p, 1, 4, 4, 8, ctypes.byref(opts), p, 64, None) == _lib.ERR_ARG  # K without V


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from open_o3_video_amd.engine import O3VEngine
    with pytest.raises(Exception):
        O3VEngine(O3VConfig.from_dict(fm.tiny_config()), None)


def test_generation_config_resolution_matches_hf(golden_dir, tmp_path):
    """hf_api.resolve_generation_config == GenerationMixin._prepare_generation_config (transformers 5.15, golden G8b) for the
    trainer's GenerationConfig (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:306-313: top_k / eos / penalty unset) against
    three checkpoint generation configs, with and without generate() kwargs; and generation_config.json is what is loaded."""
    import json
    from open_o3_video_amd.hf_api import GenerationConfigLike, load_generation_config, resolve_generation_config
    g = np.load(os.path.join(golden_dir, "g8b_top_k.npz"))
    cases = json.loads(bytes(g["gen_config_cases"]).decode())
    passed = GenerationConfigLike(**cases["trainer"])
    for name, mg in cases["models"].items():
        d = tmp_path / name
        d.mkdir()
        if mg:
            (d / "generation_config.json").write_text(json.dumps(dict(mg, transformers_version="4.37.0")))
        loaded = load_generation_config(str(d))
        assert loaded == mg
        model_gc = GenerationConfigLike(**loaded)
        for suffix, kw in (("", {}), ("+kwargs", {"top_k": 7, "eos_token_id": 3})):
            want = cases["resolved"][name + suffix]
            got = resolve_generation_config(passed, model_gc, kw, mode="tf5")
            for k, v in want.items():
                assert got[k] == v, (name + suffix, k, got[k], v)
    # the unset trainer fields fall back to HF's global defaults when the checkpoint says nothing
    r = resolve_generation_config(passed, GenerationConfigLike(), {}, mode="tf5")
    assert r["top_k"] == 50 and r["repetition_penalty"] == 1.0 and r["eos_token_id"] is None


def test_generation_config_resolution_pinned_library():
    """mode="pinned" (the default; transformers @336dc69d of R:setup.sh:4): a passed GenerationConfig keeps its own
    constructor defaults (top_k 50, penalty 1.0) and inherits ONLY eos / pad / bos from the checkpoint -- the trainer's
    config (R:grpo_trainer.py:306-313) must not pick up Qwen2.5-VL's generation_config.json top_k 1 / penalty 1.05.  With no
    passed config the checkpoint's config is the base.  Restated behaviour (the pinned commit is not installed): unpinned."""
    from open_o3_video_amd.hf_api import GenerationConfigLike, resolve_generation_config
    trainer = GenerationConfigLike(max_new_tokens=768, do_sample=True, top_p=0.95, temperature=1, num_return_sequences=4,
                                   pad_token_id=151643)
    ckpt = GenerationConfigLike(do_sample=True, eos_token_id=[151645, 151643], pad_token_id=151643, bos_token_id=151643,
                                repetition_penalty=1.05, temperature=0.1, top_k=1, top_p=0.001)
    r = resolve_generation_config(trainer, ckpt, {})
    assert r["top_k"] == 50 and r["repetition_penalty"] == 1.0 and r["temperature"] == 1 and r["top_p"] == 0.95
    assert r["eos_token_id"] == [151645, 151643] and r["bos_token_id"] == 151643 and r["num_return_sequences"] == 4
    r = resolve_generation_config(trainer, ckpt, {"top_k": 7, "eos_token_id": 3})
    assert r["top_k"] == 7 and r["eos_token_id"] == 3
    r = resolve_generation_config(None, ckpt, {"max_new_tokens": 5})          # model.generate(**inputs, max_new_tokens=5)
    assert r["top_k"] == 1 and r["repetition_penalty"] == 1.05 and r["max_new_tokens"] == 5 and r["temperature"] == 0.1
    with pytest.raises(ValueError):
        resolve_generation_config(trainer, ckpt, {}, mode="other")


def test_group_rollout_sampling_does_not_follow_checkpoint_generation_config():
    """GroupRollout over a model whose generation_config.json says top_k 1 / penalty 1.05 still samples with the trainer's
    top_k 50 / penalty 1.0, in either resolution mode; eos is inherited from the checkpoint."""
    from open_o3_video_amd.hf_api import GenerationConfigLike, Qwen2_5_VLForConditionalGeneration
    from open_o3_video_amd.rollout import GroupRollout
    seen = {}

    class Eng:
        dev = "cpu"

        def generate(self, ids, mask, **kw):
            seen.update(kw)
            G, T = kw["num_return_sequences"], 3
            row = torch.as_tensor(ids, dtype=torch.int64)
            return types.SimpleNamespace(sequences=torch.cat([row.repeat(G, 1), torch.full((G, T), 7, dtype=torch.int64)], dim=1))

        def completion_logps(self, prompt_ids, completion_ids, mask, **kw):
            return torch.zeros(completion_ids.shape, dtype=torch.float32)

    import types
    cfg = O3VConfig.from_dict(fm.tiny_config())
    for mode in ("pinned", "tf5"):
        seen.clear()
        model = Qwen2_5_VLForConditionalGeneration(cfg, Eng())
        model.generation_config_mode = mode
        model.generation_config = GenerationConfigLike(do_sample=True, eos_token_id=[510, 509], pad_token_id=511,
                                                       repetition_penalty=1.05, temperature=0.1, top_k=1, top_p=0.001)
        ro = GroupRollout(model, [lambda prompts, completions, **kw: [0.0] * len(completions)], lambda c: ["x"] * c.shape[0],
                          eos_token_id=510, pad_token_id=511, num_generations=4, max_completion_length=3)
        ro.step({"input_ids": torch.tensor([[11, 12, 13]]), "attention_mask": torch.ones(1, 3, dtype=torch.int64)}, {"prompt": "p"})
        assert seen["top_k"] == 50 and seen["repetition_penalty"] == 1.0 and seen["temperature"] == 1.0 and seen["top_p"] == 0.95
        assert seen["do_sample"] is True and seen["eos_token_ids"] == [510, 509] and seen["num_return_sequences"] == 4


# ------------------------------------------------------------------------------------------------ Qwen3-VL host logic
def test_qwen3vl_config_parsing():
    from open_o3_video_amd.config import qwen3vl_8b_dict
    import fixture_models_q3 as fq
    c = O3VConfig.from_dict(qwen3vl_8b_dict())
    assert c.arch == "qwen3_vl" and c.vision.patch_size == 16 and c.vision.head_dim == 72 and c.vision.head_dim_pad == 80
    assert c.vision.hidden_pad == 1152 and c.vision.inter_pad == 4352 and c.vision.deepstack_visual_indexes == [8, 16, 24]
    assert c.text.head_dim == 128 and c.text.qk_norm and c.text.mrope_interleaved and not c.text.attention_bias
    assert c.text.mrope_section == [24, 20, 20] and c.text.rope_theta == 5000000.0
    m = O3VConfig.from_dict(fq.medium_q3_config())
    assert m.vision.hidden_pad == 320 and m.vision.head_dim_pad == 80 and m.text.head_dim == 128
    # the HF 5.x layout keeps the rope settings under rope_parameters
    d = fq.medium_q3_config()
    tc = d["text_config"]
    tc["rope_parameters"] = {"rope_type": "default", "mrope_section": tc.pop("mrope_section"), "rope_theta": tc.pop("rope_theta"),
                             "mrope_interleaved": True}
    m2 = O3VConfig.from_dict(d)
    assert m2.text.mrope_section == [24, 20, 20] and m2.text.rope_theta == 5000000.0 and m2.text.mrope_interleaved
    # the hub's config.json layout: rope settings under text_config.rope_scaling, eos / bos inside text_config
    d = qwen3vl_8b_dict()
    tc = d["text_config"]
    tc["rope_scaling"] = {"mrope_interleaved": True, "mrope_section": tc.pop("mrope_section"), "rope_type": "default"}
    tc["eos_token_id"], tc["bos_token_id"] = d.pop("eos_token_id"), 151643
    d.pop("pad_token_id")
    h = O3VConfig.from_dict(d)
    assert h.text.mrope_section == [24, 20, 20] and h.text.mrope_interleaved and h.eos_token_id == 151645 and h.pad_token_id == 151643
    q25 = O3VConfig.from_dict(qwen25vl_7b_dict())
    assert q25.arch == "qwen2_5_vl" and not q25.text.qk_norm and q25.text.attention_bias and not q25.text.mrope_interleaved
    bad = qwen3vl_8b_dict()
    bad["text_config"]["attention_bias"] = True
    with pytest.raises(ValueError):
        O3VConfig.from_dict(bad)


def test_qwen3vl_pos_embed_taps_and_interleaved_mrope_match_oracle():
    """indexing.pos_embed_taps / mrope_axis_table_interleaved (product, numpy) == oracle/model_ref_q3 (torch), which is pinned to
    transformers' Qwen3-VL by goldens G12 / G13: table rows and fp32 weights bit for bit, rotary angles bit for bit."""
    from oracle import model_ref_q3 as q3
    for grid, side in [(((1, 4, 6), (2, 8, 4)), 8), (((3, 14, 26),), 48), (((1, 2, 2),), 12), (((2, 16, 10), (1, 6, 6)), 16)]:
        i1, w1 = indexing.pos_embed_taps(grid, side, 2)
        i2, w2 = q3.pos_embed_taps(list(grid), side, 2)
        assert np.array_equal(i1, i2.numpy()) and np.array_equal(w1, w2.numpy()), grid
        assert np.allclose(w1.sum(1), 1.0, atol=1e-6) and i1.min() >= 0 and i1.max() < side * side
    for sec, half in (([24, 20, 20], 64), ([6, 5, 5], 16)):
        cfg = {"text_config": {"head_dim": 2 * half, "rope_theta": 5e6, "mrope_section": sec}}
        ax = indexing.mrope_axis_table_interleaved(sec, half)
        pos = torch.from_numpy(np.random.RandomState(1).randint(0, 4000, (3, 2, 9)))
        c, s = q3.interleaved_mrope_cos_sin(cfg, pos, torch.float32)
        inv = 1.0 / (5e6 ** (torch.arange(0, 2 * half, 2, dtype=torch.float) / (2 * half)))
        for b in range(2):
            f = pos.float()[torch.from_numpy(ax).long(), b, :].T * inv
            assert torch.equal(torch.cat([f, f], -1).cos(), c[b]) and torch.equal(torch.cat([f, f], -1).sin(), s[b])
        assert (ax == 0).sum() == sec[0] and (ax == 1).sum() == sec[1] and (ax == 2).sum() == sec[2]


def test_deepstack_rows():
    ids = np.asarray([[5, 9, 9, 9, 7, 9, 9, 3]])
    rows, src = indexing.deepstack_rows(ids, 9)
    assert rows.tolist() == [1, 2, 3, 5, 6] and src.tolist() == [0, 1, 2, 3, 4]
    rows, src = indexing.deepstack_rows(ids, 9, first=3)          # a cached prefix of 3 tokens: later visual rows keep their ordinal
    assert rows.tolist() == [0, 2, 3] and src.tolist() == [2, 3, 4]
    rows, src = indexing.deepstack_rows(np.asarray([[1, 2], [9, 4]]), 9)   # flattened batch rows
    assert rows.tolist() == [2] and src.tolist() == [0]


def test_qwen3vl_vision_head_padding_layout():
    """weights._pack_vision_q3's head layout, restated: heads of 72 stored 80 wide as [36 | 4 zeros | 36 | 4 zeros] keep q.k and
    the rotation pairs (j, j + 36) intact when the kernels pair (j, j + 40)."""
    hd, Dp, heads = 72, 80, 3
    half, halfp = hd // 2, Dp // 2
    j = np.arange(Dp)
    inner = np.where(j < halfp, j, j - halfp)
    src = np.where(j < halfp, inner, inner + half)
    valid = inner < half
    rng = np.random.default_rng(0)
    q, k = rng.standard_normal((heads, hd)), rng.standard_normal((heads, hd))
    qp = np.where(valid, q[:, np.minimum(src, hd - 1)], 0.0)
    kp = np.where(valid, k[:, np.minimum(src, hd - 1)], 0.0)
    assert np.allclose((qp * kp).sum(1), (q * k).sum(1))
    ang = rng.standard_normal(half)
    cos, sin = np.ones(halfp), np.zeros(halfp)
    cos[:half], sin[:half] = np.cos(ang), np.sin(ang)

    def rot(x, c, s, h):       # rotate_half with pairs (j, j + h)
        return np.concatenate([x[:, :h] * c - x[:, h:] * s, x[:, h:] * c + x[:, :h] * s], axis=1)
    want = rot(q, np.cos(ang), np.sin(ang), half)
    got = rot(qp, cos, sin, halfp)
    assert np.allclose(got[:, valid], want[:, src[valid]]) and np.allclose(got[:, ~valid], 0.0)


def test_ctx_handle_owns_descriptor_copies():
    """o3v_ctx_create / destroy (host only, no GPU): the context deep-copies the descriptors and their layer arrays, so the caller's
    structs may go away; accessors hand back what the model-level entries take."""
    import ctypes as C
    lib = _lib.load()
    layers = (_lib.LlmLayerW * 3)()
    for i in range(3):
        layers[i].ln1 = 1000 + i
        layers[i].qkv_w = 2000 + i
    d = _lib.LlmDesc(hidden=64, layers=3, heads=2, kv_heads=1, head_dim=32, inter=128, vocab=100, rms_eps=1e-6, layer=layers, embed=77)
    blocks = (_lib.VitBlockW * 2)()
    blocks[1].qkv_w = 4242
    v = _lib.VitDesc(depth=2, hidden=64, heads=2, blocks=blocks)
    ctx = lib.o3v_ctx_create(C.byref(d), C.byref(v), None)
    assert ctx
    layers[1].ln1 = 0            # the caller's copies change / disappear
    d.hidden = 0
    del blocks, v
    got = lib.o3v_ctx_llm(ctx).contents
    assert got.hidden == 64 and got.layers == 3 and got.embed == 77 and got.layer[1].ln1 == 1001 and got.layer[2].qkv_w == 2002
    assert lib.o3v_ctx_vit(ctx).contents.blocks[1].qkv_w == 4242
    assert not lib.o3v_ctx_vit3(ctx)
    lib.o3v_ctx_destroy(ctx)
    lib.o3v_ctx_destroy(None)
    bad = _lib.LlmDesc(layers=2)              # layers without a layer array
    assert not lib.o3v_ctx_create(C.byref(bad), None, None)


def test_fp8_quantiser_and_fragment_packing():
    """weights.quantize_rows_fp8 (power-of-two row scales, exact bf16 dequantisation) and the two fragment-major packings the
    matrix-core decode kernels read (bf16: [N/16][K/32][64][8], fp8: [N/16][K/64][64][16]): index maps as documented."""
    from open_o3_video_amd.weights import (FP8_MAX, dequantize_rows_fp8, pack_mfma_fragments, pack_mfma_fragments_fp8,
                                           quantize_rows_fp8)
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(32, 128, generator=g) * torch.logspace(-3, 1, 32)[:, None]).to(torch.bfloat16)
    w[5] = 0
    q8, sc = quantize_rows_fp8(w)
    assert q8.dtype == torch.uint8 and sc.dtype == torch.float32 and sc[5] == 1.0
    assert torch.equal(torch.exp2(torch.round(torch.log2(sc))), sc)                      # powers of two
    deq = dequantize_rows_fp8(q8, sc)
    assert torch.equal(deq, deq.to(torch.bfloat16).float())                              # exact in bf16
    assert (deq.abs().amax(1) <= sc * FP8_MAX).all()
    rel = ((deq - w.float()).abs() / w.float().abs().clamp(min=1e-9))[w.float().abs() > sc[:, None] * 2.0 ** -6]
    assert rel.max() <= 2.0 ** -4 + 1e-6                                                 # 3 mantissa bits: half an ulp of 2^-3
    q2, s2 = quantize_rows_fp8(deq.to(torch.bfloat16))
    assert torch.equal(dequantize_rows_fp8(q2, s2), deq)                                 # value-idempotent
    # bf16 fragments: lane l of (row block nb, k-step ks) holds W[nb*16 + (l & 15)][ks*32 + (l >> 4)*8 : +8]
    wi = torch.arange(32 * 128, dtype=torch.int32).view(32, 128)
    p = pack_mfma_fragments(wi).view(2, 4, 64, 8)
    for nb, ks, l in ((0, 0, 0), (1, 3, 37), (0, 2, 63)):
        assert torch.equal(p[nb, ks, l], wi[nb * 16 + (l & 15), ks * 32 + (l >> 4) * 8: ks * 32 + (l >> 4) * 8 + 8])
    # fp8 fragments: lane l of (row block nb, double step t) holds W[nb*16 + (l & 15)][t*64 + (l >> 4)*16 : +16]
    bi = (torch.arange(32 * 128) % 251).to(torch.uint8).view(32, 128)
    p8 = pack_mfma_fragments_fp8(bi).view(2, 2, 64, 16)
    for nb, t, l in ((0, 0, 0), (1, 1, 37), (0, 1, 63)):
        assert torch.equal(p8[nb, t, l], bi[nb * 16 + (l & 15), t * 64 + (l >> 4) * 16: t * 64 + (l >> 4) * 16 + 16])


# ------------------------------------------------------------------------------------------------ native video inputs
def test_rope_index_video_matches_hf(golden_dir):
    """indexing.rope_index(mode="tf5") == get_rope_index of transformers 5.15 on native video groups (golden G5b: Qwen2.5-VL with
    second_per_grid_ts, several videos, video + image, left padding; Qwen3-VL's per-frame split)."""
    import fixture_models_q3 as fq
    g = np.load(os.path.join(golden_dir, "g5b_rope_index_video.npz"))
    tps = int(g["tokens_per_second"][0])
    tags = sorted({k[:-4] for k in g.files if k.endswith("_ids")})
    for t in tags:
        q3 = t.startswith("q3_")
        cfg = fq.tiny_q3_config() if q3 else fm.tiny_config()
        spg = list(g[f"{t}_spg"]) if bool(g[f"{t}_has_spg"][0]) and not q3 else None
        pos, delta = indexing.rope_index(g[f"{t}_ids"], g[f"{t}_mask"], g[f"{t}_igrid"] if len(g[f"{t}_igrid"]) else None,
                                         cfg["image_token_id"], video_grid_thw=g[f"{t}_vgrid"], video_token_id=cfg["video_token_id"],
                                         second_per_grid_ts=spg, tokens_per_second=tps, split_video_frames=q3, mode="tf5")
        assert np.array_equal(pos, g[f"{t}_pos"]), t
        assert np.array_equal(delta, g[f"{t}_delta"].reshape(-1)), t


def test_rope_index_video_pinned_mode():
    """mode="pinned" (transformers @336dc69d / vllm 0.7.2, restated -- parity unpinned): the worked example of that release's
    get_rope_index docstring (3 temporal patches of 2x2 merged tokens, fps 1 -> second_per_grid_t 2.0, tokens_per_second 25:
    temporal ids 0,50,100, text continues at 101), fractional seconds are truncated AFTER the product, and the mode equals "tf5"
    whenever the temporal extent stays inside the spatial one (and always for images)."""
    V, I = 501, 500
    ids = [[V] * 12 + [7, 8, 9, 10, 11]]
    pos, delta = indexing.rope_index(ids, None, None, I, video_grid_thw=[[3, 4, 4]], video_token_id=V, second_per_grid_ts=[2.0],
                                     tokens_per_second=25, mode="pinned")
    assert pos[0, 0, :12].tolist() == [0] * 4 + [50] * 4 + [100] * 4
    assert pos[1, 0, :12].tolist() == [0, 0, 1, 1] * 3 and pos[2, 0, :12].tolist() == [0, 1, 0, 1] * 3
    assert pos[:, 0, 12:].tolist() == [[101, 102, 103, 104, 105]] * 3 and int(delta[0]) == 106 - 17
    # 0.5 s per temporal patch at 2 tokens/s: 0, 1, 2, 3 (tf5 truncates the seconds first: all 0)
    p, _ = indexing.rope_index([[V] * 4 + [9]], None, None, I, video_grid_thw=[[4, 2, 2]], video_token_id=V, second_per_grid_ts=[0.5],
                               tokens_per_second=2, mode="pinned")
    assert p[0, 0].tolist() == [0, 1, 2, 3, 4]
    q, _ = indexing.rope_index([[V] * 4 + [9]], None, None, I, video_grid_thw=[[4, 2, 2]], video_token_id=V, second_per_grid_ts=[0.5],
                               tokens_per_second=2, mode="tf5")
    assert q[0, 0].tolist() == [0, 0, 0, 0, 1]
    # images, and a video whose temporal extent is below its width: both modes agree
    cfg = fm.tiny_config()
    idm = fm.make_prompt_mm(cfg, [("image", (1, 4, 6)), ("video", (2, 4, 12)), ("image", (1, 8, 4))], seed=3)
    kw = dict(video_grid_thw=[[2, 4, 12]], video_token_id=cfg["video_token_id"], second_per_grid_ts=[1.0], tokens_per_second=2)
    a = indexing.rope_index([idm], None, [[1, 4, 6], [1, 8, 4]], cfg["image_token_id"], mode="tf5", **kw)
    b = indexing.rope_index([idm], None, [[1, 4, 6], [1, 8, 4]], cfg["image_token_id"], mode="pinned", **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        indexing.rope_index([idm], None, [[1, 4, 6], [1, 8, 4]], cfg["image_token_id"], mode="x", **kw)
    with pytest.raises(ValueError):       # a video placeholder run without its grid row
        indexing.rope_index([idm], None, [[1, 4, 6], [1, 8, 4]], cfg["image_token_id"], video_token_id=cfg["video_token_id"])


def test_embed_source_rows_with_video():
    """Image placeholders take the first rows of the visual tensor, video placeholders the rows after ALL image rows, whatever
    their order in the prompt (each modality is scattered on its own, TF:1206-1215); DeepStack rows follow the same map."""
    src, ni, nv = indexing.embed_source_rows([[5, 501, 501, 500, 7, 500, 501]], 500, 501)
    assert (ni, nv) == (2, 3) and src.tolist() == [5, -3, -4, -1, 7, -2, -5]
    rows, order = indexing.deepstack_rows([[5, 501, 501, 500, 7, 500, 501]], 500, video_token_id=501)
    assert rows.tolist() == [1, 2, 3, 5, 6] and order.tolist() == [2, 3, 0, 1, 4]
    rows, order = indexing.deepstack_rows([[5, 501, 501, 500, 7, 500, 501]], 500, first=3, video_token_id=501)
    assert rows.tolist() == [0, 2, 3] and order.tolist() == [0, 1, 4]
    src, ni, nv = indexing.embed_source_rows([[5, 500, 501]], 500)          # no video id given: 501 stays a text row
    assert (ni, nv) == (1, 0) and src.tolist() == [5, -1, 501]


def test_vllm_video_placeholder_expansion():
    """One <|video_pad|> per video in the prompt (R:eval/models/model_vllm.py:41-60 via the chat template).  Qwen2.5-VL: t*gh*gw/4
    pads in place; Qwen3-VL: a "<ts seconds><|vision_start|>pads<|vision_end|>" block per temporal patch replacing the template's
    whole <|vision_start|><|video_pad|><|vision_end|> (TF:models/qwen3_vl/processing_qwen3_vl.py:81-107,178-189)."""
    import types
    from open_o3_video_amd.vllm_api import LLM
    llm = LLM.__new__(LLM)
    llm.cfg = types.SimpleNamespace(arch="qwen2_5_vl", vision=types.SimpleNamespace(merge_unit=4))
    llm.tokenizer = types.SimpleNamespace(encode=lambda text, add_special_tokens=False: text)
    ph = llm._video_placeholder((3, 4, 6), {})
    assert ph == "<|video_pad|>" * 18
    out = llm._tokenize("a<|vision_start|><|video_pad|><|vision_end|>b<|image_pad|>c", 1, 2, [ph])
    assert "".join(out) == "a<|vision_start|>" + "<|video_pad|>" * 18 + "<|vision_end|>b<|image_pad|><|image_pad|>c"
    with pytest.raises(ValueError):
        llm._tokenize("a<|video_pad|>b", 0, 0, [])
    with pytest.raises(ValueError):
        llm._tokenize("ab", 0, 0, [ph])
    llm.cfg = types.SimpleNamespace(arch="qwen3_vl", vision=types.SimpleNamespace(merge_unit=4))
    ph3 = llm._video_placeholder((2, 4, 4), {"fps": 2.0, "frames_indices": [0, 4, 8]})       # odd count: last index repeated
    blk = lambda ts: f"<{ts} seconds><|vision_start|>" + "<|video_pad|>" * 4 + "<|vision_end|>"
    assert ph3 == blk("1.0") + blk("4.0")
    out = llm._tokenize("a<|vision_start|><|video_pad|><|vision_end|>b", 0, 0, [ph3])
    assert "".join(out) == "a" + ph3 + "b"
    assert llm._video_placeholder((2, 4, 4), {"n_frames": 4}) == blk("0.0") + blk("0.1")      # no metadata: fps 24, indices 0..3


def test_prompt_identity_key_for_the_generate_to_logps_hand_over():
    """engine._prompt_key (identity of a prompt between the group generate and the policy's log-prob pass): equal bytes give equal
    keys; one flipped bit in a pixel, one changed token, another mask, another rope position, another position mode, a reshaped or
    re-typed visual tensor, a missing tensor all give different keys; lists of frames are keyed by their bytes too."""
    import types
    import numpy as np
    import torch
    from open_o3_video_amd.engine import O3VEngine
    me = types.SimpleNamespace(position_mode="pinned")
    key = lambda ids, mask, pos, tensors, who=me: O3VEngine._prompt_key(who, ids, mask, pos, tensors)
    g = torch.Generator().manual_seed(3)
    ids = np.arange(40, dtype=np.int64).reshape(1, 40)
    mask = np.ones_like(ids)
    pos = np.stack([ids, ids, ids]).astype(np.int64)
    pv = torch.randn(96, 1176, generator=g).to(torch.bfloat16)
    grid = np.asarray([[1, 8, 12]], dtype=np.int64)
    base = key(ids, mask, pos, (("pv", pv), ("grid", grid), ("frames", None)))
    assert base == key(ids.copy(), mask.copy(), pos.copy(), (("pv", pv.clone()), ("grid", grid.copy()), ("frames", None)))
    pv2 = pv.clone()
    pv2.view(torch.int16)[17, 300] ^= 1
    ids2 = ids.copy()
    ids2[0, -1] += 1
    mask2 = mask.copy()
    mask2[0, 0] = 0
    pos2 = pos.copy()
    pos2[1, 0, 5] += 1
    others = [key(ids, mask, pos, (("pv", pv2), ("grid", grid), ("frames", None))),
              key(ids2, mask, pos, (("pv", pv), ("grid", grid), ("frames", None))),
              key(ids, mask2, pos, (("pv", pv), ("grid", grid), ("frames", None))),
              key(ids, mask, pos2, (("pv", pv), ("grid", grid), ("frames", None))),
              key(ids, mask, pos, (("pv", pv), ("grid", grid), ("frames", None)), types.SimpleNamespace(position_mode="tf5")),
              key(ids, mask, pos, (("pv", pv.view(48, 2352)), ("grid", grid), ("frames", None))),
              key(ids, mask, pos, (("pv", pv.view(torch.int16)), ("grid", grid), ("frames", None))),
              key(ids, mask, pos, (("pv", None), ("grid", grid), ("frames", None))),
              key(ids, mask, pos, (("pv", pv), ("grid", np.asarray([[1, 12, 8]], dtype=np.int64)), ("frames", None)))]
    assert all(o != base for o in others) and len(set(others)) == len(others)
    # two rows swapped keep the plain sum and change the position-weighted one
    pv3 = pv.clone()
    pv3[[0, 1]] = pv3[[1, 0]]
    assert key(ids, mask, pos, (("pv", pv3), ("grid", grid), ("frames", None))) != base
    fr = [torch.randint(0, 256, (3, 28, 28), generator=g, dtype=torch.uint8) for _ in range(3)]
    k1 = key(ids, mask, pos, (("frames", fr),))
    fr2 = [f.clone() for f in fr]
    assert k1 == key(ids, mask, pos, (("frames", fr2),))
    fr2[2][1, 5, 5] ^= 4
    assert k1 != key(ids, mask, pos, (("frames", fr2),))
