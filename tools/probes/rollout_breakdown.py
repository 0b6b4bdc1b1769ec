import time, torch, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict
from open_o3_video_amd.engine import O3VEngine
from open_o3_video_amd.weights import DeviceWeights, random_getter
from open_o3_video_amd import rollout as R, hf_api
cfg = O3VConfig.from_dict(qwen25vl_7b_dict()); dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev))
ids = bench.build_prompt(cfg, 32, 120, 4490)
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
px, grid = eng.pixels_from_frames(frames)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    return w
M = hf_api.Qwen2_5_VLForConditionalGeneration
M.generate = timed("generate", M.generate)
M.completion_logps = timed("completion_logps", M.completion_logps)
R.gspo_loss = timed("gspo_loss", R.gspo_loss)
og = eng.generate
def g2(*a, **k):
    k["sync_timings"] = True
    out = og(*a, **k)
    for kk in ("vit_ms", "prefill_ms", "decode_ms"): T["gen." + kk] = T.get("gen." + kk, 0) + out.timings[kk] / 1e3
    return out
eng.generate = g2
r = bench.rollout_leg(cfg, eng, ids, px, grid, None, dev, G=8, T=768, steps=2, warmup=1)
print(json.dumps({k: r[k] for k in ("tokens_per_s", "ms_per_step")}), {k: round(v / 3 * 1e3, 1) for k, v in T.items()})
