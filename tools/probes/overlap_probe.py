import sys, time, torch
sys.path.insert(0, '/root/repo')
from bench import build_prompt
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict
from open_o3_video_amd.engine import O3VEngine
from open_o3_video_amd.weights import DeviceWeights, random_getter
cfg = O3VConfig.from_dict(qwen25vl_7b_dict()); dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False))
tpf = (224 // 28) * (420 // 28); ids = build_prompt(cfg, 32, tpf, 4490)
g = torch.Generator(device=dev).manual_seed(1)
fa = torch.randint(0, 256, (32, 3, 224, 420), generator=g, dtype=torch.uint8, device=dev)
fb = torch.randint(0, 256, (32, 3, 224, 420), generator=g, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
def prefill_only(fr):
    px, grid = eng.pixels_from_frames(fr); vis = eng.vit_forward(px, grid)
    return eng.forward_logits([ids], None, vis_embeds=vis, image_grid_thw=grid)[:, -1]
kw = dict(max_new_tokens=512, repetition_penalty=1.05, return_margins=False)
eng.generate([ids], None, frames=fa, **kw); prefill_only(fb); torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
seq = t(lambda: (eng.generate([ids], None, frames=fa, **kw), prefill_only(fb)))
def overlapped():
    eng.generate([ids], None, frames=fa, **kw)
    with torch.cuda.stream(side):
        prefill_only(fb)
ov = t(overlapped)
gen_only = t(lambda: eng.generate([ids], None, frames=fa, **kw))
pre_only = t(lambda: prefill_only(fb))
print(f"generate alone {gen_only:.1f} ms, ViT+prefill alone {pre_only:.1f} ms, back to back {seq:.1f} ms, overlapped on a side stream {ov:.1f} ms")
