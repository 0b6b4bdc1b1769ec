import sys, time, torch
sys.path.insert(0, "/root/repo")
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict
from open_o3_video_amd.engine import O3VEngine
from open_o3_video_amd.weights import DeviceWeights, random_getter
cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
dev = torch.device("cuda")
eng = O3VEngine(cfg, DeviceWeights(cfg, random_getter(cfg, 1234, dev), dev, batched_decode=False))
frames = torch.randint(0, 256, (32, 3, 224, 420), dtype=torch.uint8, device=dev)
px, grid = eng.pixels_from_frames(frames)
for tile in (0, 128, 256, 0):
    eng.w.vit.gemm_tile = tile
    for _ in range(2):
        eng.vit_forward(px, grid)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.vit_forward(px, grid)
    e1.record()
    torch.cuda.synchronize()
    print("tile", tile, "ViT ms", round(e0.elapsed_time(e1) / 5, 2), flush=True)
