import torch
from open_o3_video_amd import _lib
torch.cuda.init()
l=_lib.load()
print("cap", [l.o3v_decode_attn_block_capacity(h, q) for h, q in ((3584, 3584), (2048, 2048), (4096, 4096), (896, 1792))])
print(torch.cuda.get_device_properties(0))
