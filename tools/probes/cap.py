import torch
from open_o3_video_amd import _lib
torch.cuda.init()
l=_lib.load()
print("cap", [l.o3v_decode_attn_block_capacity(q, wb) for q, wb in ((3584, 2), (2048, 2), (4096, 2), (1792, 2), (3584, 1), (2048, 1))])
print(torch.cuda.get_device_properties(0))
