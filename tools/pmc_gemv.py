#!/usr/bin/env python3
"""Launch the dominant decode kernel (RMSNorm + gate/up GEMV) over all layers of a 7B-dims engine, for counter
collection:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_gemv.py
(one --pmc pass per counter; FETCH_SIZE is doubled afterwards as MI355X_MICROARCH.md section HBM prescribes)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_o3_video_amd import _lib  # noqa: E402
from open_o3_video_amd.config import O3VConfig, qwen25vl_7b_dict  # noqa: E402

cfg = O3VConfig.from_dict(qwen25vl_7b_dict())
tc = cfg.text
H, I, L = tc.hidden_size, tc.inter_pad, tc.num_hidden_layers
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
ws = [torch.empty(2 * I, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02, generator=g) for _ in range(L)]
nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
x = torch.randn(1, H, device=dev).to(torch.bfloat16)
out = torch.empty(1, I, dtype=torch.bfloat16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    for l in range(L):
        _lib.call("o3v_gemv_norm_bf16", C.c_void_p(x.data_ptr()), C.c_void_p(nw.data_ptr()), 1e-6, C.c_void_p(ws[l].data_ptr()),
                  None, None, C.c_void_p(out.data_ptr()), 1, 2 * I, H, H, H, I, 0, _lib.EPI_SWIGLU, st)
torch.cuda.synchronize()
print("done", 2 * L, "launches of", 2 * I * H * 2, "weight bytes")
