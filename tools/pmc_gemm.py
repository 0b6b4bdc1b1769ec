#!/usr/bin/env python3
"""Launch the prefill gate/up GEMM (7B: M=4490, N=37888, K=3584, SwiGLU epilogue) for counter collection:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out -- python3 tools/pmc_gemm.py
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs) (the gfx94x MfmaUtil formula)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_o3_video_amd import _lib  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 4490, 37888, 3584
a = torch.empty(M, K, dtype=torch.bfloat16, device=dev).uniform_(-1, 1, generator=g)
w = torch.empty(N, K, dtype=torch.bfloat16, device=dev).uniform_(-1, 1, generator=g)
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for tile in (257, 256, 128):   # 257: the phased 256-tile kernel (csrc/o3v_gemm8p.hip)
    for rep in range(4):
        _lib.call("o3v_gemm_bf16_tile", P(a), P(w), None, None, P(out), M, N, K, K, K, N // 2, 0, 3, tile, st)
torch.cuda.synchronize()
print("done: 4 launches per kernel,", 2.0 * M * N * K / 1e12, "TFLOP each")
