#!/usr/bin/env python3
"""M=8 (group rollout) linear layers: weight-streaming rate of the three code paths on 7B shapes, cold weights."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_o3_video_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
H, I = 3584, 18944
shapes = {"gate_up": (2 * I, H, 3), "down": (H, I, 1), "o_proj": (H, H, 1), "qkv": (4608, H, 0), "lm_head": (152064, H, 0)}
g = torch.Generator(device=dev).manual_seed(0)
for M in (2, 8):
    for name, (N, K, epi) in shapes.items():
        nl = 4 if name == "lm_head" else 28
        ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02, generator=g) for _ in range(nl)]
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        No = N // 2 if epi == 3 else N
        out = torch.empty(M, No, dtype=torch.bfloat16, device=dev)

        from open_o3_video_amd.weights import pack_mfma_fragments
        wps = [pack_mfma_fragments(w) for w in ws]
        nw = torch.ones(K, dtype=torch.bfloat16, device=dev)
        norm = name in ("gate_up", "qkv", "lm_head")

        def run(kind):
            for w, wp in zip(ws, wps):
                if kind == "rowmajor":
                    _lib.call("o3v_linear_decode", P(x), P(nw) if norm else None, 1e-6, P(w), None, None, P(res), P(out), M, N, K, K, No, N, epi, st)
                elif kind == "packed":
                    _lib.call("o3v_linear_decode", P(x), P(nw) if norm else None, 1e-6, P(w), P(wp), None, P(res), P(out), M, N, K, K, No, N, epi, st)
                else:
                    _lib.call("o3v_gemm_bf16", P(x), P(w), None, P(res), P(out), M, N, K, K, K, No, N, epi, st)
        line = f"M={M} {name:8s} {N * K * 2 / 1e6:7.1f} MB "
        for kind in ("rowmajor", "packed", "gemm"):
            run(kind)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(kind); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / nl)
            t = sorted(ts)[1]
            line += f" {kind}: {t:7.1f} us ({N * K * 2 / t / 1e6:5.2f} TB/s)"
        print(line, flush=True)
        del ws, wps
