#!/usr/bin/env python3
"""Throughput of o3v_gemm_bf16 on the prefill / ViT shapes of the 7B benchmark (random operands, L2-cold rotation)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_o3_video_amd import _lib  # noqa: E402

dev = torch.device("cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
shapes = [  # name, M, N(weight rows), K, epi, bias
    ("llm qkv", 4490, 4608, 3584, 0, True), ("llm o+res", 4490, 3584, 3584, 1, False),
    ("llm gate/up swiglu", 4490, 37888, 3584, 3, False), ("llm down+res", 4490, 3584, 18944, 1, False),
    ("vit qkv", 15360, 3840, 1280, 0, True), ("vit proj+res", 15360, 1280, 1280, 1, True),
    ("vit gate/up swiglu", 15360, 6912, 1280, 3, True), ("vit down+res", 15360, 1280, 3456, 1, True),
    ("square 4096", 4096, 4096, 4096, 0, False), ("square 8192", 8192, 8192, 8192, 0, False),
    ("llm gate/up S=10218", 10218, 37888, 3584, 3, False), ("llm down S=10218", 10218, 3584, 18944, 1, False),
    ("llm qkv 32 videos", 143680, 4608, 3584, 0, True), ("llm gate/up 32 videos", 143680, 37888, 3584, 3, False),
    ("llm down 32 videos", 143680, 3584, 18944, 1, False), ("vit gate/up 32 videos", 491520, 6912, 1280, 3, True),
]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if any(k in s[0] for k in sys.argv[1:])]
g = torch.Generator(device=dev).manual_seed(0)
tot_t, tot_f = 0.0, 0.0
for name, M, N, K, epi, hb in shapes:
    nrep = 3 if M * K < (1 << 29) else 1
    a = [torch.empty(M, K, dtype=torch.bfloat16, device=dev).uniform_(-1, 1, generator=g) for _ in range(nrep)]
    w = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).uniform_(-1, 1, generator=g) for _ in range(nrep)]
    bias = torch.zeros(N, dtype=torch.bfloat16, device=dev) if hb else None
    No = N // 2 if epi == 3 else N
    res = torch.zeros(M, No, dtype=torch.bfloat16, device=dev) if epi == 1 else None
    out = torch.empty(M, No, dtype=torch.bfloat16, device=dev)

    def run(tile=0):
        for i in range(nrep):
            _lib.call("o3v_gemm_bf16_tile", P(a[i]), P(w[i]), P(bias), P(res), P(out), M, N, K, K, K, No, No, epi, tile, st)
    fl = 2.0 * M * N * K
    line = f"{name:22s} M={M:6d} N={N:6d} K={K:6d} "
    for tile in (128, 256, 0):   # forced 128-tile kernel, forced 256-tile kernel, the launcher's own choice
        run(tile)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(tile); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / nrep)
        t = sorted(ts)[2]
        line += f" | {'auto' if tile == 0 else tile}: {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TFLOP/s"
    if K % 128 == 0 and K >= 256:   # the phased 256-tile kernel (csrc/o3v_gemm8p.hip)
        o2 = torch.empty_like(out)
        _lib.call("o3v_gemm_bf16_tile", P(a[0]), P(w[0]), P(bias), P(res), P(out), M, N, K, K, K, No, No, epi, 256, st)
        _lib.call("o3v_gemm_bf16_phased", P(a[0]), P(w[0]), P(bias), P(res), P(o2), M, N, K, K, K, No, No, epi, st)
        torch.cuda.synchronize()
        same = bool(torch.equal(out.view(torch.int16), o2.view(torch.int16)))

        def runp():
            for i in range(nrep):
                _lib.call("o3v_gemm_bf16_phased", P(a[i]), P(w[i]), P(bias), P(res), P(o2), M, N, K, K, K, No, No, epi, st)
        runp()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); runp(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / nrep)
        t = sorted(ts)[2]
        line += f" | phased: {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TFLOP/s {'== 256' if same else 'DIFFERS'}"
    print(line, flush=True)
    del a, w
