#!/usr/bin/env python3
"""M=8 (group rollout / batched eval) decode linears on the fragment-major weight image: A/B of the K split (KS) and of the
KiB of weight loads in flight per wave (UT), interleaved rounds, cold weights, 7B shapes.  Builds a tuning library."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "open_o3_video_amd", "csrc")
OUT = os.path.join(ROOT, "open_o3_video_amd", "libo3v_tune.so")   # in-tree: a built copy travels to the GPU box
if not os.path.exists(OUT):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                           "-DO3V_TUNE", "-shared", os.path.join(CSRC, "o3v_gemm.hip"), os.path.join(CSRC, "o3v_gemm8p.hip"), "-o", OUT])
from open_o3_video_amd.weights import pack_mfma_fragments  # noqa: E402

lib = C.CDLL(OUT)
vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
lib.o3v_linear_decode.argtypes = [vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
dev = torch.device("cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
H, I = 3584, 18944
shapes = {"gate_up": (2 * I, H, 3, True), "down": (H, I, 1, False), "o_proj": (H, H, 1, False), "qkv": (4608, H, 0, True),
          "lm_head": (152064, H, 0, True), "gate_up (norm apart)": (2 * I, H, 3, False), "lm_head (norm apart)": (152064, H, 0, False)}
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, (N, K, epi, norm) in shapes.items():
    nl = 4 if name.startswith("lm_head") else 28
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02, generator=g) for _ in range(nl)]
    wps = [pack_mfma_fragments(w) for w in ws]
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    nw = torch.ones(K, dtype=torch.bfloat16, device=dev)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    No = N // 2 if epi == 3 else N
    out = torch.empty(M, No, dtype=torch.bfloat16, device=dev)

    def run(v):
        lib.o3v_gemv_mfma_tune(*v)
        for w, wp in zip(ws, wps):
            rc = lib.o3v_linear_decode(P(x), P(nw) if norm else None, 1e-6, P(w), P(wp), None, P(res), P(out), M, N, K, K, No, N, epi, st)
            if rc:
                return rc
        return 0
    if norm and M > 16:
        continue                      # above 16 rows the linears take no fused norm
    variants = [(0, 0)] + [(ks, ut) for ks in (1, 2, 4) for ut in (4, 8, 16)]   # (0, 0): the library's own choice
    if M > 16:      # ut = 100 + U: two weight blocks per wave share the x fragments
        variants += [(ks, 100 + ut) for ks in (1, 2, 4) for ut in (4, 8, 16)]
    variants = [v for v in variants if run(v) == 0]
    res_t = {v: [] for v in variants}
    torch.cuda.synchronize()
    for rnd in range(3):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(v); e1.record(); torch.cuda.synchronize()
            res_t[v].append(e0.elapsed_time(e1) * 1e3 / nl)
    med = {v: sorted(t)[1] for v, t in res_t.items()}
    best = min(med, key=med.get)
    print(f"M={M} {name:22s} {N * K * 2 / 1e6:7.1f} MB  " + "  ".join(f"KS{v[0]}U{v[1]}:{t:6.1f}" for v, t in med.items()) +
          f"   best KS{best[0]} U{best[1]} = {N * K * 2 / med[best] / 1e6:.2f} TB/s", flush=True)
    del ws, wps
