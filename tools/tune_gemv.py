#!/usr/bin/env python3
"""A/B the GEMV decomposition (rows per wave R x intra-block K split KS) per decode shape, interleaved rounds in ONE
process on cold weights (rotating over 28 weight copies).  Builds a separate tuning library (-DO3V_TUNE, M=1 only)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open_o3_video_amd", "csrc")
OUT = os.path.join(ROOT, "open_o3_video_amd", "build", "libo3v_tune.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-DO3V_TUNE",
           "-shared", os.path.join(CSRC, "o3v_gemm.hip"), "-o", OUT]
    subprocess.check_call(cmd)


def main():
    if not os.path.exists(OUT):
        build()
    lib = C.CDLL(OUT)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.o3v_gemv_norm_bf16.argtypes = [vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.o3v_gemv_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.o3v_linear_decode_fp8.argtypes = [vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    fp8 = "--fp8" in sys.argv     # fp8 (e4m3fn) weight rows + per-row scales: the same decompositions on rows half as long
    dev = torch.device("cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    H, I = 3584, 18944
    shapes = {  # name: (N, K, epi, norm)
        "gate_up": (2 * I, H, 3, True), "down": (H, I, 1, False), "o_proj": (H, H, 1, False), "qkv": (4608, H, 0, True),
        "lm_head": (152064, H, 0, True),
    }
    if "--3b" in sys.argv:        # BASELINE config #1: Qwen2.5-VL-3B dims (36 layers of 2048 / 11008, GQA 16:2, vocabulary 151936)
        H, I = 2048, 11008
        shapes = {"gate_up": (2 * I, H, 3, True), "down": (H, I, 1, False), "o_proj": (H, H, 1, False), "qkv": (2560, H, 0, True),
                  "lm_head": (151936, H, 0, True)}
    if "--balance" in sys.argv:   # does a grid that divides evenly over the 256 CUs stream faster?  (gate/up: 2368 blocks = 9.25 per CU)
        shapes = {"gate_up": (2 * I, H, 3, True), "gu_2048blk": (2 * 16384, H, 3, True), "gu_2560blk": (2 * 20480, H, 3, True),
                  "gu_4096blk": (2 * 32768, H, 3, True), "down": (H, I, 1, False), "down_4096": (4096, I, 1, False)}
    variants = [(0, 0), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2)]   # (0, 0): the library's own choice
    if "--wg" in sys.argv:        # R = 2, KS = 1 with other workgroup sizes and trip depths: variant (100 + NW, k-steps per trip)
        variants = [(0, 0), (104, 4), (102, 4), (102, 2), (102, 6), (101, 4), (101, 2), (101, 6), (103, 4)]
        shapes = {k: shapes[k] for k in ("gate_up", "down", "lm_head")}
    if "--down" in sys.argv:      # R = 2, KS = 2 (long rows split over wave pairs) with other workgroup sizes / trip depths
        variants = [(0, 0), (314, 4), (314, 2), (314, 6), (308, 4), (306, 4), (310, 4), (316, 4), (312, 4), (304, 4)]
        shapes = {k: shapes[k] for k in ("down",)}
    if "--wg4" in sys.argv:       # R = 4 (two pairs per wave; the fp8 choice) with other workgroup sizes / trip depths
        variants = [(0, 0), (4, 1), (202, 2), (203, 2), (202, 4), (204, 4), (208, 2), (202, 1), (102, 4), (103, 4)]
        shapes = {k: shapes[k] for k in ("gate_up", "down", "lm_head")}
    L = 28
    g = torch.Generator(device=dev).manual_seed(0)
    for name, (N, K, epi, norm) in shapes.items():
        nl = 4 if name == "lm_head" else L
        if fp8:
            ws = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nl)]
            sc = torch.full((N,), 2.0 ** -8, dtype=torch.float32, device=dev)
        else:
            ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02, generator=g) for _ in range(nl)]
        x = torch.randn(1, K, device=dev).to(torch.bfloat16)
        nw = torch.ones(K, dtype=torch.bfloat16, device=dev)
        res = torch.zeros(1, N, dtype=torch.bfloat16, device=dev)
        out = torch.empty(1, N, dtype=torch.bfloat16, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        No = N // 2 if epi == 3 else N

        def run(v):
            lib.o3v_gemv_tune(*v)
            for w in ws:
                if fp8:
                    rc = lib.o3v_linear_decode_fp8(P(x), P(nw) if norm else None, 1e-6, P(w), P(sc), None, P(res), P(out), 1, N, K, K, No, N,
                                                   epi, st)
                elif norm:
                    rc = lib.o3v_gemv_norm_bf16(P(x), P(nw), 1e-6, P(w), None, P(res), P(out), 1, N, K, K, K, No, N, epi, st)
                else:
                    rc = lib.o3v_gemv_bf16(P(x), P(w), None, P(res), P(out), 1, N, K, K, K, No, N, epi, st)
                if rc:
                    return rc
            return 0
        results = {}
        for v in variants:
            if run(v):
                continue
            results[v] = []
        torch.cuda.synchronize()
        for rnd in range(5):
            for v in list(results):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(v)
                e1.record()
                torch.cuda.synchronize()
                results[v].append(e0.elapsed_time(e1) * 1e3 / nl)
        mb = N * K * (1 if fp8 else 2) / 1e6
        line = f"{name:8s} {mb:7.1f} MB  " + "  ".join(f"R{v[0]}K{v[1]}:{sorted(t)[len(t) // 2]:6.1f}us" for v, t in results.items())
        best = min(results, key=lambda v: sorted(results[v])[len(results[v]) // 2])
        bt = sorted(results[best])[len(results[best]) // 2]
        print(line + f"   best R{best[0]}K{best[1]} {mb / bt / 1e3 * 1e3:.0f} GB/s", flush=True)
        del ws


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
