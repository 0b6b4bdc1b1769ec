#!/usr/bin/env python3
"""A/B of the M = 1 decode GEMVs with four waves per workgroup vs CU-balanced workgroup sizes (csrc/o3v_gemm.hip
launch_gemv_balanced), cold weights (one matrix per layer, 28 layers), interleaved rounds in one process."""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open_o3_video_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    BF = torch.bfloat16
    L = 28
    dims = {"7b": (3584, 28, 4, 18944), "3b": (2048, 16, 2, 11008), "8b": (4096, 32, 8, 12288)}
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(0)
    for name, (H, Hq, Hkv, I) in dims.items():
        D = 128
        N, QD = (Hq + 2 * Hkv) * D, Hq * D
        rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
        x = rn(1, H)
        xm = rn(1, I)
        xa = rn(1, QD)
        nw = 1 + rn(H, sc=0.1)
        ang = torch.rand(1, 8, D // 2, generator=g, device=dev) * 30
        cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).contiguous()
        sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).contiguous()
        Wq = [rn(N, H, sc=1 / math.sqrt(H)) for _ in range(L)]
        Wo = [rn(H, QD, sc=1 / math.sqrt(QD)) for _ in range(L)]
        Wd = [rn(H, I, sc=1 / math.sqrt(I)) for _ in range(L)]
        bq = rn(N, sc=0.5)
        q = torch.zeros(1, Hq, D, dtype=BF, device=dev)
        kc = torch.zeros(1, Hkv, 64, D, dtype=BF, device=dev)
        vc = torch.zeros_like(kc)
        outs = {}

        def qkv(l):
            _lib.call("o3v_gemv_norm_qkv_rope", P(x), P(nw), 1e-6, P(Wq[l]), None, P(bq), 1, H, H, P(cos), P(sin), P(q), P(kc), P(vc),
                      5, Hq, Hkv, D, 64, 8, 3, st)

        def o(l):
            _lib.call("o3v_linear_decode", P(xa), None, 0.0, P(Wo[l]), None, None, P(x), P(outs["o"]), 1, H, QD, QD, H, H,
                      _lib.EPI_RESIDUAL, st)

        def down(l):
            _lib.call("o3v_linear_decode", P(xm), None, 0.0, P(Wd[l]), None, None, P(x), P(outs["d"]), 1, H, I, I, H, H,
                      _lib.EPI_RESIDUAL, st)

        outs["o"] = torch.zeros(1, H, dtype=BF, device=dev)
        outs["d"] = torch.zeros(1, H, dtype=BF, device=dev)

        def timeit(fn, n=5):
            for l in range(L):
                fn(l)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                for l in range(L):
                    fn(l)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (n * L) * 1e3

        # results: balanced vs four-wave
        ref = {}
        for bal in (0, 1):
            lib.o3v_gemv_set_balanced(bal)
            qkv(0); o(0); down(0)
            torch.cuda.synchronize()
            ref[bal] = (q.clone(), kc.clone(), outs["o"].clone(), outs["d"].clone())
        same = [torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref[0], ref[1])]
        dq = (ref[0][0].float() - ref[1][0].float()).abs().max().item()
        print(f"{name}: bitwise equal q/k/o/down = {same}, max |dq| = {dq:.4g}")
        for fname, fn, nbytes in (("qkv", qkv, N * H * 2), ("o_proj", o, H * QD * 2), ("down", down, H * I * 2)):
            res = {0: [], 1: []}
            for rnd in range(3):
                for bal in (0, 1):
                    lib.o3v_gemv_set_balanced(bal)
                    res[bal].append(timeit(fn))
            t0, t1 = min(res[0]), min(res[1])
            print(f"{name} {fname:7s} {nbytes / 1e6:7.1f} MB  four-wave {t0:6.2f} us ({nbytes / t0 / 1e6:5.2f} TB/s)   "
                  f"balanced {t1:6.2f} us ({nbytes / t1 / 1e6:5.2f} TB/s)")
        lib.o3v_gemv_set_balanced(1)
        del Wq, Wo, Wd


if __name__ == "__main__":
    main()
