#!/usr/bin/env python3
"""Timeline of the one-launch decode attention block (csrc/o3v_fused.hip) from in-kernel s_memrealtime stamps.

Diagnostic build only: `python tools/probes/probe_fused.py --build` (here, no GPU needed) compiles the library with
-DO3V_STAMPS into open_o3_video_amd/libo3v_hip_stamps.so; on the GPU box the script loads that variant, runs decode-like
steps at 7B dims (28 layers of distinct weights, so every launch streams cold weights; the MLP GEMVs run in between) and
prints, for the last layer's launch, when each role's workgroups started, finished waiting and ended."""
import argparse
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "open_o3_video_amd", "libo3v_hip_stamps.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--ctx", type=int, default=4600)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--dims", default="7b", choices=["7b", "3b"])
    args = ap.parse_args()
    from open_o3_video_amd import build as b
    if args.build:
        print(b.build(force=True, verbose=False, extra_flags=("-DO3V_STAMPS",), lib=VARIANT, objdir="build_stamps"))
        return
    import numpy as np
    import torch
    from open_o3_video_amd import _lib
    _lib.LIB_PATH = VARIANT
    lib = _lib.load()
    lib.o3v_fused_set_stamps.argtypes = [C.c_void_p]
    lib.o3v_fused_set_stamps.restype = None
    dev = torch.device("cuda")
    BF = torch.bfloat16
    if args.dims == "7b":
        H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
    else:
        H, Hq, Hkv, D, I = 2048, 16, 2, 128, 11008
    L, ctx = args.layers, args.ctx
    Tmax = ctx + 64
    N, QD = (Hq + 2 * Hkv) * D, Hq * D
    nsplit = max(1, min(64, (Tmax + 127) // 128, 640 // Hkv))
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
    W = [dict(ln1=1 + rn(H, sc=0.1), ln2=1 + rn(H, sc=0.1), qkv=rn(N, H, sc=1 / math.sqrt(H)), qb=rn(N, sc=0.5),
              o=rn(H, QD, sc=0.3 / math.sqrt(QD)), gu=rn(2 * I, H, sc=1 / math.sqrt(H)), down=rn(H, I, sc=0.3 / math.sqrt(I)))
         for _ in range(L)]
    kc = rn(L, 1, Hkv, Tmax, D)
    vc = rn(L, 1, Hkv, Tmax, D)
    ang = torch.rand(1, 8, D // 2, generator=g, device=dev) * 30
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(BF).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(BF).contiguous()
    x = rn(1, H)
    q = torch.zeros(1, Hq, D, dtype=BF, device=dev)
    att = torch.zeros_like(q)
    mlp = torch.zeros(1, I, dtype=BF, device=dev)
    part_o = torch.empty(Hq * 64 * D, dtype=torch.float32, device=dev)
    part_ml = torch.empty(Hq * 64 * 2, dtype=torch.float32, device=dev)
    sync = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    epoch = [0, 0]   # launches made on `sync`: attention block, MLP block
    TMO = _lib.SYNC_TMO_BYTE
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    grid = N // 8 + nsplit * Hkv + (H + 7) // 8
    stamps = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    slot = ctx - 1

    def mlp_half(l):
        _lib.call("o3v_linear_decode", P(x), P(W[l]["ln2"]), 1e-6, P(W[l]["gu"]), None, None, None, P(mlp), 1, 2 * I, H, H, I, 0,
                  _lib.EPI_SWIGLU, st)
        _lib.call("o3v_linear_decode", P(mlp), None, 0.0, P(W[l]["down"]), None, None, P(x), P(x), 1, H, I, I, H, H,
                  _lib.EPI_RESIDUAL, st)

    def step_fused():
        for l in range(L):
            epoch[0] += 1
            rc = lib.o3v_decode_attn_block(P(x), P(W[l]["ln1"]), 1e-6, P(W[l]["qkv"]), P(W[l]["qb"]), P(W[l]["o"]), P(cos), P(sin),
                                           P(q), P(att), P(kc[l]), P(vc[l]), P(part_o), P(part_ml), None, H, Hq, Hkv, D, slot, Tmax,
                                           8, 3, nsplit, scale, P(sync), epoch[0], st)
            assert rc == 0, rc
            mlp_half(l)

    def step_three():
        for l in range(L):
            _lib.call("o3v_gemv_norm_qkv_rope", P(x), P(W[l]["ln1"]), 1e-6, P(W[l]["qkv"]), None, P(W[l]["qb"]), 1, H, H, P(cos),
                      P(sin), P(q), P(kc[l]), P(vc[l]), slot, Hq, Hkv, D, Tmax, 8, 3, st)
            _lib.call("o3v_attn_decode", P(q), P(kc[l]), P(vc[l]), P(att), P(part_o), P(part_ml), None, 1, Hq, Hkv, D, ctx, Tmax,
                      nsplit, scale, st)
            _lib.call("o3v_linear_decode", P(att), None, 0.0, P(W[l]["o"]), None, None, P(x), P(x), 1, H, QD, QD, H, H,
                      _lib.EPI_RESIDUAL, st)
            mlp_half(l)

    def step_mlp_only():
        for l in range(L):
            mlp_half(l)

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3 / L   # us per layer

    res = {}
    for rnd in range(3):
        for name, fn in (("mlp_only", step_mlp_only), ("three_launches", step_three), ("fused", step_fused)):
            res.setdefault(name, []).append(timeit(fn, args.steps))
    for k, v in res.items():
        print(f"{k:16s} us/layer: " + " ".join(f"{t:7.2f}" for t in v))
    base = min(res["mlp_only"])
    print(f"attention half: three launches {min(res['three_launches']) - base:.2f} us, fused {min(res['fused']) - base:.2f} us")
    assert int(sync[TMO:TMO + 4].view(torch.int32)[0].item()) == 0, "time-out"

    # ---- timeline of the last layer's launch, for the full kernel and its ablations
    lib.o3v_fused_set_knob.argtypes = [C.c_int]
    lib.o3v_fused_set_knob.restype = None
    nb_qkv, nb_attn = N // 8, nsplit * Hkv
    roles = (("qkv", 0, nb_qkv), ("attn", nb_qkv, nb_qkv + nb_attn), ("o_proj", nb_qkv + nb_attn, grid))
    print(f"grid {grid}: qkv {nb_qkv} attn {nb_attn} o {grid - nb_qkv - nb_attn}; times in us from the first workgroup's start")
    for knob, what in ((0, "full"), (1, "q/k/v role alone"), (2, "q/k/v + attention (o_proj role exits)"),
                       (4, "no K/V request ahead of the wait"), (8, "o_proj weights after the wait"), (12, "neither"),
                       (32, "o_proj weight stream paced: 1 x s_sleep 16 per step"), (64, "paced 2x"), (128, "paced 4x")):
        lib.o3v_fused_set_knob(knob)
        tl = timeit(step_fused, args.steps) if knob not in (1, 2) else float("nan")
        stamps.zero_()
        lib.o3v_fused_set_stamps(P(stamps))
        step_fused()
        torch.cuda.synchronize()
        lib.o3v_fused_set_stamps(None)
        code = int(sync[TMO:TMO + 4].view(torch.int32)[0].item())
        print(f"        time-out word {code:#x}")
        sync.zero_()   # an ablation that removes a role leaves the tickets out of step with the epochs: start over
        epoch[0] = epoch[1] = 0
        t = stamps.cpu().numpy().reshape(grid, 8).astype(np.float64) / 100.0   # 100 MHz -> us
        t0 = t[:, 0][t[:, 0] > 0].min()

        def stat(a):
            a = a[a > 0] - t0
            return "   n/a" if a.size == 0 else f"min {a.min():6.2f} med {np.median(a):6.2f} p90 {np.percentile(a, 90):6.2f} max {a.max():6.2f}"

        print(f"---- knob {knob}: {what}   (layer incl. MLP: {tl:.2f} us)")
        for name, lo, hi in roles:
            r = t[lo:hi]
            print(f"{name:7s} start     {stat(r[:, 0])}")
            if name != "qkv":
                print(f"{name:7s} wait over {stat(r[:, 1])}")
            if name == "attn":
                print(f"{name:7s} partials  {stat(r[:, 2])}")
                print(f"{name:7s} all known {stat(r[:, 4])}")
                print(f"{name:7s} slice out {stat(r[:, 5])}")
            print(f"{name:7s} end       {stat(r[:, 3])}")
    lib.o3v_fused_set_knob(0)


if __name__ == "__main__":
    main()
