#!/usr/bin/env python3
"""Timeline of the persistent layer block (csrc/o3v_fused.hip, decode_layer_block_kernel) from in-kernel s_memrealtime stamps.

`python tools/probes/probe_layer_block.py --build` (no GPU needed) compiles the library with -DO3V_STAMPS into
open_o3_video_amd/libo3v_hip_stamps.so; on the GPU box the script runs decode-like steps at 7B dims (28 layers of distinct weights,
down_proj launched in between) and prints, for the last layer's launch, the distribution over workgroups of every stamp:
0 start, 1 x landed, 2 q/k/v pair stored, 3 E1 ticket taken, 4 attention output gathered (o_proj owners) / attention item done
(attention workgroups), 5 E4 ticket taken, 6 x' normalised (gate/up may start), 7 end.  Then us per layer of the block + down_proj
against the role-per-workgroup block + gate/up + down_proj."""
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
    args = ap.parse_args()
    from open_o3_video_amd import build as b
    if args.build:
        print(b.build(force=True, verbose=False, extra_flags=("-DO3V_STAMPS",), lib=VARIANT, objdir="build_stamps"))
        return
    import numpy as np
    import torch
    from open_o3_video_amd import _lib
    from open_o3_video_amd.weights import pack_gate_up
    _lib.LIB_PATH = VARIANT
    lib = _lib.load()
    lib.o3v_fused_set_stamps.argtypes = [C.c_void_p]
    lib.o3v_fused_set_stamps.restype = None
    dev = torch.device("cuda")
    BF = torch.bfloat16
    H, Hq, Hkv, D, I = 3584, 28, 4, 128, 18944
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
    sync2 = torch.zeros(lib.o3v_decode_sync_bytes(), dtype=torch.uint8, device=dev)
    epoch = [0, 0]
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scale = 1.0 / math.sqrt(D)
    n_wg = 3 * torch.cuda.get_device_properties(0).multi_processor_count
    stamps = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
    slot = ctx - 1

    def down(l):
        _lib.call("o3v_linear_decode", P(mlp), None, 0.0, P(W[l]["down"]), None, None, P(x), P(x), 1, H, I, I, H, H, _lib.EPI_RESIDUAL, st)

    def step_layer_block():
        for l in range(L):
            epoch[0] += 1
            rc = lib.o3v_decode_layer_block(P(x), P(W[l]["ln1"]), 1e-6, P(W[l]["qkv"]), P(W[l]["qb"]), P(W[l]["o"]), P(W[l]["ln2"]),
                                            P(W[l]["gu"]), P(mlp), P(cos), P(sin), P(q), P(att), P(kc[l]), P(vc[l]), P(part_o), P(part_ml),
                                            None, H, I, Hq, Hkv, D, slot, Tmax, 8, 3, nsplit, scale, P(sync), epoch[0], st)
            assert rc == 0, rc
            down(l)

    def step_role_block():
        for l in range(L):
            epoch[1] += 1
            rc = lib.o3v_decode_attn_block(P(x), P(W[l]["ln1"]), 1e-6, P(W[l]["qkv"]), P(W[l]["qb"]), P(W[l]["o"]), P(cos), P(sin),
                                           P(q), P(att), P(kc[l]), P(vc[l]), P(part_o), P(part_ml), None, H, Hq, Hkv, D, slot, Tmax,
                                           8, 3, nsplit, scale, P(sync2), epoch[1], st)
            assert rc == 0, rc
            _lib.call("o3v_linear_decode", P(x), P(W[l]["ln2"]), 1e-6, P(W[l]["gu"]), None, None, None, P(mlp), 1, 2 * I, H, H, I, 0,
                      _lib.EPI_SWIGLU, st)
            down(l)

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3 / L

    res = {}
    for rnd in range(3):
        for name, fn in (("role_block+gu+down", step_role_block), ("layer_block+down", step_layer_block)):
            res.setdefault(name, []).append(timeit(fn, args.steps))
    for k, v in res.items():
        print(f"{k:20s} us/layer: " + " ".join(f"{t:7.2f}" for t in v))
    for s_ in (sync, sync2):
        code = int(s_[_lib.SYNC_TMO_BYTE:_lib.SYNC_TMO_BYTE + 4].view(torch.int32)[0].item())
        assert code == 0, hex(code)
    # timeline of one launch (the last layer of one more step)
    lib.o3v_fused_set_stamps(P(stamps))
    step_layer_block()
    torch.cuda.synchronize()
    lib.o3v_fused_set_stamps(None)
    t = stamps.view(n_wg, 8).cpu().numpy().astype(np.float64)
    nb_attn = nsplit * Hkv
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0       # s_memrealtime ticks at 100 MHz
    names = ["start", "x landed", "q/k/v pair stored", "E1 ticket", "att gathered | attention done", "E4 ticket", "x' normalised", "end"]
    for grp, sel in (("attention workgroups", slice(0, nb_attn)), ("o_proj owners with a q/k/v pair", slice(nb_attn, 576)),
                     ("o_proj owners without a pair", slice(576, n_wg))):
        print(f"--- {grp}")
        for i, nm in enumerate(names):
            col = us[sel, i]
            col = col[t[sel, i] > 0]
            if col.size:
                print(f"  {i} {nm:32s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f} us")


if __name__ == "__main__":
    main()
