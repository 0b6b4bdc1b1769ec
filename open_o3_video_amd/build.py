"""Build libo3v_hip.so (gfx950) in-tree with hipcc.  `python -m open_o3_video_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libo3v_hip.so")
SOURCES = ["o3v_elem.hip", "o3v_gemm.hip", "o3v_gemm8p.hip", "o3v_fp8.hip", "o3v_attn.hip", "o3v_fused.hip", "o3v_sample.hip", "o3v_engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def find_hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "o3v.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, extra_flags=(), lib: str = LIB, objdir: str = "build") -> str:
    """extra_flags / lib / objdir: diagnostic variants (e.g. -DO3V_STAMPS for tools/probe_fused.py) built beside the product library."""
    if not force and lib == LIB and not needs_build():
        return LIB
    hipcc = find_hipcc()
    objs = []
    os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
