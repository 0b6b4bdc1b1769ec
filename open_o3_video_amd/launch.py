"""Single-node launcher: one child process per GPU, rendezvous on 127.0.0.1 (what `torchrun --nproc-per-node N` does,
R:src/scripts/run_grpo_video.sh:11-16), for entry points that are started as `python bench.py --gpus N`.

The parent never touches the GPU (no HIP call, no torch.cuda query): a process that has initialised the GPU must not be
replaced or forked into ranks.  Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT and run the same
script; rank 0's stdout is passed through, the other ranks' stdout goes to stderr.  Any child failing fails the launch
(the others are terminated), so a rank-count mismatch can never print a result line."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), O3V_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    return env


def spawn_ranks(world: int, argv: Sequence[str], timeout_s: Optional[float] = None, poll_s: float = 0.2) -> int:
    """Run `python argv...` as `world` ranks; returns 0 only if every rank exited 0."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        out = None if r == 0 else sys.stderr      # rank 0 owns stdout (the result line)
        procs.append(subprocess.Popen([sys.executable, *argv], env=rank_env(r, world, port), stdout=out))
    t0 = time.monotonic()
    rc = 0
    try:
        alive = set(range(world))
        while alive:
            for r in list(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    print(f"[launch] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for o in alive:
                        procs[o].terminate()
            if timeout_s is not None and time.monotonic() - t0 > timeout_s and alive:
                rc = rc or 124
                print(f"[launch] timeout after {timeout_s}s; stopping ranks {sorted(alive)}", file=sys.stderr, flush=True)
                for o in alive:
                    procs[o].terminate()
                timeout_s = None
            if alive:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
    return rc
