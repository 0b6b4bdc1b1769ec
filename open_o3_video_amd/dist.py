"""Data-parallel sharding of the generate path: one process per GPU, full replica, units = videos / prompts.

The reference shards the same way -- `torchrun --nproc_per_node=8` with one prompt (and all its G completions) per
rank (R:src/scripts/run_grpo_video.sh:11-16, R:…/grpo_trainer.py:410-470) and `multiprocessing` workers over static
contiguous chunks in eval (R:eval/test/test_vstar_multi_images.py:608-642) -- and needs no collective for the math.
The only exchange is the metrics gather: the six `gather_for_metrics` calls of R:…/grpo_trainer.py:711-738 become ONE
all_gather of a packed [G, 9] fp32 record (rewards_per_func[7], reward, completion_length) -- 288 B per rank at G=8,
latency-bound over xGMI, so fewer and fused is the only lever.  Backend "nccl" is RCCL on ROCm; "gloo" on CPU.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def init(backend: str = None, device: torch.device = None) -> tuple:
    """Join the torchrun world (RANK / WORLD_SIZE / MASTER_* from the env).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world


def world() -> tuple:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def contiguous_chunk(n_items: int, rank: int, world_size: int) -> range:
    """Static contiguous chunks, remainder spread over the first ranks (R:eval/test/test_vstar_multi_images.py:608-618)."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def strided_chunk(n_items: int, rank: int, world_size: int) -> range:
    """Round-robin assignment: balances better when completion lengths correlate with dataset order."""
    return range(rank, n_items, world_size)


def all_gather_records(rec: torch.Tensor) -> torch.Tensor:
    """[G, K] per rank -> [world*G, K] on every rank with a single collective."""
    r, w = world()
    if w == 1 and not (dist.is_available() and dist.is_initialized()):
        return rec
    # (a one-rank group still runs the collective: the N = 1 bench line exercises the same RCCL call the N = 8 run makes)
    out = torch.empty((w * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def all_gather_padded(t: torch.Tensor, width: int, fill) -> torch.Tensor:
    """[n, T_local] per rank (T_local may differ) -> [world*n, width]: right-pad to `width`, one collective."""
    if t.shape[1] > width:
        raise ValueError("width smaller than a local tensor")
    p = torch.full((t.shape[0], width), fill, dtype=t.dtype, device=t.device)
    p[:, :t.shape[1]] = t
    return all_gather_records(p)


def gather_objects(obj) -> list:
    r, w = world()
    if w == 1:
        return [obj]
    out = [None] * w
    dist.all_gather_object(out, obj)
    return out


_queue_serial = 0


def dynamic_indices(n_items: int):
    """Work queue over the process group's key-value store: every rank draws the next unclaimed item index with one
    atomic `add` (a few bytes over the rendezvous TCP store, no collective, no data-path traffic).  Completion lengths vary
    by more than 10x between videos, so static chunks leave ranks idle at the end of an eval; with a queue the
    imbalance is bounded by one item."""
    global _queue_serial
    r, w = world()
    if w == 1:
        yield from range(n_items)
        return
    store = dist.distributed_c10d._get_default_store()
    key = f"o3v/queue/{_queue_serial}"      # same serial on every rank: calls are made in the same order everywhere
    _queue_serial += 1
    while True:
        i = store.add(key, 1) - 1
        if i >= n_items:
            return
        yield i


def run_data_parallel(items: Sequence, fn: Callable, policy: str = "contiguous") -> List:
    """Evaluate fn(item) over this rank's shard and return ALL results in the original item order on every rank
    (the eval harness' results_list + reorder-by-original_index, R:eval/test/test_vstar_multi_images.py:661-689).
    policy: "contiguous" (the reference's static chunks), "strided", or "dynamic" (work queue, see dynamic_indices)."""
    r, w = world()
    if policy == "dynamic":
        idx = dynamic_indices(len(items))
    else:
        idx = contiguous_chunk(len(items), r, w) if policy == "contiguous" else strided_chunk(len(items), r, w)
    mine = [(i, fn(items[i])) for i in idx]
    merged = [p for part in gather_objects(mine) for p in part]
    merged.sort(key=lambda p: p[0])
    assert [p[0] for p in merged] == list(range(len(items))), "shards do not cover the item list exactly once"
    return [p[1] for p in merged]
