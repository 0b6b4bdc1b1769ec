"""CPU tests: GSPO math vs the oracle transcription, data-parallel sharding helpers, and the N>1 path over a real
2-process gloo world (the rollout metrics all_gather and the sharded eval harness)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from open_o3_video_amd import dist as od
from open_o3_video_amd import rollout
from oracle import gspo_ref


def test_completion_mask_matches_reference_rule():
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 7, (16, 12), generator=g)
    ids[3] = 1  # no EOS at all
    ids[4, 0] = 5
    assert torch.equal(rollout.completion_mask(ids, 5), gspo_ref.eos_mask(ids, 5))
    m = rollout.completion_mask(torch.tensor([[1, 5, 2, 5], [1, 2, 3, 4], [5, 5, 5, 5]]), 5)
    assert m.tolist() == [[1, 1, 0, 0], [1, 1, 1, 1], [1, 0, 0, 0]]


@pytest.mark.parametrize("gspo", [True, False])
def test_gspo_loss_matches_oracle(gspo):
    g = torch.Generator().manual_seed(1)
    G, T = 4, 9
    lp = -torch.rand(2 * G, T, generator=g) * 3
    ref = lp + 0.3 * torch.randn(2 * G, T, generator=g)
    ref[0, 0] = lp[0, 0] + 50  # exercises the clamp
    rewards = torch.rand(2 * G, generator=g) * 3
    rewards[G:] = 1.0  # zero-variance group -> advantages 0 (std + 1e-4)
    mask = gspo_ref.eos_mask(torch.randint(0, 6, (2 * G, T), generator=g), 5)
    mask[1] = 0  # empty completion: clamp(min=1) path
    loss_ref, adv_ref, kl_ref, std_ref = gspo_ref.loss_and_parts(lp, ref, rewards, mask, G, gspo=gspo)
    adv, std = rollout.group_advantages(rewards, G)
    assert torch.equal(adv, adv_ref) and torch.equal(std, std_ref)
    assert torch.equal(rollout.per_token_kl(ref, lp), kl_ref)
    loss = rollout.gspo_loss(lp, lp, ref, adv, mask, gspo=gspo)
    assert torch.allclose(loss, loss_ref, rtol=0, atol=1e-7)
    assert (adv[G:] == 0).all()


def test_chunking():
    for n in (0, 1, 7, 8, 9, 100):
        for w in (1, 2, 3, 8):
            cov = sorted(i for r in range(w) for i in od.contiguous_chunk(n, r, w))
            assert cov == list(range(n))
            cov = sorted(i for r in range(w) for i in od.strided_chunk(n, r, w))
            assert cov == list(range(n))
    assert list(od.contiguous_chunk(10, 0, 4)) == [0, 1, 2] and list(od.contiguous_chunk(10, 3, 4)) == [8, 9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    r, w = od.init("gloo")
    assert (r, w) == (rank, world)
    # sharded eval: every rank gets all results in order
    res = od.run_data_parallel(list(range(11)), lambda x: x * x + rank * 0, policy="contiguous")
    res2 = od.run_data_parallel(list(range(11)), lambda x: -x, policy="strided")
    # dynamic queue: rank 1 is slow, so rank 0 must end up with most of the items; twice in a row (fresh queue each time)
    import time as _t
    taken = []

    def slow(x):
        taken.append(x)
        _t.sleep(0.05 if rank == 1 else 0.002)
        return x + 100
    res3 = od.run_data_parallel(list(range(40)), slow, policy="dynamic")
    n_first = len(taken)
    res4 = od.run_data_parallel(list(range(7)), lambda x: x * 3, policy="dynamic")
    # rollout metrics record: rank r contributes rewards r+1
    G, nf = 4, 7
    rr = rollout.RolloutResult(None, None, torch.ones(G, 5, dtype=torch.int32), None, None, torch.full((G, 5), 0.5),
                               torch.full((G, nf), float(rank + 1)), torch.full((G,), float(nf * (rank + 1))), None, None, [])
    gr = rollout.GroupRollout(None, [lambda **k: 0] * nf, None, 0, 0, num_generations=G)
    metrics = gr.gather_metrics(rr, torch.full((G,), 0.25 * (rank + 1)))
    gathered = od.all_gather_records(torch.full((2, 3), float(rank)))
    q.put((rank, res, res2, metrics, gathered.tolist(), res3, res4, n_first))
    torch.distributed.destroy_process_group()


def test_two_process_gloo_world():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in ps], key=lambda o: o[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][7] + outs[1][7] == 40 and outs[0][7] > outs[1][7] + 10      # every item once; the fast rank took most
    for rank, res, res2, metrics, gathered, res3, res4, n_first in outs:
        assert res3 == [x + 100 for x in range(40)] and res4 == [x * 3 for x in range(7)]
        assert res == [x * x for x in range(11)]
        assert res2 == [-x for x in range(11)]
        assert gathered == [[0.0] * 3] * 2 + [[1.0] * 3] * 2
        assert metrics["reward"] == pytest.approx((7 * 1 + 7 * 2) / 2)
        assert metrics["completion_length"] == 5.0 and metrics["kl"] == pytest.approx(0.5)
        assert metrics["reward_std"] == pytest.approx(0.375)
        assert metrics["all_wrong"] == 0.0 and metrics["all_correct"] == 1.0


# ---- group-parallel rollout (SURVEY 8e partitioning B / BASELINE config #3): G completions of one prompt over the ranks
class _StubPolicy:
    """Deterministic stand-in for the HF facade: completion `i` of a group depends only on its global index i."""
    EOS, PAD = 9, 0

    def generate(self, input_ids, attention_mask, pixel_values, image_grid_thw, generation_config):
        gc = generation_config
        G, off = gc.num_return_sequences, getattr(gc, "row_id_offset", 0)
        rows = []
        for g in range(G):
            i = off + g
            body = [10 + i + k for k in range(2 + (i * 3) % 5)] + [self.EOS]
            rows.append(body)
        T = min(gc.max_new_tokens, max(len(r) for r in rows))       # generation stops when every LOCAL row has finished
        comp = torch.full((G, T), self.PAD, dtype=torch.int64)
        for g, r in enumerate(rows):
            comp[g, :min(T, len(r))] = torch.tensor(r[:T])
        return torch.cat([torch.as_tensor(input_ids).expand(G, -1), comp], dim=1)

    def completion_logps(self, prompt_ids, prompt_mask, completion_ids, pixel_values=None, image_grid_thw=None):
        c = completion_ids.float()
        return -(0.05 * c + 0.01 * torch.arange(c.shape[1])[None, :])


def _group_step(group_parallel):
    pol = _StubPolicy()
    funcs = [lambda prompts, completions, **kw: [float(len(c[0]["content"])) for c in completions],
             lambda prompts, completions, **kw: [float(c[0]["content"].count("1")) for c in completions]]
    decode = lambda ids: [str([t for t in r.tolist() if t != pol.PAD]) for r in ids]     # skip_special_tokens drops the padding
    gr = rollout.GroupRollout(pol, funcs, decode, eos_token_id=pol.EOS, pad_token_id=pol.PAD,
                              num_generations=4, max_completion_length=12, group_parallel=group_parallel)
    inputs = {"input_ids": torch.tensor([[5, 6, 7]]), "attention_mask": torch.ones(1, 3, dtype=torch.int64)}
    res = gr.step(inputs, {"prompt": "p", "task": "t"})
    return {k: getattr(res, k) for k in ("prompt_completion_ids", "completion_mask", "per_token_logps", "rewards_per_func",
                                         "rewards", "advantages", "loss")} | {"completions": res.completions}


def _group_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    od.init("gloo")
    out = _group_step(group_parallel=True)
    q.put((rank, {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in out.items()}))
    torch.distributed.destroy_process_group()


def test_group_parallel_rollout_equals_single_rank_group():
    ref = _group_step(group_parallel=False)      # world size 1: the whole group on one rank
    assert ref["prompt_completion_ids"].shape[0] == 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in ps], key=lambda o: o[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = ref["completion_mask"].float()
    for rank, out in outs:
        for k in ("prompt_completion_ids", "completion_mask", "rewards_per_func", "rewards", "completions"):
            exp = ref[k].tolist() if torch.is_tensor(ref[k]) else ref[k]
            assert out[k] == exp, (rank, k)
        # positions after a row's EOS are masked out of every statistic; a rank never computes them past its own longest row
        assert torch.allclose(torch.tensor(out["per_token_logps"]) * m, ref["per_token_logps"] * m, atol=1e-6)
        assert torch.allclose(torch.tensor(out["advantages"]), ref["advantages"], atol=1e-6)
        assert abs(out["loss"] - ref["loss"].item()) < 1e-6
    with pytest.raises(ValueError):
        # 3 completions cannot be split over 2 ranks -- checked before any collective; simulate a 2-rank world view
        gr = rollout.GroupRollout(_StubPolicy(), [], lambda ids: [], 9, 0, num_generations=3, group_parallel=True)
        orig = od.world
        od.world = lambda: (0, 2)
        try:
            gr.step({"input_ids": torch.tensor([[1]]), "attention_mask": torch.ones(1, 1, dtype=torch.int64)}, {})
        finally:
            od.world = orig


def test_gspo_against_the_reference_lines(golden_dir):
    """Golden G10b: the trainer's OWN statements (R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:591-596, :635-636, :675-681, :691-706,
    :711-738) executed from the reference source by tools/make_golden.py g10b with a stub `self` on fixed inputs -- EOS mask, clamped
    k3 KL, group advantages (unbiased std + 1e-4), sequence-level (GSPO) and token-level clipped objectives with ratio != 1, beta 0,
    equal rewards, rows without EOS -- against rollout.py (the product) and oracle/gspo_ref.py (the transcription)."""
    import numpy as np
    import os
    g = np.load(os.path.join(golden_dir, "g10b_gspo.npz"))
    n = len({k.split("_")[0] for k in g.files})
    assert n == 6
    for i in range(n):
        k = f"c{i}_"
        G, B, T, eos = (int(v) for v in g[k + "params"])
        beta, el, eh, gspo = (float(v) for v in g[k + "hyper"])
        gspo = bool(gspo)
        cids = torch.from_numpy(g[k + "completion_ids"])
        lp, old, ref = (torch.from_numpy(g[k + n_]) for n_ in ("logps", "old_logps", "ref_logps"))
        rewards = torch.from_numpy(g[k + "rewards"])
        mask = rollout.completion_mask(cids, eos)
        assert torch.equal(mask, torch.from_numpy(g[k + "completion_mask"])) and torch.equal(gspo_ref.eos_mask(cids, eos), mask)
        kl = rollout.per_token_kl(ref, lp)
        assert torch.equal(kl, torch.from_numpy(g[k + "per_token_kl"]))
        adv, std = rollout.group_advantages(rewards, G)
        assert torch.equal(adv, torch.from_numpy(g[k + "advantages"])) and torch.equal(std, torch.from_numpy(g[k + "std"]))
        loss = rollout.gspo_loss(lp, old, ref, adv, mask, beta, el, eh, gspo)
        assert abs(float(loss) - float(g[k + "loss"])) <= 1e-6 * max(1.0, abs(float(g[k + "loss"]))), (i, float(loss), float(g[k + "loss"]))
        lo, ao, ko, so = gspo_ref.loss_and_parts(lp, ref, rewards, mask, G, beta, el, eh, gspo, old_logps=old)
        assert float(lo) == float(loss) or abs(float(lo) - float(g[k + "loss"])) <= 1e-6 * max(1.0, abs(float(lo)))
        assert torch.equal(ao, adv) and torch.equal(ko, kl)
        # the logging record (R:…:711-738) from the product's gather on one rank
        gr = rollout.GroupRollout(None, [], None, eos, 0, num_generations=G)
        res = rollout.RolloutResult(None, cids, mask, lp, ref, kl, rewards[:, None], rewards, adv, loss, [])
        gr.reward_funcs = []
        m = gr.gather_metrics(res, std)
        want = dict(zip(("completion_length", "all_wrong", "all_correct", "reward", "reward_std", "kl"), g[k + "metrics"]))
        for name, v in want.items():
            assert abs(m[name] - float(v)) <= 1e-5 * max(1.0, abs(float(v))), (i, name, m[name], float(v))
