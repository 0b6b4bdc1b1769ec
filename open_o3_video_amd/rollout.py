"""GSPO group rollout around the generate path (forward side of Qwen2VLGRPOTrainer.compute_loss,
R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:402-742): G sampled completions per prompt, EOS masks, per-token
log-probs of policy and reference, KL, the seven rewards, group advantages, the sequence-level clipped GSPO
objective value, and the logging record gathered across ranks.  Backward / optimizer are out of scope.

Differences from the reference that do not change results: the ViT and the prompt prefill run once per prompt and are
shared by the G completions (the reference recomputes them G times in generate and 2G times for the log-probs,
R:…:601-606), logits are never materialised for the whole batch, and the six metric gathers are one all_gather.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import dist as o3v_dist


# ------------------------------------------------------------------------------------------------ pure math
def completion_mask(completion_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    """1 up to and including the first EOS, 0 after (R:…:590-596)."""
    is_eos = completion_ids == eos_token_id
    T = completion_ids.shape[1]
    eos_idx = torch.full((completion_ids.shape[0],), T, dtype=torch.long, device=completion_ids.device)
    any_eos = is_eos.any(dim=1)
    eos_idx[any_eos] = is_eos.int().argmax(dim=1)[any_eos]
    seq = torch.arange(T, device=completion_ids.device).expand(completion_ids.shape[0], -1)
    return (seq <= eos_idx.unsqueeze(1)).int()


def per_token_kl(ref_logps: torch.Tensor, logps: torch.Tensor) -> torch.Tensor:
    """k3 estimator with the log-ratio clamped to [-10, 10] (R:…:635-636)."""
    x = torch.clamp(ref_logps - logps, min=-10, max=10)
    return torch.exp(x) - x - 1


def group_advantages(rewards: torch.Tensor, G: int):
    """(r - mean_group) / (std_group + 1e-4), unbiased std (R:…:675-681).  Returns (advantages, std per row)."""
    grouped = rewards.view(-1, G)
    mean = grouped.mean(dim=1).repeat_interleave(G, dim=0)
    std = grouped.std(dim=1).repeat_interleave(G, dim=0)
    return (rewards - mean) / (std + 1e-4), std


def gspo_loss(logps, old_logps, ref_logps, advantages, mask, beta=0.04, eps_low=0.2, eps_high=0.2, gspo=True):
    """Sequence-level importance ratio, clipped surrogate, + beta*KL, masked mean per sequence (R:…:691-706)."""
    log_ratio = logps - old_logps
    m = mask.to(logps.dtype)
    denom = m.sum(-1).clamp(min=1.0)
    if gspo:
        liw = ((log_ratio * m).sum(-1) / denom).unsqueeze(-1)
    else:
        liw = log_ratio
    c1 = torch.exp(liw)
    c2 = torch.clamp(c1, 1 - eps_low, 1 + eps_high)
    a = advantages.unsqueeze(1)
    ptl = -torch.min(c1 * a, c2 * a) + beta * per_token_kl(ref_logps, logps)
    return ((ptl * m).sum(-1) / denom).mean()


# ------------------------------------------------------------------------------------------------ rollout
@dataclass
class RolloutResult:
    prompt_completion_ids: torch.Tensor
    completion_ids: torch.Tensor
    completion_mask: torch.Tensor
    per_token_logps: torch.Tensor
    ref_per_token_logps: torch.Tensor
    per_token_kl: torch.Tensor
    rewards_per_func: torch.Tensor
    rewards: torch.Tensor
    advantages: torch.Tensor
    loss: torch.Tensor
    completions: List[str]
    metrics: Dict[str, float] = field(default_factory=dict)


class GroupRollout:
    """model / ref_model: open_o3_video_amd.hf_api.Qwen2_5_VLForConditionalGeneration (ref_model None -> the policy
    itself, as the reference's PEFT branch does).  decode: ids -> text (tokenizer.batch_decode)."""

    def __init__(self, model, reward_funcs: Sequence[Callable], decode: Callable[[torch.Tensor], List[str]],
                 eos_token_id: int, pad_token_id: int, ref_model=None, num_generations: int = 4,
                 max_completion_length: int = 768, max_prompt_length: Optional[int] = 16384, beta: float = 0.04,
                 epsilon_low: float = 0.2, epsilon_high: float = 0.2, temperature: float = 1.0, top_p: float = 0.95,
                 gspo: bool = True, group_parallel: bool = False, top_k: int = 50):
        self.model, self.ref_model = model, ref_model
        self.reward_funcs = list(reward_funcs)
        self.decode = decode
        self.eos, self.pad = eos_token_id, pad_token_id
        self.G, self.T = num_generations, max_completion_length
        # group_parallel: the G completions of ONE prompt are sharded over the ranks (G/world rows each, every rank runs the
        # ViT and the prompt prefill itself -- cheap next to the decode) and the per-sample rewards / log-probs are
        # all-gathered, because the group mean / std of the advantages spans all G (SURVEY 8e partitioning B; BASELINE
        # config #3).  Default (False) is the reference's own layout: one prompt with all its G completions per rank.
        self.group_parallel = bool(group_parallel)
        self.max_prompt_length = max_prompt_length
        self.beta, self.el, self.eh = beta, epsilon_low, epsilon_high
        self.temperature, self.top_p, self.gspo = temperature, top_p, gspo
        self.top_k = int(top_k)

    @torch.no_grad()
    def step(self, prompt_inputs: dict, example: dict) -> RolloutResult:
        """prompt_inputs: processor output (input_ids, attention_mask, pixel_values, image_grid_thw) of ONE prompt
        (the reference hard-wires batch 1, R:…:410-470).  example: the dataset row; every column except
        prompt/completion is repeated G times and handed to the reward functions (R:…:648-656)."""
        ids, mask = prompt_inputs["input_ids"], prompt_inputs["attention_mask"]
        if self.max_prompt_length is not None:          # left truncation of ids only, as the reference (R:…:569-578)
            ids, mask = ids[:, -self.max_prompt_length:], mask[:, -self.max_prompt_length:]
        pv, grid = prompt_inputs.get("pixel_values"), prompt_inputs.get("image_grid_thw")
        # the non-multi-image branch of the trainer hands the processor's native video tensors over (R:…:555-564); generate sees
        # second_per_grid_ts, the log-prob passes do not (the trainer deletes it first, R:…:608-609)
        vid = {k: prompt_inputs[k] for k in ("pixel_values_videos", "video_grid_thw") if prompt_inputs.get(k) is not None}
        spg = prompt_inputs.get("second_per_grid_ts")
        from .hf_api import GenerationConfigLike
        rank, world = o3v_dist.world()
        sharded = self.group_parallel and world > 1
        if sharded and self.G % world:
            raise ValueError(f"num_generations={self.G} is not divisible by the {world} ranks of the group")
        G_local = self.G // world if sharded else self.G
        # the trainer's GenerationConfig (R:…:306-313) as the library the reference pins completes it (transformers @336dc69d,
        # R:setup.sh:4): top_k 50 and repetition_penalty 1.0 are that GenerationConfig's own defaults and do NOT come from the
        # checkpoint's generation_config.json (Qwen2.5-VL ships top_k 1 / penalty 1.05 there: with them all G samples would be
        # the argmax and every advantage 0).  They are passed explicitly so the rollout does not depend on the façade's
        # resolution mode; eos / pad / bos are still inherited from the checkpoint, as the pinned library does.
        gc = GenerationConfigLike(max_new_tokens=self.T, do_sample=True, top_p=self.top_p, temperature=self.temperature,
                                  top_k=self.top_k, repetition_penalty=1.0, num_return_sequences=G_local,
                                  pad_token_id=self.pad, row_id_offset=rank * G_local if sharded else 0)
        pc = self.model.generate(input_ids=ids, attention_mask=mask, pixel_values=pv, image_grid_thw=grid,
                                 generation_config=gc, **vid, **({"second_per_grid_ts": spg} if vid and spg is not None else {}))
        S = ids.shape[1]
        comp = pc[:, S:]
        cmask = completion_mask(comp, self.eos)
        # per-token log-probs of the completions under the policy and the reference model (R:…:601-632): the G rows share
        # the prompt, so each model runs the ViT and the prompt once and the lm_head only over the G x T kept positions
        lp = self._logps(self.model, ids, mask, comp, pv, grid, pc, vid)
        ref = lp if self.ref_model is None else self._logps(self.ref_model, ids, mask, comp, pv, grid, pc, vid)
        kl = per_token_kl(ref, lp)
        texts = self.decode(comp)
        completions = [[{"role": "assistant", "content": t}] for t in texts]
        cols = {k: [v] * G_local for k, v in example.items() if k not in ("prompt", "completion")}
        rpf = torch.zeros(len(texts), len(self.reward_funcs), device=pc.device)
        for i, fn in enumerate(self.reward_funcs):
            rpf[:, i] = torch.tensor(fn(prompts=[example.get("prompt")] * G_local, completions=completions, **cols),
                                     dtype=torch.float32, device=pc.device)
        if sharded:
            # one exchange step: every rank ends up with the whole group in completion-index order (rank-major == index
            # order).  Completions stop at different lengths on different ranks: right-pad to the longest.
            T_all = int(o3v_dist.all_gather_records(torch.tensor([[float(comp.shape[1])]], device=pc.device)).max().item())
            comp = o3v_dist.all_gather_padded(comp, T_all, self.pad)
            cmask = o3v_dist.all_gather_padded(cmask, T_all, 0)
            lp = o3v_dist.all_gather_padded(lp, T_all, 0.0)
            ref = lp if self.ref_model is None else o3v_dist.all_gather_padded(ref, T_all, 0.0)
            kl = per_token_kl(ref, lp)
            rpf = o3v_dist.all_gather_records(rpf.contiguous())
            texts = [t for part in o3v_dist.gather_objects(texts) for t in part]
            pc = torch.cat([pc[:1, :S].expand(self.G, -1), comp], dim=1)
        rewards = rpf.sum(dim=1)
        adv, std = group_advantages(rewards, self.G)
        loss = gspo_loss(lp, lp, ref, adv, cmask, self.beta, self.el, self.eh, self.gspo)
        res = RolloutResult(pc, comp, cmask, lp, ref, kl, rpf, rewards, adv, loss, texts)
        res.metrics = self.gather_metrics(res, std)
        return res

    def _logps(self, model, ids, mask, comp, pv, grid, pc, vid=None):
        vid = vid or {}
        if hasattr(model, "completion_logps"):
            return model.completion_logps(ids, mask, comp, pv, grid, **vid)
        # any other HF-style model: the reference's own formulation
        S = ids.shape[1]
        full_mask = torch.cat([torch.as_tensor(mask).to(pc.device).repeat_interleave(self.G, dim=0), torch.ones_like(comp)], dim=1)
        rep = lambda t: None if t is None else torch.as_tensor(t).repeat(pc.shape[0], *([1] * (torch.as_tensor(t).dim() - 1)))
        return model.per_token_logps(pc, full_mask, rep(pv), rep(grid), **{k: rep(v) for k, v in vid.items()})[:, S - 1:]

    def gather_metrics(self, res: RolloutResult, std: torch.Tensor) -> Dict[str, float]:
        """One all_gather of [G, n_funcs+4] replaces six gather_for_metrics (R:…:711-738)."""
        m = res.completion_mask.to(torch.float32)
        mean_kl = ((res.per_token_kl * m).sum(1) / m.sum(1).clamp(min=1.0))
        rec = torch.cat([res.rewards_per_func, res.rewards[:, None], m.sum(1, keepdim=True), std[:, None], mean_kl[:, None]], dim=1)
        allr = o3v_dist.all_gather_records(rec.contiguous())
        nf = res.rewards_per_func.shape[1]
        per_dev = allr[:, nf].view(-1, self.G)
        out = {"completion_length": allr[:, nf + 1].mean().item(), "reward": allr[:, nf].mean().item(),
               "reward_std": allr[:, nf + 2].mean().item(), "kl": allr[:, nf + 3].mean().item(),
               "all_wrong": (per_dev <= 1).all(dim=1).float().mean().item(),
               "all_correct": (per_dev >= 2).all(dim=1).float().mean().item()}
        for i, fn in enumerate(self.reward_funcs):
            out[f"rewards/{getattr(fn, '__name__', str(i))}"] = allr[:, i].mean().item()
        return out
