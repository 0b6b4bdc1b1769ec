"""Oracle: the loss-side arithmetic of the GSPO rollout, transcribed statement by statement from
R:src/r1-v/src/open_r1/trainer/grpo_trainer.py:591-596 (EOS mask), :635-636 (KL), :675-681 (advantages),
:691-706 (GSPO objective).  The trainer itself is not importable offline (needs trl); PINNED by golden G10b
(tests/golden/g10b_gspo.npz): tools/make_golden.py g10b executes exactly those line ranges of the reference source with a stub
`self` on fixed inputs (tests/test_rollout_dist_cpu.py::test_gspo_against_the_reference_lines).  `old_logps` (None = the
reference's forward pass, where `per_token_logps.detach()` equals the policy's own values) is what `.detach()` returns in the
optimisation steps after the first.  Test infrastructure only."""
import torch


def eos_mask(completion_ids, eos_token_id):
    is_eos = completion_ids == eos_token_id
    eos_idx = torch.full((is_eos.size(0),), is_eos.size(1), dtype=torch.long)
    eos_idx[is_eos.any(dim=1)] = is_eos.int().argmax(dim=1)[is_eos.any(dim=1)]
    sequence_indices = torch.arange(is_eos.size(1)).expand(is_eos.size(0), -1)
    return (sequence_indices <= eos_idx.unsqueeze(1)).int()


def loss_and_parts(per_token_logps, ref_per_token_logps, rewards, completion_mask, num_generations, beta=0.04,
                   epsilon_low=0.2, epsilon_high=0.2, gspo=True, old_logps=None):
    x_clamped = torch.clamp(ref_per_token_logps - per_token_logps, min=-10, max=10)
    per_token_kl = torch.exp(x_clamped) - x_clamped - 1
    mean_grouped_rewards = rewards.view(-1, num_generations).mean(dim=1)
    std_grouped_rewards = rewards.view(-1, num_generations).std(dim=1)
    mean_grouped_rewards = mean_grouped_rewards.repeat_interleave(num_generations, dim=0)
    std_grouped_rewards = std_grouped_rewards.repeat_interleave(num_generations, dim=0)
    advantages = (rewards - mean_grouped_rewards) / (std_grouped_rewards + 1e-4)
    log_ratio = per_token_logps - (per_token_logps.detach() if old_logps is None else old_logps)
    if gspo:
        log_importance_weights = (log_ratio * completion_mask).sum(-1) / completion_mask.sum(-1).clamp(min=1.0)
        log_importance_weights = log_importance_weights.unsqueeze(-1)
    else:
        log_importance_weights = log_ratio
    coef_1 = torch.exp(log_importance_weights)
    coef_2 = torch.clamp(coef_1, 1 - epsilon_low, 1 + epsilon_high)
    per_token_loss1 = coef_1 * advantages.unsqueeze(1)
    per_token_loss2 = coef_2 * advantages.unsqueeze(1)
    per_token_loss = -torch.min(per_token_loss1, per_token_loss2)
    per_token_loss = per_token_loss + beta * per_token_kl
    loss = ((per_token_loss * completion_mask).sum(-1) / completion_mask.sum(-1).clamp(min=1.0)).mean()
    return loss, advantages, per_token_kl, std_grouped_rewards
