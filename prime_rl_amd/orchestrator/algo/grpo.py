"""GRPO / MaxRL advantage plugins — mirror `prime_rl.orchestrator.algo.grpo` and `.max_rl`
(src/prime_rl/orchestrator/algo/grpo.py:16-45, max_rl.py:10-28): same classes, same `score_group` hook, same config fields
(`length_penalty.{num_output_tokens_weight, num_input_tokens_weight, num_turns_weight}`).

The arithmetic is `pm_grpo_advantage` (one wave64 per group).  `score_group` handles one cohort like the reference;
`score_groups` scores MANY cohorts in one launch — the form the trainer-side pipeline uses when raw rewards are shipped and the
`[n_groups * group_size]` reward buffer already lives in HBM (north_star: "coalesced HBM reads of the flat rollout
token/reward buffers")."""

from __future__ import annotations

from dataclasses import dataclass

import torch

from ... import _lib, ops
from .base import Algorithm, iter_trainable_traces
from .routing import assign_advantages


@dataclass
class LengthPenaltyConfig:
    """configs/algorithm.py:95-106 defaults."""

    num_output_tokens_weight: float = 0.25
    num_input_tokens_weight: float = 0.1
    num_turns_weight: float = 0.1


@dataclass
class GRPOAlgoConfig:
    type: str = "grpo"
    action_loss_type: str = "rl"
    length_penalty: LengthPenaltyConfig | None = None


def group_advantages(rewards: torch.Tensor, group_offsets: torch.Tensor, mode: int = _lib.PM_ADV_GRPO, length_penalty: LengthPenaltyConfig | None = None,
                     num_output_tokens: torch.Tensor | None = None, num_total_tokens: torch.Tensor | None = None,
                     num_turns: torch.Tensor | None = None) -> torch.Tensor:
    """Flat rewards [n_rollouts] + int32 group offsets [n_groups+1] (device tensors) -> advantages [n_rollouts]."""
    if length_penalty is None:
        return ops.grpo_advantage(rewards, group_offsets, mode)
    w = (length_penalty.num_output_tokens_weight, length_penalty.num_input_tokens_weight, length_penalty.num_turns_weight)
    return ops.grpo_advantage(rewards, group_offsets, mode, num_output_tokens, num_total_tokens, num_turns, w)


class GRPOAlgorithm(Algorithm):
    """credit = reward - group mean (optionally length-shaped); action tokens feed the `rl` loss (grpo.py:16-45)."""

    mode = _lib.PM_ADV_GRPO

    def __init__(self, config: GRPOAlgoConfig | None = None, policy_pool=None, device: str = "cuda"):
        super().__init__(config, policy_pool)
        self.length_penalty = getattr(config, "length_penalty", None)
        self.device = device

    def score_groups(self, groups: list[list]) -> None:
        """Score several cohorts (lists of episodes) with ONE kernel launch."""
        traces_per_group = [[trace for _, trace in iter_trainable_traces(eps)] for eps in groups]
        flat = [t for ts in traces_per_group for t in ts]
        if not flat:
            return
        offs = [0]
        for ts in traces_per_group:
            offs.append(offs[-1] + len(ts))
        dev = self.device
        rewards = torch.tensor([t.reward for t in flat], dtype=torch.float32, device=dev)
        offsets = torch.tensor(offs, dtype=torch.int32, device=dev)
        kw = {}
        if self.length_penalty is not None:
            kw = dict(length_penalty=self.length_penalty,
                      num_output_tokens=torch.tensor([t.num_output_tokens for t in flat], dtype=torch.float32, device=dev),
                      num_total_tokens=torch.tensor([t.num_total_tokens for t in flat], dtype=torch.float32, device=dev),
                      num_turns=torch.tensor([t.num_turns for t in flat], dtype=torch.float32, device=dev))
        adv = group_advantages(rewards, offsets, self.mode, **kw).tolist()
        for trace, a in zip(flat, adv, strict=True):
            assign_advantages(trace, a)

    async def score_group(self, episodes: list) -> None:
        self.score_groups([episodes])


class MaxRLAlgorithm(GRPOAlgorithm):
    """(reward - group mean) / group mean, zero when the mean is <= 0 (max_rl.py:10-28)."""

    mode = _lib.PM_ADV_MAXRL

    def __init__(self, config=None, policy_pool=None, device: str = "cuda"):
        super().__init__(config, policy_pool, device)
        self.length_penalty = None
