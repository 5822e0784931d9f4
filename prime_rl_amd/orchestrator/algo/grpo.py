"""GRPO / MaxRL advantage plugins — mirror `prime_rl.orchestrator.algo.grpo` and `.max_rl`
(src/prime_rl/orchestrator/algo/grpo.py:16-45, max_rl.py:10-28): same classes, same `score_group` hook, same config fields
(`length_penalty.{num_output_tokens_weight, num_input_tokens_weight, num_turns_weight}`).

Two placements of the same arithmetic, chosen by `device` — never by availability (nothing falls back silently):
  * `device="cpu"` (the default): host arithmetic in f32 torch ops, like the reference, whose orchestrator process owns no
    accelerator and scores one cohort of a few dozen rewards at a time from its event loop (grpo.py:25-45).  Launching a kernel, three
    blocking H2D copies and a D2H `.tolist()` for tens of floats would only stall that loop.
  * `device="cuda"`: `pm_grpo_advantage` (one wave64 per group) through `group_advantages` / `score_groups` — MANY cohorts in one
    launch, the form the trainer-side pipeline uses when raw rewards are shipped and the `[n_groups * group_size]` reward buffer already
    lives in HBM (north_star: "coalesced HBM reads of the flat rollout token/reward buffers").
tests/test_kernels_gpu.py::test_grpo_advantage pins the kernel to the oracle; tests/test_routing_cpu.py pins the host path to the same
known answers (zero-sum, singleton -> 0, MaxRL [3,-1,-1,-1], length-penalty monotonicity)."""

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

    def __init__(self, config: GRPOAlgoConfig | None = None, policy_pool=None, device: str = "cpu"):
        super().__init__(config, policy_pool)
        self.length_penalty = getattr(config, "length_penalty", None)
        self.device = device

    def _host_advantages(self, flat: list, offs: list[int]) -> list[float]:
        """Per-cohort advantages on the host, f32: r - mean(r); with a length penalty r is first lowered by mean(r) times the weighted,
        max-normalised token / turn counts (grpo.py:27-45); MaxRL divides the centred reward by the cohort mean, 0 when that mean is <= 0
        (max_rl.py:22-28)."""
        out: list[float] = []
        lp = self.length_penalty
        for a, b in zip(offs[:-1], offs[1:]):
            cohort = flat[a:b]
            if not cohort:
                continue
            r = torch.tensor([t.reward for t in cohort], dtype=torch.float32)
            mean = r.mean()
            if self.mode == _lib.PM_ADV_MAXRL:
                adv = (r - mean) / mean if float(mean) > 0 else torch.zeros_like(r)
            elif lp is None:
                adv = r - mean
            else:
                n_out = torch.tensor([t.num_output_tokens for t in cohort], dtype=torch.float32)
                n_in = torch.tensor([t.num_total_tokens for t in cohort], dtype=torch.float32) - n_out
                n_turn = torch.tensor([t.num_turns for t in cohort], dtype=torch.float32)
                frac = (lp.num_output_tokens_weight * (n_out / n_out.max().clamp(min=1)) + lp.num_input_tokens_weight * (n_in / n_in.max().clamp(min=1))
                        + lp.num_turns_weight * (n_turn / n_turn.max().clamp(min=1)))
                shaped = r - mean * frac
                adv = shaped - shaped.mean()
            out += adv.tolist()
        return out

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
        if torch.device(dev).type == "cpu":
            for trace, a in zip(flat, self._host_advantages(flat, offs), strict=True):
                assign_advantages(trace, a)
            return
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

    def __init__(self, config=None, policy_pool=None, device: str = "cpu"):
        super().__init__(config, policy_pool, device)
        self.length_penalty = None
