"""Hierarchical GRPO for proposer/solver environments — mirrors `prime_rl.orchestrator.algo.hierarchical_grpo`
(src/prime_rl/orchestrator/algo/hierarchical_grpo.py:16-42): same class name, same `score_group` hook, same config field
`episode_agents`.

Credit rule: a trace is compared only with its peers — traces of the same agent role, and for roles listed in `episode_agents`
(normally the solver) only those of the SAME episode; a peer set of one gets zero.  That is GRPO centering over a different
partition of the cohort, so the traces are bucketed on the host (dict insertion order = first appearance, as in the reference) and
all peer sets are centered by ONE `pm_grpo_advantage` launch over the flat reward buffer."""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from ... import _lib
from .base import Algorithm, iter_trainable_traces
from .grpo import group_advantages
from .routing import assign_advantages


@dataclass
class HierarchicalGRPOAlgoConfig:
    type: str = "hierarchical_grpo"
    action_loss_type: str = "rl"
    episode_agents: list[str] = field(default_factory=lambda: ["solver"])


class HierarchicalGRPOAlgorithm(Algorithm):
    def __init__(self, config: HierarchicalGRPOAlgoConfig | None = None, policy_pool=None, device: str = "cpu"):
        super().__init__(config, policy_pool)
        self.episode_agents = set(getattr(config, "episode_agents", ["solver"]))
        self.device = device

    async def score_group(self, episodes: list) -> None:
        peers: dict[tuple, list] = {}
        for episode, trace in iter_trainable_traces(episodes):
            scoped = trace.agent.name in self.episode_agents
            peers.setdefault((trace.agent.name, (getattr(episode, "id", None) or id(episode)) if scoped else None), []).append(trace)
        flat = [t for members in peers.values() for t in members]
        if not flat:
            return
        offs = [0]
        for members in peers.values():
            offs.append(offs[-1] + len(members))
        if torch.device(self.device).type == "cpu":  # host arithmetic like the reference (hierarchical_grpo.py:33-42): r - mean per peer set
            adv: list[float] = []
            for a, b in zip(offs[:-1], offs[1:]):
                r = torch.tensor([t.reward for t in flat[a:b]], dtype=torch.float32)
                adv += (r - r.mean()).tolist()
        else:  # every peer set in one pm_grpo_advantage launch
            rewards = torch.tensor([t.reward for t in flat], dtype=torch.float32, device=self.device)
            offsets = torch.tensor(offs, dtype=torch.int32, device=self.device)
            adv = group_advantages(rewards, offsets, _lib.PM_ADV_GRPO).tolist()
        for trace, a in zip(flat, adv, strict=True):
            assign_advantages(trace, a)
