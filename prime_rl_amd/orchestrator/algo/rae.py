"""Role-conditioned advantage estimation — mirrors `prime_rl.orchestrator.algo.rae` (src/prime_rl/orchestrator/algo/rae.py:16-43):
credit = reward minus a per-agent exponential-moving-average baseline, each trace scored BEFORE its reward updates the baseline.

This rule is a strictly sequential scalar recurrence over the cohort's traces (the k-th advantage depends on the previous k-1
rewards of that agent), a handful of floats per group: it stays on the host on purpose — there is nothing to coalesce and a kernel
launch would cost more than the arithmetic.  Double-precision Python floats, like the reference."""

from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass

from .base import Algorithm, iter_trainable_traces
from .routing import assign_advantages


@dataclass
class RAEAlgoConfig:
    type: str = "rae"
    action_loss_type: str = "rl"
    decay: float = 0.95


class RAEAlgorithm(Algorithm):
    def __init__(self, config: RAEAlgoConfig | None = None, policy_pool=None):
        super().__init__(config, policy_pool)
        self.decay = getattr(config, "decay", 0.95)
        self.baselines: dict[str, float] = defaultdict(float)

    async def score_group(self, episodes: list) -> None:
        for _, trace in iter_trainable_traces(episodes):
            name = trace.agent.name
            baseline = self.baselines[name]
            assign_advantages(trace, trace.reward - baseline)
            self.baselines[name] = self.decay * baseline + (1.0 - self.decay) * trace.reward
