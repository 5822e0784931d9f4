"""Registry keyed on `algo.type` (src/prime_rl/orchestrator/algo/__init__.py:42-62).  Only the GRPO-family credit rules are on
the §8 hot path; echo / rae / hierarchical_grpo / opd / opsd / sft share the same hook and stay with the reference."""

from .base import Algorithm, iter_trainable_traces
from .grpo import GRPOAlgoConfig, GRPOAlgorithm, LengthPenaltyConfig, MaxRLAlgorithm, group_advantages
from .routing import assign_advantages, stamp_loss_routing

ALGORITHM_CLASSES: dict[str, type[Algorithm]] = {"grpo": GRPOAlgorithm, "max_rl": MaxRLAlgorithm}


def build_algorithm(config, policy_pool=None) -> Algorithm:
    cls = ALGORITHM_CLASSES[config.type]
    assert cls.action_loss_type == getattr(config, "action_loss_type", "rl")
    return cls(config, policy_pool)


__all__ = ["Algorithm", "GRPOAlgorithm", "MaxRLAlgorithm", "GRPOAlgoConfig", "LengthPenaltyConfig", "ALGORITHM_CLASSES", "build_algorithm",
           "assign_advantages", "stamp_loss_routing", "iter_trainable_traces", "group_advantages"]
