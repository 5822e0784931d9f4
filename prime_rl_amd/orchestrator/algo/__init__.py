"""Registry keyed on `algo.type` (src/prime_rl/orchestrator/algo/__init__.py:42-62).  The advantage-producing credit rules that
share the `score_group` hook are here: grpo, max_rl, hierarchical_grpo, rae.  echo / opd / opsd / sft need a
reference-model or teacher inference pool and stay with the reference."""

from .base import Algorithm, iter_trainable_traces
from .grpo import GRPOAlgoConfig, GRPOAlgorithm, LengthPenaltyConfig, MaxRLAlgorithm, group_advantages
from .hierarchical_grpo import HierarchicalGRPOAlgoConfig, HierarchicalGRPOAlgorithm
from .rae import RAEAlgoConfig, RAEAlgorithm
from .routing import assign_advantages, stamp_loss_routing

ALGORITHM_CLASSES: dict[str, type[Algorithm]] = {"grpo": GRPOAlgorithm, "max_rl": MaxRLAlgorithm, "hierarchical_grpo": HierarchicalGRPOAlgorithm,
                                                 "rae": RAEAlgorithm}


def build_algorithm(config, policy_pool=None, device: str | None = None) -> Algorithm:
    """`device`: None keeps the class default (host arithmetic, like the reference's orchestrator); "cuda" scores on the MI355X."""
    cls = ALGORITHM_CLASSES[config.type]
    assert cls.action_loss_type == getattr(config, "action_loss_type", "rl")
    return cls(config, policy_pool) if device is None else cls(config, policy_pool, device=device)


__all__ = ["Algorithm", "GRPOAlgorithm", "MaxRLAlgorithm", "HierarchicalGRPOAlgorithm", "RAEAlgorithm", "GRPOAlgoConfig", "LengthPenaltyConfig",
           "HierarchicalGRPOAlgoConfig", "RAEAlgoConfig", "ALGORITHM_CLASSES", "build_algorithm", "assign_advantages", "stamp_loss_routing",
           "iter_trainable_traces", "group_advantages"]
