"""Trace -> training samples — mirrors `prime_rl.orchestrator.trajectories` (src/prime_rl/orchestrator/trajectories.py:69-160) and
`TrainSink._prune_zero_advantages` (orchestrator/train_sink.py:36-61) for text rollouts.  Host-side list plumbing on duck-typed
verifiers objects (`trace.branches[*].{nodes, token_ids, logprobs, reference_logprobs, advantages}`); a sampled node shared by
several branches trains once (first branch containing it), later branches carry it as context."""

from __future__ import annotations

from collections.abc import Iterator

from ..transports.rollouts.types import TrainingSample


def iter_trainable_branches(trace) -> Iterator[tuple[object, list[bool]]]:
    """Each branch that yields a training sample, with its trainable-token mask (trajectories.py:69-89)."""
    trained: set[int] = set()
    for branch in trace.branches:
        mask: list[bool] = []
        for node in branch.nodes:
            if node.sampled and any(node.mask) and id(node) in trained:
                mask.extend([False] * len(node.mask))
            else:
                if node.sampled and any(node.mask):
                    trained.add(id(node))
                mask.extend(node.mask)
        if any(mask):
            yield branch, mask


def _loss_weights(branch, name: str, trained: set[int]) -> list[float] | None:
    """Flatten one graph-native loss stream, training each shared node once (trajectories.py:92-108)."""
    weights: list[float] = []
    for node in branch.nodes:
        nw = (node.loss_weights or {}).get(name)
        if nw is None or id(node) in trained:
            weights.extend([0.0] * len(node.token_ids))
            continue
        if len(nw) != len(node.token_ids):
            raise ValueError(f"loss weight stream {name!r} must align with node token_ids: got {len(nw)}, expected {len(node.token_ids)}")
        weights.extend(nw)
        if any(nw):
            trained.add(id(node))
    return weights if any(weights) else None


def trace_to_samples(trace, *, env_name: str = "") -> list[TrainingSample]:
    """One `TrainingSample` per trainable branch (trajectories.py:111-160); `temperatures` is filled by the sink afterwards."""
    samples: list[TrainingSample] = []
    trained = {"rl": set(), "ce": set(), "ref_kl": set()}
    for branch, mask in iter_trainable_branches(trace):
        if getattr(branch, "multi_modal_data", None) is not None:
            raise NotImplementedError("multimodal branches are outside the dense GRPO path")
        samples.append(TrainingSample(token_ids=branch.token_ids, mask=mask, logprobs=branch.logprobs, temperatures=[], env_name=env_name,
                                      ref_logprobs=getattr(branch, "reference_logprobs", None), rl_weights=_loss_weights(branch, "rl", trained["rl"]),
                                      ce_weights=_loss_weights(branch, "ce", trained["ce"]), ref_kl_weights=_loss_weights(branch, "ref_kl", trained["ref_kl"]),
                                      advantages=getattr(branch, "advantages", None)))
    return samples


def prune_zero_advantages(sample: TrainingSample) -> bool:
    """Zero-advantage trainable tokens leave the RL component (rl weight 0 -> out of mask and denominator); returns whether the sample
    still trains anything (train_sink.py:36-61)."""
    if sample.advantages is None:
        return True
    rl_weights = [1.0 if t else 0.0 for t in sample.mask] if sample.rl_weights is None else list(sample.rl_weights)
    changed = False
    for i, (t, a, w) in enumerate(zip(sample.mask, sample.advantages, rl_weights, strict=True)):
        if t and a == 0.0 and w != 0.0:
            rl_weights[i] = 0.0
            changed = True
    if not changed:
        return True
    sample.rl_weights = rl_weights
    has_rl = any(t and w != 0.0 for t, w in zip(sample.mask, rl_weights, strict=True))
    has_ce = sample.ce_weights is not None and any(w != 0.0 for w in sample.ce_weights)
    has_rk = sample.ref_kl_weights is not None and any(w != 0.0 for w in sample.ref_kl_weights)
    return has_rl or has_ce or has_rk
