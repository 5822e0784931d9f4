"""Trace -> training samples: what `prime_rl.orchestrator.trajectories` (src/prime_rl/orchestrator/trajectories.py:69-160) and
`TrainSink._prune_zero_advantages` (orchestrator/train_sink.py:36-61) compute for text rollouts, written for flat arrays.

The contract (the reference's, pinned by tests/test_trajectories_cpu.py): a trace is a graph of token nodes, a branch a linear path
through it, and one `TrainingSample` leaves per branch that still has something to train.  A node that carries a training signal
delivers it ONCE — in the first branch that contains it; in every later branch its tokens are context (mask False, weight 0).  That
holds per stream: the rl mask, and each of the graph-native weight streams `rl` / `ce` / `ref_kl` independently.

Objects are duck-typed verifiers.v1 shapes (`trace.branches[*].{nodes, token_ids, logprobs, reference_logprobs, advantages}`,
`node.{token_ids, mask, sampled, loss_weights}`).  Per-node Python only touches node boundaries; per-token work is numpy on the
flattened branch (a 32k-token agentic branch is a handful of array ops, not 32k loop iterations)."""

from __future__ import annotations

from collections.abc import Iterator

import numpy as np

from ..transports.rollouts.types import TrainingSample

STREAMS = ("rl", "ce", "ref_kl")


class _Spent:
    """Which graph nodes already delivered their signal for one stream (identity-keyed: the same node object sits in many branches)."""

    def __init__(self) -> None:
        self._ids: set[int] = set()

    def __contains__(self, node) -> bool:
        return id(node) in self._ids

    def spend(self, node) -> None:
        self._ids.add(id(node))


def _train_mask(branch, spent: _Spent) -> np.ndarray:
    """The branch's trainable-token mask with every already-spent sampled node blanked."""
    parts = []
    for node in branch.nodes:
        m = np.asarray(node.mask, dtype=bool)
        if node.sampled and m.any():
            if node in spent:
                m = np.zeros(m.shape, dtype=bool)
            else:
                spent.spend(node)
        parts.append(m)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=bool)


def iter_trainable_branches(trace) -> Iterator[tuple[object, list[bool]]]:
    """(branch, mask) for every branch that yields a sample (trajectories.py:69-89 semantics)."""
    spent = _Spent()
    for branch in trace.branches:
        mask = _train_mask(branch, spent)
        if mask.any():
            yield branch, mask.tolist()


def _weight_stream(branch, stream: str, spent: _Spent) -> list[float] | None:
    """One graph-native loss-weight stream flattened over the branch, or None when it carries no weight at all (trajectories.py:92-108
    semantics: a node's declared weights count once; an undeclared or spent node contributes zeros; lengths must match the tokens)."""
    parts = []
    for node in branch.nodes:
        n_tok = len(node.token_ids)
        declared = (node.loss_weights or {}).get(stream)
        if declared is None or node in spent:
            parts.append(np.zeros(n_tok))
            continue
        w = np.asarray(declared, dtype=np.float64)
        if w.size != n_tok:
            raise ValueError(f"loss weight stream {stream!r} must align with node token_ids: got {w.size}, expected {n_tok}")
        parts.append(w)
        if w.any():
            spent.spend(node)
    flat = np.concatenate(parts) if parts else np.zeros(0)
    return flat.tolist() if flat.any() else None


def trace_to_samples(trace, *, env_name: str = "") -> list[TrainingSample]:
    """One `TrainingSample` per trainable branch (trajectories.py:111-160); `temperatures` is filled by the sink afterwards."""
    rl_spent = _Spent()
    stream_spent = {s: _Spent() for s in STREAMS}
    samples: list[TrainingSample] = []
    for branch in trace.branches:
        mask = _train_mask(branch, rl_spent)
        if not mask.any():
            continue
        if getattr(branch, "multi_modal_data", None) is not None:
            raise NotImplementedError("multimodal branches are outside the dense GRPO path")
        streams = {s: _weight_stream(branch, s, stream_spent[s]) for s in STREAMS}
        samples.append(TrainingSample(token_ids=branch.token_ids, mask=mask.tolist(), logprobs=branch.logprobs, temperatures=[], env_name=env_name,
                                      ref_logprobs=getattr(branch, "reference_logprobs", None), rl_weights=streams["rl"], ce_weights=streams["ce"],
                                      ref_kl_weights=streams["ref_kl"], advantages=getattr(branch, "advantages", None)))
    return samples


def _nonzero(weights) -> bool:
    return weights is not None and bool(np.any(np.asarray(weights, dtype=np.float64) != 0.0))


def prune_zero_advantages(sample: TrainingSample) -> bool:
    """Trainable tokens whose advantage is exactly 0 leave the rl component (weight 0: out of its mask AND its denominator).  Returns
    whether the sample still trains anything through any component (train_sink.py:36-61 semantics)."""
    if sample.advantages is None:
        return True
    mask = np.asarray(sample.mask, dtype=bool)
    adv = np.asarray(sample.advantages, dtype=np.float64)
    rl_w = mask.astype(np.float64) if sample.rl_weights is None else np.asarray(sample.rl_weights, dtype=np.float64)
    if not (mask.size == adv.size == rl_w.size):
        raise ValueError(f"mask / advantages / rl_weights must have one entry per token: {mask.size}, {adv.size}, {rl_w.size}")
    dead = mask & (adv == 0.0) & (rl_w != 0.0)
    if not dead.any():
        return True
    rl_w[dead] = 0.0
    sample.rl_weights = rl_w.tolist()
    return bool((mask & (rl_w != 0.0)).any()) or _nonzero(sample.ce_weights) or _nonzero(sample.ref_kl_weights)
