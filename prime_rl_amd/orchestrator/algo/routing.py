"""Training annotations and loss routing — mirrors `prime_rl.orchestrator.algo.routing`
(src/prime_rl/orchestrator/algo/routing.py:11-84).  Host-side list plumbing on duck-typed verifiers objects
(`trace.nodes[*].{mask, token_ids, advantages}`); the arithmetic lives in `grpo.py` / the K3 kernels."""

from __future__ import annotations


def assign_advantages(trace, values: float | list[float]) -> None:
    """Assign credit in compact sampled-token order across the trace graph (routing.py:11-34)."""
    sampled: list[tuple[object, int]] = []  # (node, number of mask-True tokens) for nodes that carry sampled tokens
    for node in trace.nodes:
        count = sum(1 for m in node.mask if m)
        if count == 0:
            continue
        if len(node.mask) != len(node.token_ids):
            raise ValueError(f"node masks must align with token ids in trace {trace.id!r}: got {len(node.mask)}, expected {len(node.token_ids)}")
        sampled.append((node, count))
    total = sum(count for _, count in sampled)
    scalar = isinstance(values, (int, float))
    if not scalar and len(values) != total:
        raise ValueError(f"advantages must align with trace {trace.id!r}'s sampled tokens: got {len(values)}, expected {total}")
    cursor = 0
    for node, count in sampled:
        node.advantages = [float(values)] * count if scalar else [float(v) for v in values[cursor:cursor + count]]
        cursor += count


def scalar_advantage(trace) -> float | None:
    """Mean nonzero token advantage, or zero for an assigned-zero trace (routing.py:53-59)."""
    advantages = [v for node in trace.nodes for v in (node.advantages or [])]
    if not advantages:
        return None
    nonzero = [v for v in advantages if v != 0.0]
    return sum(nonzero) / len(nonzero) if nonzero else 0.0


def is_trainable(trace) -> bool:
    return any(v != 0.0 for node in trace.nodes for v in (node.advantages or []))


def stamp_loss_routing(sample, action_loss_type: str) -> None:
    """Route action tokens into the algorithm's declared loss component (routing.py:67-84)."""
    if action_loss_type == "rl":
        return
    seq_len = len(sample.token_ids)
    sample.rl_weights = [0.0] * seq_len
    action_weights = sample.ce_weights if action_loss_type == "ce" and sample.ce_weights is not None else [0.0] * seq_len
    for i, trains in enumerate(sample.mask):
        if trains:
            action_weights[i] = 1.0
    if action_loss_type == "ce":
        sample.ce_weights = action_weights
    else:
        assert action_loss_type == "ref_kl"
        sample.ref_kl_weights = action_weights
