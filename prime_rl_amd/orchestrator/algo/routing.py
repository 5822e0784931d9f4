"""Training annotations and loss routing: the contract of `prime_rl.orchestrator.algo.routing`
(src/prime_rl/orchestrator/algo/routing.py:11-84) on duck-typed verifiers objects (`trace.nodes[*].{mask, token_ids,
advantages}`, `sample.{token_ids, mask, rl_weights, ce_weights, ref_kl_weights}`).  Written from the behaviour the reference's tests
pin (tests/unit/orchestrator/test_advantage.py:248-270, test_algorithms.py:112-150), not from its source; the arithmetic lives in
`grpo.py` / the K3 kernels."""

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


def _token_credit(trace) -> list[float]:
    """Every per-token advantage the trace carries, node after node."""
    credit: list[float] = []
    for node in trace.nodes:
        if node.advantages:
            credit.extend(node.advantages)
    return credit


def scalar_advantage(trace) -> float | None:
    """The trace's advantage as ONE number for logging / filtering: the mean over its nonzero token advantages; 0.0 when credit was
    assigned but is zero everywhere; None when nothing was assigned (routing.py:53-59 semantics, plain sequential float sum)."""
    credit = _token_credit(trace)
    if not credit:
        return None
    live = [c for c in credit if c != 0.0]
    if not live:
        return 0.0
    return sum(live) / len(live)


def is_trainable(trace) -> bool:
    """A trace trains iff some token of it carries nonzero credit (routing.py:62-64 semantics)."""
    return any(c != 0.0 for c in _token_credit(trace))


def stamp_loss_routing(sample, action_loss_type: str) -> None:
    """Send the sample's ACTION tokens (mask True) to the loss component the algorithm declares (routing.py:67-84 semantics):
      "rl"      nothing to do — an absent rl stream already means weight 1 on the mask;
      "ce"      rl stream all-zero; ce weight 1.0 on every action token, on top of whatever graph-native ce weights the sample has;
      "ref_kl"  rl stream all-zero; a fresh ref_kl stream with 1.0 on every action token."""
    if action_loss_type == "rl":
        return
    if action_loss_type not in ("ce", "ref_kl"):
        raise ValueError(f"unknown action loss type {action_loss_type!r}")
    n_tok = len(sample.token_ids)
    acts = [i for i, is_action in enumerate(sample.mask) if is_action]
    sample.rl_weights = [0.0] * n_tok
    stream = list(sample.ce_weights) if (action_loss_type == "ce" and sample.ce_weights is not None) else [0.0] * n_tok
    for i in acts:
        stream[i] = 1.0
    if action_loss_type == "ce":
        sample.ce_weights = stream
    else:
        sample.ref_kl_weights = stream
