"""CPU suite: trace -> samples plumbing (shared nodes train once, stream flattening, zero-advantage pruning)."""

from types import SimpleNamespace as NS

from prime_rl_amd.orchestrator.trajectories import prune_zero_advantages, trace_to_samples
from prime_rl_amd.transports.rollouts.types import TrainingSample


def _node(tokens, mask, sampled=True, lw=None):
    return NS(token_ids=tokens, mask=mask, sampled=sampled, loss_weights=lw)


def test_shared_node_trains_once_and_streams_flatten():
    prompt = _node([1, 2], [False, False], sampled=False)
    shared = _node([3, 4], [True, True])
    a = _node([5], [True], lw={"ce": [1.0]})
    b = _node([6, 7], [True, False])
    mk = lambda nodes: NS(nodes=nodes, token_ids=[t for n in nodes for t in n.token_ids], logprobs=[-0.5] * sum(len(n.token_ids) for n in nodes),
                          reference_logprobs=None, advantages=[0.1] * sum(len(n.token_ids) for n in nodes), multi_modal_data=None)
    trace = NS(branches=[mk([prompt, shared, a]), mk([prompt, shared, b])])
    s = trace_to_samples(trace, env_name="env")
    assert len(s) == 2
    assert s[0].mask == [False, False, True, True, True] and s[0].ce_weights == [0.0, 0.0, 0.0, 0.0, 1.0] and s[0].rl_weights is None
    assert s[1].mask == [False, False, False, False, True, False], "the shared sampled node is context in the second branch"
    assert s[1].ce_weights is None and s[1].env_name == "env" and s[1].token_ids == [1, 2, 3, 4, 6, 7]


def test_scalar_credit_reaches_the_training_sample_like_the_reference():
    """The reference's known answers (tests/unit/orchestrator/test_advantage.py:248-270): a scalar broadcasts over the mask-True tokens,
    mask-False positions stay 0.0 in the sample's per-token advantages, a list of the wrong length is rejected."""
    import pytest

    from prime_rl_amd.orchestrator.algo.base import SimpleNode, SimpleTrace
    from prime_rl_amd.orchestrator.algo.routing import assign_advantages

    # one masked prompt token + 2 sampled tokens
    trace = SimpleTrace("t", 0.0, [SimpleNode([11], [False], sampled=False), SimpleNode([12, 13], [True, True])])
    assign_advantages(trace, 0.7)
    assert trace_to_samples(trace)[0].advantages == [0.0, 0.7, 0.7]
    # prompt (masked) + 1 sampled + 1 observation token (masked) inside the sampled node's turn: mask [F, T, F]
    trace = SimpleTrace("t", 0.0, [SimpleNode([11], [False], sampled=False), SimpleNode([12, 13], [True, False])])
    assign_advantages(trace, 0.7)
    s = trace_to_samples(trace)[0]
    assert s.advantages == [0.0, 0.7, 0.0] and s.mask == [False, True, False] and s.token_ids == [11, 12, 13]
    with pytest.raises(ValueError, match="align"):
        assign_advantages(SimpleTrace("t", 0.0, [SimpleNode([11], [False], sampled=False), SimpleNode([12, 13], [True, True])]), [0.5])


def test_prune_zero_advantages():
    s = TrainingSample(token_ids=[1, 2, 3], mask=[False, True, True], logprobs=[0.0] * 3, temperatures=[1.0] * 3, env_name="e", advantages=[0.0, 0.0, 0.5])
    assert prune_zero_advantages(s) and s.rl_weights == [0.0, 0.0, 1.0]
    z = TrainingSample(token_ids=[1, 2], mask=[True, True], logprobs=[0.0] * 2, temperatures=[1.0] * 2, env_name="e", advantages=[0.0, 0.0])
    assert not prune_zero_advantages(z), "nothing left to train"
    z.ce_weights = [1.0, 0.0]
    z.rl_weights = None
    assert prune_zero_advantages(z)
