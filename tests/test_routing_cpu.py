"""CPU suite: loss routing and credit assignment against the reference's own known answers
(tests/unit/orchestrator/test_algorithms.py:97-240), restated on the duck-typed shapes of orchestrator/algo/base.py because
`verifiers.v1` is an un-vendored submodule of the reference.  Host-side list plumbing: exact equality."""

import pytest

from prime_rl_amd.orchestrator.algo.base import SimpleNode, SimpleTrace
from prime_rl_amd.orchestrator.algo.routing import assign_advantages, stamp_loss_routing
from prime_rl_amd.orchestrator.trajectories import trace_to_samples
from prime_rl_amd.transports.rollouts.types import TrainingSample


def _sample(ce_weights=None) -> TrainingSample:
    # 2 prompt tokens (mask False), then a 4-token completion with one env-provided observation token (position 4, mask False)
    return TrainingSample(token_ids=[1, 2, 3, 4, 5, 6], mask=[False, False, True, True, False, True], logprobs=[0.0, 0.0, -0.1, -0.2, 0.0, -0.3],
                          temperatures=[], env_name="test-env", ce_weights=ce_weights)


def test_stamp_loss_routing_uniform_rl():
    s = _sample()
    stamp_loss_routing(s, "rl")
    assert s.rl_weights is None and s.ce_weights is None and s.ref_kl_weights is None  # hot path: absent streams = rl weight 1 on the mask


def test_stamp_loss_routing_ref_kl_and_ce_actions():
    s = _sample()
    stamp_loss_routing(s, "ref_kl")
    assert s.rl_weights == [0.0] * 6 and s.ref_kl_weights == [0.0, 0.0, 1.0, 1.0, 0.0, 1.0] and s.ce_weights is None
    s = _sample()
    stamp_loss_routing(s, "ce")
    assert s.rl_weights == [0.0] * 6 and s.ce_weights == [0.0, 0.0, 1.0, 1.0, 0.0, 1.0] and s.ref_kl_weights is None


def test_stamp_loss_routing_keeps_and_merges_algorithm_written_ce_stream():
    s = _sample(ce_weights=[0.0, 0.0, 0.0, 0.0, 0.1, 0.0])
    stamp_loss_routing(s, "rl")  # rl routing must not clobber a ce stream written at group time
    assert s.rl_weights is None and s.ce_weights == [0.0, 0.0, 0.0, 0.0, 0.1, 0.0] and s.ref_kl_weights is None
    s = _sample(ce_weights=[0.0, 0.0, 0.0, 0.0, 0.1, 0.0])
    stamp_loss_routing(s, "ce")  # action tokens merge into the existing stream
    assert s.rl_weights == [0.0] * 6 and s.ce_weights == [0.0, 0.0, 1.0, 1.0, 0.1, 1.0] and s.ref_kl_weights is None


def _trace() -> SimpleTrace:
    # user [1,2] (masked) -> assistant [3,4] (sampled) -> tool [5] (masked) -> assistant [6] (sampled)
    return SimpleTrace("t", 0.0, [SimpleNode([1, 2], [False, False], sampled=False), SimpleNode([3, 4], [True, True]),
                                  SimpleNode([5], [False], sampled=False), SimpleNode([6], [True])])


def test_assign_advantages_full_length_stream_and_node_slices():
    tr = _trace()
    assign_advantages(tr, [0.5, -0.5, 1.0])
    assert tr.nodes[1].advantages == [0.5, -0.5] and tr.nodes[3].advantages == [1.0]
    assert trace_to_samples(tr)[0].advantages == [0.0, 0.0, 0.5, -0.5, 0.0, 1.0]
    tr = _trace()
    assign_advantages(tr, 1.0)
    assert trace_to_samples(tr)[0].advantages == [0.0, 0.0, 1.0, 1.0, 0.0, 1.0]


def test_unassigned_advantages_ship_none_and_misaligned_lists_raise():
    assert trace_to_samples(_trace())[0].advantages is None
    with pytest.raises(ValueError, match="align"):
        assign_advantages(_trace(), [0.5])
    with pytest.raises(ValueError, match="align"):
        assign_advantages(_trace(), [0.1, 0.2, 0.3, 0.4])
