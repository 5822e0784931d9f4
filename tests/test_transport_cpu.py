"""CPU suite: rollout wire contract (field order / omit-defaults / round trip), filesystem transport, algorithm routing."""

import asyncio
from dataclasses import fields

import msgpack
import pytest

from prime_rl_amd.orchestrator.algo.base import SimpleNode, SimpleTrace
from prime_rl_amd.orchestrator.algo.routing import assign_advantages, scalar_advantage, stamp_loss_routing
from prime_rl_amd.transports.rollouts import (FileSystemMicroBatchReceiver, FileSystemMicroBatchSender, MicroBatch, TrainingSample, decode_micro_batches,
                                              encode_micro_batches)
from prime_rl_amd.transports.rollouts.types import to_wire


def _mb(n=5, **kw):
    return MicroBatch(input_ids=list(range(n)), loss_mask=[True] * n, advantages=[0.5] * n, inference_logprobs=[-1.0] * n, position_ids=list(range(n)),
                      sequence_lengths=[n], temperatures=[1.0] * n, env_names=["e"] * n, seq_lens=[n], **kw)


def test_field_order_matches_reference_wire():
    # src/prime_rl/transports/rollouts/types.py:74-99 (array_like structs: order IS the wire format)
    assert [f.name for f in fields(MicroBatch)] == ["input_ids", "loss_mask", "advantages", "inference_logprobs", "position_ids", "sequence_lengths",
                                                    "temperatures", "env_names", "seq_lens", "ref_logprobs", "routed_experts", "mm_kwargs",
                                                    "mm_token_type_ids", "rl_weights", "ce_weights", "ref_kl_weights"]
    assert [f.name for f in fields(TrainingSample)][:5] == ["token_ids", "mask", "logprobs", "temperatures", "env_name"]


def test_omit_defaults_and_round_trip():
    plain = _mb()
    assert len(to_wire(plain)) == 9  # trailing None streams are dropped: the plain GRPO wire stays small (types.py:63-68)
    w = _mb(ce_weights=[0.0, 1.0, 0.0, 1.0, 0.0])
    assert len(to_wire(w)) == 15 and to_wire(w)[9] is None
    buf = encode_micro_batches([plain, w])
    assert isinstance(msgpack.unpackb(buf)[0], list)
    back = decode_micro_batches(buf)
    assert back[0] == plain and back[1] == w


def test_filesystem_transport_and_dataloader(tmp_path):
    from prime_rl_amd.trainer.rl.data import DataLoader

    sender = FileSystemMicroBatchSender(tmp_path, data_world_size=2, current_step=3)
    grid = [[_mb(4), _mb(6)], [_mb(5), _mb(5, rl_weights=[1.0] * 5)]]
    asyncio.run(sender.send(grid))
    assert (tmp_path / "rollouts" / "step_3" / "rank_1.bin").exists() and not list((tmp_path / "rollouts" / "step_3").glob("*.tmp"))
    recv = FileSystemMicroBatchReceiver(tmp_path, data_rank=1, current_step=3)
    assert recv.can_receive()
    dl = DataLoader(tmp_path, start_step=3, dp_world_size=2, dp_rank=1)
    dl.wait_for_batch()
    mbs = dl.get_batch()
    assert len(mbs) == 2 and mbs[0]["input_ids"].shape == (1, 5) and mbs[1]["rl_weights"].shape == (1, 5) and mbs[0]["rl_weights"] is None
    assert mbs[0]["loss_mask"].dtype.is_floating_point is False and mbs[0]["seq_lens"].tolist() == [5]
    with pytest.raises(AssertionError):
        asyncio.run(sender.send([[_mb()]]))


def test_assign_advantages_and_routing():
    # reference tests/unit/orchestrator/test_advantage.py:248-270: scalar broadcast in compact sampled-token order; misalignment raises
    tr = SimpleTrace("t", 1.0, [SimpleNode([1, 2, 3], [False, True, True]), SimpleNode([4, 5], [False, False]), SimpleNode([6], [True])])
    assign_advantages(tr, 0.7)
    assert tr.nodes[0].advantages == [0.7, 0.7] and tr.nodes[1].advantages is None and tr.nodes[2].advantages == [0.7]
    assert scalar_advantage(tr) == pytest.approx(0.7)
    assign_advantages(tr, [0.1, 0.2, 0.3])
    assert tr.nodes[2].advantages == [0.3]
    with pytest.raises(ValueError):
        assign_advantages(tr, [0.1])
    s = TrainingSample(token_ids=[1, 2, 3], mask=[False, True, True], logprobs=[0.0] * 3, temperatures=[1.0] * 3, env_name="e")
    stamp_loss_routing(s, "rl")
    assert s.rl_weights is None
    stamp_loss_routing(s, "ce")
    assert s.rl_weights == [0.0] * 3 and s.ce_weights == [0.0, 1.0, 1.0]
