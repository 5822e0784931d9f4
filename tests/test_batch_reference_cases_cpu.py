"""CPU suite: the explicit packing cases of the reference's own test file (tests/unit/orchestrator/test_batch.py:83-356, text-only
ones) restated against `prime_rl_amd.trainer.batch`.  Integer / list work: exact equality.  (tests/test_batch_cpu.py pins the same
module bit-for-bit against digests of the reference's output; this file keeps the reference's human-readable expectations.)"""

import pytest

from prime_rl_amd.trainer.batch import build_bin_cost, pad_micro_batch, prepare_batch, prepare_sample
from prime_rl_amd.transports.rollouts.types import TrainingSample
from types import SimpleNamespace

FLOPS_CFG = SimpleNamespace(hidden_size=16, num_attention_heads=2, num_key_value_heads=2, vocab_size=32, intermediate_size=64, num_hidden_layers=2, head_dim=8)


def four_token_sample(temperature=1.0, ce_weights=None, rl_weights=None, env_name="test-env") -> TrainingSample:
    return TrainingSample(token_ids=[1, 2, 3, 4], mask=[False, False, True, True], logprobs=[0.0, 0.0, -0.1, -0.2], temperatures=[temperature] * 4,
                          advantages=[0.0, 0.0, 1.0, 1.0], env_name=env_name, ce_weights=ce_weights, rl_weights=rl_weights)


def sized_sample(length: int, env_name="test-env") -> TrainingSample:
    p = length - 1
    return TrainingSample(token_ids=[1] * p + [2], mask=[False] * p + [True], logprobs=[0.0] * p + [-0.1], temperatures=[1.0] * length,
                          advantages=[0.0] * p + [1.0], env_name=env_name)


def flat(grid):
    return [mb for rank in grid for mb in rank]


def trains(mb) -> bool:
    return any(mb.loss_mask)


def test_training_sample_requires_env_name():
    with pytest.raises(TypeError, match="env_name"):
        TrainingSample(token_ids=[1, 2, 3, 4], mask=[False, False, True, True], logprobs=[0.0] * 4, temperatures=[1.0] * 4, advantages=[0.0, 0.0, 1.0, 1.0])


@pytest.mark.parametrize(("n_rollouts", "workers", "per_worker"), [(4, 2, 2), (5, 2, 3), (7, 1, 7), (11, 4, 3)])
def test_every_worker_gets_the_same_number_of_micro_batches(n_rollouts, workers, per_worker):
    grid = prepare_batch(rollouts=[four_token_sample() for _ in range(n_rollouts)], seq_len=4, num_train_workers=workers, bin_cost=build_bin_cost(None))
    assert all(len(rank) == per_worker for rank in grid)
    mbs = flat(grid)
    assert n_rollouts <= len(mbs) < n_rollouts + workers
    real, dummy = [m for m in mbs if trains(m)], [m for m in mbs if not trains(m)]
    assert len(real) == n_rollouts
    for m in real:  # a dummy can land anywhere: identify by content, not position
        assert sum(a != 0.0 for a in m.advantages) == 2 and sum(m.loss_mask) == 2
    for m in dummy:
        assert not any(m.advantages) and not any(m.loss_mask)


def test_pad_micro_batch_preserves_explicit_sequence_lengths():
    padded = pad_micro_batch(prepare_sample(sized_sample(4), seq_len=16), pad_to_multiple_of=6)
    assert len(padded.input_ids) == 6 and padded.sequence_lengths == [6] and padded.seq_lens == [6] and padded.loss_mask[-2:] == [False, False]


def test_split_to_align_avoids_dummy_micro_batches():
    grid = prepare_batch(rollouts=[sized_sample(n) for n in (6, 6, 5, 5, 4, 4)], seq_len=12, num_train_workers=4, bin_cost=build_bin_cost(None))
    assert len(flat(grid)) == 4 and all(trains(m) for m in flat(grid))


def test_token_balancing_without_a_model_config():
    grid = prepare_batch(rollouts=[sized_sample(n) for n in (100, 90, 80, 70)], seq_len=100, num_train_workers=2, bin_cost=build_bin_cost(None))
    assert [sum(len(m.input_ids) for m in rank) for rank in grid] == [170, 170]


def test_flop_aware_balancing_pairs_long_and_short_workloads():
    cost = build_bin_cost(FLOPS_CFG)
    grid = prepare_batch(rollouts=[sized_sample(n) for n in (32, 32, 16, 16, 16, 16)], seq_len=32, num_train_workers=2, bin_cost=cost)
    for rank in grid:
        assert sorted(sorted(m.sequence_lengths) for m in rank) == [[16, 16], [32]]
    assert cost([32]) > cost([16, 16])


def test_flop_aware_split_to_align_splits_the_heaviest_bin():
    grid = prepare_batch(rollouts=[sized_sample(n) for n in (20, 18, 9, 9, 8, 8, 8)], seq_len=64, num_train_workers=4, bin_cost=build_bin_cost(FLOPS_CFG))
    real = [m for m in flat(grid) if trains(m)]
    assert len(real) == 4 and sorted(n for m in real for n in m.sequence_lengths) == [8, 8, 8, 9, 9, 18, 20]
    assert sum(len(m.sequence_lengths) > 1 for m in real) == 3


def test_samples_with_different_temperatures_pack_together():
    grid = prepare_batch(rollouts=[four_token_sample(0.7, env_name="env-a"), four_token_sample(1.1, env_name="env-b")], seq_len=16, num_train_workers=1,
                         bin_cost=build_bin_cost(None))
    (mb,) = flat(grid)
    assert mb.temperatures == [0.7] * 4 + [1.1] * 4 and mb.env_names == ["env-a"] * 4 + ["env-b"] * 4 and mb.seq_lens == [4, 4]


def test_prepare_sample_weight_streams():
    mb = prepare_sample(four_token_sample(ce_weights=[0.0, 0.0, 1.0, 1.0], rl_weights=[0.0] * 4), seq_len=16)
    assert mb.ce_weights == [0.0, 0.0, 1.0, 1.0] and mb.rl_weights == [0.0] * 4
    mb = prepare_sample(four_token_sample(), seq_len=16)
    assert mb.rl_weights is None and mb.ce_weights is None and mb.ref_kl_weights is None  # uniform rl: no streams on the wire


@pytest.mark.parametrize("streams_on_longer", [True, False])
def test_mixed_components_backfill_with_stream_defaults(streams_on_longer):
    longer = TrainingSample(token_ids=[1, 2, 3, 4, 5, 6], mask=[False] * 3 + [True] * 3, logprobs=[0.0] * 3 + [-0.1] * 3, temperatures=[1.0] * 6,
                            advantages=[0.0] * 3 + [1.0] * 3, env_name="test-env", ce_weights=[0.0] * 3 + [1.0] * 3 if streams_on_longer else None,
                            rl_weights=[0.0] * 6 if streams_on_longer else None)
    shorter = four_token_sample(ce_weights=None if streams_on_longer else [0.0, 0.0, 1.0, 1.0], rl_weights=None if streams_on_longer else [0.0] * 4)
    (mb,) = flat(prepare_batch(rollouts=[longer, shorter], seq_len=16, num_train_workers=1, bin_cost=build_bin_cost(None)))
    if streams_on_longer:  # first-fit-decreasing puts the longer sample first; the stream-less side backfills with rl 1.0 / ce 0.0
        assert mb.rl_weights == [0.0] * 6 + [1.0] * 4 and mb.ce_weights == [0.0] * 3 + [1.0] * 3 + [0.0] * 4
    else:
        assert mb.rl_weights == [1.0] * 6 + [0.0] * 4 and mb.ce_weights == [0.0] * 6 + [0.0, 0.0, 1.0, 1.0]


@pytest.mark.parametrize("refs_on_longer", [True, False])
def test_ref_logprobs_stay_position_aligned_in_mixed_bins(refs_on_longer):
    longer = TrainingSample(token_ids=[1, 2, 3, 4, 5, 6], mask=[False] * 3 + [True] * 3, logprobs=[0.0] * 3 + [-0.1] * 3, temperatures=[1.0] * 6,
                            ref_logprobs=[-1.5] * 6 if refs_on_longer else None, advantages=[0.0] * 3 + [1.0] * 3, env_name="test-env")
    shorter = four_token_sample()
    shorter.ref_logprobs = None if refs_on_longer else [-1.5] * 4
    (mb,) = flat(prepare_batch(rollouts=[longer, shorter], seq_len=16, pad_to_multiple_of=1, num_train_workers=1, bin_cost=build_bin_cost(None)))
    assert len(mb.ref_logprobs) == len(mb.input_ids)
    assert mb.ref_logprobs == ([-1.5] * 6 + [0.0] * 4 if refs_on_longer else [0.0] * 6 + [-1.5] * 4)
