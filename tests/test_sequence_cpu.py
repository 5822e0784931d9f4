"""CPU suite: packed-sequence boundary helpers against the reference's known answers (tests/unit/utils/test_sequence.py:10-41) and
the host-side credit rule that has no kernel (RAE, src/prime_rl/orchestrator/algo/rae.py:39-43)."""

import asyncio

import pytest
import torch

from prime_rl_amd.utils.sequence import get_cu_seqlens_from_position_ids, get_cu_seqlens_from_seq_lens


@pytest.mark.parametrize(("position_ids", "want_cu", "want_max"), [
    (torch.arange(8).unsqueeze(0), [0, 8], 8),
    (torch.arange(8, 16).unsqueeze(0), [0, 8], 8),  # a context-parallel shard that starts mid-document
    (torch.tensor([[0, 1, 2, 3, 0, 1, 2]]), [0, 4, 7], 4),
    (torch.tensor([[5, 6, 7, 0, 1, 2]]), [0, 3, 6], 3),
    (torch.tensor([[0, 0, 0]]), [0, 1, 2, 3], 1),
])
def test_cu_seqlens_from_position_ids_is_local_relative(position_ids, want_cu, want_max):
    cu, mx = get_cu_seqlens_from_position_ids(position_ids)
    assert cu.dtype == torch.int32 and cu.tolist() == want_cu and mx == want_max


def test_cu_seqlens_from_seq_lens_known_answer_and_errors():
    cu, mx = get_cu_seqlens_from_seq_lens(torch.tensor([4, 3, 2]), total_tokens=9)
    assert cu.dtype == torch.int32 and cu.tolist() == [0, 4, 7, 9] and mx == 4
    cu, mx = get_cu_seqlens_from_seq_lens([2048], total_tokens=2048)  # host list straight from the rollout wire
    assert cu.tolist() == [0, 2048] and mx == 2048
    with pytest.raises(ValueError, match="sum must equal"):
        get_cu_seqlens_from_seq_lens(torch.tensor([4, 3]), total_tokens=9)
    with pytest.raises(ValueError, match="must be positive"):
        get_cu_seqlens_from_seq_lens(torch.tensor([4, 0]))
    with pytest.raises(ValueError, match="must not be empty"):
        get_cu_seqlens_from_seq_lens(torch.tensor([], dtype=torch.int64))
    with pytest.raises(ValueError, match="must be 1D"):
        get_cu_seqlens_from_seq_lens(torch.tensor([[4, 3]]))


def test_cu_seqlens_matches_packer_boundaries():
    """The table equals the boundaries `pack_micro_batches` derives for the kernels (same ints, same dtype)."""
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    mb = FakeDataLoader(FakeDataLoaderConfig(batch_size=1, generate_samples=True), seq_len=512).get_batch()[0]
    cu_a, mx_a = get_cu_seqlens_from_seq_lens(mb["seq_lens"], total_tokens=512)
    cu_b, mx_b = get_cu_seqlens_from_position_ids(mb["position_ids"])
    assert torch.equal(cu_a, cu_b) and mx_a == mx_b


def test_mfu_formula_matches_reference_known_answers():
    """trainer/perf.py:120-213 through the reference's own pins (tests/unit/train/test_perf.py:8-35): Qwen3-0.6B has 595 984 384
    active matmul parameters and 3 928 227 840 FLOP/token at seq_len 1024 (attention term uses hidden // heads = 64, not head_dim);
    Qwen3-8B gives the 49.03 GFLOP/token of SURVEY.md section 8d."""
    from prime_rl_amd.trainer.engine import ModelDims

    small = ModelDims.qwen3_0_6b()
    assert small.active_mm_params() == 595_984_384
    assert small.flops_per_token(1024) == 3_928_227_840
    big = ModelDims.qwen3_8b()
    assert big.flops_per_token(2048) == 6 * big.active_mm_params() + 12 * 36 * 32 * 128 * 2048
    assert abs(big.flops_per_token(2048) / 1e9 - 49.03) < 0.01


def test_rae_scores_before_updating_baseline():
    from prime_rl_amd.orchestrator.algo import RAEAlgoConfig, build_algorithm
    from prime_rl_amd.orchestrator.algo.base import SimpleAgent, SimpleEpisode, SimpleNode, SimpleTrace

    def tr(role, reward):
        return SimpleTrace(role, reward, [SimpleNode([7, 8, 9], [False, True, True])], agent=SimpleAgent(name=role))

    algo = build_algorithm(RAEAlgoConfig(decay=0.5))
    eps = [SimpleEpisode([tr("p0", 1.0), tr("p1", -1.0)]), SimpleEpisode([tr("p0", 1.0), tr("p1", -1.0)])]
    asyncio.run(algo.finalize_group(eps))
    adv = [[t.nodes[0].advantages for t in e.traces] for e in eps]
    assert adv[0] == [[1.0, 1.0], [-1.0, -1.0]]  # baselines start at 0
    assert adv[1] == [[0.5, 0.5], [-0.5, -0.5]]  # scored against 0.5*0 + 0.5*(+-1) before the second update
    assert algo.baselines == {"p0": 0.75, "p1": -0.75}
