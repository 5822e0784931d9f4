"""CPU suite: host-side data plumbing — FakeDataLoader equals the reference's value-for-value; micro-batch packing."""

import numpy as np
import torch

from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig


def test_fake_dataloader_matches_reference(golden_dir):
    g = dict(np.load(golden_dir / "fake_data.npz"))
    for gen_samples in (False, True):
        dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=4, generate_samples=gen_samples), seq_len=96, dp_world_size=2, dp_rank=1)
        for step in range(2):
            mbs = dl.get_batch()
            assert len(mbs) == 2
            for mi, mb in enumerate(mbs):
                tag = f"g{int(gen_samples)}_s{step}_m{mi}_"
                for k in ("input_ids", "position_ids", "advantages", "inference_logprobs", "temperatures", "loss_mask", "seq_lens"):
                    np.testing.assert_array_equal(mb[k].numpy(), g[tag + k], err_msg=tag + k)  # bit-exact
                assert list(mb["sequence_lengths"]) == g[tag + "sequence_lengths"].tolist()


def test_pack_micro_batches_layout():
    from prime_rl_amd.trainer.engine import pack_micro_batches

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=3, generate_samples=True), seq_len=64)
    mbs = dl.get_batch()
    pb = pack_micro_batches(mbs, "cpu")
    assert pb.n_tokens == 192 and pb.n_micro_batches == 3
    # labels: shift_tensor_left per micro-batch (last label of each micro-batch is 0)
    for i, mb in enumerate(mbs):
        ids = mb["input_ids"][0]
        assert torch.equal(pb.labels[i * 64:(i + 1) * 64 - 1], ids[1:]) and pb.labels[(i + 1) * 64 - 1] == 0
    assert pb.cu_samples[-1] == 192 and pb.cu_seqlens[-1] == 192
    starts = pb.sample_starts_mb.tolist()
    assert sum(starts) == 3 and starts[0] == 1
    assert pb.local_counts == (192, 0, 0)
    assert torch.equal(pb.inv_temp, torch.ones(192))
