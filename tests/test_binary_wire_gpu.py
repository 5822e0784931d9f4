"""GPU suite: the binary step wire feeds the device exactly what the reference-format msgpack path feeds it (bit-exact packed fields
after ONE host-to-device copy, incl. the on-device 1/temperature and shift-left labels), and `train()` over rollout files gives the
same step metrics with either wire (reference path src/prime_rl/trainer/rl/data.py:186-238)."""

import asyncio
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grid(seed, n_mb, vocab):
    from test_binary_wire_cpu import _mb

    rng = random.Random(seed)
    mbs = [_mb(rng, rng.randint(2, 5), ["math", "code"], with_ref=False, with_weights=(i == 1)) for i in range(n_mb)]
    for mb in mbs:
        mb.input_ids = [t % vocab for t in mb.input_ids]
        mb.inference_logprobs = [-6.2 - min(abs(x), 1.0) for x in mb.inference_logprobs]  # near log(1/512): finite ratios on the tiny model
    return mbs


def test_packed_batch_on_device_is_bit_identical():
    from test_binary_wire_cpu import _assert_packed_equal

    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.rl.data import DataLoader
    from prime_rl_amd.transports.rollouts import binary as B
    from prime_rl_amd.transports.rollouts import decode_micro_batches, encode_micro_batches

    mbs = _grid(3, 5, 151936)
    want = pack_micro_batches([DataLoader._micro_batch_to_tensor(mb) for mb in decode_micro_batches(encode_micro_batches(mbs))], DEV)
    blob = B.decode_step(B.encode_step(mbs))
    assert blob.payload.is_pinned()
    got = B.blob_to_packed_batch(blob, DEV)
    torch.cuda.synchronize()
    assert got.input_ids.is_cuda
    _assert_packed_equal(got, want)


def test_train_loop_same_metrics_with_either_wire(tmp_path):
    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer.rl.train import train
    from prime_rl_amd.transports.rollouts import FileSystemMicroBatchSender

    hist = {}
    for fmt in ("msgpack", "binary"):
        out = tmp_path / fmt
        sender = FileSystemMicroBatchSender(out, data_world_size=1, current_step=1, wire_format=fmt)
        for step in range(2):
            asyncio.run(sender.send([_grid(100 + step, 3, 512)]))
        cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny", seq_len=256, debug=dict(random_init=True)), max_steps=2, output_dir=str(out),
                                                tokens_per_pass=4096, rollout_transport=dict(type="filesystem", format=fmt)))
        hist[fmt] = train(cfg)
    for a, b in zip(hist["msgpack"], hist["binary"]):
        for key in ("loss/mean", "entropy/all/mean", "mismatch_kl/all/mean", "optim/grad_norm", "is_masked/mean"):
            assert a[key] == b[key], (key, a[key], b[key])
        assert any(k.startswith("entropy/math/") for k in b) and any(k.startswith("entropy/code/") for k in b)
