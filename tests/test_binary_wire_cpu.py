"""CPU suite: the binary step wire (transports/rollouts/binary.py) against the reference-format msgpack wire.

Pin: for the same `MicroBatch` objects, msgpack -> `DataLoader._micro_batch_to_tensor` -> `pack_micro_batches` (the mirrored
reference path, src/prime_rl/trainer/rl/data.py:186-238) and blob -> `blob_to_packed_batch` give bit-identical packed fields,
including f64 -> f32 rounding of the wire floats, optional-stream filling, sample / document boundaries and member-token counts."""

import asyncio
import random
import threading
import time

import pytest
import torch

from prime_rl_amd.trainer.engine import pack_micro_batches
from prime_rl_amd.trainer.rl.data import DataLoader
from prime_rl_amd.transports.rollouts import FileSystemMicroBatchSender, MicroBatch, decode_micro_batches, encode_micro_batches
from prime_rl_amd.transports.rollouts import binary as B


def _mb(rng: random.Random, n_samples: int, envs, with_ref=False, with_weights=False) -> MicroBatch:
    lens = [rng.randint(1, 40) for _ in range(n_samples)]
    n = sum(lens)
    env_names, pos, docs = [], [], []
    for ln in lens:
        env_names += [rng.choice(envs)] * ln
        cut = rng.randint(1, ln)  # a sample may hold two attention documents
        docs += [cut] + ([ln - cut] if ln > cut else [])
        pos += list(range(cut)) + list(range(ln - cut))
    f = lambda: [rng.gauss(0, 1) * 10 ** rng.randint(-6, 2) for _ in range(n)]  # f64 values that need rounding to f32
    return MicroBatch(input_ids=[rng.randrange(151936) for _ in range(n)], loss_mask=[rng.random() < 0.7 for _ in range(n)], advantages=f(),
                      inference_logprobs=f(), position_ids=pos, sequence_lengths=lens, temperatures=[rng.choice([0.6, 0.7, 1.0, 1.3]) for _ in range(n)],
                      env_names=env_names, seq_lens=docs, ref_logprobs=f() if with_ref else None,
                      rl_weights=[float(rng.random() < 0.5) for _ in range(n)] if with_weights else None,
                      ce_weights=[rng.choice([0.0, 0.5, 1.0]) for _ in range(n)] if with_weights else None)


def _assert_packed_equal(a, b):
    for name in ("input_ids", "position_ids", "labels", "inv_temp", "advantages", "inference_logprobs", "loss_mask", "cu_seqlens", "cu_samples",
                 "sample_starts_mb", "ref_logprobs", "rl_weights", "ce_weights", "ref_kl_weights", "rl_every_sample"):
        x, y = getattr(a, name), getattr(b, name)
        assert (x is None) == (y is None), name
        if x is not None:
            assert x.dtype == y.dtype and torch.equal(x, y), name
    assert a.n_micro_batches == b.n_micro_batches and a.local_counts == b.local_counts


@pytest.mark.parametrize("case", ["plain", "ref", "mixed_streams"])
def test_binary_wire_equals_msgpack_path(case):
    rng = random.Random(hash(case) % 1000)
    mbs = [_mb(rng, rng.randint(1, 5), ["math", "code", "wiki"], with_ref=case == "ref", with_weights=(case == "mixed_streams" and i % 2 == 0))
           for i in range(4)]
    ref_tensors = [DataLoader._micro_batch_to_tensor(mb) for mb in decode_micro_batches(encode_micro_batches(mbs))]
    want = pack_micro_batches(ref_tensors, "cpu")
    buf = B.encode_step(mbs)
    blob = B.decode_step(buf)
    assert all(off % B.ALIGN == 0 for off, _, _ in blob.offsets.values())
    _assert_packed_equal(B.blob_to_packed_batch(blob, "cpu"), want)
    # the host-side TensorMicroBatch view also matches the reference layout field by field (optional streams are filled per step)
    for got, ref in zip(B.blob_to_tensor_micro_batches(blob), ref_tensors):
        for k in ("input_ids", "position_ids", "advantages", "inference_logprobs", "loss_mask", "temperatures", "seq_lens", "lora_num_tokens"):
            assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and torch.equal(got[k], ref[k]), k
        assert got["env_names"] == ref["env_names"] and got["sequence_lengths"] == ref["sequence_lengths"]
    # the blob is smaller than msgpack-of-lists (9-byte floats, per-token env strings) even at these toy sizes with 64-byte padding
    assert len(buf) < len(encode_micro_batches(mbs))


def test_binary_wire_keeps_which_micro_batches_had_no_rl_stream():
    """A step that mixes micro-batches with and without an rl_weights stream: the wire fills the absent ones with 1.0, and the per-sample flag the rl
    metric rows depend on (`PackedBatch.rl_every_sample`, stats.collect_step_tensors) survives the trip exactly as pack_micro_batches builds it."""
    rng = random.Random(5)
    mbs = [_mb(rng, 3, ["math"], with_weights=(i == 1)) for i in range(3)]
    want = pack_micro_batches([DataLoader._micro_batch_to_tensor(mb) for mb in decode_micro_batches(encode_micro_batches(mbs))], "cpu")
    got = B.blob_to_packed_batch(B.decode_step(B.encode_step(mbs)), "cpu")
    assert want.rl_every_sample is not None and want.rl_every_sample.tolist() == [True] * 3 + [False] * 3 + [True] * 3
    assert got.rl_every_sample is not None and torch.equal(got.rl_every_sample, want.rl_every_sample)
    none = [_mb(rng, 2, ["math"]) for _ in range(2)]  # no rl stream anywhere: no flag on either path
    assert B.blob_to_packed_batch(B.decode_step(B.encode_step(none)), "cpu").rl_every_sample is None


def test_binary_blob_rejects_corruption():
    rng = random.Random(1)
    buf = B.encode_step([_mb(rng, 2, ["e"])])
    with pytest.raises(ValueError, match="not a PMRB"):
        B.decode_step(b"XXXX" + buf[4:])
    with pytest.raises(ValueError, match="version"):
        B.decode_step(buf[:4] + (99).to_bytes(4, "little") + buf[8:])
    with pytest.raises(ValueError, match="shorter"):
        B.decode_step(buf[:-200])
    bad = MicroBatch(input_ids=[1, 2, 3], loss_mask=[True] * 3, advantages=[0.0] * 3, inference_logprobs=[0.0] * 3, position_ids=[0, 1, 2],
                     sequence_lengths=[2], temperatures=[1.0] * 3, env_names=["e"] * 3, seq_lens=[3])
    with pytest.raises(ValueError, match="must sum"):
        B.encode_step([bad])
    with pytest.raises(ValueError, match="empty step"):
        B.encode_step([])


def test_binary_receiver_reads_ahead_in_order(tmp_path):
    """Sender (binary format) -> BinaryStepReceiver: step k+1 is read by the background thread while step k is being consumed; a step
    that appears late is waited for, and the DataLoader mirror serves both the tensor view and the packed batch."""
    rng = random.Random(7)
    grids = [[[_mb(rng, 2, ["a", "b"]) for _ in range(2)] for _ in range(2)] for _ in range(3)]  # [step][rank][micro-batch]
    sender = FileSystemMicroBatchSender(tmp_path, data_world_size=2, wire_format="binary")
    asyncio.run(sender.send(grids[0]))

    def late():
        for g in grids[1:]:
            time.sleep(0.15)
            asyncio.run(sender.send(g))

    th = threading.Thread(target=late)
    th.start()
    from types import SimpleNamespace

    loader = DataLoader(tmp_path, start_step=0, dp_world_size=2, dp_rank=1, config=SimpleNamespace(type="filesystem", format="binary"))
    for step in range(3):
        loader.wait_for_batch()
        packed, mbs = loader.get_packed_batch("cpu")
        want = pack_micro_batches([DataLoader._micro_batch_to_tensor(mb) for mb in grids[step][1]], "cpu")
        _assert_packed_equal(packed, want)
        assert len(mbs) == 2 and mbs[0]["env_names"] == grids[step][1][0].env_names
    th.join()
    assert loader.receiver.current_step == 3


@pytest.mark.parametrize("dtype", ["uint8", "int16", "int32"])
def test_binary_wire_carries_router_replay_payloads(dtype):
    """`MicroBatch.routed_experts` (raw [tokens, layers, top_k] blocks) through the step blob: the packed batch and the per-micro-batch tensors equal
    the msgpack path's, i32 on arrival whatever the sender's integer width; a step that mixes micro-batches with and without a payload is refused."""
    import numpy as np

    from prime_rl_amd.transports.rollouts.types import RoutedExperts

    rng = random.Random(11)
    nrng = np.random.default_rng(11)
    mbs = [_mb(rng, rng.randint(1, 4), ["math", "code"]) for _ in range(3)]
    for mb in mbs:
        n = len(mb.input_ids)
        arr = nrng.integers(0, 100, size=(n, 3, 4)).astype(dtype)
        mb.routed_experts = RoutedExperts(data=arr.tobytes(), shape=[n, 3, 4], dtype=dtype)
    ref_tensors = [DataLoader._micro_batch_to_tensor(mb) for mb in decode_micro_batches(encode_micro_batches(mbs))]
    want = pack_micro_batches(ref_tensors, "cpu")
    assert want.routed_experts.shape == (want.n_tokens, 3, 4) and want.routed_experts.dtype == torch.int32
    blob = B.decode_step(B.encode_step(mbs))
    got = B.blob_to_packed_batch(blob, "cpu")
    _assert_packed_equal(got, want)
    assert got.routed_experts.dtype == torch.int32 and torch.equal(got.routed_experts, want.routed_experts)
    for g, r in zip(B.blob_to_tensor_micro_batches(blob), ref_tensors):
        assert g["routed_experts"].dtype == torch.int32 and torch.equal(g["routed_experts"], r["routed_experts"])
    mbs[1].routed_experts = None
    with pytest.raises(ValueError, match="router replay"):
        B.encode_step(mbs)
