"""GPU suite: the data-parallel plumbing of the step (RCCL reduce-scatter of f32 gradient buckets on the comm stream, sharded
clip + AdamW, in-place bf16 all-gather, per-bucket gather events awaited by the next forward) executed on ONE GPU through a
world_size-1 `nccl` (= RCCL) process group.  With one rank the collectives are copies, so two optimizer steps with
`force_collectives=True` must reproduce the plain dp=1 engine bit for bit; what this covers is the part the world_size-2 gloo
tests (tests/test_zero_gloo_cpu.py) cannot: RCCL itself, the side stream and the event ordering on the device.
Reference: FSDP2 wrapping + gradient post-processing, src/prime_rl/trainer/model.py:823-960, trainer/rl/train.py:557-570."""

import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def rccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    if created:
        dist.destroy_process_group()


def _batches(vocab, n_steps):
    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=4, generate_samples=True), seq_len=128)
    out = []
    for _ in range(n_steps):
        mbs = dl.get_batch()
        for mb in mbs:
            mb["input_ids"] = mb["input_ids"] % vocab
        out.append([pack_micro_batches(mbs[:2], DEV), pack_micro_batches(mbs[2:], DEV)])
    return out


def test_forced_collectives_step_equals_plain_step(rccl_world1):
    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims, OptimConfig

    dims = ModelDims.tiny(layers=2, vocab=512)
    optim = OptimConfig(lr=1e-3, max_norm=0.5)
    plain = GRPOEngine(dims, device=DEV, optim=optim, seed=5)
    forced = GRPOEngine(dims, device=DEV, optim=optim, seed=5, force_collectives=True)
    assert not plain.use_collectives and forced.use_collectives
    assert forced.comm_stream is not None and forced.store.grad_shard.data_ptr() != forced.store.main_grad.data_ptr()
    assert torch.equal(plain.store.params, forced.store.params)
    for passes in _batches(dims.vocab_size, 3):
        a = plain.step(passes)
        b = forced.step(passes)
        torch.cuda.synchronize()
        assert torch.equal(a["grad_norm"], b["grad_norm"])
        for la, lb in zip(a["loss"], b["loss"]):
            assert torch.equal(la, lb)
        assert torch.equal(plain.store.master, forced.store.master)
        assert torch.equal(plain.store.params, forced.store.params)
    # the last step left one gather event per bucket for the next forward to wait on
    assert set(forced._gather_events) == {bk.name for bk in forced.store.buckets}


def test_rccl_weight_push_trainer_side_on_device(rccl_world1, tmp_path):
    """Trainer half of the sharded push (transports/weights RCCLWeightBroadcast) over RCCL with an empty inference pool: the
    NCCL_READY handshake, the LAYOUT.json manifest and one async device broadcast per bucket run on the GPU and leave the
    parameters untouched.  (Sender + receiver together: tests/test_weight_push_cpu.py, gloo world_size 3.)"""
    import json
    import math

    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims
    from prime_rl_amd.transports import weights as W
    from prime_rl_amd.utils.pathing import get_broadcast_dir, get_step_path

    dims = ModelDims.tiny(layers=1, vocab=512)
    eng = GRPOEngine(dims, device=DEV, seed=9)
    before = eng.store.params.clone()
    groups = W.init_push_groups(1, 0)
    step_dir = get_step_path(get_broadcast_dir(tmp_path), 1)
    step_dir.mkdir(parents=True)
    (step_dir / W.NCCL_READY_MARKER).touch()  # what the inference pool writes once its receives are posted
    W.RCCLWeightBroadcast(tmp_path, 0, 1, groups, ready_timeout=10).broadcast_weights(eng, 1)
    torch.cuda.synchronize()
    assert torch.equal(before, eng.store.params)
    layout = json.loads((step_dir / "LAYOUT.json").read_text())
    sd = eng.store.state_dict()
    names = {n for b in layout for n, _, _ in b["params"]}
    assert names == set(sd)
    for b, bk in zip(layout, eng.store.buckets):
        for name, off, shape in b["params"]:
            view = eng.store.params[bk.offset + off: bk.offset + off + math.prod(shape)].view(shape)
            assert torch.equal(view, sd[name])
