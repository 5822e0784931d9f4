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


@pytest.mark.parametrize("one_pass,tied", [(False, False), (True, False), (False, True)])
def test_forced_collectives_step_equals_plain_step(rccl_world1, one_pass, tied, monkeypatch):
    """Also the split reduction of the embedding bucket (engine._plan_embed_split) on the device, comm stream and all: dense part of the earlier
    pass reduce-scattered before the last pass, the last pass's rows all-gathered and added by pm_embedding_bwd_window — same f32 additions
    in the same order as the plain engine's accumulate-in-place, so still bit for bit.  `one_pass`: nothing dense; `tied`: the head gradient
    is the dense part."""
    import dataclasses

    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims, OptimConfig

    dims = dataclasses.replace(ModelDims.tiny(layers=2, vocab=512), tie_word_embeddings=tied)
    optim = OptimConfig(lr=1e-3, max_norm=0.5)
    # the plain engine with the streaming norm pass: the epilogue-fused norm (its default) adds the same squares in another order, and this test is
    # about the collective machinery being bit-transparent (fused vs streaming norm: test_fused_gradient_norm_equals_the_separate_pass)
    plain = GRPOEngine(dims, device=DEV, optim=optim, seed=5, fused_sumsq=False)
    forced = GRPOEngine(dims, device=DEV, optim=optim, seed=5, force_collectives=True)
    assert not plain.use_collectives and forced.use_collectives
    assert forced.comm_stream is not None and forced.store.grad_shard.data_ptr() != forced.store.main_grad.data_ptr()
    assert torch.equal(plain.store.params, forced.store.params)
    for passes in _batches(dims.vocab_size, 3):
        passes = passes[:1] if one_pass else passes
        a = plain.step(passes)
        b = forced.step(passes)
        torch.cuda.synchronize()
        assert forced.embed_split and forced._embed_bufs is not None
        assert torch.equal(a["grad_norm"], b["grad_norm"])
        for la, lb in zip(a["loss"], b["loss"]):
            assert torch.equal(la, lb)
        assert torch.equal(plain.store.master, forced.store.master)
        assert torch.equal(plain.store.params, forced.store.params)
    # the last step left one gather event per bucket for the next forward to wait on
    assert set(forced._gather_events) == {bk.name for bk in forced.store.buckets}


def test_rccl_weight_push_trainer_side_on_device(rccl_world1, tmp_path):
    """Trainer half of the sharded push (transports/weights RCCLWeightBroadcast) on a raw RCCL communicator (pm_comm_*, bootstrapped from
    a TCP store like the reference's StatelessProcessGroup) with an empty inference pool: the NCCL_READY handshake, the LAYOUT.json
    manifest, the asynchronous push stream + the engine's reader event run on the GPU and leave the parameters untouched.
    (Sender + receivers together: tests/test_weight_push_cpu.py, gloo world_size 4.)"""
    import json
    import math

    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims
    from prime_rl_amd.transports import weights as W
    from prime_rl_amd.utils.pathing import get_broadcast_dir, get_step_path

    dims = ModelDims.tiny(layers=1, vocab=512)
    eng = GRPOEngine(dims, device=DEV, seed=9)
    before = eng.store.params.clone()
    import socket

    from tests._ports import free_port

    port = free_port()
    comm = W.init_push_communicator("127.0.0.1", port, 0, 1, 0, DEV, timeout=60)
    assert (comm.rank, comm.size) == (0, 1)
    step_dir = get_step_path(get_broadcast_dir(tmp_path), 1)
    step_dir.mkdir(parents=True)
    (step_dir / W.NCCL_READY_MARKER).touch()  # what the inference pool writes once its receives are posted
    wb = W.RCCLWeightBroadcast(tmp_path, comm, 0, 1, 0, ready_timeout=10)
    wb.broadcast_weights(eng, 1)
    assert len(eng._param_readers) == 1  # the next optimizer step will wait for the push on the device
    assert wb.last_transfer_seconds() is not None
    torch.cuda.synchronize()
    assert torch.equal(before, eng.store.params)
    comm.close()
    layout = json.loads((step_dir / "LAYOUT.json").read_text())
    sd = eng.store.state_dict()
    names = {n for b in layout for n, _, _ in b["params"]}
    assert names == set(sd)
    for b, bk in zip(layout, eng.store.buckets):
        for name, off, shape in b["params"]:
            view = eng.store.params[bk.offset + off: bk.offset + off + math.prod(shape)].view(shape)
            assert torch.equal(view, sd[name])


class _Loopback:
    """A world-1 raw RCCL communicator presented as the 1 trainer + 1 inference push world: every peer is this rank.  RCCL refuses two ranks on one
    device, but a send and the matching receive of ONE rank posted inside one ncclGroup are a legal pair — the bytes travel device -> device through
    RCCL's own copy kernel, which is as much of the a25 data path as a one-GPU box can exercise."""

    def __init__(self, comm):
        self.comm, self.rank, self.size = comm, 0, 2

    def send(self, t, peer):
        self.comm.send(t, 0)

    def recv(self, t, peer):
        self.comm.recv(t, 0)

    def group_start(self):
        self.comm.group_start()  # ncclGroupStart / End nest: only the outermost End launches

    def group_end(self):
        self.comm.group_end()


def test_rccl_weight_push_moves_real_bytes_device_to_device(tmp_path):
    """a25 with DATA on the GPU (the other on-device push test runs with an empty inference pool): the sharded push protocol end to end on one
    device — trainer side `push_sharded` and inference side `receive_sharded` of the SAME process posted inside one RCCL group on a world-1 raw
    communicator, so every bucket of the flat bf16 parameter buffer crosses RCCL (ncclSend / ncclRecv self pair) and must come out bit for bit under
    its HF name.  Also times a 256 MiB self pair (informational: an on-device copy rate, not an xGMI number)."""
    import socket
    import time

    from prime_rl_amd._lib import PrimeMI355XError
    from prime_rl_amd.trainer.engine import GRPOEngine, ModelDims
    from prime_rl_amd.transports import weights as W

    dims = ModelDims.tiny(layers=2, vocab=512)
    eng = GRPOEngine(dims, device=DEV, seed=13)
    from tests._ports import free_port

    port = free_port()
    raw = W.init_push_communicator("127.0.0.1", port, 0, 1, 0, DEV, timeout=60)
    loop = _Loopback(raw)
    layout = W.build_layout(eng.store)
    try:
        loop.group_start()
        W.push_sharded(loop, eng.store, 0, 1, 1)
        received = list(W.receive_sharded(loop, layout, 1, torch.device(DEV)))  # views of the receive buffers: filled when the outer group ends
        loop.group_end()
        torch.cuda.synchronize()
    except PrimeMI355XError as e:  # pragma: no cover - an RCCL build that refuses self pairs: say so instead of passing silently
        raw.close()
        pytest.skip(f"RCCL refused a self send / recv pair on one device: {e} (a25 on-device bytes stay blocked on a one-GPU box)")
    sd = eng.store.state_dict()
    assert {n for n, _ in received} == set(sd)
    n_bytes = 0
    for name, t in received:
        assert torch.equal(t, sd[name]), name
        n_bytes += t.numel() * t.element_size()
    assert n_bytes > 0
    a = torch.randint(0, 2 ** 15, (128 * 1024 * 1024,), dtype=torch.int16, device=DEV).view(torch.bfloat16)
    b = torch.empty_like(a)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw.group_start()
        raw.send(a, 0)
        raw.recv(b, 0)
        raw.group_end()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    print(f"[a25 self pair] {n_bytes} parameter bytes bit-exact through RCCL on one device; 256 MiB self pair {dt * 1e3:.2f} ms = {a.numel() * 2 / dt / 1e9:.0f} GB/s")
    raw.close()


def test_train_loop_with_nccl_weight_broadcast_from_the_config_path(tmp_path):
    """`[weight_broadcast] type = "nccl"` through train(config) (this path used to raise KeyError): train() bootstraps the raw
    RCCL push communicator from host:port, pushes the startup policy v0 and one policy per step except the last (train.py:582-587), the
    engine's optimizer waits for the asynchronous push, and the ZeRO shards stay on the trainer ranks.  The 'inference pool' here is a
    thread that answers STABLE with NCCL_READY (inference_world_size = 0: no receiver GPU on a one-GPU box)."""
    import socket
    import threading
    import time

    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer.rl.train import train

    from tests._ports import free_port

    port = free_port()
    stop = threading.Event()

    def pool():
        while not stop.is_set():
            for d in (tmp_path / "broadcasts").glob("step_*"):
                if (d / "STABLE").exists() and not (d / "NCCL_READY").exists():
                    (d / "NCCL_READY").touch()
            time.sleep(0.02)

    th = threading.Thread(target=pool, daemon=True)
    th.start()
    try:
        cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny", seq_len=128, debug=dict(random_init=True)), max_steps=3, output_dir=str(tmp_path),
                                                data=dict(fake=dict(batch_size=2, generate_samples=True)), tokens_per_pass=256,
                                                weight_broadcast=dict(type="nccl", host="127.0.0.1", port=port, inference_world_size=0, timeout=60)))
        hist = train(cfg)
    finally:
        stop.set()
        th.join()
    assert len(hist) == 3
    pushed = sorted(int(d.name.split("_")[1]) for d in (tmp_path / "broadcasts").glob("step_*") if (d / "LAYOUT.json").exists())
    assert pushed == [0, 1, 2]  # v0 at start-up, v1, v2; the final v3 is skipped for nccl
    assert "time/broadcast_weights" in hist[1] and "time/broadcast_weights_transfer" in hist[1]


def test_setup_model_and_forward_keep_the_reference_signatures():
    """trainer/model.py:1314 setup_model(config, parallel_dims, loading_from_checkpoint_later) and :1431 forward(model, input_ids,
    position_ids, *, seq_lens, labels, temperature) -> PrimeLmOutput{logprobs, entropy} f32 contiguous; the fused head is injected with
    the config's chunk size; the autograd path reaches every parameter."""
    import inspect

    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer import model as M
    from prime_rl_amd.trainer.models.layers.lm_head import FusedOutputLinear, VanillaOutputLinear
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims

    assert list(inspect.signature(M.setup_model).parameters) == ["config", "parallel_dims", "loading_from_checkpoint_later"]
    fp = inspect.signature(M.forward).parameters
    assert list(fp)[:3] == ["model", "input_ids", "position_ids"] and all(fp[k].kind is inspect.Parameter.KEYWORD_ONLY for k in ("seq_lens", "labels", "temperature"))
    cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny", seq_len=64, debug=dict(random_init=True, num_layers=1), fused_lm_head_token_chunk_size=32)))
    pd = get_parallel_dims(cfg.model, cfg.model.seq_len, world_size=1)
    model = M.setup_model(cfg.model, pd)
    assert isinstance(model.lm_head, FusedOutputLinear) and cfg.model.attn == "hip_varlen" and len(model.model.layers) == 1
    n = 64
    ids = torch.randint(0, 512, (1, n), device=DEV)
    labels = torch.cat([ids[:, 1:], ids.new_zeros(1, 1)], 1)
    out = M.forward(model, ids, torch.arange(n, device=DEV)[None], seq_lens=torch.tensor([40, 24], device=DEV), labels=labels,
                    temperature=torch.ones(1, n, device=DEV))
    assert out["logprobs"].shape == (1, n) and out["logprobs"].dtype == torch.float32 and out["entropy"].is_contiguous() and out.get("logits") is None
    out["logprobs"].sum().backward()
    assert all(p.grad is not None for p in model.parameters())
    cfg2 = TrainerConfig.model_validate(dict(model=dict(name="tiny", debug=dict(random_init=True, num_layers=1), fused_lm_head_token_chunk_size="disabled")))
    assert isinstance(M.setup_model(cfg2.model, pd).lm_head, VanillaOutputLinear)
    with pytest.raises(FileNotFoundError):
        M.setup_model(TrainerConfig.model_validate(dict(model=dict(name="tiny"))).model, pd)  # no weights and no random_init: loud
