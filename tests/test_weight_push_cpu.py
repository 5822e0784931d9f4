"""CPU suite (gloo, 2 trainers + 2 inference ranks): the weight push (sharded multi-source protocol and the reference's byte protocol) delivers every HF-named parameter bit-exactly
("bytes in == bytes out" is the only property the reference's skipped test_nccl_broadcast.py states), follows the
STABLE -> NCCL_READY rendezvous, and the filesystem transport writes a loadable checkpoint + marker."""

import os
import socket
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    from tests._ports import free_port

    return free_port()


N_TR, N_INF = 2, 2


class _FakeVllmModel:
    """What the worker extension hands the stream to: vLLM models expose `load_weights(iterable of (name, tensor))`."""

    def __init__(self):
        self.loaded: dict = {}

    def load_weights(self, it):
        for name, t in it:
            self.loaded[name] = t.clone()


def _worker(rank, world, port, out_dir, protocol):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.comm import TorchGroupCommunicator
    from prime_rl_amd.trainer.engine import ModelDims, ParamStore
    from prime_rl_amd.transports.weights import NCCL_READY_MARKER, RCCLWeightBroadcast, RCCLWeightReceiver

    # push world: trainer ranks first, then the inference ranks.  The reference protocol runs on {trainer master} + inference ranks.
    if protocol == "reference":
        members = [0] + list(range(N_TR, N_TR + N_INF))
        grp = dist.new_group(members)
        comm = TorchGroupCommunicator(grp) if rank in members else None
    else:
        comm = TorchGroupCommunicator(None)
    trainers = dist.new_group(list(range(N_TR)))
    out = Path(out_dir)
    step_dir = out / "broadcasts" / "step_7"
    if rank < N_TR:
        store = ParamStore(ModelDims.tiny(layers=2, vocab=320), "cpu", dp_rank=rank, dp_size=N_TR)
        store.random_init(seed=11)
        if rank == 0:
            torch.save({k: v.clone() for k, v in store.state_dict().items()}, out / "sent.pt")
        RCCLWeightBroadcast(out, comm, rank, N_TR, N_INF, protocol=protocol, ready_timeout=60, trainer_group=trainers).broadcast_weights(store, 7)
    else:
        t0 = time.time()
        while not (step_dir / "STABLE").exists():  # the orchestrator's WeightWatcher polls this marker (watcher.py:91-158)
            time.sleep(0.05)
            assert time.time() - t0 < 60
        if rank == N_TR:
            (step_dir / NCCL_READY_MARKER).touch()  # inference engines are paused and listening
        if rank == N_TR:  # plain receiver
            got = {name: t.clone() for name, t in RCCLWeightReceiver(comm, N_TR, "cpu", protocol).receive_state_dict(step_dir)}
        else:  # the vLLM-worker-extension-shaped receiver, minus the communicator bootstrap (raw RCCL: GPU suite)
            from types import SimpleNamespace

            from prime_rl_amd.inference.vllm.worker.rccl import RCCLWeightUpdateWorker

            w = RCCLWeightUpdateWorker()
            w.device, w.model_runner = torch.device("cpu"), SimpleNamespace(model=_FakeVllmModel())
            w.rccl_weight_receiver = RCCLWeightReceiver(comm, N_TR, "cpu", protocol)
            assert w.liveness_probe() is None
            w.update_weights_from_path(str(step_dir))
            got = w.model_runner.model.loaded
        torch.save(got, out / f"received_{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("protocol", ["sharded", "reference"])
def test_push_round_trip(tmp_path, protocol):
    """2 trainer ranks + 2 inference ranks over gloo: both inference ranks (a plain receiver and the worker-extension-shaped one) end up
    with every HF-named parameter bit-exactly, in checkpoint order."""
    world = N_TR + N_INF
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), protocol), nprocs=world, join=True, start_method="spawn")
    sent = torch.load(tmp_path / "sent.pt")
    for r in range(N_TR, world):
        got = torch.load(tmp_path / f"received_{r}.pt")
        assert set(sent) == set(got) and "model.layers.1.self_attn.k_proj.weight" in got and "lm_head.weight" in got
        for k in sent:
            assert got[k].shape == sent[k].shape and torch.equal(got[k], sent[k]), k
        names = list(got)
        if protocol == "sharded":  # bucket order on the wire: embed, layers ascending, head
            assert names[0] == "model.embed_tokens.weight" and names[-1] == "lm_head.weight"
        else:  # the reference's order: the non-layer state dict first, then layer 0, 1, ... (nccl.py:70-84)
            first_layer = next(i for i, n in enumerate(names) if n.startswith("model.layers."))
            assert all(not n.startswith("model.layers.") for n in names[:first_layer]) and all(n.startswith("model.layers.") for n in names[first_layer:])
            assert [n for n in names if n.startswith("model.layers.0.")] and names.index("model.layers.0.mlp.up_proj.weight") < names.index("model.layers.1.mlp.up_proj.weight")
    assert (tmp_path / "broadcasts" / "step_7" / "STABLE").exists() and (tmp_path / "broadcasts" / "step_7" / "LAYOUT.json").exists()


def test_reference_protocol_stream_is_the_reference_byte_layout():
    """The 'reference' protocol's stream, field by field: i64 count, then per state dict i64 pickle length, the
    pickled {dtype: [(key, shape, numel)]} and one flat tensor per dtype — recorded from the sender through a loop-back communicator."""
    import pickle

    from prime_rl_amd.transports.weights import receive_reference_protocol, send_reference_protocol

    class Tape:
        def __init__(self):
            self.sent, self.i = [], 0

        def broadcast(self, t, src):
            assert src == 0
            if self.i < len(self.sent):  # replay
                t.copy_(self.sent[self.i].view(t.dtype).reshape(t.shape))
            else:
                self.sent.append(t.clone())
            self.i += 1

    sd = {"lm_head.weight": torch.arange(6, dtype=torch.bfloat16).view(2, 3), "model.norm.weight": torch.ones(3, dtype=torch.bfloat16),
          "model.layers.0.a": torch.full((2, 2), 2.0, dtype=torch.bfloat16), "model.layers.0.b": torch.arange(3, dtype=torch.float32),
          "model.layers.1.a": torch.full((2, 2), 3.0, dtype=torch.bfloat16)}
    tape = Tape()
    send_reference_protocol(tape, sd, "cpu")
    assert tape.sent[0].dtype == torch.int64 and tape.sent[0].tolist() == [3]  # non-layer + 2 layers (nccl.py:147-150)
    assert tape.sent[1].dtype == torch.int64 and tape.sent[2].dtype == torch.uint8 and tape.sent[2].numel() == int(tape.sent[1])
    meta = pickle.loads(bytes(tape.sent[2].numpy()))
    assert meta == {torch.bfloat16: [("lm_head.weight", torch.Size([2, 3]), 6), ("model.norm.weight", torch.Size([3]), 3)]}
    assert tape.sent[3].dtype == torch.bfloat16 and tape.sent[3].numel() == 9  # one flat tensor per dtype (nccl.py:59-63)
    n_sent = len(tape.sent)
    tape.i = 0
    got = dict(receive_reference_protocol(tape, "cpu"))
    assert tape.i == n_sent and set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_filesystem_broadcast(tmp_path):
    from safetensors.torch import load_file

    from prime_rl_amd.trainer.engine import ModelDims, ParamStore
    from prime_rl_amd.transports.weights import setup_weight_broadcast

    store = ParamStore(ModelDims.tiny(layers=1, vocab=320), "cpu")
    store.random_init(seed=3)
    wb = setup_weight_broadcast(tmp_path, type("C", (), {"type": "filesystem"})())
    for step in (1, 2, 3):
        wb.broadcast_weights(store, step)
    sd = load_file(str(tmp_path / "broadcasts" / "step_3" / "model.safetensors"))
    assert torch.equal(sd["model.layers.0.mlp.up_proj.weight"], store.state_dict()["model.layers.0.mlp.up_proj.weight"])
    assert (tmp_path / "broadcasts" / "step_3" / "STABLE").exists() and not (tmp_path / "broadcasts" / "step_1").exists()


def _moe_worker(rank, world, port, out_dir):
    """2 expert-parallel trainer ranks (each holds half of the experts of a Qwen3-MoE model in a ModuleEngine) + 1 inference rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from prime_rl_amd.comm import TorchGroupCommunicator
    from prime_rl_amd.transports.weights import NCCL_READY_MARKER, RCCLWeightBroadcast, RCCLWeightReceiver

    grp = dist.new_group([0, 2])  # the reference protocol's communicator: trainer master + inference ranks
    comm = TorchGroupCommunicator(grp) if rank in (0, 2) else None
    trainers = dist.new_group([0, 1])
    out = Path(out_dir)
    step_dir = out / "broadcasts" / "step_3"
    if rank < 2:
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.models.qwen3_moe.converting_qwen3_moe import prime_to_hf
        from prime_rl_amd.trainer.module_engine import ModuleEngine

        cfg = Qwen3MoeConfig(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                             num_experts=4, num_experts_per_tok=2, moe_intermediate_size=32, use_grouped_mm=False)
        torch.manual_seed(3)
        full = Qwen3MoeForCausalLM(cfg)
        for p in full.parameters():
            torch.nn.init.normal_(p, std=0.1)
        full = full.to(torch.bfloat16)
        local = Qwen3MoeForCausalLM(cfg, ep_group=trainers).to(torch.bfloat16)
        local.load_state_dict({k: (v[rank * 2:(rank + 1) * 2].clone() if ".experts.w" in k else v.clone()) for k, v in full.state_dict().items()})
        eng = ModuleEngine(local, "cpu", dp_group=trainers, dp_size=2, ep_group=trainers, ep_size=2)
        if rank == 0:
            torch.save({k: v.clone() for k, v in prime_to_hf({k: v.detach() for k, v in full.state_dict().items()}).items()}, out / "sent.pt")
        wb = RCCLWeightBroadcast(out, comm, rank, 2, 1, protocol="reference", ready_timeout=60, trainer_group=trainers)
        wb.broadcast_weights(eng, 3)
    else:
        t0 = time.time()
        while not (step_dir / "STABLE").exists():
            time.sleep(0.05)
            assert time.time() - t0 < 60
        (step_dir / NCCL_READY_MARKER).touch()
        got = {name: t.clone() for name, t in RCCLWeightReceiver(comm, 2, "cpu", "reference").receive_state_dict(step_dir)}
        torch.save(got, out / "received.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_push_of_an_expert_parallel_moe_model(tmp_path):
    """The module engine's push in the REFERENCE byte protocol (for a receiver that speaks only that; the multi-source form is
    test_sharded_push_of_an_expert_parallel_module_engine): the expert shards of the two ep ranks are gathered, names
    converted to HF's per-expert form (what vLLM's `load_weights` expects; the reference does this with `revert_weight_conversion`,
    transports/weights/nccl.py:87-98), and the inference rank receives the WHOLE model bit-exactly, layer by layer."""
    mp.start_processes(_moe_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True, start_method="spawn")
    sent, got = torch.load(tmp_path / "sent.pt"), torch.load(tmp_path / "received.pt")
    assert set(sent) == set(got) and "model.layers.1.mlp.experts.3.down_proj.weight" in got and "model.layers.0.mlp.gate.weight" in got
    for k in sent:
        assert got[k].shape == sent[k].shape and torch.equal(got[k], sent[k].to(got[k].dtype)), k
    assert not (tmp_path / "broadcasts" / "step_3" / "LAYOUT.json").exists()  # the manifest belongs to the sharded protocol


def test_sharded_layout_speaks_hf_names_for_attention_bias_models():
    """Qwen2 / Qwen2.5 carry a fused q|k|v BIAS next to the fused weight: the LAYOUT.json of the sharded push must split both into the HF checkpoint names
    that `state_dict()` and the filesystem broadcast emit, and the (offset, shape) of every entry must address exactly that tensor in the flat buffer."""
    import dataclasses

    from prime_rl_amd.trainer.engine import ModelDims, ParamStore
    from prime_rl_amd.transports.weights import build_layout

    dims = dataclasses.replace(ModelDims.tiny(layers=2, vocab=320), qk_norm=False, attention_bias=True, name="tiny-qwen2")
    store = ParamStore(dims, "cpu")
    store.random_init(seed=5)
    sd = store.state_dict()
    names, seen = set(), 0
    for bucket, b in zip(build_layout(store), store.buckets):
        flat = store.params[b.offset:b.offset + b.size]
        for name, rel, shape in bucket["params"]:
            n = 1
            for d in shape:
                n *= d
            assert torch.equal(flat[rel:rel + n].view(shape), sd[name]), name
            names.add(name)
            seen += 1
    assert names == set(sd) - ({"lm_head.weight"} if dims.tie_word_embeddings else set())
    assert any(n.endswith("q_proj.bias") for n in names) and not any("qkv_proj" in n for n in names)


def _moe_sharded_worker(rank, world, port, out_dir, n_tr):
    """n_tr trainer ranks run an expert-parallel (ep = 2) Qwen3-MoE ModuleEngine; the other ranks are inference receivers."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from prime_rl_amd.comm import TorchGroupCommunicator
    from prime_rl_amd.transports.weights import NCCL_READY_MARKER, RCCLWeightBroadcast, RCCLWeightReceiver

    comm = TorchGroupCommunicator(None)
    trainers = dist.new_group(list(range(n_tr)))
    ep_groups = [dist.new_group(list(range(b, b + 2))) for b in range(0, n_tr, 2)]            # blocks of 2 consecutive trainer ranks
    replica_groups = [dist.new_group(list(range(pos, n_tr, 2))) for pos in range(2)] if n_tr > 2 else [None, None]
    out = Path(out_dir)
    step_dir = out / "broadcasts" / "step_3"
    if rank < n_tr:
        from prime_rl_amd.trainer.engine import OptimConfig
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.module_engine import ModuleEngine

        cfg = Qwen3MoeConfig(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                             num_experts=4, num_experts_per_tok=2, moe_intermediate_size=32, mlp_only_layers=[0], use_grouped_mm=False)
        torch.manual_seed(5)
        full = Qwen3MoeForCausalLM(cfg)
        for p in full.parameters():
            torch.nn.init.normal_(p, std=0.1)
        full = full.to(torch.bfloat16)
        ep_group = ep_groups[rank // 2]
        local = Qwen3MoeForCausalLM(cfg, ep_group=ep_group).to(torch.bfloat16)
        er = rank % 2
        local.load_state_dict({k: (v[er * 2:(er + 1) * 2].clone() if ".experts.w" in k else v.clone()) for k, v in full.state_dict().items()})
        eng = ModuleEngine(local, "cpu", optim=OptimConfig(type="sign_sgd"), dp_group=trainers, dp_size=n_tr, ep_group=ep_group, ep_size=2,
                           ep_replica_group=replica_groups[er])
        if rank == 0:
            ref = ModuleEngine(full, "cpu", optim=OptimConfig(type="sign_sgd"))
            torch.save({k: v.clone() for k, v in ref.state_dict().items()}, out / "sent_moe.pt")
            plan = eng.push_plan(n_tr)
            # every stacked expert tensor is sent by the rank that owns the shard; replicated tensors are spread over all trainers
            srcs = {m["src"] for g in plan for m in g["messages"] if "experts" in m}
            assert srcs == {0, 1} and {m["src"] for g in plan for m in g["messages"] if "names" in m} == set(range(n_tr))
        RCCLWeightBroadcast(out, comm, rank, n_tr, world - n_tr, protocol="sharded", ready_timeout=60, trainer_group=trainers).broadcast_weights(eng, 3)
    else:
        t0 = time.time()
        while not (step_dir / "STABLE").exists():
            time.sleep(0.05)
            assert time.time() - t0 < 60
        if rank == n_tr:
            (step_dir / NCCL_READY_MARKER).touch()
        got = {name: t.clone() for name, t in RCCLWeightReceiver(comm, n_tr, "cpu", "sharded").receive_state_dict(step_dir)}
        torch.save(got, out / f"received_moe_{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_tr", [2, 4])
def test_sharded_push_of_an_expert_parallel_module_engine(tmp_path, n_tr):
    """The multi-source push of an nn.Module engine (MoE, ep = 2; n_tr = 4: two ep groups): nothing is gathered — the stacked expert shards travel from the
    ranks that own them, replicated tensors round-robin over the trainers — and both inference ranks end up with every per-expert HF-named tensor of the
    WHOLE model bit for bit, non-layer tensors first, then layer by layer."""
    world = n_tr + 2
    mp.start_processes(_moe_sharded_worker, args=(world, _free_port(), str(tmp_path), n_tr), nprocs=world, join=True, start_method="spawn")
    sent = torch.load(tmp_path / "sent_moe.pt")
    assert "model.layers.1.mlp.experts.3.down_proj.weight" in sent and "model.layers.1.mlp.gate.weight" in sent
    for r in range(n_tr, world):
        got = torch.load(tmp_path / f"received_moe_{r}.pt")
        assert set(got) == set(sent), (sorted(set(sent) - set(got))[:4], sorted(set(got) - set(sent))[:4])
        for k in sent:
            assert got[k].shape == sent[k].shape and torch.equal(got[k], sent[k]), k
        names = list(got)
        first_layer = next(i for i, n in enumerate(names) if n.startswith("model.layers."))
        assert all(not n.startswith("model.layers.") for n in names[:first_layer])
        assert max(i for i, n in enumerate(names) if n.startswith("model.layers.0.")) < min(i for i, n in enumerate(names) if n.startswith("model.layers.1."))
    import json

    layout = json.loads((tmp_path / "broadcasts" / "step_3" / "LAYOUT.json").read_text())
    assert layout["kind"] == "named" and [g["name"] for g in layout["groups"]] == ["non_layer", "model.layers.0", "model.layers.1"]
