"""CPU suite (gloo, 2 trainers + 1 inference rank): the sharded weight push delivers every HF-named parameter bit-exactly
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
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.trainer.engine import ModelDims, ParamStore
    from prime_rl_amd.transports.weights import NCCL_READY_MARKER, RCCLWeightBroadcast, RCCLWeightReceiver, init_push_groups

    n_tr, n_inf = 2, 1
    groups = init_push_groups(n_tr, n_inf)
    out = Path(out_dir)
    step_dir = out / "broadcasts" / "step_7"
    if rank < n_tr:
        store = ParamStore(ModelDims.tiny(layers=2, vocab=320), "cpu", dp_rank=rank, dp_size=n_tr)
        store.random_init(seed=11)
        if rank == 0:
            torch.save({k: v.clone() for k, v in store.state_dict().items()}, out / "sent.pt")
        RCCLWeightBroadcast(out, rank, n_tr, groups, ready_timeout=60).broadcast_weights(store, 7)
    else:
        t0 = time.time()
        while not (step_dir / "STABLE").exists():  # the orchestrator's WeightWatcher polls this marker (watcher.py:91-158)
            time.sleep(0.05)
            assert time.time() - t0 < 60
        (step_dir / NCCL_READY_MARKER).touch()  # inference engines are paused and listening
        got = {name: t.clone() for name, t in RCCLWeightReceiver(n_tr, groups, "cpu").receive_state_dict(step_dir)}
        torch.save(got, out / "received.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_push_round_trip(tmp_path):
    mp.start_processes(_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True, start_method="spawn")
    sent = torch.load(tmp_path / "sent.pt")
    got = torch.load(tmp_path / "received.pt")
    assert set(sent) == set(got) and "model.layers.1.self_attn.k_proj.weight" in got and "lm_head.weight" in got
    for k in sent:
        assert got[k].shape == sent[k].shape and torch.equal(got[k], sent[k]), k
    assert (tmp_path / "broadcasts" / "step_7" / "STABLE").exists() and (tmp_path / "broadcasts" / "step_7" / "LAYOUT.json").exists()
    # layer order on the wire: non-layer (embed) first, then layers ascending, head last
    names = list(got)
    assert names[0] == "model.embed_tokens.weight" and names[-1] == "lm_head.weight"


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
