"""Trainer -> inference weight push — mirrors `prime_rl.transports.weights` (base.py:10-18, __init__.py:13-26, nccl.py:33-224,
filesystem broadcast) with the transport re-planned for the xGMI mesh.

Reference (nccl.py): every trainer rank all-gathers every parameter (`full_tensor()`), then rank 0 ALONE broadcasts 37 state
dicts (non-layer + 36 layers), per dict a pickled metadata blob and one flat tensor per dtype, to all inference GPUs — a
single-source funnel bound by one xGMI link (16.4 GB at ~150 GB/s ~ 107 ms for Qwen3-8B).

Here: the flat bf16 parameter buffer is already bucketed per decoder layer and replicated on every trainer rank (ZeRO-2), so
no gather is needed.  Each bucket is cut into `dp` equal chunks and trainer rank r broadcasts chunk r to the inference ranks
over its own group {trainer r} + inference ranks: every inference GPU receives `dp` chunks over `dp` different links at once
(4+4 split: ~27 ms of wire time).  Buckets go out in layer order so the receiver can load layer-wise
(`load_weights_checkpoint_layerwise` contract, inference/vllm/worker/nccl.py:138-158).  The rendezvous is the reference's:
`broadcasts/step_N/STABLE` written by the trainer master, `NCCL_READY` awaited before any collective is enqueued
(nccl.py:195-224), names in HF checkpoint format.  The layout manifest (`LAYOUT.json`: bucket -> [(name, offset, shape)])
replaces the per-dict pickled metadata broadcast.
"""

from __future__ import annotations

import json
import math
import time
from abc import ABC, abstractmethod
from pathlib import Path

import torch
import torch.distributed as dist

from ...utils.pathing import get_broadcast_dir, get_step_path, sync_wait_for_path

NCCL_READY_MARKER = "NCCL_READY"
STABLE_MARKER = "STABLE"


class WeightBroadcast(ABC):
    def __init__(self, output_dir: Path):
        self.output_dir = Path(output_dir)

    @abstractmethod
    def broadcast_weights(self, model, step: int) -> None: ...

    def _notify_orchestrator(self, save_dir: Path) -> None:
        """Create the STABLE marker the orchestrator's weight watcher polls for (nccl.py:214-217)."""
        save_dir.mkdir(parents=True, exist_ok=True)
        (save_dir / STABLE_MARKER).touch()


def _named_tensors(model) -> dict[str, torch.Tensor]:
    """HF-named bf16 tensors of either a GRPOEngine / ParamStore (zero-copy views) or an nn.Module."""
    store = getattr(model, "store", model)
    if hasattr(model, "wait_params_ready"):
        model.wait_params_ready()  # the last optimizer step's parameter all-gathers run on the engine's comm stream
    if hasattr(store, "state_dict") and hasattr(store, "buckets"):
        return store.state_dict()
    return {k: v.detach() for k, v in model.state_dict().items()}


class FileSystemWeightBroadcast(WeightBroadcast):
    """HF-format safetensors checkpoint under `broadcasts/step_N/` + STABLE marker (the reference's fallback transport);
    broadcasts older than `keep` steps are deleted like trainer/utils.py:349-363."""

    def __init__(self, output_dir: Path, is_master: bool = True, keep: int = 2):
        super().__init__(output_dir)
        self.is_master, self.keep = is_master, keep

    @torch.no_grad()
    def broadcast_weights(self, model, step: int) -> None:
        from safetensors.torch import save_file

        if self.is_master:
            save_dir = get_step_path(get_broadcast_dir(self.output_dir), step)
            save_dir.mkdir(parents=True, exist_ok=True)
            tensors = {k: v.detach().to("cpu").contiguous().clone() for k, v in _named_tensors(model).items()}
            tmp = save_dir / "model.safetensors.tmp"
            save_file(tensors, str(tmp))
            tmp.rename(save_dir / "model.safetensors")
            self._notify_orchestrator(save_dir)
            for old in get_broadcast_dir(self.output_dir).glob("step_*"):
                try:
                    n = int(old.name.split("_")[1])
                except ValueError:
                    continue
                if n < step - self.keep + 1:
                    for f in old.iterdir():
                        f.unlink()
                    old.rmdir()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()


def build_layout(store) -> list[dict]:
    """Manifest of the flat parameter buffer: one entry per bucket with HF-named (name, offset-in-bucket, shape) triples."""
    d = store.dims
    out = []
    for b in store.buckets:
        entries = []
        for name, off, shape in b.params:
            rel = off - b.offset
            if name.endswith("qkv_proj.weight"):
                base, h = name[: -len("qkv_proj.weight")], shape[1]
                entries += [(base + "q_proj.weight", rel, [d.q_dim, h]), (base + "k_proj.weight", rel + d.q_dim * h, [d.kv_dim, h]),
                            (base + "v_proj.weight", rel + (d.q_dim + d.kv_dim) * h, [d.kv_dim, h])]
            elif name.endswith("gate_up_proj.weight"):
                base, h = name[: -len("gate_up_proj.weight")], shape[1]
                entries += [(base + "gate_proj.weight", rel, [d.intermediate_size, h]), (base + "up_proj.weight", rel + d.intermediate_size * h, [d.intermediate_size, h])]
            else:
                entries.append((name, rel, list(shape)))
        out.append(dict(name=b.name, size=b.size, params=entries))
    return out


class RCCLWeightBroadcast(WeightBroadcast):
    """Sharded multi-source push.  World layout of the push communicator: trainer ranks [0, dp) then inference ranks
    [dp, dp + n_inf) — created by `init_push_groups` on BOTH sides (collective call)."""

    def __init__(self, output_dir: Path, trainer_rank: int, n_trainers: int, groups: list, ready_timeout: float | None = None):
        super().__init__(output_dir)
        self.rank, self.dp, self.groups, self.ready_timeout = trainer_rank, n_trainers, groups, ready_timeout

    @torch.no_grad()
    def broadcast_weights(self, model, step: int) -> None:
        store = getattr(model, "store", model)
        if hasattr(model, "wait_params_ready"):
            model.wait_params_ready()
        save_dir = get_step_path(get_broadcast_dir(self.output_dir), step)
        if self.rank == 0:
            save_dir.mkdir(parents=True, exist_ok=True)
            (save_dir / "LAYOUT.json").write_text(json.dumps(build_layout(store)))
            self._notify_orchestrator(save_dir)
            sync_wait_for_path(save_dir / NCCL_READY_MARKER, interval=0.05, timeout=self.ready_timeout)
        # every trainer rank waits for the inference pool before enqueuing collectives (nccl.py:203-211)
        dist.barrier(group=self.groups[-1])
        works = []
        for b in store.buckets:
            chunk = b.size // self.dp
            view = store.params[b.offset + self.rank * chunk: b.offset + (self.rank + 1) * chunk]
            works.append(dist.broadcast(view, src=self.rank, group=self.groups[self.rank], async_op=True))
        for w in works:
            w.wait()


class RCCLWeightReceiver:
    """Inference-side half (replaces inference/vllm/worker/nccl.py:28-89): posts one receive per (bucket, trainer chunk) and yields
    `(hf_name, tensor)` bucket by bucket for a layer-wise `load_weights`."""

    def __init__(self, n_trainers: int, groups: list, device):
        self.dp, self.groups, self.device = n_trainers, groups, torch.device(device)

    @torch.no_grad()
    def receive_state_dict(self, weight_dir: Path):
        layout = json.loads((Path(weight_dir) / "LAYOUT.json").read_text())
        dist.barrier(group=self.groups[-1])
        pending = []
        for b in layout:
            buf = torch.empty(b["size"], dtype=torch.bfloat16, device=self.device)
            chunk = b["size"] // self.dp
            works = [dist.broadcast(buf[r * chunk:(r + 1) * chunk], src=r, group=self.groups[r], async_op=True) for r in range(self.dp)]
            pending.append((b, buf, works))
        for b, buf, works in pending:  # all receives are in flight; hand layers over in order as they complete
            for w in works:
                w.wait()
            for name, off, shape in b["params"]:
                yield name, buf[off: off + math.prod(shape)].view(shape)


def init_push_groups(n_trainers: int, n_inference: int) -> list:
    """Collective on every rank of the push world (trainers first): groups[r] = {trainer r} + all inference ranks, groups[-1] = world."""
    inf = list(range(n_trainers, n_trainers + n_inference))
    groups = [dist.new_group([r] + inf) for r in range(n_trainers)]
    groups.append(dist.new_group(list(range(n_trainers + n_inference))))
    return groups


def setup_weight_broadcast(output_dir: Path, config, **kw) -> WeightBroadcast:
    """`weight_broadcast.type` dispatch (transports/weights/__init__.py:13-26): "nccl" -> RCCL sharded push, "filesystem" -> checkpoint."""
    kind = getattr(config, "type", "filesystem")
    if kind == "filesystem":
        return FileSystemWeightBroadcast(output_dir, is_master=kw.get("is_master", True))
    if kind in ("nccl", "rccl"):
        return RCCLWeightBroadcast(output_dir, kw["trainer_rank"], kw["n_trainers"], kw["groups"], kw.get("ready_timeout"))
    raise ValueError(f"Invalid weight broadcast type: {kind} (nixl / ModelExpress RDMA is multi-node and stays with the reference)")
