"""Trainer -> inference weight push — mirrors `prime_rl.transports.weights` (base.py:10-18, __init__.py:13-26, nccl.py:33-224,
filesystem broadcast) with the transport re-planned for the xGMI mesh.

Reference (nccl.py): every trainer rank all-gathers every parameter (`full_tensor()`), then rank 0 ALONE broadcasts 37 state
dicts (non-layer + 36 layers), per dict a pickled metadata blob and one flat tensor per dtype, to all inference GPUs — a
single-source funnel bound by one xGMI link (16.4 GB at ~150 GB/s ~ 107 ms for Qwen3-8B).

Here: the flat bf16 parameter buffer is already bucketed per decoder layer and replicated on every trainer rank (ZeRO-2), so
no gather is needed.  Each bucket is cut into `dp` equal chunks and trainer rank r SENDS chunk r to every inference rank
(grouped point-to-point over a raw RCCL communicator, prime_rl_amd/comm.py): every inference GPU receives `dp` chunks over `dp`
different direct xGMI links at once (4+4 split: ~27 ms of wire time), nothing is forwarded.  Buckets go out in layer order so
the receiver can load layer-wise (`load_weights_checkpoint_layerwise` contract, inference/vllm/worker/nccl.py:138-158).  The
rendezvous is the reference's: a TCP store at `weight_broadcast.host:port` bootstraps the communicator, `broadcasts/step_N/STABLE`
is written by the trainer master and `NCCL_READY` awaited before anything is enqueued (nccl.py:195-224), names in HF checkpoint
format.  The layout manifest (`LAYOUT.json`: bucket -> [(name, offset, shape)]) replaces the per-dict pickled metadata broadcast.
The push is asynchronous (own stream; the next optimizer step waits for it on the device).  For a receiver that speaks the
reference's byte protocol there is `protocol="reference"` (single source, pickled metadata, one flat tensor per dtype).
nn.Module engines (MoE models; trainer/module_engine.py) push sharded too, on NAMED tensors: `ModuleEngine.push_plan()` lists one message per tensor
with its source rank — replicated tensors round-robin over the trainers, every stacked expert tensor from the ep rank that owns the shard — so nothing is
gathered before the push and every inference rank receives each tensor straight from its owner (`push_named` / `receive_named`).
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

    def __init__(self, output_dir: Path, is_master: bool = True, keep: int = 2, group=None):
        super().__init__(output_dir)
        self.is_master, self.keep, self.group = is_master, keep, group

    @torch.no_grad()
    def broadcast_weights(self, model, step: int) -> None:
        from safetensors.torch import save_file

        # an expert-parallel model assembles its full state dict with collectives: then every rank takes part, the master writes
        named = _named_tensors(model) if (self.is_master or getattr(model, "collective_state_dict", False)) else None
        if self.is_master:
            save_dir = get_step_path(get_broadcast_dir(self.output_dir), step)
            save_dir.mkdir(parents=True, exist_ok=True)
            tensors = {k: v.detach().to("cpu").contiguous().clone() for k, v in named.items()}
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
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.barrier(group=self.group)


def build_layout(store) -> list[dict]:
    """Manifest of the flat parameter buffer: one entry per bucket with HF-named (name, offset-in-bucket, shape) triples."""
    d = store.dims
    if getattr(d, "head_padded", False):
        raise NotImplementedError("the sharded push sends the flat buffers as they are stored; models whose 64-wide heads are stored zero-padded to 128 push "
                                  "de-padded HF tensors with the reference protocol (PRIME_PUSH_PROTOCOL=reference) or the filesystem broadcast")
    out = []
    for b in store.buckets:
        entries = []
        for name, off, shape in b.params:
            rel = off - b.offset
            if name.endswith("qkv_proj.weight"):
                base, h = name[: -len("qkv_proj.weight")], shape[1]
                entries += [(base + "q_proj.weight", rel, [d.q_dim, h]), (base + "k_proj.weight", rel + d.q_dim * h, [d.kv_dim, h]),
                            (base + "v_proj.weight", rel + (d.q_dim + d.kv_dim) * h, [d.kv_dim, h])]
            elif name.endswith("qkv_proj.bias"):  # Qwen2 / Qwen2.5: the receiver sees HF checkpoint names, like state_dict() and the filesystem broadcast
                base = name[: -len("qkv_proj.bias")]
                entries += [(base + "q_proj.bias", rel, [d.q_dim]), (base + "k_proj.bias", rel + d.q_dim, [d.kv_dim]),
                            (base + "v_proj.bias", rel + d.q_dim + d.kv_dim, [d.kv_dim])]
            elif name.endswith("gate_up_proj.weight"):
                base, h = name[: -len("gate_up_proj.weight")], shape[1]
                entries += [(base + "gate_proj.weight", rel, [d.intermediate_size, h]), (base + "up_proj.weight", rel + d.intermediate_size * h, [d.intermediate_size, h])]
            else:
                entries.append((name, rel, list(shape)))
        out.append(dict(name=b.name, size=b.size, params=entries))
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Protocol 1 — "sharded": the xGMI-native push.  Push world = trainer ranks [0, T) then inference ranks [T, T + I).  Every bucket of
# the flat bf16 buffer is cut into T equal chunks; trainer r SENDS chunk r to every inference rank, every inference rank RECEIVES T
# chunks from T different trainers — point-to-point, grouped (ncclGroupStart/End), so on the xGMI full mesh every (trainer,
# inference) pair uses its own direct link and no GPU forwards anything.  No metadata travels: LAYOUT.json sits next to STABLE.
def push_sharded(comm, store, trainer_rank: int, n_trainers: int, n_inference: int) -> None:
    for b in store.buckets:
        chunk = b.size // n_trainers
        view = store.params[b.offset + trainer_rank * chunk: b.offset + (trainer_rank + 1) * chunk]
        comm.group_start()
        for i in range(n_inference):
            comm.send(view, n_trainers + i)
        comm.group_end()


def receive_sharded(comm, layout: list[dict], n_trainers: int, device):
    """Yields (hf_name, tensor) bucket by bucket (layer order) for a layer-wise `load_weights`."""
    for b in layout:
        buf = torch.empty(b["size"], dtype=torch.bfloat16, device=device)
        chunk = b["size"] // n_trainers
        comm.group_start()
        for r in range(n_trainers):
            comm.recv(buf[r * chunk:(r + 1) * chunk], r)
        comm.group_end()
        for name, off, shape in b["params"]:
            yield name, buf[off: off + math.prod(shape)].view(shape)


# Protocol 1b — "sharded" for nn.Module engines (MoE models, expert parallel): the same idea on NAMED tensors.  `plan` (ModuleEngine.push_plan) lists,
# group by group, one message per tensor with its source rank: replicated tensors round-robin over the trainers, the stacked expert tensors from the rank
# that owns the shard.  Every inference rank receives each message straight from its source; per-expert HF tensors are views of the received stack.
def push_named(comm, plan: list[dict], tensors: list[list], trainer_rank: int, n_trainers: int, n_inference: int) -> None:
    for g, row in zip(plan, tensors):
        comm.group_start()
        for m, t in zip(g["messages"], row):
            if m["src"] == trainer_rank:
                for i in range(n_inference):
                    comm.send(t, n_trainers + i)
        comm.group_end()


def receive_named(comm, plan: list[dict], device):
    """Yields (hf_name, tensor) group by group: the non-layer tensors, then layer 0, 1, ..."""
    for g in plan:
        bufs = [torch.empty(m["shape"], dtype=torch.bfloat16, device=device) for m in g["messages"]]
        comm.group_start()
        for m, b in zip(g["messages"], bufs):
            comm.recv(b, m["src"])
        comm.group_end()
        for m, b in zip(g["messages"], bufs):
            if "experts" in m:
                ex = m["experts"]
                for e in range(b.shape[0]):
                    yield f"{ex['base']}{ex['first'] + e}.{ex['proj']}.weight", b[e]
            else:
                for name in m["names"]:
                    yield name, b


# ---------------------------------------------------------------------------------------------------------------------------
# Protocol 2 — "reference": the byte stream of the reference's single-source broadcast, for a receiver that speaks it
# (transports/weights/nccl.py:27-84 sender, inference/vllm/worker/nccl.py:28-89 receiver).  World =
# trainer master (rank 0) + inference ranks; per step: i64[1] number of state dicts, then per dict: i64[1] pickle length, u8[len]
# pickled {dtype: [(key, shape, numel), ...]}, then one flat tensor per dtype.  Dict order: non-layer weights, then layer 0, 1, ...
def _layer_state_dicts(state_dict: dict[str, torch.Tensor], layer_prefix: str = "model.layers."):
    layer_ids = sorted({int(k[len(layer_prefix):].split(".")[0]) for k in state_dict if k.startswith(layer_prefix)})
    yield {k: v for k, v in state_dict.items() if not k.startswith(layer_prefix)}
    for i in layer_ids:
        yield {k: v for k, v in state_dict.items() if k.startswith(f"{layer_prefix}{i}.")}


def send_reference_protocol(comm, state_dict: dict[str, torch.Tensor], device) -> None:
    import pickle

    dicts = list(_layer_state_dicts(state_dict))
    comm.broadcast(torch.tensor([len(dicts)], dtype=torch.int64, device=device), 0)
    for sd in dicts:
        by_dtype: dict[torch.dtype, list[tuple[str, torch.Tensor]]] = {}
        for k, v in sd.items():
            by_dtype.setdefault(v.dtype, []).append((k, v))
        meta = pickle.dumps({dt: [(k, v.shape, v.numel()) for k, v in items] for dt, items in by_dtype.items()})
        comm.broadcast(torch.tensor([len(meta)], dtype=torch.int64, device=device), 0)
        comm.broadcast(torch.frombuffer(bytearray(meta), dtype=torch.uint8).to(device), 0)
        for items in by_dtype.values():
            comm.broadcast(torch.cat([v.reshape(-1) for _, v in items]).contiguous(), 0)


def receive_reference_protocol(comm, device):
    import pickle

    count = torch.zeros(1, dtype=torch.int64, device=device)
    comm.broadcast(count, 0)
    for _ in range(int(count.item())):
        size = torch.zeros(1, dtype=torch.int64, device=device)
        comm.broadcast(size, 0)
        blob = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
        comm.broadcast(blob, 0)
        meta = pickle.loads(bytes(blob.cpu().numpy()))
        for dtype, infos in meta.items():
            flat = torch.empty(sum(n for _, _, n in infos), dtype=dtype, device=device)
            comm.broadcast(flat, 0)
            off = 0
            for key, shape, numel in infos:
                yield key, flat[off: off + numel].view(shape).clone()
                off += numel


class RCCLWeightBroadcast(WeightBroadcast):
    """Trainer half of the push.  `comm` is a communicator of prime_rl_amd.comm (RCCLCommunicator on the GPU).

    Asynchronous on the GPU: the sends are enqueued on a dedicated push stream behind an event that marks the parameters final, and
    the call returns; the engine only waits for the push (`add_param_reader`) right before the NEXT optimizer step rewrites the
    parameters, so the transfer overlaps with the next step's data wait, forward and backward.  `last_transfer_seconds()` reports the
    device time of the previous push from stream events."""

    def __init__(self, output_dir: Path, comm, trainer_rank: int, n_trainers: int, n_inference: int, protocol: str = "sharded",
                 ready_timeout: float | None = None, trainer_group=None):
        super().__init__(output_dir)
        if protocol not in ("sharded", "reference"):
            raise ValueError(f"unknown push protocol {protocol!r}")
        if protocol == "reference" and n_trainers != 1 and comm is not None and comm.size != n_inference + 1:
            raise ValueError("the reference protocol runs on a communicator of the trainer master + the inference ranks")
        self.comm, self.rank, self.dp, self.n_inf, self.protocol = comm, trainer_rank, n_trainers, n_inference, protocol
        self.ready_timeout, self.trainer_group = ready_timeout, trainer_group
        self._stream = None
        self._events: tuple | None = None
        self._in_flight: tuple | None = None  # (completion event, None) of the previous asynchronous push (its temporaries are owned by record_stream)

    @torch.no_grad()
    def broadcast_weights(self, model, step: int) -> None:
        store = getattr(model, "store", model)
        flat = hasattr(store, "buckets")  # GRPOEngine / ParamStore; otherwise an nn.Module engine (trainer/module_engine.py: MoE, expert parallel)
        if not flat and self.protocol != "reference" and not hasattr(model, "push_plan"):
            raise NotImplementedError("the sharded push needs either the flat parameter buckets of the dense engine or an engine with push_plan() / push_tensors() "
                                      "(trainer/module_engine.py); anything else pushes with the reference byte protocol: PRIME_PUSH_PROTOCOL=reference")
        if hasattr(model, "wait_params_ready"):
            model.wait_params_ready()
        plan = model.push_plan(self.dp) if (not flat and self.protocol == "sharded") else None
        # reference protocol of an nn.Module engine: HF-named tensors; an expert-parallel model assembles them with collectives over its ep group, so every
        # trainer rank takes part.  The sharded protocol gathers nothing: every rank sends what it holds (`plan`).
        if flat or plan is not None:
            named = None
        else:
            named = model.state_dict() if (self.rank == 0 or getattr(model, "collective_state_dict", False)) else None
        if plan is not None:
            named = {f"{gi}.{mi}": t for gi, row in enumerate(model.push_tensors(plan, self.rank)) for mi, t in enumerate(row) if t is not None}
        device = store.params.device if flat else torch.device(model.device)
        save_dir = get_step_path(get_broadcast_dir(self.output_dir), step)
        if self.rank == 0:
            save_dir.mkdir(parents=True, exist_ok=True)
            if flat and (self.protocol == "sharded" or not getattr(store.dims, "head_padded", False)):
                (save_dir / "LAYOUT.json").write_text(json.dumps(build_layout(store)))  # refuses padded-head storage: those models push by name
            elif plan is not None:
                (save_dir / "LAYOUT.json").write_text(json.dumps(dict(kind="named", groups=[dict(name=g["name"], messages=[{k: v for k, v in m.items() if k != "key"}
                                                                                                                          for m in g["messages"]]) for g in plan])))
            self._notify_orchestrator(save_dir)
            sync_wait_for_path(save_dir / NCCL_READY_MARKER, interval=0.05, timeout=self.ready_timeout)
        # every trainer rank waits for the inference pool before enqueuing collectives (nccl.py:203-211)
        if self.dp > 1 and dist.is_initialized():
            dist.barrier(group=self.trainer_group)
        if device.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=device)
            if self._in_flight is not None:  # the previous push has long finished (the optimizer step waited for it)
                self._in_flight[0].synchronize()
                self._in_flight = None
            final = torch.cuda.Event()
            final.record(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(final)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self._stream)
                self._push(store, named, device, plan)
                t1.record(self._stream)
            self._events = (t0, t1)
            # `named` of an nn.Module engine holds TEMPORARIES allocated on the main stream (bf16 copies of f32 router gates, all-gathered expert
            # tensors) that the push stream reads later.  record_stream() hands their lifetime to the caching allocator: a block is reused only after the
            # push-stream work enqueued so far has finished, so the references can be dropped NOW — an expert-parallel MoE model's gathered state dict
            # (tens of GB) is released when the push completes instead of being pinned through the whole next training step.
            if named is not None:
                for t in named.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(self._stream)
            del named
            self._in_flight = (t1, None)
            if hasattr(model, "add_param_reader"):
                model.add_param_reader(t1)  # the next optimizer step waits for the push on the device, the host never blocks
            else:
                torch.cuda.current_stream().wait_event(t1)
        else:
            self._push(store, named, device, plan)

    def _push(self, store, named, device, plan=None) -> None:
        if plan is not None:
            rows = [[named.get(f"{gi}.{mi}") for mi in range(len(g["messages"]))] for gi, g in enumerate(plan)]
            push_named(self.comm, plan, rows, self.rank, self.dp, self.n_inf)
        elif self.protocol == "sharded":
            push_sharded(self.comm, store, self.rank, self.dp, self.n_inf)
        elif self.rank == 0:
            send_reference_protocol(self.comm, named if named is not None else store.state_dict(), device)

    def last_transfer_seconds(self) -> float | None:
        """Device time of the most recent push (synchronises with its completion)."""
        if self._events is None:
            return None
        t0, t1 = self._events
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e-3


class RCCLWeightReceiver:
    """Inference-side half (replaces inference/vllm/worker/nccl.py:28-89): yields `(hf_name, tensor)` in layer order."""

    def __init__(self, comm, n_trainers: int, device, protocol: str = "sharded"):
        self.comm, self.dp, self.device, self.protocol = comm, n_trainers, torch.device(device), protocol

    @torch.no_grad()
    def receive_state_dict(self, weight_dir: Path | None = None):
        if self.protocol == "reference":
            yield from receive_reference_protocol(self.comm, self.device)
            return
        layout = json.loads((Path(weight_dir) / "LAYOUT.json").read_text())
        if isinstance(layout, dict) and layout.get("kind") == "named":  # an nn.Module engine's push: named messages, one source rank each
            yield from receive_named(self.comm, layout["groups"], self.device)
        else:
            yield from receive_sharded(self.comm, layout, self.dp, self.device)


def init_push_communicator(host: str, port: int, rank: int, n_trainers: int, n_inference: int, device, timeout: float = 1200.0,
                           protocol: str = "sharded"):
    """Both sides call this (trainers with rank in [0, T), inference GPU g with rank T + g): a raw RCCL communicator bootstrapped from a
    TCP store at host:port, the reference's rendezvous (transports/weights/nccl.py:128-136, configs/trainer.py:557-578).  With the
    reference protocol only the trainer master joins (world = 1 + I); other trainer ranks get None."""
    from ...comm import RCCLCommunicator, StatelessRendezvous

    if protocol == "reference":
        if rank != 0 and rank < n_trainers:
            return None
        world = n_inference + 1
        rank = 0 if rank == 0 else rank - n_trainers + 1
    else:
        world = n_trainers + n_inference
    return RCCLCommunicator(StatelessRendezvous(host, port, rank, world, timeout), device)


def setup_weight_broadcast(output_dir: Path, config, **kw) -> WeightBroadcast:
    """`weight_broadcast.type` dispatch (transports/weights/__init__.py:13-26): "nccl" -> RCCL push, "filesystem" -> checkpoint.
    For "nccl" pass `comm` (a prime_rl_amd.comm communicator), `trainer_rank`, `n_trainers`, `n_inference`."""
    kind = getattr(config, "type", "filesystem")
    if kind == "filesystem":
        return FileSystemWeightBroadcast(output_dir, is_master=kw.get("is_master", True), group=kw.get("group"))
    if kind in ("nccl", "rccl"):
        missing = [k for k in ("comm", "trainer_rank", "n_trainers", "n_inference") if k not in kw]
        if missing:
            raise TypeError(f"weight_broadcast.type = {kind!r} needs {missing}: build the push communicator with init_push_communicator() "
                            "(trainer/rl/train.py does) and pass it here")
        return RCCLWeightBroadcast(output_dir, kw["comm"], kw["trainer_rank"], kw["n_trainers"], kw["n_inference"], kw.get("protocol", "sharded"),
                                   kw.get("ready_timeout"), kw.get("trainer_group"))
    raise ValueError(f"Invalid weight broadcast type: {kind} (nixl / ModelExpress RDMA is multi-node and stays with the reference)")
