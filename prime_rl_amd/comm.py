"""Communicators for the trainer -> inference weight push (and any other exchange that must live NEXT to the host framework's
own process group).

The reference builds, for the push, a raw NCCL communicator on a stateless TCP rendezvous — vllm's `PyNcclCommunicator` over
`StatelessProcessGroup.create(host, port, rank, world_size)` (src/prime_rl/transports/weights/nccl.py:128-136 on the trainer,
inference/vllm/worker/nccl.py:63-76 inside every vLLM worker) — because a vLLM worker already owns a default torch process group
that the trainer is not part of.  `RCCLCommunicator` is that object for MI355X: a raw RCCL communicator behind the C ABI
(`pm_comm_*`, include/prime_mi355x.h) bootstrapped from a `torch.distributed.TCPStore` at `weight_broadcast.host:port`
(rank 0 = the trainer master hosts the store, like the reference).  All operations are enqueued on the CURRENT torch stream.

`TorchGroupCommunicator` has the same surface on top of a torch.distributed group (gloo on CPU for the test-suite, or an existing
nccl group); the push protocols in transports/weights are written against this surface only.
"""

from __future__ import annotations

import ctypes
import time
from datetime import timedelta

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from ._lib import PrimeMI355XError, check

_PM_DT = {torch.float32: _lib.PM_DTYPE_F32, torch.bfloat16: _lib.PM_DTYPE_BF16, torch.int64: _lib.PM_DTYPE_I64, torch.uint8: _lib.PM_DTYPE_U8}


class StatelessRendezvous:
    """Out-of-band key/value exchange for bootstrapping a communicator: a TCPStore hosted by rank 0 at host:port."""

    def __init__(self, host: str, port: int, rank: int, world_size: int, timeout: float = 1200.0):
        self.rank, self.world_size = rank, world_size
        self.store = dist.TCPStore(host, port, world_size, is_master=(rank == 0), timeout=timedelta(seconds=timeout), wait_for_workers=False)
        self._round = 0

    def broadcast_bytes(self, payload: bytes | None, key: str) -> bytes:
        """Rank 0's `payload` to everyone."""
        if self.rank == 0:
            self.store.set(key, payload)
            return payload
        return self.store.get(key)

    def barrier(self, tag: str = "barrier") -> None:
        self._round += 1
        key = f"{tag}/{self._round}"
        self.store.add(key, 1)
        deadline = time.monotonic() + 1200.0
        while int(self.store.add(key, 0)) < self.world_size:
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous barrier {key} timed out")
            time.sleep(0.005)


class RCCLCommunicator:
    """Raw RCCL communicator over xGMI (pm_comm_*).  `device` must be this process' GPU; tensors must live on it."""

    def __init__(self, rendezvous: StatelessRendezvous, device: torch.device | int | str, name: str = "push"):
        lib = _lib.load()
        if not lib.pm_comm_available():
            raise PrimeMI355XError("librccl is not loadable: " + (lib.pm_last_error() or b"").decode())
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.rank, self.size = rendezvous.rank, rendezvous.world_size
        uid = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(128)
            check(lib.pm_comm_unique_id(buf), "pm_comm_unique_id")
            uid = buf.raw
        uid = rendezvous.broadcast_bytes(uid, f"pm_comm/{name}/unique_id")
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.pm_comm_init(ctypes.byref(handle), uid, self.size, self.rank), "pm_comm_init")
        self._comm = handle
        rendezvous.barrier(f"pm_comm/{name}/ready")

    # -- helpers ------------------------------------------------------------------------------------------------------------
    def _args(self, t: Tensor):
        if not t.is_cuda or t.device != self.device or not t.is_contiguous():
            raise PrimeMI355XError(f"communicator tensors must be contiguous and live on {self.device}, got {t.device}")
        if t.dtype not in _PM_DT:
            raise PrimeMI355XError(f"unsupported dtype {t.dtype}")
        return t.data_ptr(), t.numel(), _PM_DT[t.dtype], torch.cuda.current_stream(self.device).cuda_stream

    # -- operations (enqueued on the current stream) ---------------------------------------------------------------------------
    def broadcast(self, t: Tensor, src: int) -> None:
        p, n, dt, st = self._args(t)
        check(_lib.load().pm_comm_broadcast(self._comm, p, p, n, dt, src, st), "pm_comm_broadcast")

    def all_reduce(self, t: Tensor) -> None:
        p, n, dt, st = self._args(t)
        check(_lib.load().pm_comm_all_reduce(self._comm, p, p, n, dt, st), "pm_comm_all_reduce")

    def reduce_scatter(self, out: Tensor, inp: Tensor) -> None:
        po, no, dt, st = self._args(out)
        pi, ni, _, _ = self._args(inp)
        if ni != no * self.size:
            raise PrimeMI355XError("reduce_scatter: input must hold size * output elements")
        check(_lib.load().pm_comm_reduce_scatter(self._comm, pi, po, no, dt, st), "pm_comm_reduce_scatter")

    def all_gather(self, out: Tensor, inp: Tensor) -> None:
        po, no, dt, st = self._args(out)
        pi, ni, _, _ = self._args(inp)
        if no != ni * self.size:
            raise PrimeMI355XError("all_gather: output must hold size * input elements")
        check(_lib.load().pm_comm_all_gather(self._comm, pi, po, ni, dt, st), "pm_comm_all_gather")

    def all_to_all(self, out: Tensor, inp: Tensor) -> None:
        po, no, dt, st = self._args(out)
        pi, ni, _, _ = self._args(inp)
        if no != ni or ni % self.size:
            raise PrimeMI355XError("all_to_all: equal sizes divisible by the communicator size")
        check(_lib.load().pm_comm_all_to_all(self._comm, pi, po, ni // self.size, dt, st), "pm_comm_all_to_all")

    def send(self, t: Tensor, peer: int) -> None:
        p, n, dt, st = self._args(t)
        check(_lib.load().pm_comm_send(self._comm, p, n, dt, peer, st), "pm_comm_send")

    def recv(self, t: Tensor, peer: int) -> None:
        p, n, dt, st = self._args(t)
        check(_lib.load().pm_comm_recv(self._comm, p, n, dt, peer, st), "pm_comm_recv")

    def group_start(self) -> None:
        check(_lib.load().pm_comm_group_start(), "pm_comm_group_start")

    def group_end(self) -> None:
        check(_lib.load().pm_comm_group_end(), "pm_comm_group_end")

    def barrier(self) -> None:
        t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.all_reduce(t)
        torch.cuda.current_stream(self.device).synchronize()

    def close(self) -> None:
        if self._comm:
            check(_lib.load().pm_comm_destroy(self._comm), "pm_comm_destroy")
            self._comm = ctypes.c_void_p()


class TorchGroupCommunicator:
    """The same surface over a torch.distributed group (gloo for the CPU suite).  Ranks are GROUP ranks."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self._pending: list | None = None

    def _global(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def broadcast(self, t: Tensor, src: int) -> None:
        dist.broadcast(t, src=self._global(src), group=self.group)

    def all_reduce(self, t: Tensor) -> None:
        dist.all_reduce(t, group=self.group)

    def reduce_scatter(self, out: Tensor, inp: Tensor) -> None:
        dist.reduce_scatter_tensor(out, inp, group=self.group)

    def all_gather(self, out: Tensor, inp: Tensor) -> None:
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_to_all(self, out: Tensor, inp: Tensor) -> None:
        dist.all_to_all_single(out, inp, group=self.group)

    def send(self, t: Tensor, peer: int) -> None:
        op = dist.P2POp(dist.isend, t, self._global(peer), group=self.group)
        if self._pending is not None:
            self._pending.append(op)
        else:
            for w in dist.batch_isend_irecv([op]):
                w.wait()

    def recv(self, t: Tensor, peer: int) -> None:
        op = dist.P2POp(dist.irecv, t, self._global(peer), group=self.group)
        if self._pending is not None:
            self._pending.append(op)
        else:
            for w in dist.batch_isend_irecv([op]):
                w.wait()

    def group_start(self) -> None:
        self._pending = []

    def group_end(self) -> None:
        ops, self._pending = self._pending or [], None
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    def close(self) -> None:
        pass
