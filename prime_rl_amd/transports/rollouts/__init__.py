"""Rollout transport — mirrors `prime_rl.transports.rollouts` (base.py:10-55, filesystem.py:8-62): `MicroBatchSender.send(grid)` /
`MicroBatchReceiver.{wait, can_receive, receive}`; filesystem layout `rollouts/step_{k}/rank_{r}.bin` written tmp+rename.
ZMQ PUB/SUB (rollouts/zmq.py) is control-plane networking and stays with the reference."""

from __future__ import annotations

import asyncio
from abc import ABC, abstractmethod
from pathlib import Path

import msgpack

from ...utils.pathing import get_rollout_dir, get_step_path, sync_wait_for_path
from .types import EncodedTensor, MicroBatch, RoutedExperts, TrainingSample, from_wire, to_wire


def encode_micro_batches(mbs: list[MicroBatch]) -> bytes:
    return msgpack.packb([to_wire(mb) for mb in mbs], use_bin_type=True)


def decode_micro_batches(buf: bytes) -> list[MicroBatch]:
    return [from_wire(MicroBatch, arr) for arr in msgpack.unpackb(buf, raw=False)]


class MicroBatchSender(ABC):
    def __init__(self, output_dir: Path, data_world_size: int):
        self.output_dir, self.data_world_size = Path(output_dir), data_world_size

    @abstractmethod
    async def send(self, micro_batch_grid: list[list[MicroBatch]]) -> None: ...

    def close(self) -> None:
        pass


class MicroBatchReceiver(ABC):
    def __init__(self, output_dir: Path, data_rank: int):
        self.output_dir, self.data_rank = Path(output_dir), data_rank

    @abstractmethod
    def wait(self) -> None: ...

    @abstractmethod
    def can_receive(self) -> bool: ...

    @abstractmethod
    def receive(self) -> list[MicroBatch]: ...

    def close(self) -> None:
        pass


class FileSystemMicroBatchSender(MicroBatchSender):
    def __init__(self, output_dir: Path, data_world_size: int, current_step: int = 0, wire_format: str = "msgpack"):
        super().__init__(output_dir, data_world_size)
        self.rollout_dir = get_rollout_dir(output_dir)
        self.current_step = current_step
        self.wire_format = wire_format  # "binary": raw-array step blobs (binary.py) instead of msgpack-of-lists

    async def send(self, micro_batch_grid: list[list[MicroBatch]]) -> None:
        assert len(micro_batch_grid) == self.data_world_size, "Number of micro batch lists must match data world size"
        for lst in micro_batch_grid:
            assert len(lst) == len(micro_batch_grid[0]), "All micro batch lists must have the same length"
        step_path = get_step_path(self.rollout_dir, self.current_step)
        step_path.mkdir(parents=True, exist_ok=True)
        await asyncio.to_thread(self._encode_and_write, micro_batch_grid, step_path)
        self.current_step += 1

    def _encode_and_write(self, grid, step_path: Path) -> None:
        if self.wire_format == "binary":
            from .binary import write_step

            write_step(self.output_dir, self.current_step, grid)
            return
        for rank in range(self.data_world_size):
            tmp = step_path / f"rank_{rank}.bin.tmp"
            tmp.write_bytes(encode_micro_batches(grid[rank]))
            tmp.rename(step_path / f"rank_{rank}.bin")


class FileSystemMicroBatchReceiver(MicroBatchReceiver):
    def __init__(self, output_dir: Path, data_rank: int, current_step: int = 0):
        super().__init__(output_dir, data_rank)
        self.rollout_dir = get_rollout_dir(output_dir)
        self.current_step = current_step

    def _path(self) -> Path:
        return get_step_path(self.rollout_dir, self.current_step) / f"rank_{self.data_rank}.bin"

    def wait(self, timeout: float | None = None) -> None:
        sync_wait_for_path(self._path(), interval=0.05, timeout=timeout)

    def can_receive(self) -> bool:
        return self._path().exists()

    def receive(self) -> list[MicroBatch]:
        mbs = decode_micro_batches(self._path().read_bytes())
        self.current_step += 1
        return mbs


def setup_micro_batch_receiver(output_dir: Path, data_rank: int, current_step: int, config=None) -> MicroBatchReceiver:
    kind = getattr(config, "type", "filesystem")
    if kind != "filesystem":
        raise NotImplementedError(f"rollout transport '{kind}': only the filesystem transport is mirrored (zmq stays with the reference)")
    return FileSystemMicroBatchReceiver(output_dir, data_rank, current_step)


__all__ = ["MicroBatch", "TrainingSample", "EncodedTensor", "RoutedExperts", "MicroBatchSender", "MicroBatchReceiver", "FileSystemMicroBatchSender",
           "FileSystemMicroBatchReceiver", "setup_micro_batch_receiver", "encode_micro_batches", "decode_micro_batches"]
