"""Binary step wire + pinned prefetch for micro-batch ingest.

What it replaces: the reference ships `msgpack(list[MicroBatch])` whose token streams are Python lists
(src/prime_rl/transports/rollouts/types.py:74-99, base.py:15,34) and the trainer rebuilds tensors with seven
`torch.tensor(python_list)` calls per micro-batch (src/prime_rl/trainer/rl/data.py:186-238) — CPU work that grows with the token
count and sits on the step's critical path.  Here one rank's micro-batches of one step travel as ONE blob of raw little-endian
arrays (the `EncodedTensor` idea of types.py:4-10 applied to the token streams), laid out structure-of-arrays over the whole step:

    magic "PMRB" | u32 version | u32 header_len | msgpack header | pad to 64 | section 0 | pad | section 1 | ...

    sections (each 64-byte aligned, concatenated over the step's micro-batches in order):
      input_ids i32 [n] | position_ids i32 [n] | loss_mask u8 [n] | advantages f32 [n] | inference_logprobs f32 [n] |
      temperatures f32 [n] | optional f32 [n]: ref_logprobs, rl_weights, ce_weights, ref_kl_weights |
      optional routed_experts [n, layers, top_k] in the sender's integer dtype (router replay; all micro-batches of the step or none)
    header: tokens per micro-batch, sample lengths (`sequence_lengths`) and document lengths (`seq_lens`) with per-micro-batch
      counts, the env-name table + one env id per SAMPLE (the per-token `env_names` list of the reference is constant within a
      sample), which optional sections are present.

The receiver reads the blob straight into page-locked host memory (a background thread does this for step k+1 while step k
trains), issues ONE host-to-device copy of the payload, and derives everything else on the device (int64 widening, shift-left
labels per micro-batch, 1/temperature) — no per-token Python anywhere.  f32 rounding of the f64 wire floats happens at the sender
(`np.float32`) instead of at the receiver (`torch.tensor(list, dtype=float)`): same IEEE rounding, same bits; pinned by
tests/test_binary_wire_cpu.py against the msgpack path.  A stream missing from some micro-batches of a step is filled the way
the packer fills it (rl_weights 1.0, others 0.0; types.py:96-99)."""

from __future__ import annotations

import struct
import threading
from dataclasses import dataclass
from pathlib import Path

import msgpack
import numpy as np
import torch

from ...utils.pathing import get_rollout_dir, get_step_path, sync_wait_for_path
from .types import MicroBatch

MAGIC = b"PMRB"
VERSION = 1
ALIGN = 64
_REQUIRED = (("input_ids", np.int32), ("position_ids", np.int32), ("loss_mask", np.uint8), ("advantages", np.float32),
             ("inference_logprobs", np.float32), ("temperatures", np.float32))
_OPTIONAL = ("ref_logprobs", "rl_weights", "ce_weights", "ref_kl_weights")
_FILL = {"rl_weights": 1.0}


def _pad(n: int) -> int:
    return (-n) % ALIGN


def encode_step(mbs: list[MicroBatch]) -> bytes:
    """One rank's micro-batches of one step -> blob."""
    if not mbs:
        raise ValueError("empty step")
    for mb in mbs:
        if mb.mm_kwargs or mb.mm_token_type_ids is not None:
            raise NotImplementedError("multimodal payloads are outside the GRPO hot path")
    counts = [len(mb.input_ids) for mb in mbs]
    routed = None
    if any(mb.routed_experts is not None for mb in mbs):  # router replay: raw [tokens, layers, top_k] blocks, concatenated like every other stream
        if not all(mb.routed_experts is not None for mb in mbs):
            raise ValueError("router replay: every micro-batch of a step carries routed_experts, or none does (one packed pass routes all of its tokens)")
        r0 = mbs[0].routed_experts
        for mb, n in zip(mbs, counts):
            r = mb.routed_experts
            if r.dtype != r0.dtype or list(r.shape[1:]) != list(r0.shape[1:]) or r.shape[0] != n:
                raise ValueError(f"routed_experts: inconsistent payloads in one step ({r.shape} {r.dtype} vs {r0.shape} {r0.dtype}, {n} tokens)")
        routed = dict(layers=int(r0.shape[1]), top_k=int(r0.shape[2]), dtype=str(np.dtype(r0.dtype)))
    env_table: dict[str, int] = {}
    sample_env: list[int] = []
    for mb in mbs:
        if not (len(mb.loss_mask) == len(mb.advantages) == len(mb.inference_logprobs) == len(mb.position_ids) == len(mb.temperatures) == len(mb.input_ids)):
            raise ValueError("micro-batch token streams must have equal length")
        if sum(mb.sequence_lengths) != len(mb.input_ids) or sum(mb.seq_lens) != len(mb.input_ids):
            raise ValueError("sequence_lengths / seq_lens must sum to the micro-batch length")
        off = 0
        for sl in mb.sequence_lengths:
            sample_env.append(env_table.setdefault(mb.env_names[off] if mb.env_names else "", len(env_table)))
            off += sl
    present = [name for name in _OPTIONAL if any(getattr(mb, name) is not None for mb in mbs)]
    header = dict(n_tokens=sum(counts), mb_tokens=counts, sample_lens=[s for mb in mbs for s in mb.sequence_lengths],
                  mb_samples=[len(mb.sequence_lengths) for mb in mbs], seq_lens=[s for mb in mbs for s in mb.seq_lens],
                  mb_seqs=[len(mb.seq_lens) for mb in mbs], env_names=list(env_table), sample_env=sample_env, optional=present,
                  # which micro-batches carried NO rl_weights stream (the filled 1.0s below lose that): `PackedBatch.rl_every_sample` needs it
                  mb_no_rl=[mb.rl_weights is None for mb in mbs])
    if routed is not None:
        header["routed"] = routed
    hbytes = msgpack.packb(header, use_bin_type=True)
    out = bytearray(MAGIC + struct.pack("<II", VERSION, len(hbytes)) + hbytes)
    out += b"\0" * _pad(len(out))

    def section(arr: np.ndarray) -> None:
        out.extend(arr.tobytes())
        out.extend(b"\0" * _pad(len(out)))

    for name, dt in _REQUIRED:
        section(np.concatenate([np.asarray(getattr(mb, name), dtype=dt) for mb in mbs]))
    for name in present:
        parts = []
        for mb, n in zip(mbs, counts):
            v = getattr(mb, name)
            parts.append(np.full(n, _FILL.get(name, 0.0), dtype=np.float32) if v is None else np.asarray(v, dtype=np.float32))
        section(np.concatenate(parts))
    if routed is not None:
        section(np.frombuffer(b"".join(bytes(mb.routed_experts.data) for mb in mbs), dtype=np.uint8))
    return bytes(out)


@dataclass
class StepBlob:
    """A received step: parsed header + the payload in (pinned, when a GPU is present) host memory."""

    header: dict
    payload: torch.Tensor  # u8, host; starts at the first section
    offsets: dict[str, tuple[int, int, np.dtype]]  # name -> (byte offset in payload, element count, dtype)

    def host_array(self, name: str) -> np.ndarray:
        off, n, dt = self.offsets[name]
        return self.payload.numpy()[off:off + n * np.dtype(dt).itemsize].view(dt)

    def env_names_per_token(self) -> list[str]:
        """The reference's per-token `env_names` list (data.py:48), rebuilt on demand."""
        table = self.header["env_names"]
        out: list[str] = []
        for eid, sl in zip(self.header["sample_env"], self.header["sample_lens"]):
            out += [table[eid]] * sl
        return out


def _alloc_host(nbytes: int) -> torch.Tensor:
    return torch.empty(nbytes, dtype=torch.uint8, pin_memory=torch.cuda.is_available())


def read_step(path: Path) -> StepBlob:
    """File -> StepBlob, the payload read straight into page-locked memory (no intermediate bytes object)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12 or head[:4] != MAGIC:
            raise ValueError(f"{path}: not a PMRB step blob")
        version, hlen = struct.unpack("<II", head[4:])
        if version != VERSION:
            raise ValueError(f"{path}: PMRB version {version}, expected {VERSION}")
        header = msgpack.unpackb(f.read(hlen), raw=False)
        f.seek(_pad(12 + hlen), 1)
        start = f.tell()
        f.seek(0, 2)
        nbytes = f.tell() - start
        f.seek(start)
        payload = _alloc_host(nbytes)
        if f.readinto(payload.numpy()) != nbytes:
            raise ValueError(f"{path}: truncated payload")
    return _index(header, payload)


def decode_step(buf: bytes) -> StepBlob:
    if buf[:4] != MAGIC:
        raise ValueError("not a PMRB step blob")
    version, hlen = struct.unpack("<II", buf[4:12])
    if version != VERSION:
        raise ValueError(f"PMRB version {version}, expected {VERSION}")
    header = msgpack.unpackb(buf[12:12 + hlen], raw=False)
    start = 12 + hlen + _pad(12 + hlen)
    payload = _alloc_host(len(buf) - start)
    payload.numpy()[:] = np.frombuffer(buf, dtype=np.uint8, offset=start)
    return _index(header, payload)


def _index(header: dict, payload: torch.Tensor) -> StepBlob:
    n = header["n_tokens"]
    if sum(header["mb_tokens"]) != n or sum(header["sample_lens"]) != n or sum(header["seq_lens"]) != n:
        raise ValueError("PMRB header: token counts do not add up")
    offsets, cur = {}, 0
    for name, dt in list(_REQUIRED) + [(o, np.float32) for o in header["optional"]]:
        offsets[name] = (cur, n, np.dtype(dt))
        cur += n * np.dtype(dt).itemsize
        cur += _pad(cur)
    if header.get("routed"):
        r = header["routed"]
        offsets["routed_experts"] = (cur, n * r["layers"] * r["top_k"], np.dtype(r["dtype"]))
        cur += n * r["layers"] * r["top_k"] * np.dtype(r["dtype"]).itemsize
    if cur > payload.numel():
        raise ValueError("PMRB payload shorter than its header says")
    return StepBlob(header, payload, offsets)


def blob_to_tensor_micro_batches(blob: StepBlob) -> list[dict]:
    """Host-side view in the reference's `TensorMicroBatch` layout (data.py:18-52) — for parity tests and non-packed consumers."""
    h = blob.header
    names = blob.env_names_per_token()
    out, t0, s0, d0 = [], 0, 0, 0
    r = h.get("routed")
    routed_all = blob.host_array("routed_experts").reshape(h["n_tokens"], r["layers"], r["top_k"]) if r else None
    for n, ns, nd in zip(h["mb_tokens"], h["mb_samples"], h["mb_seqs"]):
        def f(name, dtype):
            return torch.from_numpy(blob.host_array(name)[t0:t0 + n].copy()).to(dtype).unsqueeze(0)

        def opt(name):
            return f(name, torch.float) if name in blob.offsets else None

        out.append(dict(input_ids=f("input_ids", torch.long), position_ids=f("position_ids", torch.long), advantages=f("advantages", torch.float),
                        inference_logprobs=f("inference_logprobs", torch.float), ref_logprobs=opt("ref_logprobs"),
                        loss_mask=f("loss_mask", torch.bool), temperatures=f("temperatures", torch.float), env_names=names[t0:t0 + n],
                        sequence_lengths=list(h["sample_lens"][s0:s0 + ns]), lora_num_tokens=torch.tensor([n], dtype=torch.int32),
                        seq_lens=torch.tensor(h["seq_lens"][d0:d0 + nd], dtype=torch.long), mm_kwargs=None, mm_token_type_ids=None,
                        routed_experts=None if routed_all is None else torch.from_numpy(routed_all[t0:t0 + n].astype(np.int32)).unsqueeze(0),
                        rl_weights=opt("rl_weights"), ce_weights=opt("ce_weights"), ref_kl_weights=opt("ref_kl_weights")))
        t0, s0, d0 = t0 + n, s0 + ns, d0 + nd
    return out


def blob_to_packed_batch(blob: StepBlob, device):
    """StepBlob -> `PackedBatch` on `device`: one H2D copy of the payload, everything else derived on the device.  Field-for-field
    identical to `pack_micro_batches` over the msgpack path (tests/test_binary_wire_gpu.py)."""
    from ...trainer.engine import PackedBatch

    h = blob.header
    n = h["n_tokens"]
    mask_h = blob.host_array("loss_mask").astype(bool)

    def nz(name):
        return blob.host_array(name) != 0

    # per-component member-token counts on the host arrays, as train.py:307-317 takes them
    n_rl = int((mask_h & nz("rl_weights")).sum()) if "rl_weights" in blob.offsets else int(mask_h.sum())
    n_ce = int(nz("ce_weights").sum()) if "ce_weights" in blob.offsets else 0
    n_rk = int(nz("ref_kl_weights").sum()) if "ref_kl_weights" in blob.offsets else 0
    dev_payload = blob.payload.to(device, non_blocking=True)

    def sec(name, dtype):
        off, cnt, dt = blob.offsets[name]
        return dev_payload[off:off + cnt * dt.itemsize].view(dtype)

    def opt(name):
        return sec(name, torch.float32) if name in blob.offsets else None

    ids = sec("input_ids", torch.int32).to(torch.int64)
    mb_ends = torch.tensor(np.cumsum(h["mb_tokens"]) - 1, dtype=torch.int64).to(device, non_blocking=True)
    labels = torch.roll(ids, -1)
    labels[mb_ends] = 0  # shift_tensor_left per micro-batch: the last position predicts nothing (loss.py:66-73)
    cu_seq = np.zeros(len(h["seq_lens"]) + 1, dtype=np.int32)
    np.cumsum(h["seq_lens"], out=cu_seq[1:])
    cu_smp = np.zeros(len(h["sample_lens"]) + 1, dtype=np.int32)
    np.cumsum(h["sample_lens"], out=cu_smp[1:])
    starts = np.zeros(len(h["sample_lens"]), dtype=np.uint8)
    starts[np.cumsum([0] + h["mb_samples"][:-1])] = 1

    def small(a):
        return torch.from_numpy(a).to(device, non_blocking=True)

    routed = None
    if h.get("routed"):
        r = h["routed"]
        off, cnt, dt = blob.offsets["routed_experts"]
        tdt = {"uint8": torch.uint8, "int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64}[dt.name]
        routed = dev_payload[off:off + cnt * dt.itemsize].view(tdt).view(n, r["layers"], r["top_k"]).to(torch.int32)

    # rows of samples whose micro-batch had no rl_weights stream count for the rl metrics whatever their mask (pack_micro_batches sets the same flag)
    no_rl = h.get("mb_no_rl")
    every = None
    if "rl_weights" in blob.offsets and no_rl is not None and any(no_rl):
        every = small(np.repeat(np.asarray(no_rl, dtype=bool), h["mb_samples"]))
    return PackedBatch(input_ids=ids, position_ids=sec("position_ids", torch.int32).to(torch.int64), labels=labels,
                       inv_temp=1.0 / sec("temperatures", torch.float32), advantages=sec("advantages", torch.float32),
                       inference_logprobs=sec("inference_logprobs", torch.float32), loss_mask=sec("loss_mask", torch.uint8).view(torch.bool),
                       cu_seqlens=small(cu_seq), cu_samples=small(cu_smp), sample_starts_mb=small(starts), n_micro_batches=len(h["mb_tokens"]),
                       ref_logprobs=opt("ref_logprobs"), rl_weights=opt("rl_weights"), ce_weights=opt("ce_weights"),
                       ref_kl_weights=opt("ref_kl_weights"), local_counts=(n_rl, n_ce, n_rk), routed_experts=routed, rl_every_sample=every)


def step_file(output_dir: Path, step: int, rank: int) -> Path:
    return get_step_path(get_rollout_dir(Path(output_dir)), step) / f"rank_{rank}.pmrb"


def write_step(output_dir: Path, step: int, grid: list[list[MicroBatch]]) -> None:
    """Sender half: `rollouts/step_{k}/rank_{r}.pmrb`, tmp + rename like the reference's filesystem sender (filesystem.py:17-36)."""
    for rank, mbs in enumerate(grid):
        path = step_file(output_dir, step, rank)
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(".pmrb.tmp")
        tmp.write_bytes(encode_step(mbs))
        tmp.rename(path)


class BinaryStepReceiver:
    """`MicroBatchReceiver`-shaped (wait / can_receive / receive) receiver of step blobs with one step of read-ahead: while the
    trainer works on step k, a daemon thread waits for `step_{k+1}/rank_r.pmrb` and reads it into pinned memory."""

    def __init__(self, output_dir: Path, data_rank: int, current_step: int = 0, prefetch: bool = True, timeout: float | None = None):
        self.output_dir, self.data_rank, self.current_step, self.timeout = Path(output_dir), data_rank, current_step, timeout
        self.prefetch = prefetch
        self._ahead: tuple[int, threading.Thread, list] | None = None

    def _path(self, step: int) -> Path:
        return step_file(self.output_dir, step, self.data_rank)

    def can_receive(self) -> bool:
        return self._path(self.current_step).exists()

    def wait(self) -> None:
        if self._ahead is None or self._ahead[0] != self.current_step:
            sync_wait_for_path(self._path(self.current_step), interval=0.05, timeout=self.timeout)

    def _start_read_ahead(self, step: int) -> None:
        box: list = []

        def work():
            try:
                sync_wait_for_path(self._path(step), interval=0.05, timeout=self.timeout)
                box.append(read_step(self._path(step)))
            except Exception as e:  # surfaced by the consumer
                box.append(e)

        th = threading.Thread(target=work, daemon=True, name=f"pmrb-prefetch-{step}")
        th.start()
        self._ahead = (step, th, box)

    def receive(self) -> StepBlob:
        step = self.current_step
        if self._ahead is not None and self._ahead[0] == step:
            _, th, box = self._ahead
            th.join()
            self._ahead = None
            if isinstance(box[0], Exception):
                raise box[0]
            blob = box[0]
        else:
            blob = read_step(self._path(step))
        self.current_step += 1
        if self.prefetch:
            self._start_read_ahead(self.current_step)
        return blob
