"""Packed-sequence boundary helpers — mirror `prime_rl.utils.sequence` (src/prime_rl/utils/sequence.py:6-52): same names, same
return `(cu_seqlens int32 [S+1], max_seqlen int)`, same `ValueError`s; pinned by the reference's known answers
(tests/unit/utils/test_sequence.py:10-41) in tests/test_sequence_cpu.py.

MI355X note: the reference validates on the device and pays three host syncs per forward (`.any().item()`, `.sum().item()`,
`.max().item()`).  Document lengths arrive from the rollout wire as host integers, so here validation and the
prefix sum run on the host copy and only the finished int32 table is placed on `device`: zero device syncs when the input is a
list or a CPU tensor, exactly one D2H copy when the caller hands over a device tensor.  Per-token document lookup inside the kernels
is `pm_seq_start_from_cu` (binary search over this table)."""

from __future__ import annotations

from typing import Sequence

import torch


def _host_list(x) -> tuple[list[int], torch.device]:
    if isinstance(x, torch.Tensor):
        return x.reshape(-1).tolist() if x.ndim != 1 else x.tolist(), x.device
    return [int(v) for v in x], torch.device("cpu")


def get_cu_seqlens_from_seq_lens(seq_lens: torch.Tensor | Sequence[int], total_tokens: int | None = None, device=None) -> tuple[torch.Tensor, int]:
    if isinstance(seq_lens, torch.Tensor) and seq_lens.ndim != 1:
        raise ValueError(f"seq_lens must be 1D, got shape={tuple(seq_lens.shape)}")
    lens, src_device = _host_list(seq_lens)
    if len(lens) == 0:
        raise ValueError("seq_lens must not be empty")
    if any(n <= 0 for n in lens):
        raise ValueError(f"seq_lens must be positive, got {lens}")
    if total_tokens is not None and sum(lens) != total_tokens:
        raise ValueError(f"seq_lens sum must equal sequence length: {lens} vs {total_tokens}")
    cu = [0] * (len(lens) + 1)
    for i, n in enumerate(lens):
        cu[i + 1] = cu[i] + n
    return torch.tensor(cu, dtype=torch.int32, device=device if device is not None else src_device), max(lens)


def get_cu_seqlens_from_position_ids(position_ids: torch.Tensor, device=None) -> tuple[torch.Tensor, int]:
    """Local-relative boundaries from packed position ids: a new document starts at index 0 and wherever the id is 0 (a shard may
    begin mid-document with a non-zero id, sequence.py:7-35)."""
    flat, src_device = _host_list(position_ids)
    if len(flat) == 0:
        raise AssertionError("Cannot build cu_seqlens for an empty position_ids tensor")
    starts = [0] + [i for i, p in enumerate(flat) if p == 0 and i > 0]
    bounds = starts + [len(flat)]
    lens = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
    return torch.tensor(bounds, dtype=torch.int32, device=device if device is not None else src_device), max(lens)
