"""Orchestrator -> trainer wire contract — mirrors `prime_rl.transports.rollouts.types`
(src/prime_rl/transports/rollouts/types.py:4-99): `EncodedTensor`, `RoutedExperts`, `TrainingSample`, `MicroBatch` with the
same field names and ORDER.  The reference declares them `msgspec.Struct(array_like=True, omit_defaults=True)` and ships
`msgpack(list[MicroBatch])`; msgspec is not installed here, so `to_wire`/`from_wire` restate that encoding for the `msgpack`
package: an array of field values in declaration order with trailing default-valued fields dropped.  No byte-level golden file
exists in the reference (wire bytes unpinned), so the pin here is structural: round trip + field order."""

from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any


@dataclass
class EncodedTensor:
    dtype: str
    shape: list[int]
    data: bytes


@dataclass
class RoutedExperts:
    data: bytes
    shape: list[int]
    dtype: str


@dataclass
class TrainingSample:
    """One branch of a rollout as a flat token sequence (types.py:23-71)."""

    token_ids: list[int]
    mask: list[bool]
    logprobs: list[float]
    temperatures: list[float]
    env_name: str
    ref_logprobs: list[float] | None = None
    mm_kwargs: dict[str, EncodedTensor] | None = None
    routed_experts: RoutedExperts | None = None
    mm_token_type_ids: list[int] | None = None
    rl_weights: list[float] | None = None
    ce_weights: list[float] | None = None
    ref_kl_weights: list[float] | None = None
    advantages: list[float] | None = None


@dataclass
class MicroBatch:
    """A packed micro batch (types.py:74-99)."""

    input_ids: list[int]
    loss_mask: list[bool]
    advantages: list[float]
    inference_logprobs: list[float]
    position_ids: list[int]
    sequence_lengths: list[int]
    temperatures: list[float]
    env_names: list[str]
    seq_lens: list[int]
    ref_logprobs: list[float] | None = None
    routed_experts: RoutedExperts | None = None
    mm_kwargs: dict[str, EncodedTensor] | None = None
    mm_token_type_ids: list[int] | None = None
    rl_weights: list[float] | None = None
    ce_weights: list[float] | None = None
    ref_kl_weights: list[float] | None = None


_NESTED = {"routed_experts": RoutedExperts}


def to_wire(obj: Any) -> list:
    """array_like + omit_defaults: values in field order, trailing defaults trimmed."""
    vals = []
    for f in fields(obj):
        v = getattr(obj, f.name)
        if isinstance(v, (RoutedExperts, EncodedTensor)):
            v = to_wire(v)
        elif isinstance(v, dict) and f.name == "mm_kwargs":
            v = {k: to_wire(t) for k, t in v.items()}
        vals.append(v)
    fl = fields(obj)
    while vals and fl[len(vals) - 1].default is None and vals[-1] is None:
        vals.pop()
    return vals


def from_wire(cls, arr: list):
    fl = fields(cls)
    if len(arr) > len(fl):
        raise ValueError(f"{cls.__name__}: {len(arr)} wire fields > {len(fl)} declared")
    kw = {}
    for f, v in zip(fl, arr):
        if v is not None and f.name in _NESTED:
            v = from_wire(_NESTED[f.name], v)
        elif v is not None and f.name == "mm_kwargs":
            v = {k: from_wire(EncodedTensor, t) for k, t in v.items()}
        kw[f.name] = v
    return cls(**kw)
