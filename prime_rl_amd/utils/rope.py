"""RoPE inverse frequencies with the scaling rules of the dense families on this path.

The reference's `RotaryEmbedding` (src/prime_rl/trainer/models/layers/rotary_emb.py:41-57) takes `inv_freq` from transformers'
`ROPE_INIT_FUNCTIONS[rope_type]`; the table kernel (`pm_rope_table`) only needs the resulting vector, so the scaling is host arithmetic done once
per model.  Restated here (pinned against transformers in tests/test_config_cpu.py):

  "default"  inv_freq[i] = theta^(-2i/d)                                                          (rotary_emb.py:10-26)
  "linear"   inv_freq / factor                                                                    (position interpolation)
  "llama3"   Llama-3.1 / 3.2: wavelengths longer than original_max / low_freq_factor are divided by `factor`, shorter than
             original_max / high_freq_factor kept, the band in between interpolated smoothly.
All three keep `attention_scaling = 1`; the types that rescale cos / sin (yarn, longrope) or depend on the sequence length (dynamic) are not built."""

from __future__ import annotations

import math

import torch

SUPPORTED = ("default", "linear", "llama3")


def rope_type_of(rope_scaling: dict | None) -> str:
    if not rope_scaling:
        return "default"
    return rope_scaling.get("rope_type", rope_scaling.get("type", "default")) or "default"


def scaled_inv_freq(head_dim: int, theta: float, rope_scaling: dict | None = None, device=None) -> torch.Tensor:
    kind = rope_type_of(rope_scaling)
    if kind not in SUPPORTED:
        raise NotImplementedError(f"rope_scaling type {kind!r}: built are {SUPPORTED} (cos / sin rescaling and length-dependent types stay with the reference)")
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))
    if kind == "linear":
        inv_freq = inv_freq / float(rope_scaling["factor"])
    elif kind == "llama3":
        factor, low, high = float(rope_scaling["factor"]), float(rope_scaling["low_freq_factor"]), float(rope_scaling["high_freq_factor"])
        old_len = float(rope_scaling["original_max_position_embeddings"])
        low_wavelen, high_wavelen = old_len / low, old_len / high
        wavelen = 2 * math.pi / inv_freq
        scaled = torch.where(wavelen > low_wavelen, inv_freq / factor, inv_freq)
        smooth = (old_len / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * scaled / factor + smooth * scaled
        medium = ~(wavelen < high_wavelen) & ~(wavelen > low_wavelen)
        inv_freq = torch.where(medium, smoothed, scaled)
    return inv_freq.to(device) if device is not None else inv_freq
