"""Rotary embedding seam — mirrors `prime_rl.trainer.models.layers.rotary_emb` (rotary_emb.py:41-73): `RotaryEmbedding`
keeps the `inv_freq` buffer and returns `(cos, sin)` in the activation dtype.  The table is one HIP kernel
(`pm_rope_table`: angle = inv_freq * position in f32, cos/sin rounded to bf16); it returns the HALF table `[N, D/2]` —
the reference's `cat(freqs, freqs)` duplicate is never materialised, the fused q/k-norm+RoPE kernel indexes it modulo D/2."""

from __future__ import annotations

import torch
from torch import nn

from .... import ops
from ....utils.rope import scaled_inv_freq


class RotaryEmbedding(nn.Module):
    inv_freq: torch.Tensor

    def __init__(self, head_dim: int, rope_theta: float = 1_000_000.0, device=None, rope_scaling: dict | None = None):
        super().__init__()
        inv_freq = scaled_inv_freq(head_dim, rope_theta, rope_scaling)  # rotary_emb.py:24 + ROPE_INIT_FUNCTIONS["linear" | "llama3"]
        self.register_buffer("inv_freq", inv_freq.to(device) if device is not None else inv_freq, persistent=False)
        self.attention_scaling = 1.0

    @torch.no_grad()
    def forward(self, x: torch.Tensor, position_ids: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        return ops.rope_table(position_ids.reshape(-1).contiguous(), self.inv_freq.to(torch.float32))
