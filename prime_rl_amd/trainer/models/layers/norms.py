"""RMSNorm seam — mirrors `prime_rl.trainer.models.layers.norms` (src/prime_rl/trainer/models/layers/norms.py:23-47).

Same module surface (`RMSNormConfig`, `RMSNorm(config)` with a `weight` parameter and `variance_epsilon`); the forward and
backward are the HIP kernels `pm_rmsnorm_fwd/bwd` (one wave per row, f32 accumulation, the reference's bf16 rounding
points).  The reference dispatches to an optional fast function `(x, weight, eps=...)` (norms.py:10-20,36-39);
`rms_norm(x, weight, eps)` below has that signature."""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from .... import ops


@dataclass
class RMSNormConfig:
    hidden_size: int
    eps: float = 1e-6


class _RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        y, rstd, _ = ops.rmsnorm_fwd(x2, weight.contiguous(), eps)
        ctx.save_for_backward(x2, weight, rstd)
        ctx.shape = shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        x2, weight, rstd = ctx.saved_tensors
        gw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
        gx = ops.rmsnorm_bwd(gy.reshape(x2.shape).contiguous(), x2, weight.contiguous(), rstd, gw)
        return gx.view(ctx.shape), gw.to(weight.dtype), None


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """bf16 `x [..., H]`, bf16 `weight [H]` -> `weight * (x * rsqrt(mean(x^2) + eps)).to(bf16)`; differentiable."""
    return _RMSNormFn.apply(x, weight, eps)


class RMSNorm(nn.Module):
    def __init__(self, config: RMSNormConfig) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(config.hidden_size))
        self.variance_epsilon = config.eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return rms_norm(hidden_states, self.weight, self.variance_epsilon)

    def extra_repr(self):
        return f"{tuple(self.weight.shape)}, eps={self.variance_epsilon}"
