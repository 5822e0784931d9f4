"""SwiGLU MLP seam — mirrors `prime_rl.trainer.models.layers.mlp` (src/prime_rl/trainer/models/layers/mlp.py:8-29):
`MLPConfig`, `MLP` with `gate_proj`, `up_proj`, `down_proj` (state-dict compatible).  `silu(gate) * up` is one HIP kernel
(`pm_swiglu_fwd/bwd`) on the concatenated gate|up activations; the three projections stay library GEMMs."""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from .... import ops


@dataclass
class MLPConfig:
    hidden_size: int
    intermediate_size: int
    gate_act: str = "silu"
    bias: bool = False


class _SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate_up):
        ctx.save_for_backward(gate_up)
        return ops.swiglu_fwd(gate_up)

    @staticmethod
    def backward(ctx, g):
        (gate_up,) = ctx.saved_tensors
        return ops.swiglu_bwd(g.contiguous(), gate_up)


def swiglu(gate_up: torch.Tensor) -> torch.Tensor:
    """[N, 2I] (gate | up) -> [N, I] = bf16(silu(gate)) * up."""
    return _SwiGLUFn.apply(gate_up)


class MLP(nn.Module):
    def __init__(self, config: MLPConfig):
        super().__init__()
        if config.gate_act != "silu" or config.bias:
            raise NotImplementedError("the HIP path implements the bias-free SwiGLU MLP of Qwen3/Llama (mlp.py:27-29)")
        self.config = config
        self.hidden_size, self.intermediate_size = config.hidden_size, config.intermediate_size
        self.gate_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=False)

    def forward(self, x, routed_experts: torch.Tensor | None = None):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        gate_up = torch.cat([self.gate_proj(x2), self.up_proj(x2)], dim=-1)
        return self.down_proj(swiglu(gate_up)).view(*shape[:-1], self.hidden_size)
