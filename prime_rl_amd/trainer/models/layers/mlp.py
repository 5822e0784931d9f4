"""SwiGLU MLP seam — mirrors `prime_rl.trainer.models.layers.mlp` (src/prime_rl/trainer/models/layers/mlp.py:8-29):
`MLPConfig`, `MLP` with `gate_proj`, `up_proj`, `down_proj` (state-dict compatible).  the projections run on the hand-written MFMA core (layers/linear.py: `ops.gemm`), `silu(gate) * up`
in the gate|up product's epilogue and its backward in the down row gradient's epilogue where gate_proj / up_proj are adjacent in memory, else as
`pm_swiglu_fwd/bwd` on the concatenated gate|up activations."""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from .... import ops
from .linear import fused_linear, linear, swiglu_mlp


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
        # one node on the hand-written core (gate|up product + SwiGLU epilogue, down product; SwiGLU backward in the down row gradient's epilogue) when
        # gate_proj / up_proj lie back to back (a trainer engine's flat buffer); else the three projections on the core and the SwiGLU kernel
        y = swiglu_mlp(x2.contiguous(), self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight) if x2.is_cuda else None
        if y is None:
            gate_up = fused_linear(x2, [self.gate_proj.weight, self.up_proj.weight])
            y = linear(swiglu(gate_up), self.down_proj.weight)
        return y.view(*shape[:-1], self.hidden_size)
