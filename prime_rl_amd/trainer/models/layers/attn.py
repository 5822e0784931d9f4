"""Attention seam — mirrors `prime_rl.trainer.models.layers.attn` (src/prime_rl/trainer/models/layers/attn.py:29-189).

`HipVarlenAttention(AttentionConfig)` has the reference's parameter names (`q_proj, k_proj, v_proj, o_proj, q_norm, k_norm`) and
forward contract `forward(hidden_states [1,N,H], position_embeddings=(cos,sin), cu_seqlens i32 [S+1], max_seqlen) ->
(out [1,N,H], None)`; it registers itself as `ATTN_IMPL2CLASS["hip_varlen"]` so `config._attn_implementation = "hip_varlen"`
selects it (attn.py:185-189).  Inside: fused per-head q/k RMSNorm + RoPE (no transposes, attn.py:147-158 does two) and the
MFMA varlen causal attention kernels; `max_seqlen` is accepted and ignored (the kernels need no host-side schedule).
`use_qk_norm=False` (the Llama block, models/llama/modeling_llama.py:52-62) runs the same kernel as RoPE only; `attention_bias=True`
(Qwen2) puts a bias on q/k/v_proj (attn.py:64-72) — o_proj keeps `output_bias`, False everywhere on this path."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch
from torch import nn

from .... import ops
from .linear import fused_linear, linear
from .norms import RMSNorm, RMSNormConfig


@dataclass
class AttentionConfig:
    hidden_size: int
    head_dim: int
    num_attention_heads: int
    num_key_value_heads: int
    is_causal: bool = True
    attention_bias: bool = False
    use_qk_norm: bool = True
    rms_norm_eps: float = 1e-6
    qk_norm_type: Literal["per_head", "per_layer"] = "per_head"
    output_bias: bool = False


class _RopeFn(torch.autograd.Function):
    """RoPE on the q and k column blocks of the fused projection output, no per-head norm."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, hq, hkv, d):
        q, k, _ = ops.qknorm_rope_fwd(qkv, None, None, cos, sin, hq, hkv, d, 0.0)
        ctx.save_for_backward(cos, sin)
        ctx.dims = (hq, hkv, d, qkv.shape)
        return q, k

    @staticmethod
    def backward(ctx, gq, gk):
        cos, sin = ctx.saved_tensors
        hq, hkv, d, shape = ctx.dims
        gqkv = torch.zeros(shape, dtype=torch.bfloat16, device=gq.device)  # v columns carry no gradient through this op
        ops.qknorm_rope_bwd(gq.contiguous(), gk.contiguous(), None, None, None, cos, sin, None, gqkv, None, None, hq, hkv, d)
        return gqkv, None, None, None, None, None


class _QKNormRopeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qw, kw, cos, sin, hq, hkv, d, eps):
        q, k, rstd = ops.qknorm_rope_fwd(qkv, qw.contiguous(), kw.contiguous(), cos, sin, hq, hkv, d, eps)
        ctx.save_for_backward(qkv, qw, kw, cos, sin, rstd)
        ctx.dims = (hq, hkv, d)
        return q, k

    @staticmethod
    def backward(ctx, gq, gk):
        qkv, qw, kw, cos, sin, rstd = ctx.saved_tensors
        hq, hkv, d = ctx.dims
        gqkv = torch.zeros_like(qkv)  # v columns carry no gradient through this op
        gqw = torch.zeros(d, dtype=torch.float32, device=qkv.device)
        gkw = torch.zeros(d, dtype=torch.float32, device=qkv.device)
        ops.qknorm_rope_bwd(gq.contiguous(), gk.contiguous(), qkv, qw.contiguous(), kw.contiguous(), cos, sin, rstd, gqkv, gqw, gkw, hq, hkv, d)
        return gqkv, gqw.to(qw.dtype), gkw.to(kw.dtype), None, None, None, None, None, None


class _VarlenAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, seq_start, scale):
        out, lse = ops.attn_varlen_fwd(q, k, v, seq_start, scale)
        ctx.save_for_backward(q, k, v, out, lse, seq_start)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, go):
        q, k, v, out, lse, seq_start = ctx.saved_tensors
        gq, gk, gv = ops.attn_varlen_bwd(go.contiguous(), q, k, v, out, lse, seq_start, ctx.scale)
        return gq, gk, gv, None, None


class _QKVAttnFn(torch.autograd.Function):
    """per-head q/k RMSNorm + RoPE + varlen causal attention on the fused projection output `qkv [N, (hq + 2 hkv) d]` as ONE autograd node: its backward
    writes d(qkv) once — the q | k columns by `pm_qknorm_rope_bwd`, the v columns by the attention backward itself (`grad_v_out`) — instead of autograd
    assembling it from a zero-filled buffer, a slice copy and an add (three passes over [N, qkv] per layer).  `qw` / `kw` None = RoPE only (Llama, Qwen2)."""

    @staticmethod
    def forward(ctx, qkv, qw, kw, cos, sin, seq_start, hq, hkv, d, eps, scale):
        n = qkv.shape[0]
        q, k, rstd = ops.qknorm_rope_fwd(qkv, qw.contiguous() if qw is not None else None, kw.contiguous() if kw is not None else None, cos, sin, hq, hkv, d, eps)
        v = qkv[:, (hq + hkv) * d:].view(n, hkv, d)
        out, lse = ops.attn_varlen_fwd(q, k, v, seq_start, scale)
        ctx.save_for_backward(qkv, qw, kw, cos, sin, rstd, q, k, out, lse, seq_start)
        ctx.dims = (hq, hkv, d, scale)
        return out

    @staticmethod
    def backward(ctx, go):
        qkv, qw, kw, cos, sin, rstd, q, k, out, lse, seq_start = ctx.saved_tensors
        hq, hkv, d, scale = ctx.dims
        n = qkv.shape[0]
        gqkv = torch.empty_like(qkv)
        v = qkv[:, (hq + hkv) * d:].view(n, hkv, d)
        gv_view = gqkv[:, (hq + hkv) * d:].view(n, hkv, d)
        gq, gk, _ = ops.attn_varlen_bwd(go.contiguous(), q, k, v, out, lse, seq_start, scale, grad_v_out=gv_view)
        gqw = gkw = None
        if qw is not None:
            gqw = torch.zeros(d, dtype=torch.float32, device=qkv.device)
            gkw = torch.zeros(d, dtype=torch.float32, device=qkv.device)
            ops.qknorm_rope_bwd(gq, gk, qkv, qw.contiguous(), kw.contiguous(), cos, sin, rstd, gqkv, gqw, gkw, hq, hkv, d)
            gqw, gkw = gqw.to(qw.dtype), gkw.to(kw.dtype)
        else:
            ops.qknorm_rope_bwd(gq, gk, None, None, None, cos, sin, None, gqkv, None, None, hq, hkv, d)
        return gqkv, gqw, gkw, None, None, None, None, None, None, None, None


def varlen_causal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens: torch.Tensor, scale: float,
                            sliding_window: int | None = None) -> torch.Tensor:
    """flash_attn_varlen_func(q, k, v, cu, cu, max, max, causal=True[, window_size=(sliding_window - 1, 0)]) equivalent on packed
    [N, heads, 128] bf16 tensors (attn.py:94-110).  The window only raises each row's first visible key: same kernels."""
    seq_start = ops.seq_start_from_cu(cu_seqlens.to(torch.int32).contiguous(), q.shape[0])
    if sliding_window is not None:
        seq_start = ops.seq_start_with_window(seq_start, sliding_window)
    return _VarlenAttnFn.apply(q.contiguous(), k.contiguous(), v, seq_start, scale)


class HipVarlenAttention(nn.Module):
    def __init__(self, config: AttentionConfig):
        super().__init__()
        if config.head_dim != 128 or not config.is_causal or (config.use_qk_norm and config.qk_norm_type != "per_head"):
            raise NotImplementedError("hip_varlen implements head_dim 128 causal attention with per-head q/k RMSNorm (Qwen3) or none (Llama, Qwen2)")
        if config.output_bias:
            raise NotImplementedError("o_proj bias is not on this path (output_bias is False for Qwen3, Qwen2 and Llama)")
        self.head_dim = config.head_dim
        self.hq, self.hkv = config.num_attention_heads, config.num_key_value_heads
        self.num_key_value_groups = self.hq // self.hkv
        self.scaling = self.head_dim**-0.5
        self.eps = config.rms_norm_eps
        self.q_proj = nn.Linear(config.hidden_size, self.hq * self.head_dim, bias=config.attention_bias)
        self.k_proj = nn.Linear(config.hidden_size, self.hkv * self.head_dim, bias=config.attention_bias)
        self.v_proj = nn.Linear(config.hidden_size, self.hkv * self.head_dim, bias=config.attention_bias)
        self.o_proj = nn.Linear(self.hq * self.head_dim, config.hidden_size, bias=False)
        self.use_qk_norm = config.use_qk_norm
        if self.use_qk_norm:
            self.q_norm = RMSNorm(RMSNormConfig(hidden_size=self.head_dim, eps=config.rms_norm_eps))
            self.k_norm = RMSNorm(RMSNormConfig(hidden_size=self.head_dim, eps=config.rms_norm_eps))

    def forward(self, hidden_states, position_embeddings=None, cu_seqlens=None, max_seqlen=None):
        b, n, _ = hidden_states.shape
        assert b == 1, "packed batches are [1, N, H] (trainer/batch.py)"
        x = hidden_states[0]
        # one q|k|v product on the hand-written core when the three weights lie back to back (a trainer engine's flat buffer); else three + a concatenation
        qkv = fused_linear(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias])
        cos, sin = position_embeddings
        half = self.head_dim // 2
        # accept the reference's cat(f,f) tables too; the kernels read bf16 (the reference casts cos/sin to x.dtype, rotary_emb.py:72-73)
        cos, sin = cos.reshape(n, -1)[:, :half].to(torch.bfloat16).contiguous(), sin.reshape(n, -1)[:, :half].to(torch.bfloat16).contiguous()
        seq_start = ops.seq_start_from_cu(cu_seqlens.to(torch.int32).contiguous(), n)
        sliding_window = getattr(self, "sliding_window", None)  # attn.py:97-99
        if sliding_window is not None:
            seq_start = ops.seq_start_with_window(seq_start, sliding_window)
        qw, kw = (self.q_norm.weight, self.k_norm.weight) if self.use_qk_norm else (None, None)
        out = _QKVAttnFn.apply(qkv.contiguous(), qw, kw, cos, sin, seq_start, self.hq, self.hkv, self.head_dim, self.eps if self.use_qk_norm else 0.0, self.scaling)
        return linear(out.view(n, self.hq * self.head_dim), self.o_proj.weight).unsqueeze(0), None


ATTN_IMPL2CLASS = {"hip_varlen": HipVarlenAttention}
