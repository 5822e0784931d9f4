"""The dense projections of the nn.Module seams on the hand-written MFMA core — what `nn.Linear` / `F.linear` is in the reference's layers
(src/prime_rl/trainer/models/layers/attn.py:60-84 q/k/v/o_proj, mlp.py:27-29 gate/up/down_proj, moe.py:674-680 the router gate; cuBLAS there).

`linear(x, weight, bias)` is `F.linear` as ONE autograd node over `ops.gemm` (csrc/gemm.hip, `pm_gemm_bf16*`): forward `x @ W^T (+ b)`, row gradient
`dy @ W` on the stored `[out, in]` weight read K-major (no transposed copies), weight gradient `dy^T @ x` with f32 accumulation — written STRAIGHT into
the parameter's f32 `main_grad` when a trainer engine hung one on it (Megatron-style: `p.main_grad`, `p.main_grad_written`; trainer/module_engine.py),
else returned as the usual `.grad`.  `fused_linear(x, [W_a, W_b, ...])` runs several projections of the same input as ONE product when their weights lie
back to back in one allocation (the q | k | v and gate | up projections of a model whose parameters a trainer engine re-homed in its flat buffer do): one
read of x, one launch, and the output is the concatenated layout the next kernel wants (no `torch.cat`).  Shapes / dtypes / devices the core does not
take fall back to `F.linear` — `gemm_supported` decides, never silently on a GPU box with bf16 operands of the production shapes.

`add_rms_norm(x, residual, weight, eps)` = `(RMSNorm(x + residual), x + residual)` as one kernel each way (the residual add of the decoder block fused
into the norm that follows it, as the flat engine does): no stand-alone add kernels in forward, no gradient-accumulation adds in backward."""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

from .... import ops


def _direct_of(params) -> list | None:
    """[(main_grad, written_flag), ...] when EVERY parameter carries a trainer engine's f32 main gradient, else None."""
    out = []
    for p in params:
        mg, st = getattr(p, "main_grad", None), getattr(p, "main_grad_written", None)
        if mg is None or st is None:
            return None
        out.append((mg, st))
    return out


def _adjacent_view(tensors: list[Tensor]) -> Tensor | None:
    """One [sum(out_i), in] view over 2-D row-major tensors that lie back to back in ONE storage (None otherwise)."""
    t0 = tensors[0]
    if any(t.dim() != 2 or not t.is_contiguous() or t.dtype != t0.dtype or t.device != t0.device or t.shape[1] != t0.shape[1] for t in tensors):
        return None
    try:
        s0 = t0.untyped_storage().data_ptr()
        if any(t.untyped_storage().data_ptr() != s0 for t in tensors):
            return None
    except RuntimeError:
        return None
    cursor = t0.data_ptr()
    for t in tensors:
        if t.data_ptr() != cursor:
            return None
        cursor += t.numel() * t.element_size()
    rows = sum(t.shape[0] for t in tensors)
    if t0.storage_offset() + rows * t0.shape[1] > t0.untyped_storage().nbytes() // t0.element_size():
        return None
    return torch.as_strided(t0.detach(), (rows, t0.shape[1]), (t0.shape[1], 1))


def _own_ok(x: Tensor, w: Tensor) -> bool:
    """Forward, row-gradient and weight-gradient products of this (x, w) pair all run on the core."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(1) == 1):
        return False
    n, k = x.shape
    o = w.shape[0]
    return (ops.gemm_supported(x, w) and ops.gemm_shape_supported(n, k, o, False, True) and ops.gemm_shape_supported(o, k, n, True, True)
            and x.data_ptr() % 16 == 0)


class _LinearFn(torch.autograd.Function):
    """y = x @ Wf^T (+ bias) where Wf is the (possibly fused) weight view of `n_w` parameters passed after it for autograd's bookkeeping."""

    @staticmethod
    def forward(ctx, x, w_view, bias, variant, *weights):
        y = ops.gemm(x, w_view, bias=bias, variant=variant)
        ctx.save_for_backward(x, w_view)
        ctx.direct = _direct_of(weights)
        if ctx.direct is not None and len(weights) > 1:
            mg = _adjacent_view([d[0] for d in ctx.direct])
            ctx.direct_view = mg
        else:
            ctx.direct_view = ctx.direct[0][0] if ctx.direct is not None else None
        ctx.rows = [w.shape[0] for w in weights]
        ctx.w_dtype = weights[0].dtype
        ctx.has_bias = bias is not None
        ctx.variant = variant
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_view = ctx.saved_tensors
        dy = dy if (dy.stride(1) == 1 and dy.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
        dx = ops.gemm(dy, w_view.t(), variant=ctx.variant) if ctx.needs_input_grad[0] else None
        n_w = len(ctx.rows)
        grads: list = [None] * n_w
        if any(ctx.needs_input_grad[4:4 + n_w]):
            if ctx.direct is not None and ctx.direct_view is not None:
                flags = [d[1] for d in ctx.direct]
                ops.gemm(dy.t(), x.t(), out=ctx.direct_view, accumulate=all(f[0] for f in flags), variant=ctx.variant)
                for f in flags:
                    f[0] = True
            elif ctx.direct is not None:  # main gradients that are not adjacent: one launch per parameter on its column block of dy
                r0 = 0
                for (mg, f), r in zip(ctx.direct, ctx.rows):
                    ops.gemm(dy[:, r0:r0 + r].t(), x.t(), out=mg, accumulate=f[0], variant=ctx.variant)
                    f[0] = True
                    r0 += r
            else:
                gw = ops.gemm(dy.t(), x.t(), out_dtype=torch.float32, variant=ctx.variant).to(ctx.w_dtype)
                r0 = 0
                for i, r in enumerate(ctx.rows):
                    grads[i] = gw[r0:r0 + r]
                    r0 += r
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            acc = torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
            ops.column_sum(dy, acc)
            db = acc.to(ctx.w_dtype)
        return (dx, None, db, None, *grads)


def linear(x: Tensor, weight: Tensor, bias: Tensor | None = None, variant: int | None = None) -> Tensor:
    """F.linear on the hand-written core for [N, in] bf16 rows (any leading dims are flattened); falls back to F.linear where the core does not apply."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if _own_ok(x2, weight) and weight.is_contiguous():
        y = _LinearFn.apply(x2, weight.detach(), bias, variant, weight)
        return y.view(*shape[:-1], weight.shape[0])
    return F.linear(x, weight, bias)


def fused_linear(x: Tensor, weights: list[Tensor], biases: list[Tensor | None] | None = None, variant: int | None = None) -> Tensor:
    """cat([F.linear(x, w_i, b_i)], dim=-1) for x [N, in]: ONE product when the weights are adjacent in memory (and so are the biases, if any); else one
    product per weight on the core (or F.linear) and a concatenation."""
    biases = biases if biases is not None else [None] * len(weights)
    wv = _adjacent_view([w for w in weights]) if all(w.is_contiguous() for w in weights) else None
    if wv is not None and _own_ok(x, wv):
        bias = None
        if any(b is not None for b in biases):
            bias = torch.cat([b if b is not None else x.new_zeros(w.shape[0]) for b, w in zip(biases, weights)])  # (tiny; differentiable)
        return _LinearFn.apply(x, wv, bias, variant, *weights)
    return torch.cat([linear(x, w, b, variant) for w, b in zip(weights, biases)], dim=-1)


class _GateUpSwiGLUDownFn(torch.autograd.Function):
    """down(silu(gate(x)) * up(x)) (mlp.py:27-29) as ONE node on the core: the gate|up product with the SwiGLU activation in its epilogue
    (pm_gemm_swiglu_bf16), the down projection, and in backward the down projection's row gradient with the SwiGLU BACKWARD in its epilogue
    (pm_gemm_swiglu_bwd_bf16) — `act`'s gradient is never written.  Same values as linear / swiglu / linear bit for bit."""

    @staticmethod
    def forward(ctx, x, wgu_view, w_gate, w_up, w_down, variant):
        gu, act = ops.gemm_swiglu(x, wgu_view, variant=variant)
        y = ops.gemm(act, w_down, variant=variant)
        ctx.save_for_backward(x, wgu_view, w_down, gu, act)
        ctx.d_gu, ctx.d_down = _direct_of([w_gate, w_up]), _direct_of([w_down])
        ctx.inter = w_gate.shape[0]
        ctx.variant = variant
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wgu, w_down, gu, act = ctx.saved_tensors
        v = ctx.variant
        dy = dy.contiguous()
        w_dtype = w_down.dtype
        g_gate = g_up = g_down = None
        if ctx.needs_input_grad[4]:
            if ctx.d_down is not None:
                mg, f = ctx.d_down[0]
                ops.gemm(dy.t(), act.t(), out=mg, accumulate=f[0], variant=v)
                f[0] = True
            else:
                g_down = ops.gemm(dy.t(), act.t(), out_dtype=torch.float32, variant=v).to(w_dtype)
        if ops.gemm_swiglu_bwd_supported(dy, w_down):
            dgu = ops.gemm_swiglu_bwd(dy, w_down, gu, out=gu, variant=v)  # in place over the saved gate|up
        else:
            dgu = ops.swiglu_bwd(ops.gemm(dy, w_down.t(), variant=v), gu)
        dx = ops.gemm(dgu, wgu.t(), variant=v) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            mgv = _adjacent_view([d[0] for d in ctx.d_gu]) if ctx.d_gu is not None else None
            if mgv is not None:
                flags = [d[1] for d in ctx.d_gu]
                ops.gemm(dgu.t(), x.t(), out=mgv, accumulate=all(f[0] for f in flags), variant=v)
                for f in flags:
                    f[0] = True
            else:
                gw = ops.gemm(dgu.t(), x.t(), out_dtype=torch.float32, variant=v)
                if ctx.d_gu is not None:
                    for (mg, f), part in zip(ctx.d_gu, (gw[:ctx.inter], gw[ctx.inter:])):
                        if f[0]:
                            mg.add_(part)
                        else:
                            mg.copy_(part)
                        f[0] = True
                else:
                    g_gate, g_up = gw[:ctx.inter].to(w_dtype), gw[ctx.inter:].to(w_dtype)
        return dx, None, g_gate, g_up, g_down, None


def swiglu_mlp(x: Tensor, w_gate: Tensor, w_up: Tensor, w_down: Tensor, variant: int | None = None) -> Tensor | None:
    """The fused MLP node for x [N, hidden] when gate_proj / up_proj are adjacent in memory and every product fits the core; None otherwise (the caller keeps
    the three-projection form)."""
    if not (w_gate.is_contiguous() and w_up.is_contiguous() and w_down.is_contiguous()):
        return None
    wgu = _adjacent_view([w_gate, w_up])
    if wgu is None or not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()):
        return None
    n, h = x.shape
    inter = w_gate.shape[0]
    if not (ops.gemm_swiglu_supported(x, wgu) and ops.gemm_shape_supported(n, h, inter, False, False) and ops.gemm_shape_supported(n, h, 2 * inter, False, True)
            and ops.gemm_shape_supported(2 * inter, h, n, True, True) and ops.gemm_shape_supported(h, inter, n, True, True) and w_down.dtype == torch.bfloat16):
        return None
    return _GateUpSwiGLUDownFn.apply(x, wgu, w_gate, w_up, w_down, variant)


class _AddRMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, eps):
        y, rstd, s = ops.rmsnorm_fwd(x, weight.contiguous(), eps, residual=residual)
        ctx.save_for_backward(s, weight, rstd)
        return y, s

    @staticmethod
    def backward(ctx, gy, gs):
        s, weight, rstd = ctx.saved_tensors
        gw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
        if gy is None:
            return gs, gs, None, None
        gx = ops.rmsnorm_bwd(gy.contiguous(), s, weight.contiguous(), rstd, gw, dresidual=gs.contiguous() if gs is not None else None)
        return gx, gx, gw.to(weight.dtype), None


def add_rms_norm(x: Tensor, residual: Tensor, weight: Tensor, eps: float) -> tuple[Tensor, Tensor]:
    """(RMSNorm(x + residual) * weight, x + residual) for [N, H] bf16 rows — `pm_rmsnorm_fwd` with its residual input; backward = `pm_rmsnorm_bwd` with the
    incoming gradient of the sum as its `dresidual` (both inputs receive the same gradient)."""
    if x.is_cuda and x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_contiguous() and residual.is_contiguous():
        return _AddRMSNormFn.apply(x, residual, weight, eps)
    s = x + residual
    from .norms import rms_norm

    return rms_norm(s, weight, eps), s


def final_norm(norm, hidden_states: Tensor, pending: Tensor | None) -> Tensor:
    """The model's last RMSNorm over `hidden_states (+ pending)`: the last block's second residual add fused into it."""
    if pending is None:
        return norm(hidden_states)
    shape = hidden_states.shape
    y, _ = add_rms_norm(pending.reshape(-1, shape[-1]), hidden_states.reshape(-1, shape[-1]), norm.weight, norm.variance_epsilon)
    return y.view(shape)
