"""Token-choice MoE layer on the HIP seams — mirrors `prime_rl.trainer.models.layers.moe`
(src/prime_rl/trainer/models/layers/moe.py): `MoEArgs` (:21-37), `BCFeedForward` (:71-90, the shared expert), `GroupedExperts` (:431-485,
parameters `w1 [E, hidden, dim]`, `w2 [E, dim, hidden]`, `w3 [E, hidden, dim]`), `TokenChoiceTopKRouter` (:644-751, `gate`),
`TokenReorderer` (:754-809), `MoE` (:812-1054: `experts`, `router`, `reorderer`, `shared_expert`, buffers `expert_bias`,
`tokens_per_expert`, `routing_confidence_sum`) — same module tree, parameter names and forward contract (`x [bs, slen, dim]`, optional
`routed_experts [bs, slen, top_k]`).

What runs where: the gate GEMM is a library call; the expert GEMMs are the hand-written grouped MFMA kernels of grouped_gemm.hip
(`pm_grouped_gemm_nt / nn / tn`, device-side offsets; `torch._grouped_mm` for shapes that are not multiples of 128, a per-expert loop when
`use_grouped_mm=False`: the reference's two paths); everything between them is `moe.hip`:
score function + top-k + normalisation + histogram in one launch per layer (`pm_moe_router_fwd`, the reference: softmax, topk, gather,
sum, div, mul, histc), a four-launch stable counting sort (`pm_moe_sort`; the reference: histc + argsort + index + floor-divide), the
expert-sorted row gather with the optional score multiply (`pm_moe_gather_rows`) and the weighted combine as a deterministic gather with
f32 accumulation (`pm_moe_combine`; the reference: f32 multiply, cast, zeros_like, scatter_add with atomics).
Expert parallelism (`ep_group`): `prime_rl_amd.trainer.distributed.expert_parallel` dispatches the sorted rows to the ranks that own
the experts and brings the outputs back (all-to-all over xGMI), the experts module then holds `E / ep` experts.

Not built (the reference's other branches): DeepEP, the fused Triton MoE kernel, FP8 / MXFP8 experts, GPT-OSS and non-gated experts,
NemotronH router."""

from __future__ import annotations

import os

from dataclasses import dataclass

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from .... import ops
from ...distributed import expert_parallel as EP


@dataclass
class MoEArgs:
    num_experts: int = 8
    num_shared_experts: int = 1
    # router
    score_func: str = "sigmoid"
    route_norm: bool = False
    route_scale: float = 1.0
    score_before_experts: bool = True
    # token-choice
    top_k: int = 1
    use_grouped_mm: bool = True
    load_balance_coeff: float | None = 1e-3
    fp8: bool = False


class BCFeedForward(nn.Module):
    """The shared expert: w2(silu(w1 x) * w3 x) with bare parameter matrices (moe.py:71-90)."""

    def __init__(self, dim: int, hidden_dim: int) -> None:
        super().__init__()
        self.w1 = nn.Parameter(torch.empty(hidden_dim, dim))
        self.w2 = nn.Parameter(torch.empty(dim, hidden_dim))
        self.w3 = nn.Parameter(torch.empty(hidden_dim, dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.matmul(F.silu(torch.matmul(x, self.w1.T)) * torch.matmul(x, self.w3.T), self.w2.T)

    def init_weights(self, init_std: float):
        nn.init.trunc_normal_(self.w1, mean=0.0, std=0.02)
        nn.init.trunc_normal_(self.w2, mean=0.0, std=init_std)
        nn.init.trunc_normal_(self.w3, mean=0.0, std=init_std)


def _main_grads(*params):
    """Megatron-style main gradients: a trainer engine that owns flat f32 gradient storage hangs `main_grad` (the f32 view of this parameter's slice)
    and `main_grad_written` (a one-element list: has this step already written the slice?) on the expert parameters; the weight-gradient GEMM then
    writes / accumulates there in f32 and autograd gets no `.grad` for them (trainer/module_engine.py).  None unless EVERY given parameter has one."""
    out = []
    for p in params:
        mg, st = getattr(p, "main_grad", None), getattr(p, "main_grad_written", None)
        if mg is None or st is None:
            return None
        out.append((mg, st))
    return out


def _main_grad_sumsq(p):
    """The engine's per-launch workspace for the gradient norm fused into the weight-gradient epilogue (`ops.grouped_gemm_tn_f32(tile_sumsq=)`), hung on
    the FIRST parameter a launch writes (w1 for the gate|up launch, w2 for the down launch) as `main_grad_sumsq` = (slots tensor, [valid this step?]);
    None when the engine computes the norm with a pass over the gradients instead."""
    ws = getattr(p, "main_grad_sumsq", None)
    if ws is None:
        return None
    ws[1][0] = True  # this step's norm of the launch's outputs comes from the slots
    return ws[0]


class _GroupedLinearFn(torch.autograd.Function):
    """y[m] = x[m] @ w[expert of m].T over expert-sorted rows — `torch._grouped_mm(x, w.transpose(-2, -1), offs=...)` (moe.py:181-183) and its
    two gradients on the hand-written MFMA kernels of grouped_gemm.hip (pm_grouped_gemm_nt / nn / tn)."""

    @staticmethod
    def forward(ctx, x, w, offsets):
        ctx.save_for_backward(x, w, offsets)
        ctx.main_grads = _main_grads(w)
        return ops.grouped_gemm_nt(x, w, offsets)

    @staticmethod
    def backward(ctx, g):
        x, w, offsets = ctx.saved_tensors
        g = g.contiguous()
        gx = ops.grouped_gemm_nn(g, w, offsets) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            mg = ctx.main_grads
            if mg is not None and ops.grouped_gemm_tn_f32_supported(w.shape[1], w.shape[1], w.shape[2]):
                ops.grouped_gemm_tn_f32(g, x, offsets, mg[0][0], None, accumulate=mg[0][1][0], tile_sumsq=_main_grad_sumsq(w))
                mg[0][1][0] = True
            else:
                gw = ops.grouped_gemm_tn(g, x, offsets)
        return gx, gw, None


class _GroupedSwiGLUFn(torch.autograd.Function):
    """h = silu(x @ w1[e].T) * (x @ w3[e].T) (moe.py:181-182) with both projections in ONE grouped GEMM that writes the gate|up layout, the
    activation as `pm_swiglu_fwd`; backward: `pm_swiglu_bwd`, one row-gradient GEMM contracting over gate and up, one weight-gradient launch
    with two outputs.  Saves, per layer, two [rows, hidden] round trips, the silu / mul / silu-backward / add kernels and one read of x."""

    @staticmethod
    def forward(ctx, x, w1, w3, offsets):
        gate_up = ops.grouped_gemm_nt(x, w1, offsets, w2=w3)
        ctx.save_for_backward(x, w1, w3, offsets, gate_up)
        ctx.main_grads = _main_grads(w1, w3)
        return ops.swiglu_fwd(gate_up)

    @staticmethod
    def backward(ctx, g):
        x, w1, w3, offsets, gate_up = ctx.saved_tensors
        d_gate_up = ops.swiglu_bwd(g.contiguous(), gate_up)
        gx = ops.grouped_gemm_nn(d_gate_up, w1, offsets, w2=w3) if ctx.needs_input_grad[0] else None
        gw1 = gw3 = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            mg = ctx.main_grads
            if mg is not None and ops.grouped_gemm_tn_f32_supported(w1.shape[1], w1.shape[1] + w3.shape[1], w1.shape[2]):
                acc = mg[0][1][0]  # both halves were written by the same launches so far
                ops.grouped_gemm_tn_f32(d_gate_up, x, offsets, mg[0][0], mg[1][0], accumulate=acc, tile_sumsq=_main_grad_sumsq(w1))
                mg[0][1][0] = mg[1][1][0] = True
            else:
                gw1, gw3 = ops.grouped_gemm_tn(d_gate_up, x, offsets, n_first=w1.shape[1])
        return gx, gw1, gw3, None


class _GroupedFFNFn(torch.autograd.Function):
    """The whole expert FFN (moe.py:181-183) as ONE node, so that its backward can run the down projection's row-gradient GEMM with the SwiGLU backward
    in the epilogue (`pm_grouped_gemm_nn_swiglu_bwd`): the gradient of h is never written, `pm_swiglu_bwd`'s pass over three [rows, hidden] tensors is
    gone.  Same launches otherwise, same results bit for bit as _GroupedLinearFn(_GroupedSwiGLUFn(x))."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, offsets):
        if ops.grouped_gemm_nt_swiglu_supported(x, w1, w3):  # (w3 above w1 within 2 GiB: parameters of a trainer's flat buffer are)
            gate_up, h = ops.grouped_gemm_nt_swiglu(x, w1, w3, offsets)
        else:
            gate_up = ops.grouped_gemm_nt(x, w1, offsets, w2=w3)
            h = ops.swiglu_fwd(gate_up)
        ctx.save_for_backward(x, w1, w2, w3, offsets, gate_up, h)
        ctx.main_grads_13, ctx.main_grads_2 = _main_grads(w1, w3), _main_grads(w2)
        return ops.grouped_gemm_nt(h, w2, offsets)

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, w3, offsets, gate_up, h = ctx.saved_tensors
        g = g.contiguous()
        d_gate_up = ops.grouped_gemm_nn_swiglu_bwd(g, w2, offsets, gate_up)
        gw1 = gw2 = gw3 = None
        if ctx.needs_input_grad[2]:
            mg = ctx.main_grads_2
            if mg is not None and ops.grouped_gemm_tn_f32_supported(w2.shape[1], w2.shape[1], w2.shape[2]):
                ops.grouped_gemm_tn_f32(g, h, offsets, mg[0][0], None, accumulate=mg[0][1][0], tile_sumsq=_main_grad_sumsq(w2))
                mg[0][1][0] = True
            else:
                gw2 = ops.grouped_gemm_tn(g, h, offsets)
        gx = ops.grouped_gemm_nn(d_gate_up, w1, offsets, w2=w3) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            mg = ctx.main_grads_13
            if mg is not None and ops.grouped_gemm_tn_f32_supported(w1.shape[1], w1.shape[1] + w3.shape[1], w1.shape[2]):
                ops.grouped_gemm_tn_f32(d_gate_up, x, offsets, mg[0][0], mg[1][0], accumulate=mg[0][1][0], tile_sumsq=_main_grad_sumsq(w1))
                mg[0][1][0] = mg[1][1][0] = True
            else:
                gw1, gw3 = ops.grouped_gemm_tn(d_gate_up, x, offsets, n_first=w1.shape[1])
        return gx, gw1, gw2, gw3, None


def _experts_grouped_hip(w1, w2, w3, x, offsets):
    """_run_experts_grouped_mm_impl (moe.py:161-188) on pm_grouped_gemm_*: offsets stay on the device, no transposed weight copies."""
    if ops.grouped_gemm_nn_swiglu_bwd_supported(x.shape[0], w2.shape[1], w2.shape[2]):
        return _GroupedFFNFn.apply(x, w1, w2, w3, offsets)
    return _GroupedLinearFn.apply(_GroupedSwiGLUFn.apply(x, w1, w3, offsets), w2, offsets)


def _experts_grouped(w1, w2, w3, x, offsets):
    """The same through the library's grouped GEMM (`torch._grouped_mm`), for shapes the MFMA kernels are not built for."""
    offs = offsets[1:]  # cumulative END of every group, int32
    h = F.silu(torch._grouped_mm(x, w1.transpose(-2, -1), offs=offs))
    h = h * torch._grouped_mm(x, w3.transpose(-2, -1), offs=offs)
    return torch._grouped_mm(h, w2.transpose(-2, -1), offs=offs)


def _experts_for_loop(w1, w2, w3, x, offsets):
    """_run_experts_for_loop_impl (moe.py:93-126); like the reference this path synchronises with the host for the group sizes."""
    bounds = offsets.tolist()
    outs = []
    for e in range(len(bounds) - 1):
        xe = x[bounds[e]:bounds[e + 1]]
        h = F.silu(torch.matmul(xe, w1[e].transpose(-2, -1))) * torch.matmul(xe, w3[e].transpose(-2, -1))
        outs.append(torch.matmul(h, w2[e].transpose(-2, -1)))
    return torch.cat(outs, dim=0) if outs else x.new_zeros((0, w2.shape[1]))


class GroupedExperts(nn.Module):
    def __init__(self, dim: int, hidden_dim: int, num_experts: int, use_grouped_mm: bool, fp8: bool = False):
        super().__init__()
        if fp8:
            raise NotImplementedError("FP8 expert GEMMs (moe.py:151-158) are not built")
        self.num_experts = num_experts  # the LOCAL expert count under expert parallelism
        self.w1 = nn.Parameter(torch.empty(num_experts, hidden_dim, dim))
        self.w2 = nn.Parameter(torch.empty(num_experts, dim, hidden_dim))
        self.w3 = nn.Parameter(torch.empty(num_experts, hidden_dim, dim))
        self.use_grouped_mm = use_grouped_mm
        # "auto": the MFMA kernels when the shapes are multiples of 128 (every production MoE), else the library; "hip" / "torch" force one
        self.grouped_impl = "auto"

    def _apply(self, fn, *args, **kwargs):
        """After any `.to(device)` / `.bfloat16()`: keep w3 directly above w1 in ONE allocation, so that the grouped gate|up forward can read both through one
        buffer descriptor and run SwiGLU in its epilogue (ops.grouped_gemm_nt_swiglu; a trainer engine that re-homes the parameters in its flat buffer keeps
        that order by construction).  The parameters stay separate `nn.Parameter`s with the reference's names and shapes."""
        super()._apply(fn, *args, **kwargs)
        w1, w3 = self.w1, self.w3
        if w1.is_cuda and w1.dtype == torch.bfloat16 and w3.dtype == torch.bfloat16 and w1.shape == w3.shape and w1.device == w3.device and w1.numel() > 0:
            # re-stack only when the kernel's REAL placement requirement fails (pm_grouped_gemm_nt_swiglu: w3 above w1, a whole number of weight rows away,
            # within 2 GiB) — parameters a trainer engine re-homed in its flat buffer (w2 lies between them there) satisfy it and must NOT be moved: a fresh
            # allocation would detach them from the buffer the optimizer updates and the weight push reads
            delta, row_bytes = w3.data_ptr() - w1.data_ptr(), w1.shape[2] * w1.element_size()
            placed = delta > 0 and delta % row_bytes == 0 and delta + 128 * row_bytes < 2**31 - 4096
            engine_owned = hasattr(w1, "main_grad") or hasattr(w3, "main_grad")
            if not placed and not engine_owned:
                with torch.no_grad():
                    both = torch.stack([w1.detach(), w3.detach()])
                    self.w1.data, self.w3.data = both[0], both[1]
        return self

    def forward(self, x: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
        """x: rows sorted by (local) expert; offsets [E_local + 1] i32 on the device."""
        if not self.use_grouped_mm:
            return _experts_for_loop(self.w1, self.w2, self.w3, x, offsets)
        hidden, dim = self.w1.shape[1], self.w1.shape[2]
        if self.grouped_impl == "hip" or (self.grouped_impl == "auto" and x.is_cuda and x.dtype == torch.bfloat16
                                          and ops.grouped_gemm_supported(x.shape[0], hidden, dim)):
            return _experts_grouped_hip(self.w1, self.w2, self.w3, x.contiguous(), offsets)
        return _experts_grouped(self.w1, self.w2, self.w3, x, offsets)

    def init_weights(self, init_std: float):
        nn.init.trunc_normal_(self.w1, mean=0.0, std=0.02)
        nn.init.trunc_normal_(self.w2, mean=0.0, std=init_std)
        nn.init.trunc_normal_(self.w3, mean=0.0, std=init_std)


class _RouterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, expert_bias, forced, top_k, score_func, route_norm, route_scale):
        scores, top_scores, top_idx, _, conf = ops.moe_router_fwd(logits, top_k, score_func, route_norm, route_scale, expert_bias, forced,
                                                                  want_counts=False)
        ctx.save_for_backward(scores, top_idx)
        ctx.cfg = (score_func, route_norm, route_scale, logits.dtype)
        ctx.mark_non_differentiable(top_idx, conf)
        return top_scores, top_idx, conf

    @staticmethod
    def backward(ctx, g_top, *_):
        scores, top_idx = ctx.saved_tensors
        score_func, route_norm, route_scale, dtype = ctx.cfg
        g = ops.moe_router_bwd(scores, top_idx, g_top.contiguous().float(), score_func, route_norm, route_scale)
        return g.to(dtype), None, None, None, None, None, None


class _GateLinearFn(torch.autograd.Function):
    """logits = x @ w_gate^T (the router's nn.Linear, moe.py:674-680) with the weight gradient on the hand-written core: dw[E, dim] = g^T x contracts over
    ALL tokens into 128 x 2048 outputs — 8 tiles; the library runs it without a K split on 16 workgroups (177 us at 16k tokens), the core splits the
    contraction over the XCDs (pm_gemm_bf16_ws).  f32 accumulation, one rounding to the weight's dtype, like the library's."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.main_grads = _main_grads(w)
        return ops.gemm(x, w) if ops.gemm_supported(x, w) else torch.mm(x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.gemm(g, w.t()) if ops.gemm_supported(g, w.t()) else torch.mm(g, w)
        gw = None
        if ctx.needs_input_grad[1]:
            mg = ctx.main_grads
            if mg is not None:  # a trainer engine's f32 main gradient: written / accumulated by the GEMM itself
                ops.gemm(g.t(), x.t(), out=mg[0][0], accumulate=mg[0][1][0])
                mg[0][1][0] = True
            else:
                gw = ops.gemm(g.t(), x.t(), out_dtype=torch.float32).to(w.dtype)
        return gx, gw


def _gate_linear(gate: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    w = gate.weight
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and gate.bias is None and x.is_contiguous() and torch.is_grad_enabled()
            and w.requires_grad):
        if ops.gemm_shape_supported(w.shape[0], w.shape[1], x.shape[0], True, True) and x.data_ptr() % 16 == 0:  # dw = g^T x: both operands token-major
            return _GateLinearFn.apply(x, w)
    return gate(x)


class TokenChoiceTopKRouter(nn.Module):
    def __init__(self, dim: int, num_experts: int, top_k: int, score_func: str, route_norm: bool, route_scale: float):
        super().__init__()
        self.gate = nn.Linear(dim, num_experts, bias=False)
        self.num_experts, self.top_k = num_experts, top_k
        self.score_func, self.route_norm, self.route_scale = score_func, route_norm, route_scale
        self.force_balanced = False  # debug.force_balanced_routing (moe.py:721-725): round-robin experts instead of the top-k
        self.fp32_gate = False  # model.moe_router_dtype = 'float32' (moe.py:674-676): f32 gate weight and gate GEMM

    def forward(self, x: torch.Tensor, expert_bias: torch.Tensor | None = None, routed_experts: torch.Tensor | None = None):
        """-> (top_scores [N,k] f32, selected_experts_indices [N,k] i32, num_tokens_per_expert [E] i32, routing_confidence_sum)."""
        if routed_experts is not None and routed_experts.shape[-1] != self.top_k:
            raise AssertionError(f"routed_experts shape: {routed_experts.shape}, top_k: {self.top_k}")
        logits = self.gate(x.to(torch.float32)) if self.fp32_gate else _gate_linear(self.gate, x)
        forced = None if routed_experts is None else routed_experts.reshape(-1, self.top_k).to(torch.int32).contiguous()
        if forced is None and self.force_balanced:
            n = logits.shape[0]
            forced = (torch.arange(n * self.top_k, device=logits.device, dtype=torch.int32) % self.num_experts).view(n, self.top_k)
        top_scores, idx, conf = _RouterFn.apply(logits.contiguous(), expert_bias, forced, self.top_k, self.score_func, self.route_norm, self.route_scale)
        # the per-expert histogram (histc, moe.py:739-744) falls out of the stable sort the reorderer needs anyway: sort once, here, and
        # hand the maps to TokenReorderer through `last_sort` (an in-kernel histogram is 1k-way contended atomics at 16k tokens x 8 slots)
        self.last_sort = (idx, *ops.moe_sort(idx.reshape(-1), self.num_experts))
        offsets = self.last_sort[3]
        return top_scores, idx, offsets[1:] - offsets[:-1], conf.sum()

    def init_weights(self, init_std: float):
        nn.init.trunc_normal_(self.gate.weight, mean=0.0, std=init_std)


class TokenReorderer(nn.Module):
    """Stable sort of the (token, slot) pairs by expert (moe.py:754-809).  Returns the reference's triple plus the two index maps the
    permutation kernels consume (`slot_of_dest` = the argsort, `dest_of_slot` = its inverse) and the device-side group offsets."""

    def __init__(self, num_experts: int, top_k: int):
        super().__init__()
        self.num_experts, self.top_k = num_experts, top_k

    def forward(self, top_scores: torch.Tensor, selected_experts_indices: torch.Tensor, presorted=None):
        if presorted is not None and presorted[0] is selected_experts_indices:
            return presorted[1:]
        return ops.moe_sort(selected_experts_indices.reshape(-1).contiguous(), self.num_experts)


class _GatherFn(torch.autograd.Function):
    """routed_input = x[token of every sorted row] (* score, rounded to bf16): moe.py:891-895."""

    @staticmethod
    def forward(ctx, x, scores_flat, slot_of_dest, dest_of_slot, top_k):
        ctx.save_for_backward(x, scores_flat, slot_of_dest, dest_of_slot)
        ctx.top_k = top_k
        return ops.moe_gather_rows(x, slot_of_dest, top_k, scores_flat)

    @staticmethod
    def backward(ctx, g):
        x, scores_flat, slot_of_dest, dest_of_slot = ctx.saved_tensors
        g = g.contiguous()
        gx = ops.moe_combine(g, dest_of_slot, ctx.top_k, scores=scores_flat)  # the gather's scatter_add backward, fixed order
        gs = None
        if scores_flat is not None:
            _, gs = ops.moe_slot_grads(x, g, slot_of_dest, ctx.top_k, None, want_rows=False, want_dscore=True)
        return gx, gs, None, None, None


class _CombineFn(torch.autograd.Function):
    """out = base + scatter_add of bf16(routed_output * score) over token rows: moe.py:899-900, 1043-1051."""

    @staticmethod
    def forward(ctx, y, scores_flat, base, slot_of_dest, dest_of_slot, top_k):
        ctx.save_for_backward(y, scores_flat, slot_of_dest)
        ctx.top_k, ctx.has_base = top_k, base is not None
        return ops.moe_combine(y, dest_of_slot, top_k, scores=scores_flat, base=base)

    @staticmethod
    def backward(ctx, g):
        y, scores_flat, slot_of_dest = ctx.saved_tensors
        g = g.contiguous()
        gy, gs = ops.moe_slot_grads(g, y, slot_of_dest, ctx.top_k, scores_flat, want_rows=True, want_dscore=scores_flat is not None)
        return gy, gs, (g if ctx.has_base else None), None, None, None


class MoE(nn.Module):
    def __init__(self, moe_args: MoEArgs, dim: int, hidden_dim: int, ep_group: dist.ProcessGroup | None = None):
        super().__init__()
        num_experts = moe_args.num_experts
        self.ep_group = ep_group
        self.ep_size = dist.get_world_size(ep_group) if ep_group is not None else 1
        if num_experts % self.ep_size:
            raise ValueError(f"num_experts {num_experts} is not divisible by the expert-parallel degree {self.ep_size}")
        self.experts = GroupedExperts(dim=dim, hidden_dim=hidden_dim, num_experts=num_experts // self.ep_size, use_grouped_mm=moe_args.use_grouped_mm,
                                      fp8=moe_args.fp8)
        self.router = TokenChoiceTopKRouter(dim=dim, num_experts=num_experts, top_k=moe_args.top_k, score_func=moe_args.score_func,
                                            route_norm=moe_args.route_norm, route_scale=moe_args.route_scale)
        self.reorderer = TokenReorderer(num_experts=num_experts, top_k=moe_args.top_k)
        self.shared_expert = BCFeedForward(dim=dim, hidden_dim=hidden_dim * moe_args.num_shared_experts) if moe_args.num_shared_experts > 0 else None
        self.score_before_experts = moe_args.score_before_experts
        self.top_k, self.num_experts = moe_args.top_k, num_experts
        self.load_balance_coeff = moe_args.load_balance_coeff
        if self.load_balance_coeff is not None:
            assert self.load_balance_coeff > 0.0
            self.register_buffer("expert_bias", torch.zeros(num_experts, dtype=torch.float32), persistent=True)
        else:
            self.expert_bias = None
        self.register_buffer("tokens_per_expert", torch.zeros(num_experts, dtype=torch.float32), persistent=False)
        self.register_buffer("routing_confidence_sum", torch.tensor(0.0, dtype=torch.float32), persistent=False)

    def _apply(self, fn, *args, **kwargs):
        """`.to(torch.bfloat16)` / `.bfloat16()` on the model must not round the routing buffers: token counts above 256 and the load-balance bias are
        f32 quantities (the reference keeps buffers out of its bf16 cast, model.py:1079-1083)."""
        super()._apply(fn, *args, **kwargs)
        for name in ("tokens_per_expert", "routing_confidence_sum", "expert_bias"):
            buf = getattr(self, name, None)
            if isinstance(buf, torch.Tensor) and buf.is_floating_point() and buf.dtype != torch.float32:
                self._buffers[name] = buf.float()
        return self

    def _forward_ep_chunked(self, x: torch.Tensor, routed_experts, chunks: int) -> torch.Tensor:
        """Expert parallelism with the dispatch / combine all-to-alls hidden under expert compute: the router runs ONCE over all tokens (its gate GEMM must
        not depend on the chunking: a library GEMM picks its algorithm by row count), then the tokens are cut into `chunks` pieces that are sorted and
        gathered first (so that ONE exchange of counts and ONE host copy of split sizes serve the layer) and `EP.run_experts_chunked` pipelines dispatch ->
        experts -> combine over the pieces.  Same output and input gradient as the one-piece path bit for bit: sort, gather and combine are per token, an
        expert's rows are independent of each other; the weight gradients accumulate the pieces in f32 main-grads (or bf16 `.grad`s like two micro-batches
        would)."""
        n = x.shape[0]
        top_scores, idx, counts, conf = self.router(x, self.expert_bias, routed_experts=routed_experts)
        with torch.no_grad():  # moe.py:1010-1012
            self.tokens_per_expert.add_(counts)
            self.routing_confidence_sum.add_(conf)
        cuts = [n * i // chunks for i in range(chunks + 1)]
        pieces, routed = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            xc = x[a:b]
            if b == a:  # fewer tokens than pieces on this rank: the piece still takes part in every collective, with zero rows
                routed.append((xc, torch.zeros(self.num_experts, dtype=torch.int32, device=x.device)))  # xc = x[a:a]: zero rows that stay in the autograd graph
                pieces.append(None)
                continue
            dest_of_slot, slot_of_dest, offsets = self.reorderer(top_scores[a:b], idx[a:b].contiguous())
            scores_flat = top_scores[a:b].reshape(-1)
            routed.append((_GatherFn.apply(xc, scores_flat if self.score_before_experts else None, slot_of_dest, dest_of_slot, self.top_k),
                           (offsets[1:] - offsets[:-1]).to(torch.int32)))
            pieces.append((xc, scores_flat, dest_of_slot, slot_of_dest))
        outs = EP.run_experts_chunked(self.experts, routed, self.ep_group)
        res = []
        for piece, routed_out in zip(pieces, outs):
            if piece is None:  # zero rows came back; they still belong to the output's graph, so that this rank issues the gradients' all-to-alls as well
                res.append(routed_out)
                continue
            xc, scores_flat, dest_of_slot, slot_of_dest = piece
            base = self.shared_expert(xc) if self.shared_expert is not None else None
            res.append(_CombineFn.apply(routed_out.contiguous(), None if self.score_before_experts else scores_flat, base, slot_of_dest, dest_of_slot, self.top_k))
        return torch.cat(res, dim=0)

    def forward(self, x: torch.Tensor, routed_experts: torch.Tensor | None = None) -> torch.Tensor:
        bs, slen, dim = x.shape
        x = x.reshape(-1, dim)
        if x.dtype != torch.bfloat16:
            raise NotImplementedError("the HIP MoE permutation kernels move bf16 rows (the reference trains in bf16)")
        if routed_experts is not None:
            routed_experts = routed_experts.reshape(-1, routed_experts.shape[-1])
        if self.ep_size > 1:
            # PM_EP_OVERLAP_CHUNKS: pieces the layer's tokens are cut into so that the row all-to-alls travel under expert compute (default 2 on a GPU; 1 = the
            # one-piece synchronous exchange).  Every rank must take the same branch: the decision uses only the environment and the layer's configuration.
            chunks = int(os.environ.get("PM_EP_OVERLAP_CHUNKS", "2" if x.is_cuda else "1"))
            if chunks > 1:
                return self._forward_ep_chunked(x, routed_experts, chunks).reshape(bs, slen, dim)
        top_scores, idx, counts, conf = self.router(x, self.expert_bias, routed_experts=routed_experts)
        with torch.no_grad():  # moe.py:1010-1012
            self.tokens_per_expert.add_(counts)
            self.routing_confidence_sum.add_(conf)
        dest_of_slot, slot_of_dest, offsets = self.reorderer(top_scores, idx, presorted=getattr(self.router, "last_sort", None))
        scores_flat = top_scores.reshape(-1)
        routed_in = _GatherFn.apply(x, scores_flat if self.score_before_experts else None, slot_of_dest, dest_of_slot, self.top_k)
        if self.ep_size > 1:
            routed_out = EP.run_experts(self.experts, routed_in, counts, self.ep_group)
        else:
            routed_out = self.experts(routed_in, offsets)
        base = self.shared_expert(x) if self.shared_expert is not None else None
        out = _CombineFn.apply(routed_out.contiguous(), None if self.score_before_experts else scores_flat, base, slot_of_dest, dest_of_slot, self.top_k)
        return out.reshape(bs, slen, dim)

    def init_weights(self, init_std: float, buffer_device: torch.device | None = None):
        self.experts.init_weights(init_std)
        self.router.init_weights(init_std)
        if self.shared_expert is not None:
            self.shared_expert.init_weights(init_std)
        with torch.no_grad():
            self.tokens_per_expert.zero_()
            self.routing_confidence_sum.zero_()
            if self.expert_bias is not None:
                self.expert_bias.zero_()
