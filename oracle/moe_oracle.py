"""CPU restatement of the reference's token-choice MoE layer — TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

Follows /root/reference/src/prime_rl/trainer/models/layers/moe.py (line numbers below) in plain torch ops, device-agnostic, autograd
supplying the backward; pinned by tests/test_moe_oracle_cpu.py against tests/golden/moe.npz, which tests/golden/make_moe_golden.py
produced by running the reference's own `MoE` module (router outputs, layer output, input and weight gradients, four argument sets).
The expert-parallel dispatch / combine of torchtitan 0.x `ExpertParallel` (third party, not in /root/reference; used through
trainer/distributed/expert_parallel.py:47-70) is restated in `ep_dispatch_layout`: counts all-to-all, row all-to-all with the summed
splits, rows regrouped from (source rank, local expert) to (local expert, source rank) order.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class MoEArgs:  # moe.py:21-37
    num_experts: int = 8
    num_shared_experts: int = 1
    score_func: str = "sigmoid"
    route_norm: bool = False
    route_scale: float = 1.0
    score_before_experts: bool = True
    top_k: int = 1
    load_balance_coeff: float | None = 1e-3


def router(x, gate_w, args: MoEArgs, expert_bias=None, routed_experts=None):
    """TokenChoiceTopKRouter.forward (moe.py:678-746): gate GEMM in the activation dtype, score function in f32, top-k (on biased scores
    when a load-balance bias is present; the gating value is always the unbiased score), optional normalisation and scale, per-expert
    histogram, and the selected probability mass (moe.py:634-641)."""
    scores = F.linear(x, gate_w).float()
    scores = torch.sigmoid(scores) if args.score_func == "sigmoid" else torch.softmax(scores, dim=1)
    if routed_experts is not None:
        idx = routed_experts.long()
        top = scores.gather(1, idx)
    elif expert_bias is not None:
        idx = torch.topk(scores + expert_bias, args.top_k, dim=1).indices
        top = scores.gather(1, idx)
    else:
        top, idx = torch.topk(scores, args.top_k, dim=1)
    with torch.no_grad():
        conf = top.sum() if args.score_func == "softmax" else (top / (scores.sum(-1, keepdim=True) + 1e-20)).sum()
    if args.route_norm:
        top = top / (top.sum(-1, keepdim=True) + 1e-20)
    top = top * args.route_scale
    counts = torch.bincount(idx.reshape(-1), minlength=args.num_experts)
    return top, idx, counts, conf, scores


def reorder(top_scores, idx, top_k: int):
    """TokenReorderer.forward (moe.py:769-809): stable sort of the flattened (token, slot) pairs by expert."""
    order = torch.argsort(idx.reshape(-1), stable=True)
    return top_scores.reshape(-1)[order], order // top_k, order


def experts_for_loop(w1, w2, w3, x_sorted, counts):
    """_run_experts_for_loop_impl (moe.py:93-126): per expert silu(x w1^T) * (x w3^T) then w2^T, every product in the activation dtype."""
    outs, o = [], 0
    for e, c in enumerate(counts.tolist()):
        xe = x_sorted[o:o + c]
        h = F.silu(xe @ w1[e].T) * (xe @ w3[e].T)
        outs.append(h @ w2[e].T)
        o += c
    return torch.cat(outs, 0) if outs else x_sorted.new_zeros((0, w2.shape[1]))


def moe_forward(x, params: dict, args: MoEArgs, expert_bias=None, routed_experts=None):
    """MoE.forward (moe.py:976-1054) on [N, dim] tokens.  `params`: router.gate.weight, experts.w1/w2/w3, shared_expert.w1/w2/w3."""
    top, idx, counts, conf, _ = router(x, params["router.gate.weight"], args, expert_bias, routed_experts)
    s_sorted, tok_sorted, _ = reorder(top, idx, args.top_k)
    rows = tok_sorted.reshape(-1, 1).expand(-1, x.shape[-1])
    routed_in = torch.gather(x, 0, rows)  # moe.py:891-892 (the same op as the reference: its backward fixes the bf16 accumulation order)
    if args.score_before_experts:
        routed_in = (routed_in.float() * s_sorted[:, None]).to(x.dtype)
    routed_out = experts_for_loop(params["experts.w1"], params["experts.w2"], params["experts.w3"], routed_in, counts)
    if not args.score_before_experts:
        routed_out = (routed_out.float() * s_sorted[:, None]).to(x.dtype)
    if args.num_shared_experts > 0:  # BCFeedForward (moe.py:82-83)
        w1, w2, w3 = (params[f"shared_expert.{k}"] for k in ("w1", "w2", "w3"))
        out = (F.silu(x @ w1.T) * (x @ w3.T)) @ w2.T
    else:
        out = torch.zeros_like(x)
    out = out.scatter_add(0, rows, routed_out)  # moe.py:1050-1051, accumulated in the activation dtype
    return out, dict(top_scores=top, top_idx=idx, counts=counts, confidence=conf)


def ep_dispatch_layout(counts_per_rank: list[torch.Tensor], ep: int):
    """What expert parallelism does to the expert-sorted rows (torchtitan ExpertParallel._token_dispatch as driven by
    expert_parallel.py:47-63): rank r sends to rank d the rows of experts d*E/ep .. (d+1)*E/ep - 1; rank d receives them in
    (source rank, local expert) order and regroups them to (local expert, source rank).  Returns, for every destination rank d, the list
    of (source rank, first row in the source's expert-sorted buffer, row count) in the order the rows are consumed by d's local experts,
    and d's per-local-expert totals.  Pure index bookkeeping: used to check the device-side layout kernels and the gloo test."""
    e_total = counts_per_rank[0].numel()
    le = e_total // ep
    offs = [torch.cat([c.new_zeros(1), c.cumsum(0)]) for c in counts_per_rank]
    plan, totals = [], []
    for d in range(ep):
        segs, tot = [], []
        for j in range(le):
            e = d * le + j
            tot.append(sum(int(c[e]) for c in counts_per_rank))
            for r in range(ep):
                segs.append((r, int(offs[r][e]), int(counts_per_rank[r][e])))
        plan.append(segs)
        totals.append(tot)
    return plan, totals
