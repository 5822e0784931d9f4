"""Expert parallelism for the routed experts — the role of `prime_rl.trainer.distributed.expert_parallel`
(src/prime_rl/trainer/distributed/expert_parallel.py:47-70) and of torchtitan's `ExpertParallel` it builds on (third party, not in the
reference tree; its `_token_dispatch` / `_token_combine` are restated in oracle/moe_oracle.py::ep_dispatch_layout).

Every rank routes its own tokens over ALL experts and sorts the (token, slot) rows by expert; rank d owns experts
[d * E/ep, (d+1) * E/ep).  Dispatch: (1) all-to-all of the per-expert counts, (2) all-to-all of the rows with the per-rank sums as
splits — xGMI is a full mesh of point-to-point links, so an all-to-all is one direct exchange per pair, no ring —, (3) the received rows
arrive in (source rank, local expert) order and are regrouped to (local expert, source rank) by `pm_moe_ep_row_map` +
`pm_moe_gather_rows`, so the local grouped GEMM sees one contiguous group per expert.  Combine is the exact inverse.  Like the reference
(`.tolist()` on the split sizes, expert_parallel.py:52-53) the row all-to-all needs its split sizes on the host: one small D2H per layer.
The backward of an all-to-all is the all-to-all with the splits swapped; of a row permutation, the inverse permutation."""

from __future__ import annotations

import torch
import torch.distributed as dist

from ... import ops


class _AllToAllRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_splits, in_splits, group):
        ctx.splits, ctx.group = (out_splits, in_splits), group
        out = x.new_empty((sum(out_splits),) + tuple(x.shape[1:]))
        dist.all_to_all_single(out, x.contiguous(), out_splits, in_splits, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        out_splits, in_splits = ctx.splits
        gi = g.new_empty((sum(in_splits),) + tuple(g.shape[1:]))
        dist.all_to_all_single(gi, g.contiguous(), in_splits, out_splits, group=ctx.group)
        return gi, None, None, None


class _PermuteRows(torch.autograd.Function):
    """out[r] = x[row_map[r]]; the backward applies the inverse permutation."""

    @staticmethod
    def forward(ctx, x, row_map, inv_map):
        ctx.save_for_backward(row_map, inv_map)
        return ops.moe_gather_rows(x, row_map)

    @staticmethod
    def backward(ctx, g):
        row_map, inv_map = ctx.saved_tensors
        return ops.moe_gather_rows(g.contiguous(), inv_map), None, None


def exchange_counts(counts: torch.Tensor, group) -> torch.Tensor:
    """counts [E] (this rank's rows per GLOBAL expert) -> recv_counts [ep, E/ep] (rows this rank receives per (source, local expert))."""
    ep = dist.get_world_size(group)
    recv = torch.empty_like(counts)
    dist.all_to_all_single(recv, counts.contiguous(), group=group)
    return recv.view(ep, -1)


def run_experts(experts, routed_in: torch.Tensor, counts: torch.Tensor, group) -> torch.Tensor:
    """routed_in: this rank's rows sorted by global expert, counts [E] i32.  Returns the expert outputs in the same row order."""
    ep = dist.get_world_size(group)
    with torch.no_grad():
        recv_counts = exchange_counts(counts, group)                              # [ep, E/ep]
        in_splits = counts.view(ep, -1).sum(dim=1).tolist()                       # rows sent to each rank (host sync, as the reference)
        out_splits = recv_counts.sum(dim=1).tolist()                              # rows received from each rank
        row_map, inv_map, local_offsets = ops.moe_ep_row_map(recv_counts.to(torch.int32).contiguous(), sum(out_splits))
    received = _AllToAllRows.apply(routed_in, out_splits, in_splits, group)       # (source, local expert) order
    local_in = _PermuteRows.apply(received, row_map, inv_map)                     # (local expert, source) order
    local_out = experts(local_in, local_offsets)
    back = _PermuteRows.apply(local_out.contiguous(), inv_map, row_map)           # (source, local expert) order again
    return _AllToAllRows.apply(back, in_splits, out_splits, group)
