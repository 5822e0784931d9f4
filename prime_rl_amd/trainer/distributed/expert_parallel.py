"""Expert parallelism for the routed experts — the role of `prime_rl.trainer.distributed.expert_parallel`
(src/prime_rl/trainer/distributed/expert_parallel.py:47-70) and of torchtitan's `ExpertParallel` it builds on (third party, not in the
reference tree; its `_token_dispatch` / `_token_combine` are restated in oracle/moe_oracle.py::ep_dispatch_layout).

Every rank routes its own tokens over ALL experts and sorts the (token, slot) rows by expert; rank d owns experts
[d * E/ep, (d+1) * E/ep).  Dispatch: (1) all-to-all of the per-expert counts, (2) all-to-all of the rows with the per-rank sums as
splits — xGMI is a full mesh of point-to-point links, so an all-to-all is one direct exchange per pair, no ring —, (3) the received rows
arrive in (source rank, local expert) order and are regrouped to (local expert, source rank) by `pm_moe_ep_row_map` +
`pm_moe_gather_rows`, so the local grouped GEMM sees one contiguous group per expert.  Combine is the exact inverse.  Like the reference
(`.tolist()` on the split sizes, expert_parallel.py:52-53) the row all-to-all needs its split sizes on the host: one small D2H per layer.
The backward of an all-to-all is the all-to-all with the splits swapped; of a row permutation, the inverse permutation.

Overlap (`run_experts_chunked`, the reference runs `_token_dispatch` / `_token_combine` synchronously, expert_parallel.py:44-66): the layer's tokens are cut
into C chunks that are routed, dispatched, computed and combined independently; every row all-to-all is ISSUED asynchronously (c10d runs it on the
communicator's own stream — on MI355X a direct xGMI exchange per pair) and WAITED for right before its consumer, so chunk c + 1's dispatch travels under chunk
c's expert GEMMs and chunk c's combine under chunk c + 1's.  In the backward the same holds with the roles swapped: the node that waited in the forward
issues the gradient's all-to-all, the node that issued in the forward waits for it — autograd runs the chunks' subgraphs in reverse creation order, which
interleaves them the same way.  One exchange of ALL chunks' counts and ONE device-to-host copy of the split sizes per layer.  Exposed per layer and pass:
the first chunk's dispatch and the last chunk's combine, i.e. 1 / C of the row traffic."""

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


class _AllToAllStart(torch.autograd.Function):
    """Issue the row all-to-all asynchronously; `st` carries the work handle to _AllToAllFinish (and, in the backward, the gradient's from it)."""

    @staticmethod
    def forward(ctx, x, out_splits, in_splits, group, st):
        ctx.st = st
        out = x.new_empty((sum(out_splits),) + tuple(x.shape[1:]))
        st["splits"], st["group"] = (out_splits, in_splits), group
        st["src"] = x.contiguous()  # kept alive until the exchange has completed
        st["work"] = dist.all_to_all_single(out, st["src"], out_splits, in_splits, group=group, async_op=True)
        return out

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        st.pop("bwd_work").wait()  # issued by _AllToAllFinish.backward, which autograd runs first
        st.pop("bwd_src", None)
        return st.pop("bwd_out"), None, None, None, None


class _AllToAllFinish(torch.autograd.Function):
    """Wait for the exchange _AllToAllStart issued; the backward issues the gradient's exchange (splits swapped) and leaves the wait to _AllToAllStart."""

    @staticmethod
    def forward(ctx, out, st):
        ctx.st = st
        st.pop("work").wait()
        st.pop("src", None)
        return out

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        out_splits, in_splits = st["splits"]
        st["bwd_src"] = g.contiguous()
        st["bwd_out"] = g.new_empty((sum(in_splits),) + tuple(g.shape[1:]))
        st["bwd_work"] = dist.all_to_all_single(st["bwd_out"], st["bwd_src"], in_splits, out_splits, group=st["group"], async_op=True)
        return g, None


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


def run_experts_chunked(experts, routed: list[tuple[torch.Tensor, torch.Tensor]], group) -> list[torch.Tensor]:
    """`routed`: per token chunk (rows sorted by global expert, counts [E] i32).  Returns the expert outputs per chunk in the same row order; the row
    all-to-alls of the chunks overlap each other's expert compute (module docstring)."""
    ep = dist.get_world_size(group)
    nc = len(routed)
    with torch.no_grad():
        all_counts = torch.stack([c for _, c in routed])                          # [C, E]
        recv = torch.empty_like(all_counts)
        for i in range(nc):  # one small exchange per chunk, no host involvement
            dist.all_to_all_single(recv[i], all_counts[i].contiguous(), group=group)
        recv = recv.view(nc, ep, -1)                                              # [C, ep, E / ep]
        splits = torch.stack([all_counts.view(nc, ep, -1).sum(dim=2), recv.sum(dim=2)]).tolist()  # ONE device-to-host copy: [2][C][ep]
        maps = [ops.moe_ep_row_map(recv[i].to(torch.int32).contiguous(), sum(splits[1][i])) for i in range(nc)]
    states = [dict() for _ in range(nc)]
    received = [_AllToAllStart.apply(routed[i][0], splits[1][i], splits[0][i], group, states[i]) for i in range(nc)]  # every dispatch is in flight
    combines, cstates = [], [dict() for _ in range(nc)]
    for i in range(nc):
        row_map, inv_map, local_offsets = maps[i]
        local_in = _PermuteRows.apply(_AllToAllFinish.apply(received[i], states[i]), row_map, inv_map)  # waits for chunk i only
        local_out = experts(local_in, local_offsets)
        back = _PermuteRows.apply(local_out.contiguous(), inv_map, row_map)
        combines.append(_AllToAllStart.apply(back, splits[0][i], splits[1][i], group, cstates[i]))     # travels under chunk i + 1's experts
    return [_AllToAllFinish.apply(combines[i], cstates[i]) for i in range(nc)]
