"""Ulysses context parallelism for the packed scoring forward — SURVEY.md section 8 row f4 (BASELINE config 5: 32k-token agentic rollouts,
context-parallel over the 8 GPUs of a node).

What the reference does (src/prime_rl/trainer/models/layers/ulysses_attn.py:72-170, utils/cp.py:97-190, trainer/rl/train.py:385-448): every
rank of a cp group receives the SAME packed row; input_ids / position_ids / labels / temperatures are cut into `cp` contiguous chunks
(`torch.chunk`, utils/cp.py:97-120) and rank r runs the whole model on chunk r — except attention, where an all-to-all swaps the sharded
axis: [S/cp, H, D] sequence-sharded -> [S, H/cp, D] head-sharded, so an unmodified varlen kernel sees the FULL sequence (full cu_seqlens)
for H/cp heads, and a second all-to-all swaps back.  KV heads are replicated first when there are fewer of them than cp ranks
(ulysses_attn.py:51-70).  Log-probs / entropy are all-gathered over the cp group and the loss is evaluated on the full row
(train.py:446-448).

Here the same exchange feeds `attn_fwd_v4` / `attn_bwd_*` unchanged (they take any head count).  On MI355X the 8 GPUs of a node are a
full xGMI mesh: an all-to-all sends every pair's block over that pair's own direct link, which is why Ulysses (all-to-all) rather
than ring attention (cp sequential neighbour exchanges) is the first CP style here.  Functions are device-agnostic (gloo on CPU in
tests/test_ulysses_gloo_cpu.py) and work on plain tensors: the engine has no autograd graph, so the backward exchange is an explicit
call of the inverse function on the gradients."""

from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor


def shard_for_cp(t: Tensor, cp_rank: int, cp_size: int) -> Tensor:
    """Contiguous chunk `cp_rank` of a flat per-token tensor (utils/cp.py:97-120; uneven shards would deadlock the all-to-all)."""
    if t.shape[0] % cp_size:
        raise ValueError(f"CP requires the packed length to be divisible by cp size: {t.shape[0]} tokens, cp_size={cp_size}")
    per = t.shape[0] // cp_size
    return t[cp_rank * per:(cp_rank + 1) * per]


def replicate_kv_heads(t: Tensor, cp_size: int) -> Tensor:
    """[S_local, H_kv, D] -> [S_local, cp, D] when H_kv < cp (each KV head repeated cp / H_kv times: ulysses_attn.py:51-70)."""
    h = t.shape[1]
    if cp_size % h:
        raise ValueError(f"num_key_value_heads ({h}) must divide cp_size ({cp_size}) for the KV-replication path")
    return t.repeat_interleave(cp_size // h, dim=1)


def reduce_kv_replicas(g: Tensor, n_kv_heads: int) -> Tensor:
    """Backward of `replicate_kv_heads`: [S_local, cp, D] -> [S_local, H_kv, D], summing the replicas of each head in f32."""
    s, c, d = g.shape
    return g.float().view(s, n_kv_heads, c // n_kv_heads, d).sum(2).to(g.dtype)


def seq_to_head(t: Tensor, cp_size: int, group) -> Tensor:
    """[S_local, H, D] sequence-sharded -> [S_global, H / cp, D] head-sharded (rank r keeps heads [r H/cp, (r+1) H/cp))."""
    s_local, h, d = t.shape
    if h % cp_size:
        raise ValueError(f"num_heads ({h}) must be divisible by cp_size ({cp_size}); replicate KV heads first")
    h_local = h // cp_size
    send = t.reshape(s_local, cp_size, h_local, d).transpose(0, 1).contiguous()  # block p = my tokens, rank p's heads
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)  # block i = rank i's tokens, my heads; ranks hold consecutive token chunks
    return recv.reshape(cp_size * s_local, h_local, d)


def head_to_seq(t: Tensor, cp_size: int, group) -> Tensor:
    """Inverse of `seq_to_head`: [S_global, H / cp, D] -> [S_local, H, D]."""
    s_global, h_local, d = t.shape
    if s_global % cp_size:
        raise ValueError("global sequence length must be divisible by cp_size")
    s_local = s_global // cp_size
    send = t.reshape(cp_size, s_local, h_local, d).contiguous()  # block p = rank p's tokens, my heads
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)  # block i = my tokens, rank i's heads
    return recv.transpose(0, 1).contiguous().reshape(s_local, cp_size * h_local, d)


def all_gather_tokens(t: Tensor, cp_size: int, group) -> Tensor:
    """Per-token vector sharded over the cp group -> the full vector on every rank (gather_for_cp, utils/cp.py:129-139)."""
    out = torch.empty((cp_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def exchange_prefix_chunks(part: Tensor, cp_rank: int, cp_size: int, n_loc: int, group) -> Tensor:
    """The dK / dV exchange of `cp_style = "ring"` (gathered K / V) without padding.  Rank r attended its q rows against the kv prefix [0, (r + 1) n_loc), so
    `part` [(r + 1) n_loc, ...] holds its contribution to the rows of ranks 0..r and NOTHING for later ranks: chunk j goes to rank j, and rank r receives
    chunk r of every rank >= r.  One all-to-all with ragged splits (each pair's block on its own xGMI link) moves r chunks per rank instead of the
    cp chunks of a reduce-scatter over zero-padded [cp n_loc] buffers (the reference: ring_attn.py:252-262).  Returns [(cp - r) n_loc, ...]: block i = rank
    r + i's contribution to MY rows; `sum_prefix_chunks` adds the blocks in rank order."""
    if part.shape[0] != (cp_rank + 1) * n_loc:
        raise ValueError(f"rank {cp_rank}: expected {(cp_rank + 1) * n_loc} prefix rows, got {part.shape[0]}")
    in_splits = [n_loc] * (cp_rank + 1) + [0] * (cp_size - 1 - cp_rank)
    out_splits = [0] * cp_rank + [n_loc] * (cp_size - cp_rank)
    recv = part.new_empty(((cp_size - cp_rank) * n_loc,) + tuple(part.shape[1:]))
    dist.all_to_all_single(recv, part.contiguous(), out_splits, in_splits, group=group)
    return recv


def sum_prefix_chunks(recv: Tensor, n_loc: int) -> Tensor:
    """[(blocks) n_loc, ...] -> [n_loc, ...]: the blocks summed in f32 in rank order, rounded once (deterministic; a bf16 ring reduction rounds per hop)."""
    blocks = recv.shape[0] // n_loc
    if blocks == 1:
        return recv
    return recv.float().view((blocks, n_loc) + tuple(recv.shape[1:])).sum(0).to(recv.dtype)
