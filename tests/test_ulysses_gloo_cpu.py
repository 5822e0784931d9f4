"""CPU suite, gloo world_size 2 and 4: the Ulysses exchange (prime_rl_amd/trainer/ulysses.py) — sequence-sharded q/k/v -> all-to-all ->
varlen attention over the FULL packed row for a slice of the heads -> all-to-all back — equals attention on the unsharded tensors, forward
and backward, including the GQA case with fewer KV heads than cp ranks (KV replication, gradients summed over replicas).
Contract: src/prime_rl/trainer/models/layers/ulysses_attn.py:72-170."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import grpo_oracle as O

SEQ_LENS = [40, 24, 56, 8]  # packed documents; boundaries fall INSIDE cp shards (128 tokens / cp)
HQ, D = 8, 16


def _free_port():
    from tests._ports import free_port

    return free_port()


def _full(hkv, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = sum(SEQ_LENS)
    q, k, v = torch.randn(n, HQ, D, generator=g), torch.randn(n, hkv, D, generator=g), torch.randn(n, hkv, D, generator=g)
    go = torch.randn(n, HQ, D, generator=g)
    return q, k, v, go


def _worker(rank, world, port, hkv, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.trainer import ulysses as U

    q, k, v, go = _full(hkv)
    ql, kl, vl, gol = (U.shard_for_cp(t, rank, world).clone() for t in (q, k, v, go))
    # forward exchange
    kx, vx = (U.replicate_kv_heads(t, world) for t in (kl, vl)) if hkv < world else (kl, vl)
    qh, kh, vh = U.seq_to_head(ql, world, None), U.seq_to_head(kx, world, None), U.seq_to_head(vx, world, None)
    assert qh.shape == (q.shape[0], HQ // world, D) and kh.shape[1] == max(hkv // world, 1)
    qh, kh, vh = (t.requires_grad_(True) for t in (qh, kh, vh))
    oh = O.varlen_causal_attention(qh, kh, vh, SEQ_LENS, D**-0.5)  # full sequence, my heads
    ol = U.head_to_seq(oh.detach(), world, None)
    # backward exchange: the inverse functions applied to the gradients
    goh = U.seq_to_head(gol, world, None)
    gqh, gkh, gvh = torch.autograd.grad(oh, (qh, kh, vh), goh)
    gql, gkl, gvl = U.head_to_seq(gqh, world, None), U.head_to_seq(gkh, world, None), U.head_to_seq(gvh, world, None)
    if hkv < world:
        gkl, gvl = U.reduce_kv_replicas(gkl, hkv), U.reduce_kv_replicas(gvl, hkv)
    lp = U.all_gather_tokens(torch.arange(4, dtype=torch.float32) + 10 * rank, world, None)
    torch.save(dict(o=ol, gq=gql, gk=gkl, gv=gvl, lp=lp), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hkv", [(2, 2), (4, 4), (4, 2)])
def test_ulysses_exchange_equals_unsharded_attention(tmp_path, world, hkv):
    mp.start_processes(_worker, args=(world, _free_port(), hkv, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    q, k, v, go = _full(hkv)
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))
    o = O.varlen_causal_attention(q, k, v, SEQ_LENS, D**-0.5)
    gq, gk, gv = torch.autograd.grad(o, (q, k, v), go)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    for name, want in (("o", o.detach()), ("gq", gq), ("gk", gk), ("gv", gv)):
        got = torch.cat([x[name] for x in r])
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5, msg=lambda m: f"{name}: {m}")
    want_lp = torch.cat([torch.arange(4, dtype=torch.float32) + 10 * i for i in range(world)])
    assert all(torch.equal(x["lp"], want_lp) for x in r)


def test_shard_for_cp_rejects_uneven_lengths():
    from prime_rl_amd.trainer.ulysses import shard_for_cp

    with pytest.raises(ValueError, match="divisible"):
        shard_for_cp(torch.zeros(10), 0, 4)


def _prefix_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.trainer import ulysses as U

    n_loc = 6
    g = torch.Generator().manual_seed(100 + rank)
    part = torch.randn((rank + 1) * n_loc, 2, 4, generator=g).to(torch.bfloat16)  # this rank's contribution to the kv rows of ranks 0..rank
    recv = U.exchange_prefix_chunks(part, rank, world, n_loc, None)
    assert recv.shape[0] == (world - rank) * n_loc
    mine = U.sum_prefix_chunks(recv, n_loc)
    # what the reference's form does (ring_attn.py:252-262): reduce-scatter of zero-padded [world * n_loc] buffers — here in f32 so the comparison is exact
    full = torch.zeros(world * n_loc, 2, 4)
    full[: (rank + 1) * n_loc] = part.float()
    out = torch.empty(n_loc, 2, 4)
    dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM)
    torch.save(dict(mine=mine, ref=out), os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_prefix_chunk_exchange_equals_the_padded_reduce_scatter(tmp_path, world):
    """cp_style "ring" backward: the ragged all-to-all that sends each peer only the dK / dV chunk it owns equals the reference's reduce-scatter over
    zero-padded buffers (sum in f32 in rank order, rounded once to bf16)."""
    mp.start_processes(_prefix_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for i in range(world):
        r = torch.load(tmp_path / f"p{i}.pt")
        assert torch.equal(r["mine"], r["ref"].to(torch.bfloat16)), i
