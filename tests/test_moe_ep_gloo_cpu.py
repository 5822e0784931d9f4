"""Expert parallelism of the MoE layer over gloo (world 2 and 4) with CPU stand-in kernels: the UNCHANGED
`prime_rl_amd.trainer.models.layers.moe.MoE` + `trainer/distributed/expert_parallel.py` (counts all-to-all, row all-to-all, regrouping,
local experts, inverse) must give every rank exactly what the single-rank layer with all experts gives for that rank's tokens —
outputs, input gradients, and the local experts' weight gradients summed over every rank's tokens."""

from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    from tests._ports import free_port

    return free_port()


def _patch_ops():
    sys.path.insert(0, str(ROOT))
    from prime_rl_amd import ops
    from tests import cpu_standin_ops as S

    for name in ("moe_router_fwd", "moe_router_bwd", "moe_sort", "moe_gather_rows", "moe_combine", "moe_slot_grads", "moe_ep_row_map"):
        setattr(ops, name, getattr(S, name))


def _make(world_experts, dim, hidden, k, ep_group, seed):
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    args = MoEArgs(num_experts=world_experts, num_shared_experts=1, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False,
                   top_k=k, use_grouped_mm=False, load_balance_coeff=None)
    torch.manual_seed(seed)
    full = MoE(args, dim, hidden)  # every rank builds the SAME full layer first (same seed), then keeps its expert slice
    for p in full.parameters():
        torch.nn.init.normal_(p, std=0.3)
    full = full.to(torch.bfloat16)
    if ep_group is None:
        return full, full
    local = MoE(args, dim, hidden, ep_group=ep_group).to(torch.bfloat16)
    ep, r = dist.get_world_size(ep_group), dist.get_rank(ep_group)
    le = world_experts // ep
    sd = {k_: v for k_, v in full.state_dict().items() if not k_.startswith("experts.")}
    for w in ("w1", "w2", "w3"):
        sd[f"experts.{w}"] = getattr(full.experts, w).detach()[r * le:(r + 1) * le].clone()
    local.load_state_dict(sd)
    return local, full


def _worker(rank, world, port, n_experts, chunks=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PM_EP_OVERLAP_CHUNKS=str(chunks))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # small tensors: a full thread pool per worker only oversubscribes the host
    try:
        _patch_ops()
        dim, hidden, k = 32, 16, 2
        local, full = _make(n_experts, dim, hidden, k, dist.group.WORLD, seed=7)
        g = torch.Generator().manual_seed(100 + rank)
        n = 24 + 5 * rank  # ragged token counts across ranks
        x = torch.randn(1, n, dim, generator=g).to(torch.bfloat16)
        gy = torch.randn(1, n, dim, generator=g).to(torch.bfloat16)
        xa = x.clone().requires_grad_(True)
        ya = local(xa)
        ya.backward(gy)
        xb = x.clone().requires_grad_(True)
        yb = full(xb)  # single-rank layer with all experts on this rank's tokens (same stand-in kernels, no communication)
        yb.backward(gy)
        assert torch.equal(ya, yb), f"rank {rank}: outputs differ by {(ya.float() - yb.float()).abs().max()}"
        assert torch.equal(xa.grad, xb.grad), f"rank {rank}: input gradients differ"
        if chunks == 1:
            assert torch.equal(local.router.gate.weight.grad, full.router.gate.weight.grad)
        else:  # the pieces' gate gradients are summed in bf16 by autograd: one rounding per piece instead of one
            ga, gb = local.router.gate.weight.grad.float(), full.router.gate.weight.grad.float()
            assert (ga - gb).abs().max() <= 2.0 ** -6 * gb.abs().max(), f"rank {rank}: gate gradient {(ga - gb).abs().max()} of {gb.abs().max()}"
        assert torch.equal(local.tokens_per_expert, full.tokens_per_expert)
        # expert weight gradients: the local slice holds the sum over EVERY rank's tokens; the full layer only this rank's -> all-reduce it
        le = n_experts // world
        for w in ("w1", "w2", "w3"):
            ref = getattr(full.experts, w).grad.float()
            dist.all_reduce(ref)
            got = getattr(local.experts, w).grad.float()
            want = ref[rank * le:(rank + 1) * le]
            # each side rounds to bf16 at a different point of the sum over ranks: one bf16 ulp of the largest entry
            # (every further piece of a chunked exchange adds one bf16 rounding of the running `.grad` sum)
            assert (got - want).abs().max() <= 2.0 ** -7 * max(1, chunks / 4) * want.abs().max(), f"rank {rank} {w}: {(got - want).abs().max()} of {want.abs().max()}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_experts", [(2, 8), (4, 8), (2, 2)])
def test_expert_parallel_layer_equals_single_rank_layer(world, n_experts):
    mp.spawn(_worker, args=(world, _free_port(), n_experts), nprocs=world, join=True)


@pytest.mark.parametrize("world,n_experts,chunks", [(2, 8, 2), (2, 8, 3), (4, 8, 2), (2, 2, 40)])
def test_expert_parallel_with_overlapped_chunked_exchange_equals_single_rank_layer(world, n_experts, chunks):
    """PM_EP_OVERLAP_CHUNKS > 1 (the default on a GPU): the layer's tokens are cut into pieces whose dispatch / combine all-to-alls are issued
    asynchronously and waited for right before their consumers, forward and backward (EP.run_experts_chunked).  Same outputs and input gradients bit for
    bit, same expert gradients up to the bf16 sum over pieces; 40 pieces over 24-29 tokens: pieces without tokens still take part in every collective."""
    mp.spawn(_worker, args=(world, _free_port(), n_experts, chunks), nprocs=world, join=True)


def _groups_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # small tensors: a full thread pool per worker only oversubscribes the host
    try:
        sys.path.insert(0, str(ROOT))
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=1, pp=1, ep=2, world_size=world)
        assert pd.ep_enabled and pd.dp_size == world
        g = pd.get_group("ep")
        assert dist.get_world_size(g) == 2 and dist.get_process_group_ranks(g) == [rank - rank % 2, rank - rank % 2 + 1]
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, group=g)
        assert float(t) == (rank - rank % 2) * 2 + 1
    finally:
        dist.destroy_process_group()


def test_ep_process_groups_are_consecutive_rank_blocks():
    mp.spawn(_groups_worker, args=(4, _free_port()), nprocs=4, join=True)


def test_moe_model_names_resolve_and_the_dense_engine_refuses_them():
    sys.path.insert(0, str(ROOT))
    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer import model as MM

    cfg = TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen3-30B-A3B")))
    moe = MM.resolve_moe_config(cfg.model)
    assert (moe.num_experts, moe.num_experts_per_tok, moe.moe_intermediate_size, moe.hidden_size, moe.num_hidden_layers) == (128, 8, 768, 2048, 48)
    assert MM.resolve_moe_config(TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen3-8B"))).model) is None


def _patch_all_ops():
    """Every stand-in of tests/cpu_standin_ops.py replaces its namesake in prime_rl_amd.ops (worker processes only)."""
    import inspect

    sys.path.insert(0, str(ROOT))
    from prime_rl_amd import ops
    from tests import cpu_standin_ops as S

    for name, fn in inspect.getmembers(S, inspect.isfunction):
        if not name.startswith("_") and hasattr(ops, name):
            setattr(ops, name, fn)


def _model_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # small tensors: a full thread pool per worker only oversubscribes the host
    try:
        _patch_all_ops()
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        cfg = Qwen3MoeConfig(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                             num_experts=4, num_experts_per_tok=2, moe_intermediate_size=32, mlp_only_layers=[0], use_grouped_mm=False)
        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=1, pp=1, ep=2, world_size=world)
        group = pd.get_group("ep")
        torch.manual_seed(11)
        full = Qwen3MoeForCausalLM(cfg)
        for p in full.parameters():
            torch.nn.init.normal_(p, std=0.3 if p.dim() == 3 else 0.08)
        for n_, p in full.named_parameters():
            if n_.endswith("norm.weight") or n_.endswith("layernorm.weight"):
                torch.nn.init.ones_(p)
        full = full.to(torch.bfloat16)
        local = Qwen3MoeForCausalLM(cfg, ep_group=group).to(torch.bfloat16)
        er, le = dist.get_rank(group), cfg.num_experts // 2
        sd = {}
        for k_, v in full.state_dict().items():
            sd[k_] = v[er * le:(er + 1) * le].clone() if ".experts.w" in k_ else v
        local.load_state_dict(sd)
        g = torch.Generator().manual_seed(50 + rank)
        n = 40 + 8 * rank
        ids = torch.randint(0, 64, (1, n), generator=g)
        gy = torch.randn(1, n, 64, generator=g)
        outs = []
        for m_ in (local, full):
            logits = m_(input_ids=ids, position_ids=torch.arange(n)[None], seq_lens=torch.tensor([n - 12, 12]))["logits"]
            (logits.float() * gy).sum().backward()
            outs.append(logits.detach())
        assert torch.equal(outs[0], outs[1]), f"rank {rank}: logits differ by {(outs[0].float() - outs[1].float()).abs().max()}"
        for (na, pa), (nb, pb) in zip(local.named_parameters(), full.named_parameters()):
            ga, gb = pa.grad.float(), pb.grad.float()
            if ".experts.w" in na:  # local experts see every rank's tokens: compare with the all-reduced single-rank gradient
                dist.all_reduce(gb, group=group)
                gb = gb[er * le:(er + 1) * le]
                assert (ga - gb).abs().max() <= 2.0 ** -6 * gb.abs().max() + 1e-6, (rank, na)
            else:
                assert torch.equal(ga, gb), (rank, na)
    finally:
        dist.destroy_process_group()


def test_qwen3_moe_model_with_expert_parallelism_equals_the_single_rank_model():
    """The whole Qwen3-MoE autograd model (dense layer 0, MoE layer 1) with its experts sharded over ParallelDims' "ep" group (world 2) against
    the same model holding all experts, on each rank's own ragged packed batch: logits and every replicated gradient bit-identical,
    expert gradients equal to the all-reduced single-rank ones."""
    mp.spawn(_model_worker, args=(2, _free_port()), nprocs=2, join=True)
