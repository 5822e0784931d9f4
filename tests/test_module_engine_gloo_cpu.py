"""CPU suite, gloo world 2 and 4: the nn.Module trainer step (trainer/module_engine.py) of a Qwen3-MoE model under expert parallelism
against the same step of ONE process that holds every expert and consumes every rank's rollouts.

World 2, ep 2: every rank holds half of the experts; replicated parameters are all-reduced, expert gradients are complete after the dispatch
all-to-all and reduced nowhere, the gradient norm adds the ep group's expert shares.  World 4, ep 2: two ep groups — expert gradients are
summed over the `ep_replica` pairs (ranks 0+2, 1+3).  Arithmetic kernels are the CPU stand-ins (tests/cpu_standin_ops.py); the model,
the module engine, the expert-parallel dispatch and the collectives are the shipped code."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
SEQ, LR, STEPS = 48, 2e-3, 2


def _free_port():
    from tests._ports import free_port

    return free_port()


def _batches(world, rank, step):
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=2 * world, generate_samples=True), seq_len=SEQ, dp_world_size=world, dp_rank=rank)
    dl.batch_counter = step
    mbs = dl.get_batch()
    g = torch.Generator().manual_seed(77 * rank + step)
    for mb in mbs:
        mb["input_ids"] = mb["input_ids"] % 64
        mb["inference_logprobs"] = -torch.rand(1, SEQ, generator=g) * 0.1 - 4.1  # near log(1/64)
    return mbs


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # tiny tensors: more threads per worker only oversubscribe the host
    try:
        sys.path.insert(0, str(ROOT))
        from tests.test_moe_ep_gloo_cpu import _patch_all_ops

        _patch_all_ops()
        from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
        from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.module_engine import ModuleEngine
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        cfg = Qwen3MoeConfig(vocab_size=64, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                             num_experts=4, num_experts_per_tok=2, moe_intermediate_size=32, mlp_only_layers=[0], use_grouped_mm=False)
        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=1, pp=1, ep=2, world_size=world)
        ep_group, replica = pd.get_group("ep"), pd.get_group("ep_replica")
        assert (replica is None) == (world == 2)
        torch.manual_seed(11)
        full = Qwen3MoeForCausalLM(cfg)
        for p in full.parameters():
            torch.nn.init.normal_(p, std=0.3 if p.dim() == 3 else 0.08)
        for n_, p in full.named_parameters():
            if n_.endswith("norm.weight") or n_.endswith("layernorm.weight"):
                torch.nn.init.ones_(p)
        full = full.to(torch.bfloat16)
        local = Qwen3MoeForCausalLM(cfg, ep_group=ep_group).to(torch.bfloat16)
        er, le = dist.get_rank(ep_group), cfg.num_experts // 2
        local.load_state_dict({k: (v[er * le:(er + 1) * le].clone() if ".experts.w" in k else v.clone()) for k, v in full.state_dict().items()})
        inject_prime_lm_head(full, chunk_size=32)
        inject_prime_lm_head(local, chunk_size=32)
        optim = OptimConfig(lr=LR, weight_decay=0.01, max_norm=0.5)
        eng = ModuleEngine(local, "cpu", optim=optim, dp_group=pd.get_group("dp"), dp_size=world, ep_group=ep_group, ep_size=2, ep_replica_group=replica)
        ref = ModuleEngine(full, "cpu", optim=OptimConfig(lr=LR, weight_decay=0.01, max_norm=0.5))
        assert 0 < eng.n_replicated < eng.total and ref.n_replicated == ref.total and eng.collective_state_dict and not ref.collective_state_dict
        for step in range(STEPS):
            mine = _batches(world, rank, step)
            out = eng.step([pack_micro_batches(mine[:1], "cpu"), pack_micro_batches(mine[1:], "cpu")])
            everyone = [pack_micro_batches(_batches(world, r, step)[i:i + 1], "cpu") for r in range(world) for i in range(2)]
            want = ref.step(everyone)
            loss = torch.stack([x.reshape(()) for x in out["loss"]]).sum()
            dist.all_reduce(loss)
            want_loss = sum(x.item() for x in want["loss"])
            assert out["scales"] == want["scales"]
            assert abs(loss.item() - want_loss) < (2e-4 if step == 0 else 2e-3), (rank, step, loss.item(), want_loss)
            gn, wn = out["grad_norm"].item(), want["grad_norm"].item()
            assert abs(gn - wn) / wn < (2e-3 if step == 0 else 3e-2), (rank, step, gn, wn)
            # the log-probs of this rank's own rollouts: the dispatch all-to-all changes nothing (bit-identical at step 0, same weights)
            mine_ref = torch.cat(want["logprobs"][2 * rank: 2 * rank + 2])
            d = (torch.cat(out["logprobs"]) - mine_ref).abs()
            assert (d.max() == 0) if step == 0 else (d.mean() < 2e-2), (rank, step, d.max().item())
        # parameters after two optimizer steps: replicated ones track the single process, each expert shard tracks its slice
        got = dict(local.named_parameters())
        for name, p in full.named_parameters():
            w = p.detach().float()
            if ".experts.w" in name:
                w = w[er * le:(er + 1) * le]
            diff = (got[name].detach().float() - w).abs()
            assert float(diff.max()) <= 4.5 * LR and float((diff > 0).float().mean()) < 0.2, (rank, name, float(diff.max()), float((diff > 0).float().mean()))
        # the assembled HF-named state dict is the whole model on every rank (experts gathered over the ep group)
        sd, sd_ref = eng.state_dict(), ref.state_dict()
        assert set(sd) == set(sd_ref) and "model.layers.1.mlp.experts.3.up_proj.weight" in sd and "model.layers.1.mlp.gate.weight" in sd
        for k in sd:
            assert sd[k].shape == sd_ref[k].shape and float((sd[k].float() - sd_ref[k].float()).abs().max()) <= 4.5 * LR, (rank, k)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_module_engine_expert_parallel_step_equals_single_process(world):
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)
