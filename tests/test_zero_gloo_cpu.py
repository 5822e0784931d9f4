"""CPU suite, world_size 2 over gloo: the data-parallel path of prime_rl_amd/trainer/zero.py (reduce-scatter of f32 gradient
buckets, global-norm clip, sharded AdamW, in-place bf16 all-gather, token-count all-reduce) gives the same parameters on every
rank as a single-process update with the summed gradient.  The arithmetic hooks are CPU stand-ins built on the oracle; the
collective / sharding code under test is the one the GPU engine runs over RCCL."""

import math
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import grpo_oracle as O


class CpuKernels:
    """CPU stand-ins with the signatures of prime_rl_amd.ops.{sumsq_accumulate, clip_coef, adamw_step}."""

    @staticmethod
    def sumsq_accumulate(x, out):
        out += x.double().pow(2).sum().float()

    @staticmethod
    def clip_coef(total_sumsq, grad_scale, max_norm):
        norm = total_sumsq.sqrt() * grad_scale
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0) if max_norm > 0 else torch.ones(1)
        return coef, norm

    @staticmethod
    def adamw_step(master, m, v, grad, param_bf16, lr, b1, b2, eps, wd, step, grad_scale, clip, zero_grad=False):
        g = (grad * grad_scale * clip).numpy()
        p, mm, vv = O.adamw_step(master.numpy(), m.numpy(), v.numpy(), g, lr, b1, b2, eps, wd, step)
        master.copy_(torch.from_numpy(p)); m.copy_(torch.from_numpy(mm)); v.copy_(torch.from_numpy(vv))
        param_bf16.copy_(master.to(torch.bfloat16))


def _free_port():
    from tests._ports import free_port

    return free_port()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.trainer import zero
    from prime_rl_amd.trainer.engine import ModelDims, OptimConfig, ParamStore

    dims = ModelDims.tiny(layers=2, vocab=320)
    store = ParamStore(dims, "cpu", dp_rank=rank, dp_size=world)
    store.random_init(seed=0)
    init = store.params.clone()
    g = torch.Generator().manual_seed(100 + rank)
    store.main_grad.copy_(torch.randn(store.total, generator=g) * 0.01)
    # keep the padding between parameters gradient-free, as the engine does
    keep = torch.zeros(store.total, dtype=torch.bool)
    for b in store.buckets:
        for _, off, shape in b.params:
            keep[off:off + math.prod(shape)] = True
    store.main_grad[~keep] = 0
    my_grad = store.main_grad.clone()
    for b in reversed(store.buckets):  # backward order
        zero.reduce_scatter_bucket(store, b, None, None)
    optim = OptimConfig(lr=1e-2, max_norm=0.05)
    norm = zero.sharded_optimizer_step(store, optim, 1, None, CpuKernels())
    counts = zero.all_reduce_counts((100 + rank, 0, 7 * rank), None, "cpu")
    torch.save(dict(init=init, grad=my_grad, params=store.params.clone(), norm=norm.clone(), counts=counts, master=store.master.clone(),
                    shard=[store.shard_slices(b) for b in store.buckets]), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_two_ranks_match_single_process(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=False) for i in range(world)]
    assert torch.equal(r[0]["init"], r[1]["init"]), "random init must be identical on every rank"
    assert torch.equal(r[0]["params"], r[1]["params"]), "all-gathered parameters differ between ranks"
    assert r[0]["counts"] == r[1]["counts"] == (201, 1, 7)
    # single-process reference: summed gradient (loss is pre-divided by the GLOBAL token count -> SUM, train.py:557-558)
    gsum = (r[0]["grad"] + r[1]["grad"]).numpy()
    total, coef = O.clip_grad_norm([gsum], 0.05)
    assert abs(r[0]["norm"].item() - total) / total < 1e-5 and torch.equal(r[0]["norm"], r[1]["norm"])
    p0 = r[0]["init"].float().numpy()
    p1, _, _ = O.adamw_step(p0, np.zeros_like(p0), np.zeros_like(p0), (gsum * np.float32(coef)).astype(np.float32), 1e-2, 0.9, 0.999, 1e-8, 0.01, 1)
    want = torch.from_numpy(p1).to(torch.bfloat16)
    diff = (r[0]["params"].float() - want.float()).abs().max().item()
    assert diff <= 2e-3, diff  # identical up to bf16 rounding of an identical f32 update (clip coefficient in f32 vs f64)
    assert (r[0]["params"] != r[0]["init"]).float().mean() > 0.5, "parameters did not move"
    # shards tile each bucket exactly once
    for i in range(len(r[0]["shard"])):
        (f0, l0), (f1, l1) = r[0]["shard"][i], r[1]["shard"][i]
        assert f0.stop == f1.start and (f0.stop - f0.start) == (f1.stop - f1.start) == (l0.stop - l0.start)


def _worker_other_optimizers(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tests.cpu_standin_ops as standin
    from prime_rl_amd.trainer import zero
    from prime_rl_amd.trainer.engine import ModelDims, OptimConfig, ParamStore

    res = {}
    for kind in ("sgd", "sign_sgd"):
        store = ParamStore(ModelDims.tiny(layers=1, vocab=320), "cpu", dp_rank=rank, dp_size=world, optimizer=kind)
        assert store.exp_avg_sq.numel() == 0 and store.exp_avg.numel() == (store.shard_total if kind == "sgd" else 0)
        store.random_init(seed=0)
        init = store.master.clone()
        optim = OptimConfig(lr=1e-2, weight_decay=0.0, max_norm=0.0, type=kind, momentum=0.9, nesterov=True)
        grads = []
        for step in (1, 2):
            g = torch.Generator().manual_seed(10 * step + rank)
            store.main_grad.copy_(torch.randn(store.total, generator=g) * 0.01)
            grads.append(store.main_grad.clone())
            for b in reversed(store.buckets):
                zero.reduce_scatter_bucket(store, b, None, None)
            zero.sharded_optimizer_step(store, optim, step, None, standin)
        res[kind] = dict(init=init, grads=grads, master=store.master.clone(), params=store.params.clone(), slices=[store.shard_slices(b) for b in store.buckets])
    torch.save(res, os.path.join(out_dir, f"opt{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_with_sgd_and_sign_sgd(tmp_path):
    """`optim.type = "sgd" | "sign_sgd"` under ZeRO-2 at world 2 (zero.apply_update on shard slices, momentum buffer sharded like the master weights, no
    second moment allocated): every rank's master shard after two steps is the single-process rule applied to the summed gradients."""
    world = 2
    mp.start_processes(_worker_other_optimizers, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"opt{i}.pt", weights_only=False) for i in range(world)]
    for kind in ("sgd", "sign_sgd"):
        assert torch.equal(r[0][kind]["params"], r[1][kind]["params"])
        g1 = r[0][kind]["grads"][0] + r[1][kind]["grads"][0]
        g2 = r[0][kind]["grads"][1] + r[1][kind]["grads"][1]
        for rank in range(world):
            for full, local in r[rank][kind]["slices"]:
                p0 = r[rank][kind]["init"][local]
                a, b = g1[full], g2[full]
                if kind == "sgd":  # Nesterov, momentum 0.9: buf1 = g1, step1 = g1 + 0.9 g1; buf2 = 0.9 g1 + g2, step2 = g2 + 0.9 buf2
                    want = p0 - 1e-2 * (1.9 * a) - 1e-2 * (b + 0.9 * (0.9 * a + b))
                else:
                    want = p0 - 1e-2 * torch.sign(a) - 1e-2 * torch.sign(b)
                torch.testing.assert_close(r[rank][kind]["master"][local], want, rtol=1e-5, atol=1e-7)
