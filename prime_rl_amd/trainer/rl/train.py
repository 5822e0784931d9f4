"""The GRPO trainer loop — mirrors `prime_rl.trainer.rl.train.train(config)` (src/prime_rl/trainer/rl/train.py:75-743) for the
dense-model / data-parallel path, on the MI355X engine.

Same step and version indexing as the reference: at start the incoming policy `v{start-1}` is broadcast
(train.py:271-277); step `n` consumes batch `n`, produces `pi_n` and broadcasts it as `broadcasts/step_n` (train.py:575-594).
Same observable surface (section 9.5): console line `Step N | Time | Loss | Entropy | Mismatch KL | Grad. Norm | LR |
Throughput | MFU | Peak Mem.` and metric keys `perf/*`, `optim/*`, `time/*`, `{loss, entropy/all, mismatch_kl/all,
masked_mismatch_kl, ...}/{mean,median,std,min,max}` with the reference's `Tensors.compute_stats` semantics (trainer/utils.py:238-281:
all-rank gather, unbiased std, NaN for empty keys).  Step statistics are reduced ON the GPU and fetched with one D2H copy per
step (the reference moves per-token vectors to the host and groups them in Python, train.py:483-539: next-row f3).

Launch: `python -m prime_rl_amd.trainer.rl.train` (1 GPU) or `torchrun --nproc-per-node N -m prime_rl_amd.trainer.rl.train`.
"""

from __future__ import annotations

import json
import os
import time
from datetime import timedelta
from pathlib import Path

import torch
import torch.distributed as dist

from ... import _lib
from ...configs.trainer import TrainerConfig, parse_argv
from ...transports.weights import setup_weight_broadcast
from ..engine import pack_micro_batches
from ..model import setup_engine, setup_module_engine, uses_module_engine
from ..parallel_dims import get_parallel_dims
from ..stats import STAT_SUFFIXES, Tensors, collect_step_tensors
from .data import DataLoader, FakeDataLoader

PEAK_BF16_FLOPS = 2.5e15  # MI355X dense bf16 MFMA (the reference's table stops at MI325X and would fall back to A100, perf.py:99-109)


def setup_scheduler(config: TrainerConfig):
    """LR schedule exactly as trainer/scheduler.py builds it — the same torch.optim.lr_scheduler objects on a one-parameter stand-in
    optimizer (the fused AdamW kernel takes `lr` as an argument per step); returns a callable step -> lr for 1-based steps."""
    from torch.optim.lr_scheduler import ConstantLR, CosineAnnealingLR, LinearLR, SequentialLR

    lr, sc, max_steps = config.optim.lr, config.scheduler, config.max_steps
    opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=lr)
    if sc.type == "constant":
        sched = ConstantLR(opt, factor=1.0)
    else:
        floor = sc.min_lr / lr if sc.min_lr > 0 else 1e-8
        parts, milestones = [], []
        if sc.warmup_steps > 0:
            parts.append(LinearLR(opt, start_factor=floor, end_factor=1.0, total_iters=sc.warmup_steps))
            milestones.append(sc.warmup_steps)
        if sc.type == "linear":
            if sc.decay_steps > 0:
                start = max_steps - sc.decay_steps
                parts.append(LinearLR(opt, start_factor=1.0, end_factor=1.0, total_iters=start - sc.warmup_steps))
                parts.append(LinearLR(opt, start_factor=1.0, end_factor=floor, total_iters=sc.decay_steps - 1))
                milestones.append(start)
        else:
            parts.append(CosineAnnealingLR(opt, T_max=max_steps - sc.warmup_steps, eta_min=sc.min_lr))
        sched = parts[0] if len(parts) == 1 else SequentialLR(opt, parts, milestones=milestones)
    cache: list[float] = [sched.get_last_lr()[0]]

    def lr_at(step: int) -> float:
        while len(cache) < step:
            opt.step()
            sched.step()
            cache.append(sched.get_last_lr()[0])
        return cache[step - 1]

    return lr_at


def train(config: TrainerConfig) -> list[dict]:
    """`prime_rl.trainer.rl.train.train(config)` (train.py:75) for the dense data-parallel path: one process per GPU under torchrun
    (RANK / LOCAL_RANK / WORLD_SIZE), every rank a trainer rank."""
    config.validate_for_train_loop()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = int(os.environ.get("WORLD_SIZE", "1"))
    # PRIME_TRAINER_SHARED_GPU=1: rehearsal of a multi-rank launch on a ONE-GPU box — every rank uses device 0 and the trainer group runs over gloo
    # (RCCL refuses two ranks on one device); same loop, same engines, same collectives (tests/test_multirank_one_gpu.py)
    shared_gpu = os.environ.get("PRIME_TRAINER_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    torch.set_float32_matmul_precision(config.matmul_precision)  # train.py:117-120
    wbc = config.weight_broadcast
    trainer_world = launched
    if launched > 1 and not dist.is_initialized():
        if shared_gpu:
            dist.init_process_group(backend="gloo", timeout=timedelta(seconds=config.dist_timeout_seconds))
        else:
            dist.init_process_group(backend="nccl", device_id=device, timeout=timedelta(seconds=config.dist_timeout_seconds))
    is_master = rank == 0
    # the torchrun world holds trainer ranks only; the inference GPUs are reached through a SEPARATE raw RCCL communicator bootstrapped at
    # weight_broadcast.host:port (below), as in the reference (transports/weights/nccl.py:128-136) — ZeRO shards can never span them
    parallel_dims = get_parallel_dims(config.model, config.model.seq_len, world_size=trainer_world)
    # dense models: the flat-buffer engine; MoE families / expert parallelism: the same step on the nn.Module tree (trainer/module_engine.py)
    engine = (setup_module_engine if uses_module_engine(config, parallel_dims) else setup_engine)(config, parallel_dims, device=device)
    dims = engine.dims
    dp_rank, dp_world = parallel_dims.dp_rank_of(rank), parallel_dims.dp_size
    out_dir = Path(config.output_dir)
    seq_len = config.model.seq_len
    if config.data.fake is not None:
        loader = FakeDataLoader(config.data.fake, seq_len, dp_world_size=dp_world, dp_rank=dp_rank)
    else:
        loader = DataLoader(out_dir, start_step=1, dp_world_size=dp_world, dp_rank=dp_rank, config=config.rollout_transport)
    wb = None
    if wbc is not None:
        if wbc.type == "nccl":
            from ...transports.weights import init_push_communicator

            # "sharded" = the multi-source push (dense engine: 1 / T of every flat bucket per trainer rank; nn.Module engines: named tensors, the expert
            # shards straight from the ranks that own them); "reference" = the reference's single-source byte stream for a receiver that speaks only that
            protocol = os.environ.get("PRIME_PUSH_PROTOCOL", "sharded")
            comm = init_push_communicator(wbc.host, wbc.port, rank, trainer_world, wbc.inference_world_size, device, wbc.timeout, protocol)
            wb = setup_weight_broadcast(out_dir, wbc, comm=comm, trainer_rank=rank, n_trainers=trainer_world, n_inference=wbc.inference_world_size,
                                        protocol=protocol, ready_timeout=wbc.timeout, trainer_group=parallel_dims.get_group("dp_cp"))
        else:
            wb = setup_weight_broadcast(out_dir, wbc, is_master=is_master)
    lr_at = setup_scheduler(config)
    per_pass = max(config.tokens_per_pass // seq_len, 1)
    history: list[dict] = []
    step, start_step = 1, 1
    mon = (config.monitors.get("file") or {}).get("path") if isinstance(config.monitors, dict) else None
    metrics_path = out_dir / (mon or "metrics.jsonl")
    if is_master:
        out_dir.mkdir(parents=True, exist_ok=True)
    while config.max_steps is None or step <= config.max_steps:
        torch.cuda.reset_peak_memory_stats()
        t_step = time.perf_counter()
        t_bcast = 0.0
        if step == start_step and wb is not None:  # startup policy v{start-1} (train.py:271-277)
            t0 = time.perf_counter()
            wb.broadcast_weights(engine, step - 1)
            t_bcast += time.perf_counter() - t0
        t0 = time.perf_counter()
        loader.wait_for_batch()
        t_wait = time.perf_counter() - t0
        t0 = time.perf_counter()
        packed = None
        if getattr(loader, "binary", False):
            packed, mbs = loader.get_packed_batch(device)  # one H2D copy of the step blob; `mbs` is the host view used for the env split
        else:
            mbs = loader.get_batch()
        if config.data.fake is not None and dims.vocab_size < 120000:  # FakeDataLoader draws ids up to 120000 (data.py:101)
            for mb in mbs:
                mb["input_ids"] = mb["input_ids"] % dims.vocab_size
        if packed is not None and len(mbs) <= per_pass:
            passes = [packed]
        else:
            passes = [pack_micro_batches(mbs[i:i + per_pass], device) for i in range(0, len(mbs), per_pass)]
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        engine.optim.lr = lr_at(step)
        out = engine.step(passes)
        # --- step statistics (trainer/stats.py): per-token vectors stay on the device, gathered over all ranks, ONE D2H copy ------
        tensors = Tensors(group=engine.dp_group)
        env_ids: dict[str, int] = {}
        for pi, (p, lp, en) in enumerate(zip(passes, out["logprobs"], out["entropy"])):
            # per-env split (train.py:486-494) with integer env ids: one id per SAMPLE on the host (env_names is constant within a
            # sample), expanded to tokens on the device — no per-token Python
            sample_env = []
            for mb in mbs[pi * per_pass:(pi + 1) * per_pass]:
                off = 0
                for sl in mb["sequence_lengths"]:
                    sample_env.append(env_ids.setdefault(mb["env_names"][off], len(env_ids)))
                    off += sl
            names = list(env_ids)
            collect_step_tensors(tensors, p, lp, en, out["sample_stats"][pi], out["sample_metrics"][pi], out["scales"], dims.vocab_size, config.loss.type,
                                 torch.tensor(sample_env), names, _lib.METRIC_KEYS, _lib.STAT)
        for lb in out.get("load_balance", ()):  # MoE models (train.py:531-535): one value per MoE layer and pass
            for name, v in lb.items():
                if v is not None:
                    tensors[name].append(v)
        keys, stat_dev = tensors.compute_stats_device()
        flat = torch.cat([stat_dev.reshape(-1), out["grad_norm"].reshape(1).float()]).cpu()  # ONE device-to-host copy per step
        torch.cuda.synchronize()
        t_fb = time.perf_counter() - t0
        vals = flat.tolist()
        rec: dict[str, float] = {"step": step}
        for i, name in enumerate(keys):
            for j, suffix in enumerate(STAT_SUFFIXES):
                rec[f"{name}/{suffix}"] = vals[i * 5 + j]
        grad_norm = vals[-1]
        t_transfer = None
        if wb is not None and not (config.max_steps is not None and step == config.max_steps and getattr(config.weight_broadcast, "type", "") == "nccl"):
            if hasattr(wb, "last_transfer_seconds"):
                t_transfer = wb.last_transfer_seconds()  # device time of the PREVIOUS (asynchronous) push, long finished by now
            t0 = time.perf_counter()
            wb.broadcast_weights(engine, step)  # pi_step -> broadcasts/step_{step}; the final nccl push is skipped (train.py:582-587)
            t_bcast += time.perf_counter() - t0
        n_tokens = sum(p.n_tokens for p in passes) * dp_world
        thr = n_tokens / t_fb
        world = trainer_world
        mfu = thr * dims.flops_per_token(seq_len) / (world * PEAK_BF16_FLOPS) * 100
        peak = torch.cuda.max_memory_allocated() / 2**30
        rec.update({"perf/throughput": thr, "perf/throughput_per_gpu": thr / world, "perf/mfu": mfu, "perf/peak_memory": peak, "optim/lr": engine.optim.lr,
                    "optim/grad_norm": grad_norm, "time/step": time.perf_counter() - t_step, "time/wait_for_batch": t_wait, "time/load_data": t_load,
                    "time/broadcast_weights": t_bcast, "time/forward_backward": t_fb,
                    })
        if t_transfer is not None:
            rec["time/broadcast_weights_transfer"] = t_transfer  # extension: wire time of the asynchronous push (overlapped with this step)
        if rec.get("entropy/all/mean", 0.0) > 0 and "mismatch_kl/all/mean" in rec:  # train.py:669-672
            rec["kl_ent_ratio/mean"] = rec["mismatch_kl/all/mean"] / rec["entropy/all/mean"]
        if is_master:
            print(f"Step {step} | Time: {rec['time/step']:.2f}s | Loss: {rec['loss/mean']:.4f} | Entropy: {rec['entropy/all/mean']:.4f} | "
                  f"Mismatch KL: {rec['mismatch_kl/all/mean']:.4f} | Grad. Norm: {grad_norm:.4f} | LR: {engine.optim.lr:.2e} | "
                  f"Throughput: {thr:.0f} tokens/s | MFU: {mfu:.1f}% | Peak Mem.: {peak:.1f} GiB"
                  + (f" | Max Vio {rec['max_vio/mean']:.4f}" if "max_vio/mean" in rec else ""), flush=True)  # train.py:644-645
            with open(metrics_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        history.append(rec)
        step += 1
    return history


def main(argv: list[str] | None = None) -> None:
    """`torchrun -m prime_rl_amd.trainer.rl.train @ cfg.toml --model.seq-len 2048 ...` — the reference's launch grammar
    (entrypoints/rl.py:320-335, benchmarks/scripts/run_single_benchmark.py:111-191)."""
    train(parse_argv(argv))


if __name__ == "__main__":
    main()
