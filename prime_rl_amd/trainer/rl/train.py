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
import math
import os
import time
from dataclasses import asdict, dataclass, field
from pathlib import Path
from types import SimpleNamespace

import torch
import torch.distributed as dist

from ... import _lib
from ...transports.weights import setup_weight_broadcast
from ..engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches
from ..stats import STAT_SUFFIXES, Tensors, collect_step_tensors
from .data import DataLoader, FakeDataLoader, FakeDataLoaderConfig

PEAK_BF16_FLOPS = 2.5e15  # MI355X dense bf16 MFMA (the reference's table stops at MI325X and would fall back to A100, perf.py:99-109)


@dataclass
class WeightBroadcastConfig:
    type: str = "filesystem"  # "filesystem" | "nccl" (trainer/configs: standalone trainer defaults to filesystem, configs/trainer.py:617)


@dataclass
class TrainerConfig:
    model: str = "qwen3-8b"  # qwen3-8b | qwen3-0.6b | tiny
    seq_len: int = 2048
    max_steps: int | None = 4
    output_dir: str = "outputs"
    fake_data: FakeDataLoaderConfig | None = field(default_factory=lambda: FakeDataLoaderConfig(batch_size=8))
    loss: LossConfig = field(default_factory=LossConfig)
    optim: OptimConfig = field(default_factory=OptimConfig)
    weight_broadcast: WeightBroadcastConfig | None = field(default_factory=WeightBroadcastConfig)
    tokens_per_pass: int = 16384
    seed: int = 0
    rollout_format: str = "msgpack"  # "msgpack" = the reference wire; "binary" = raw-array step blobs + pinned read-ahead (transports/rollouts/binary.py)


def train(config: TrainerConfig) -> list[dict]:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend="nccl", device_id=device)
    is_master = rank == 0
    dims = {"qwen3-8b": ModelDims.qwen3_8b, "qwen3-0.6b": ModelDims.qwen3_0_6b, "tiny": ModelDims.tiny}[config.model]()
    engine = GRPOEngine(dims, device=device, optim=config.optim, loss=config.loss, seed=config.seed)
    out_dir = Path(config.output_dir)
    if config.fake_data is not None:
        loader = FakeDataLoader(config.fake_data, config.seq_len, dp_world_size=world, dp_rank=rank)
    else:
        loader = DataLoader(out_dir, start_step=1, dp_world_size=world, dp_rank=rank, config=SimpleNamespace(type="filesystem", format=config.rollout_format))
    wb = setup_weight_broadcast(out_dir, config.weight_broadcast, is_master=is_master) if config.weight_broadcast is not None else None
    per_pass = max(config.tokens_per_pass // config.seq_len, 1)
    history: list[dict] = []
    step, start_step = 1, 1
    metrics_path = out_dir / "metrics.jsonl"
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
        if config.fake_data is not None and dims.vocab_size < 120000:  # FakeDataLoader draws ids up to 120000 (data.py:101)
            for mb in mbs:
                mb["input_ids"] = mb["input_ids"] % dims.vocab_size
        if packed is not None and len(mbs) <= per_pass:
            passes = [packed]
        else:
            passes = [pack_micro_batches(mbs[i:i + per_pass], device) for i in range(0, len(mbs), per_pass)]
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
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
        if wb is not None and not (config.max_steps is not None and step == config.max_steps and getattr(config.weight_broadcast, "type", "") == "nccl"):
            t0 = time.perf_counter()
            wb.broadcast_weights(engine, step)  # pi_step -> broadcasts/step_{step}; the final nccl push is skipped (train.py:582-587)
            t_bcast += time.perf_counter() - t0
        n_tokens = sum(p.n_tokens for p in passes) * world
        thr = n_tokens / t_fb
        mfu = thr * dims.flops_per_token(config.seq_len) / (world * PEAK_BF16_FLOPS) * 100
        peak = torch.cuda.max_memory_allocated() / 2**30
        rec.update({"perf/throughput": thr, "perf/throughput_per_gpu": thr / world, "perf/mfu": mfu, "perf/peak_memory": peak, "optim/lr": config.optim.lr,
                    "optim/grad_norm": grad_norm, "time/step": time.perf_counter() - t_step, "time/wait_for_batch": t_wait, "time/load_data": t_load,
                    "time/broadcast_weights": t_bcast, "time/forward_backward": t_fb,
                    })
        if rec.get("entropy/all/mean", 0.0) > 0 and "mismatch_kl/all/mean" in rec:  # train.py:669-672
            rec["kl_ent_ratio/mean"] = rec["mismatch_kl/all/mean"] / rec["entropy/all/mean"]
        if is_master:
            print(f"Step {step} | Time: {rec['time/step']:.2f}s | Loss: {rec['loss/mean']:.4f} | Entropy: {rec['entropy/all/mean']:.4f} | "
                  f"Mismatch KL: {rec['mismatch_kl/all/mean']:.4f} | Grad. Norm: {grad_norm:.4f} | LR: {config.optim.lr:.2e} | "
                  f"Throughput: {thr:.0f} tokens/s | MFU: {mfu:.1f}% | Peak Mem.: {peak:.1f} GiB", flush=True)
            with open(metrics_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        history.append(rec)
        step += 1
    return history


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--max-steps", type=int, default=4)
    ap.add_argument("--micro-batches", type=int, default=8, help="per GPU")
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--output-dir", default="outputs")
    a = ap.parse_args()
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    train(TrainerConfig(model=a.model, seq_len=a.seq_len, max_steps=a.max_steps, output_dir=a.output_dir,
                        fake_data=FakeDataLoaderConfig(batch_size=a.micro_batches * ws)))
