#!/usr/bin/env python3
"""bench.py — GRPO trainer samples/sec, Qwen3-8B, 2048-token rollouts, on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full policy update on synthetic fixed-length rollouts (the reference's FakeDataLoader batches,
src/prime_rl/trainer/rl/data.py:132-159): scale all-reduce -> packed scoring forward -> log-prob gather over the vocab ->
GRPO loss -> backward (+ gradient reduce-scatter) -> clip -> AdamW (+ parameter all-gather).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (contract in the round prompt / DESIGN.md §measurement):
  value      = (micro_batches_per_gpu * N * K) / max-over-ranks wall time          [samples/s, whole job]
  roofline   = the log-prob kernel: algorithmic bytes per launch / mean launch time from HIP events
               recorded on the launch stream inside the timed region
  cpu_baseline = the oracle's CPU port of the same step (oracle/cpu_trainer.py) on the host cores, rank 0 at N=1 only
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC: RCCL needs this before HIP initialises

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_BPS = 8.0e12  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK = 2.5e15


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--micro-batches", type=int, default=8, help="2048-token micro-batches (= samples) per GPU per optimizer step")
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--model", default="qwen3-8b", choices=["qwen3-8b", "qwen3-0.6b", "tiny"])
    ap.add_argument("--tokens-per-pass", type=int, default=16384, help="micro-batches are concatenated into passes of at most this many tokens")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic: at N=1 run the RCCL reduce-scatter/all-gather + comm-stream path on a world_size-1 group (not the headline run)")
    args = ap.parse_args()

    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" IS RCCL on ROCm
    elif args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    dims = {"qwen3-8b": ModelDims.qwen3_8b, "qwen3-0.6b": ModelDims.qwen3_0_6b, "tiny": ModelDims.tiny}[args.model]()
    engine = GRPOEngine(dims, device=device, optim=OptimConfig(), loss=LossConfig(), seed=0, force_collectives=args.force_collectives and world == 1)
    loader = FakeDataLoader(FakeDataLoaderConfig(batch_size=args.micro_batches * world), seq_len=args.seq_len, dp_world_size=world, dp_rank=rank)

    per_pass = max(args.tokens_per_pass // args.seq_len, 1)

    def make_step_batches():
        mbs = loader.get_batch()
        return [pack_micro_batches(mbs[i:i + per_pass], device) for i in range(0, len(mbs), per_pass)]

    total_steps = args.warmup + args.steps
    batches = [make_step_batches() for _ in range(total_steps)]  # resident in HBM before timing
    torch.cuda.synchronize()

    def barrier():
        if dist.is_initialized():  # N > 1, or the world_size-1 group of --force-collectives (same code path on one GPU)
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for s in range(args.warmup):
        out = engine.step(batches[s])
    barrier()
    engine.k1_events = []
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        out = engine.step(batches[s])
        losses.append(out["loss"])
    host_enqueue = time.perf_counter() - t0  # how long the host needed to enqueue the timed steps (GPU-bound when << elapsed)
    barrier()
    elapsed = time.perf_counter() - t0
    k1_events, engine.k1_events = engine.k1_events, None
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples = args.micro_batches * world * args.steps
        value = samples / elapsed
        # roofline of the log-prob kernel: algorithmic bytes (sizeof(T)*V + 20 per token) / mean launch time
        durs = [a.elapsed_time(b) * 1e-3 for a, b, _, _ in k1_events]
        rows = k1_events[0][2]
        esz = k1_events[0][3]
        alg_bytes = rows * (esz * dims.vocab_size + 20)
        mean_t = sum(durs) / len(durs)
        achieved = alg_bytes / mean_t
        traffic = None
        prof = ROOT / "profiles" / "k1_hbm_traffic.json"  # per-launch HBM bytes from rocprofv3 --pmc passes (see profiles/README.md)
        if prof.exists():
            try:
                rec = json.loads(prof.read_text())
                if rec.get("rows") == rows and rec.get("elem_size") == esz:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        flops_per_sample = dims.flops_per_token(args.seq_len) * args.seq_len
        result = {
            "metric": "GRPO trainer samples/sec (Qwen3-8B, 2k-token rollouts)" if args.model == "qwen3-8b" else f"GRPO trainer samples/sec ({dims.name})",
            "value": round(value, 4),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic (reference FakeDataLoader batches: ids<100, randn advantages/inference logprobs, T=1, all tokens trainable); random-init weights",
            "config": {
                "workload": f"{dims.name} GRPO policy-update step, fixed-length {args.seq_len}-token rollouts, {args.micro_batches} micro-batches/GPU/step "
                            f"(configs[2] trainer side; no 8-GPU-only config fits one bench line)",
                "global_batch": args.micro_batches * world,
                "seq_len": args.seq_len,
                "parallelism": f"dp{world} (ZeRO-2: replicated bf16 params, reduce-scatter f32 grads, sharded AdamW)",
                "tokens_per_pass": min(args.tokens_per_pass, args.micro_batches * args.seq_len),
                "step_includes": "scale all-reduce, forward, log-prob, GRPO loss, backward, grad reduce-scatter, clip, AdamW, param all-gather",
                "mfu_vs_2.5PF_bf16": round(value * flops_per_sample / (world * MFMA_BF16_PEAK), 4),
                "tflop_per_sample": round(flops_per_sample / 1e12, 2),
                "loss_last": float(sum(l.item() for l in losses[-1])),
                "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 1),
            },
            "roofline": {
                "kernel": f"logprob_fwd_row_kernel<{'bf16' if esz == 2 else 'f32'}> (selective log-softmax + entropy)",
                "bound": "hbm",
                "achieved": round(achieved / 1e9, 1),
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BPS, 4),
                "traffic": traffic,
                "bytes_per_launch": alg_bytes,
                "rows_per_launch": rows,
                "launches_timed": len(durs),
                "mean_launch_ms": round(mean_t * 1e3, 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_trainer import time_cpu_baseline

            cpu = time_cpu_baseline(dims, seq_len=args.seq_len)
            result["cpu_baseline"] = {
                "value": round(1.0 / cpu["seconds_per_sample"], 6),
                "unit": "samples/s",
                "cores": cpu["threads"],
                "kind": "port",
                "sample": f"oracle/cpu_trainer.py (f32 torch-CPU port of the reference step) on a {cpu['sample_tokens']}-token rollout prefix: one decoder layer fwd+bwd "
                          f"{cpu['t_layer']:.2f}s (x{dims.num_hidden_layers} layers), full-vocab LM head + log-probs + loss fwd+bwd {cpu['t_head_loss']:.2f}s, AdamW on "
                          f"{cpu['opt_params'] >> 20}Mi params {cpu['t_opt']:.2f}s (x params/{cpu['opt_params'] >> 20}Mi); scaled linearly to {args.seq_len} tokens "
                          f"(ignores quadratic attention: favours the CPU)",
            }
        if args.force_collectives and world == 1:
            result["config"]["diagnostic"] = "force_collectives: world_size-1 RCCL reduce-scatter/all-gather on the comm stream (not the headline run)"
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (flushed at exit when stdout is a pipe/file): flush it first so that the
        # JSON line is the LAST line of stdout
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
