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


def measure_extras(args, engine, loader, device, world, rank, barrier) -> dict:
    """Measurements the judge asked for next to the headline (VERDICT r1 'Next round' 3), each guarded so that a failure is REPORTED in
    the JSON instead of taking the headline down:
      mb2        the reference benchmark's default batch, 2 micro-batches per GPU (benchmarks/scripts/run_single_benchmark.py:91,187);
      weight_push  the sharded trainer -> inference push pattern (transports/weights: every rank sends its 1/dp slice of every layer
                 bucket point-to-point to every peer) timed with the other trainer ranks as STAND-IN receivers — bench.py runs all N GPUs
                 as trainers, so there is no inference pool to push to; samples/s including that push is reported next to the headline;
      j1         the fused MFMA LM head timed stand-alone at the step's shape (MFMA roofline of the one hand-written GEMM)."""
    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    extra: dict = {}
    dims = engine.dims
    # -- 2 micro-batches per GPU -------------------------------------------------------------------------------------------------------
    try:
        if args.micro_batches != 2:
            l2 = FakeDataLoader(FakeDataLoaderConfig(batch_size=2 * world), seq_len=args.seq_len, dp_world_size=world, dp_rank=rank)
            k2 = max(2, min(args.steps, 6))
            b2 = [[pack_micro_batches(l2.get_batch(), device)] for _ in range(k2 + 1)]
            engine.step(b2[0])
            barrier()
            t0 = time.perf_counter()
            for i in range(1, k2 + 1):
                engine.step(b2[i])
            barrier()
            dt = time.perf_counter() - t0
            if dist.is_initialized():
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            extra["mb2"] = {"micro_batches_per_gpu": 2, "steps": k2, "ms_per_step": round(dt / k2 * 1e3, 3), "samples_per_s": round(2 * world * k2 / dt, 4),
                            "note": "reference benchmark default batch (run_single_benchmark.py:91,187); same engine, same build, measured after the headline"}
    except Exception as e:  # pragma: no cover
        extra["mb2"] = {"error": repr(e)[:300]}
    # -- weight push with stand-in receivers -----------------------------------------------------------------------------------------------
    try:
        store = engine.store
        if world > 1:
            from prime_rl_amd.comm import TorchGroupCommunicator

            comm = TorchGroupCommunicator(None)
            engine.wait_params_ready()
            scratch = torch.empty(max(b.size for b in store.buckets), dtype=torch.bfloat16, device=device)

            def push_once():
                for b in store.buckets:
                    chunk = b.size // world
                    mine = store.params[b.offset + rank * chunk: b.offset + (rank + 1) * chunk]
                    comm.group_start()
                    for peer in range(world):
                        if peer != rank:
                            comm.send(mine, peer)
                            comm.recv(scratch[peer * chunk:(peer + 1) * chunk], peer)
                    comm.group_end()

            push_once()
            barrier()
            t0 = time.perf_counter()
            push_once()
            barrier()
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            sent = store.total * 2 / world * (world - 1)
            extra["weight_push"] = {"ms": round(dt * 1e3, 2), "bytes_sent_per_rank": int(sent), "GBps_per_rank": round(sent / dt / 1e9, 1),
                                    "pattern": f"sharded multi-source push, every rank sends its 1/{world} slice of each layer bucket to {world - 1} stand-in receivers "
                                               "(point-to-point over xGMI); in training it runs on its own stream and overlaps the next step"}
        else:
            extra["weight_push"] = {"ms": 0.0, "note": "N=1: no peer GPU to push to (the push communicator, protocol and async stream are exercised by "
                                                       "tests/test_collectives_gpu.py and tests/test_weight_push_cpu.py)"}
    except Exception as e:  # pragma: no cover
        extra["weight_push"] = {"error": repr(e)[:300]}
    # -- J1 stand-alone ---------------------------------------------------------------------------------------------------------------------
    try:
        from prime_rl_amd import ops

        if rank == 0 and dims.hidden_size % 64 == 0:
            n = min(args.tokens_per_pass, args.micro_batches * args.seq_len)
            hid = torch.randn(n, dims.hidden_size, device=device).to(torch.bfloat16)
            lab = torch.randint(0, dims.vocab_size, (n,), device=device)
            it = torch.ones(n, device=device)
            w = store.head_weight()
            ops.lmhead_logprob_fwd(hid, w, lab, it)
            evs = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.lmhead_logprob_fwd(hid, w, lab, it)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            flop = 2.0 * n * dims.vocab_size * dims.hidden_size
            extra["j1_roofline"] = {"kernel": "lmhead_kernel<fwd> (J1: fused MFMA LM head, GEMM + log-prob epilogue, logits never in HBM)", "bound": "mfma",
                                    "achieved": round(flop / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                    "frac": round(flop / (ms * 1e-3) / MFMA_BF16_PEAK, 4), "traffic": None, "flops_per_launch": flop, "tokens_per_launch": n,
                                    "mean_launch_ms": round(ms, 3)}
    except Exception as e:  # pragma: no cover
        extra["j1_roofline"] = {"error": repr(e)[:300]}
    # -- one MoE layer of config 4 (Qwen3-30B-A3B shapes), forward + backward, on the row-f4 kernels -----------------------------------------
    try:
        from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

        n, hidden_dim, e, k, inter = 16384, 2048, 128, 8, 768
        layer = MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k,
                            use_grouped_mm=True, load_balance_coeff=None), hidden_dim, inter).to(device).to(torch.bfloat16)
        layer.init_weights(0.02)
        x = torch.randn(1, n, hidden_dim, device=device).to(torch.bfloat16).requires_grad_(True)
        gy = torch.randn(1, n, hidden_dim, device=device).to(torch.bfloat16)

        def fb():
            x.grad = None
            layer(x).backward(gy)

        fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fb()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        flop = 3 * 3 * 2.0 * n * k * hidden_dim * inter  # three expert GEMMs, forward + two gradients each
        extra["moe_layer"] = {"shape": f"{n} tokens, {e} experts top-{k}, hidden {hidden_dim}, expert intermediate {inter} (Qwen3-30B-A3B layer), forward + backward",
                              "ms": round(ms, 3), "expert_gemm_TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1),
                              "note": "router, stable sort, gather, grouped MFMA GEMMs (pm_grouped_gemm_nt/nn/tn), combine; per-kernel numbers in profiles/r02_moe.md"}
        del layer, x, gy
    except Exception as ex:  # pragma: no cover
        extra["moe_layer"] = {"error": repr(ex)[:300]}
    if dist.is_initialized():
        dist.barrier()
    return extra


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--micro-batches", type=int, default=8, help="2048-token micro-batches (= samples) per GPU per optimizer step")
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--model", default="qwen3-8b", choices=["qwen3-8b", "qwen3-0.6b", "qwen2.5-1.5b", "qwen2.5-0.5b", "tiny"])
    ap.add_argument("--tokens-per-pass", type=int, default=16384, help="micro-batches are concatenated into passes of at most this many tokens")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic: at N=1 run the RCCL reduce-scatter/all-gather + comm-stream path on a world_size-1 group (not the headline run)")
    ap.add_argument("--recompute", default="none", choices=["none", "layer"],
                    help="layer = block-level activation recomputation (the reference's default model.ac): keep layer inputs only, re-run each layer's forward in backward")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements reported under config.extra (2 micro-batches/GPU line, weight push)")
    args = ap.parse_args()

    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run (one rank per GPU)")
    # PM_BENCH_DRYRUN_SHARED_GPU=1: rehearsal of the N > 1 launch on a ONE-GPU box — every rank uses device 0 and the group runs over gloo (RCCL refuses
    # two ranks on one device).  Same script, same engine, same barriers / reductions / extras; the line it prints carries config.diagnostic and is
    # not a measurement (tests/test_multirank_one_gpu.py runs it).
    dryrun = os.environ.get("PM_BENCH_DRYRUN_SHARED_GPU") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and dryrun:
        dist.init_process_group(backend="gloo")
    elif world > 1:
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" IS RCCL on ROCm
    elif args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    dims = {"qwen3-8b": ModelDims.qwen3_8b, "qwen3-0.6b": ModelDims.qwen3_0_6b, "qwen2.5-1.5b": ModelDims.qwen2_5_1_5b, "qwen2.5-0.5b": ModelDims.qwen2_5_0_5b,
            "tiny": ModelDims.tiny}[args.model]()
    engine = GRPOEngine(dims, device=device, optim=OptimConfig(), loss=LossConfig(), seed=0, force_collectives=args.force_collectives and world == 1,
                        recompute=args.recompute)
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
    if rank == 0 and dist.is_initialized():
        print(f"[bench preflight] RCCL group up: world={dist.get_world_size()} backend={dist.get_backend()} dp={engine.dp_size} "
              f"collectives={'reduce-scatter/all-gather on a side stream' if engine.use_collectives else 'none (single rank)'}", file=sys.stderr, flush=True)
    engine.k1_events = []
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        out = engine.step(batches[s])
        losses.append(out["loss"])
    barrier()
    elapsed = time.perf_counter() - t0
    k1_events, engine.k1_events = engine.k1_events, None
    # host cost of enqueuing ONE step on an EMPTY queue: an extra, untimed step after the timed region (inside it the HIP queue fills up and
    # the host blocks on back-pressure, so wall time there says nothing about the host; the first warm-up step is cold)
    th = time.perf_counter()
    engine.step(batches[-1])
    host_enqueue_empty_queue = time.perf_counter() - th
    barrier()
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = args.micro_batches * world * args.steps
    # everything the JSON line needs from the device is read NOW, on this thread: the watchdog below must be able to print the line while
    # the main thread sits in a wedged collective (a .item() or an event query from there would queue behind it)
    k1_durs = [a.elapsed_time(b) * 1e-3 for a, b, _, _ in k1_events] if k1_events else []
    loss_last = float(sum(l.item() for l in losses[-1]))

    def assemble(extra: dict) -> dict:
        """The contract's JSON line (rank 0) from the timed region alone; `extra` = whatever extra measurements finished."""
        value = samples / elapsed
        if k1_events:
            # roofline of the log-prob kernel: algorithmic bytes (sizeof(T)*V + 20 per token) / mean launch time
            durs = k1_durs
            rows = k1_events[0][2]
            esz = k1_events[0][3]
            alg_bytes = rows * (esz * dims.vocab_size + 20)
            mean_t = sum(durs) / len(durs)
            achieved = alg_bytes / mean_t
            traffic, traffic_source = None, None
            prof = ROOT / "profiles" / "k1_hbm_traffic.json"  # per-launch HBM bytes from rocprofv3 --pmc passes (see profiles/README.md)
            if prof.exists():
                try:
                    rec = json.loads(prof.read_text())
                    if rec.get("rows") == rows and rec.get("elem_size") == esz:
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_source = ("profiles/k1_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel at this launch shape, gfx950 "
                                          "correction applied (MI355X guide, HBM section); PMC counters cannot be collected inside a timed run, so this is a recorded "
                                          "profile of the same binary kernel, not a live reading")
                except Exception:
                    traffic = None
        flops_per_sample = dims.flops_per_token(args.seq_len) * args.seq_len
        result = {
            "metric": "GRPO trainer samples/sec (Qwen3-8B, 2k-token rollouts)" if (args.model == "qwen3-8b" and args.seq_len == 2048)
            else f"GRPO trainer samples/sec ({dims.name}, {args.seq_len}-token rollouts)",
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
                "loss_last": loss_last,
                "host_enqueue_ms_per_step": round(host_enqueue_empty_queue * 1e3, 1) if host_enqueue_empty_queue is not None else None,
                "lm_head": engine.lm_head_impl,
                "extra": extra,
            },
        }
        if k1_events:
            result["roofline"] = {
                "kernel": f"logprob_fwd_row_kernel<{'bf16' if esz == 2 else 'f32'}> (selective log-softmax + entropy)",
                "bound": "hbm",
                "achieved": round(achieved / 1e9, 1),
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "bytes_per_launch": alg_bytes,
                "rows_per_launch": rows,
                "launches_timed": len(durs),
                "mean_launch_ms": round(mean_t * 1e3, 4),
            }
        else:  # PM_LM_HEAD=fused: no stand-alone log-prob kernel runs; the dominant hand-written kernel is the fused MFMA LM head (J1)
            result["roofline"] = extra.get("j1_roofline")
        return result

    # ---- extra measurements (all ranks take part; reported under config.extra, never part of `value`) ----------------------------------
    extra: dict = {}
    if not args.no_extra:
        import threading

        done = threading.Event()

        def watchdog():  # extras are best-effort: if one of them wedges (e.g. a collective at N > 1), rank 0 prints the line it already has
            if not done.wait(float(os.environ.get("PM_BENCH_EXTRA_TIMEOUT", "180"))):  # (headline + roofline) and EVERY rank leaves
                if rank == 0:
                    line = assemble({"error": "an extra measurement did not finish in time; headline and roofline only"})
                    line["cpu_baseline"] = None
                    print(json.dumps(line), flush=True)
                else:
                    time.sleep(3.0)
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        extra = measure_extras(args, engine, loader, device, world, rank, barrier)
        done.set()

    if rank == 0:
        result = assemble(extra)
        wp = extra.get("weight_push") or {}
        if "ms" in wp:  # samples/s excluding (= value) and including the weight push, if the push were NOT overlapped
            extra["samples_per_s_incl_serial_weight_push"] = round(samples / (elapsed + args.steps * wp["ms"] * 1e-3), 4)
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
        if args.recompute != "none":
            result["config"]["recompute"] = args.recompute + " (one extra forward per layer in backward; not the headline configuration)"
        if dryrun:
            result["config"]["diagnostic"] = f"PM_BENCH_DRYRUN_SHARED_GPU: {world} ranks share ONE GPU over gloo — a rehearsal of the launch path, not a measurement"
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
