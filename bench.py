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

DETAIL_DIR = ROOT / "gpurun_out"  # verbose blocks (notes, per-leg parity tables) go to side files here; the JSON line keeps scalars
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
    # -- BASELINE.json configs[4]: 32 768-token rollouts through the SAME engine with block-level recomputation (VERDICT r5 'next' 3) ---------------------
    try:
        if rank == 0 and world == 1 and getattr(args, "long_context", True) and dims.head_dim == dims.kernel_head_dim:
            extra["long_context"] = long_context_extra(engine, dims, device)
    except Exception as e:  # pragma: no cover
        extra["long_context"] = {"error": repr(e)[:300]}
    finally:
        torch.cuda.empty_cache()
    # -- the GEMM routing of rounds 3-4 on the SAME engine: library (hipBLASLt) forward + dgrad on [in, out] weight copies, own weight gradients --------
    # (VERDICT r4 'next' 1: "add config.extra.all_own so the hand-written-only step is driver-measured".  Since round 5 the hand-written-only step IS
    # the headline (PM_GEMM_OWN=all); `all_own` repeats its numbers under the name the review asked for and `library_mix` times the old default next to it.)
    try:
        if engine.gemm_own == {"fwd", "dgrad", "wgrad"} and hasattr(engine, "set_gemm_own"):
            kx = max(2, min(args.steps, 4))
            per_pass = max(args.tokens_per_pass // args.seq_len, 1)

            def time_steps(k):
                bs = []
                for _ in range(k + 1):
                    mbs = loader.get_batch()
                    bs.append([pack_micro_batches(mbs[i:i + per_pass], device) for i in range(0, len(mbs), per_pass)])
                engine.step(bs[0])
                barrier()
                t0 = time.perf_counter()
                for i in range(1, k + 1):
                    engine.step(bs[i])
                barrier()
                dt = time.perf_counter() - t0
                if dist.is_initialized():
                    t = torch.tensor([dt], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                return dt

            engine.set_gemm_own("wgrad")
            try:
                dt = time_steps(kx)
            finally:
                engine.set_gemm_own("all")
                torch.cuda.empty_cache()
            extra["library_mix"] = {"gemm_own": ["wgrad"], "steps": kx, "ms_per_step": round(dt / kx * 1e3, 3),
                                    "samples_per_s": round(args.micro_batches * world * kx / dt, 4),
                                    "note": "PM_GEMM_OWN=wgrad, the default of rounds 3-4: hipBLASLt forward + dgrad on K-contiguous [in, out] weight copies "
                                            "(refreshed by pm_transpose_2d after every update), weight gradients on the own core, stand-alone SwiGLU pass; "
                                            "same engine, same build, measured after the headline"}
            dt = time_steps(kx)
            extra["all_own"] = {"gemm_own": sorted(engine.gemm_own), "steps": kx, "ms_per_step": round(dt / kx * 1e3, 3),
                                "samples_per_s": round(args.micro_batches * world * kx / dt, 4),
                                "note": "every GEMM of the step on csrc/gemm.hip (= the headline configuration since round 5), re-timed right after library_mix"}
    except Exception as e:  # pragma: no cover
        extra["library_mix"] = {"error": repr(e)[:300]}
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
            extra["j1_roofline"] = {"kernel": "gemm_kernel<..., MODE 8> + lmhead_finalize_parts_kernel (J1: fused MFMA LM head = the GEMM core's tile stream with the log-prob epilogue, logits never in HBM)", "bound": "mfma",
                                    "achieved": round(flop / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                    "frac": round(flop / (ms * 1e-3) / MFMA_BF16_PEAK, 4), "traffic": None, "flops_per_launch": flop, "tokens_per_launch": n,
                                    "mean_launch_ms": round(ms, 3)}
    except Exception as e:  # pragma: no cover
        extra["j1_roofline"] = {"error": repr(e)[:300]}
    # -- the own MFMA GEMM core stand-alone: the weight gradient of gate|up at this step's token count (the largest own kernel of the step) --------------
    try:
        from prime_rl_amd import ops

        if rank == 0 and "wgrad" in getattr(engine, "gemm_own", ()):
            n = min(args.tokens_per_pass, args.micro_batches * args.seq_len)
            dy = torch.randn(n, 2 * dims.intermediate_size, device=device).to(torch.bfloat16)
            xin = torch.randn(n, dims.hidden_size, device=device).to(torch.bfloat16)
            gbuf = torch.empty(2 * dims.intermediate_size, dims.hidden_size, dtype=torch.float32, device=device)
            ops.gemm(dy.t(), xin.t(), out=gbuf)
            evs = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(dy.t(), xin.t(), out=gbuf)  # as in a one-pass step: the first (and last) pass overwrites
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
            flop = 2.0 * n * 2 * dims.intermediate_size * dims.hidden_size
            extra["gemm_roofline"] = {"kernel": "gemm_kernel<3, K-major, K-major> (pm_gemm_bf16: gate|up weight gradient, bf16 x bf16 -> f32, random operands)",
                                      "bound": "mfma", "achieved": round(flop / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                      "frac": round(flop / (ms * 1e-3) / MFMA_BF16_PEAK, 4), "traffic": None, "flops_per_launch": flop, "tokens_per_launch": n,
                                      "median_launch_ms": round(ms, 3),
                                      "note": "the step's 145 weight-gradient GEMMs run on this kernel (27 % of the step); PMC: profiles/r03_gemm.md (MFMA busy 87.6 %)"}
            del dy, xin, gbuf
    except Exception as e:  # pragma: no cover
        extra["gemm_roofline"] = {"error": repr(e)[:300]}
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
            for p_ in layer.parameters():  # (a trainer's weight gradients go straight into the f32 main-grad buffer: no `.grad +=` passes over 1.2 GB of bf16)
                p_.grad = None
            layer(x).backward(gy)

        for _ in range(3):  # (the caching allocator and the clocks settle: the first passes after the 8 B engine's step allocate fresh blocks)
            fb()
        torch.cuda.synchronize()
        per_iter = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fb()
            e1.record()
            e1.synchronize()
            per_iter.append(e0.elapsed_time(e1))
        ms = sorted(per_iter)[len(per_iter) // 2]
        flop = 3 * 3 * 2.0 * n * k * hidden_dim * inter  # three expert GEMMs, forward + two gradients each
        extra["moe_layer"] = {"shape": f"{n} tokens, {e} experts top-{k}, hidden {hidden_dim}, expert intermediate {inter} (Qwen3-30B-A3B layer), forward + backward",
                              "ms": round(ms, 3), "ms_mean": round(sum(per_iter) / len(per_iter), 3), "ms_min": round(min(per_iter), 3), "iters": len(per_iter),
                              "expert_gemm_TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1),
                              "note": "median of 12 passes (HIP events around each); router, stable sort, gather, grouped MFMA GEMMs on the gemm.hip core (the SwiGLU backward in the "
                                      "down row gradient's epilogue, expert offsets held in registers), combine; per-kernel numbers in profiles/r05_moe_layer.md"}
        del layer, x, gy
    except Exception as ex:  # pragma: no cover
        extra["moe_layer"] = {"error": repr(ex)[:300]}
    if dist.is_initialized():
        dist.barrier()
    return extra


def long_context_extra(engine, dims, device, seq: int = 32768, rollouts=(1, 4), steps: int = 2) -> dict:
    """BASELINE.json configs[4] (32k-token agentic rollouts) on ONE GPU, driver-visible: the policy-update step of the SAME engine on `r` rollouts of `seq`
    tokens (one document per pass: the causal attention of a pass is seq^2 / 2 per head), `recompute="layer"` (the reference's default model.ac; at 32k
    tokens the kept activations of 36 layers would be ~140 GB per pass).  Reports ms per step, samples/s, MFU by the reference formula
    (trainer/perf.py:177-213: 12 L H q T counts the attention as 3 dense N^2 passes) AND by the FLOPs the step executes (causal half; forward twice under
    recomputation), the attention kernels' share of the step (their stand-alone time at this shape x launches / step time; the kernel trace of the same
    step is profiles/r06_bench_32k.md), and the peak HBM bytes.  Context-parallel variants (utils/cp.py:140-190, ring_attn.py:99-280) need > 1 GPU:
    correctness at world 2 on a shared GPU is tests/test_multirank_one_gpu.py."""
    from prime_rl_amd import ops
    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    res: dict = {"seq_len": seq, "recompute": "layer"}
    old, old_freq = engine.recompute, engine.recompute_freq
    engine.recompute, engine.recompute_freq = "layer", 1
    try:
        # the attention kernels stand-alone at this shape (one 32k-token document, all heads)
        hq, hkv, hd = dims.num_attention_heads, dims.num_key_value_heads, dims.kernel_head_dim
        q = torch.randn(seq, hq, hd, device=device).to(torch.bfloat16)
        k = torch.randn(seq, hkv, hd, device=device).to(torch.bfloat16)
        v = torch.randn(seq, hkv, hd, device=device).to(torch.bfloat16)
        do = torch.randn(seq, hq, hd, device=device).to(torch.bfloat16)
        ss = torch.zeros(seq, dtype=torch.int32, device=device)
        scale = hd**-0.5

        def timed(fn, reps=3):
            fn()
            evs = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            return sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]

        o, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
        t_f = timed(lambda: ops.attn_varlen_fwd(q, k, v, ss, scale))
        t_b = timed(lambda: ops.attn_varlen_bwd(do, q, k, v, o, lse, ss, scale))
        causal_fwd = 2.0 * seq * seq * hq * dims.head_dim  # QK^T + PV, causal half
        res["attn_fwd_ms"], res["attn_bwd_ms"] = round(t_f, 3), round(t_b, 3)
        res["attn_fwd_frac_of_2.5PF"] = round(causal_fwd / (t_f * 1e-3) / MFMA_BF16_PEAK, 4)
        res["attn_bwd_frac_of_2.5PF"] = round(2.5 * causal_fwd / (t_b * 1e-3) / MFMA_BF16_PEAK, 4)
        del q, k, v, do, o, lse
        L = dims.num_hidden_layers
        for r in rollouts:
            try:
                dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=r), seq_len=seq)
                bs = [[pack_micro_batches([mb], device) for mb in dl.get_batch()] for _ in range(steps + 1)]
                torch.cuda.reset_peak_memory_stats()
                engine.step(bs[0])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in bs[1:]:
                    engine.step(b)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                tok = r * seq
                ref_flops = dims.flops_per_token(seq) * tok
                mm = dims.active_mm_params()
                head = dims.vocab_size * dims.hidden_size
                executed = (8.0 * (mm - head) + 6.0 * head) * tok + r * L * (2 + 2.5) * causal_fwd  # layers: fwd + recomputed fwd + 2 x bwd; head: no recompute
                attn_ms = r * L * (2 * t_f + t_b)
                res[f"r{r}"] = {"rollouts": r, "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(r / dt, 4), "tokens_per_s": round(tok / dt, 1),
                                "mfu_ref_formula": round(ref_flops / dt / MFMA_BF16_PEAK, 4), "mfu_executed_causal": round(executed / dt / MFMA_BF16_PEAK, 4),
                                "attention_ms": round(attn_ms, 2), "attention_share": round(attn_ms / (dt * 1e3), 4),
                                "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
                del bs
            except Exception as e:  # pragma: no cover
                res[f"r{r}"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
        # the reference's `ac.mode = "selective"` (norm / attn_proj / mlp re-run, the attention core's output kept): the same step with one attention forward
        # per layer less; 288 GB hold the kept outputs.  Then with `ac.freq = 2` (trainer/model.py:1151: only every second layer is checkpointed, the others
        # keep their activations: 253 GB peak, half the recompute)
        for key, freq in (("r1_selective", 1), ("r1_selective_freq2", 2)):
            try:
                engine.recompute, engine.recompute_freq = "selective", freq
                dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=1), seq_len=seq)
                bs = [[pack_micro_batches([mb], device) for mb in dl.get_batch()] for _ in range(steps + 1)]
                torch.cuda.reset_peak_memory_stats()
                engine.step(bs[0])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in bs[1:]:
                    engine.step(b)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                res[key] = {"rollouts": 1, "recompute": "selective", "recompute_freq": freq, "ms_per_step": round(dt * 1e3, 2), "tokens_per_s": round(seq / dt, 1),
                            "mfu_ref_formula": round(dims.flops_per_token(seq) * seq / dt / MFMA_BF16_PEAK, 4), "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
                del bs
            except Exception as e:  # pragma: no cover
                res[key] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
    finally:
        engine.recompute, engine.recompute_freq = old, old_freq
    return res


def moe_step_extra(device, layers: int = 8, n_mb: int = 8, seq: int = 2048, steps: int = 3) -> dict:
    """BASELINE.json configs[3] on one GPU, driver-visible (VERDICT r4 'next' 4): the module-engine trainer step (trainer/module_engine.py: forward + loss +
    backward + clip + fused AdamW) of a Qwen3-30B-A3B-width model cut to `layers` of its 48 layers — the full model needs the expert-parallel node (30.5 B
    parameters x 18 B of f32 state).  Runs after the headline engine has been released."""
    try:
        from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
        from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.module_engine import ModuleEngine
        from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

        cfg = Qwen3MoeConfig(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=4,
                             num_experts=128, num_experts_per_tok=8, moe_intermediate_size=768, norm_topk_prob=True)
        with torch.device(device):
            model = Qwen3MoeForCausalLM(cfg)
        model = model.to(torch.bfloat16)
        gen = torch.Generator(device=device).manual_seed(0)
        with torch.no_grad():
            for _, p in model.named_parameters():
                if p.dim() == 1:
                    p.fill_(1.0)
                else:
                    p.copy_((torch.randn(p.shape, device=device, generator=gen) * 0.02).to(p.dtype))
        inject_prime_lm_head(model, chunk_size=8192)
        eng = ModuleEngine(model, device, optim=OptimConfig())
        dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=n_mb), seq_len=seq)
        batches = [[pack_micro_batches(dl.get_batch(), device)] for _ in range(steps + 1)]
        eng.step(batches[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[1:]:
            eng.step(b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tok = n_mb * seq
        flops = eng.dims.flops_per_token(seq) * tok
        res = {"model": f"Qwen3-30B-A3B width (hidden 2048, 128 experts top-8, expert intermediate 768), {layers} of 48 layers, {n_mb} x {seq}-token rollouts",
               "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(n_mb / dt, 3), "tokens_per_s": round(tok / dt, 1),
               "mfu_vs_2.5PF_bf16": round(flops / dt / MFMA_BF16_PEAK, 4), "params_B": round(sum(p.numel() for p in model.parameters()) / 1e9, 2),
               "note": "nn.Module trainer step (ModuleEngine): router / sort / gather / combine kernels, grouped expert GEMMs on the gemm.hip tile stream, f32 expert "
                       "main-grads written by the grouped weight-gradient GEMM, stored-logits LM head, one fused AdamW launch over 5.6 B parameters (31 ms of the step)"}
        del eng, model, batches
        return res
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)[:300]}


def parity_block(probe: dict, dims, seq_len: int) -> dict:
    """The parity gates of SURVEY.md section 8(d), evaluated on what ONE EXTRA UNTIMED STEP of the same engine (right after the timed region, same batch
    stream) left in `engine.parity_probe` (device tensors captured inside the step without a host sync; rank 0 only):
      gather_bit_exact   token-index gather of the embedding (all rows) AND the label gather over the vocab (all rows): K1's log-prob must equal
                         (logit[label] * inv_T) - logz bit for bit
      logprob_max_abs /  K1 (selective log-softmax + entropy on the step's bf16 logits) vs torch.log_softmax in f32 on 64 sampled rows
      entropy_max_abs    (test_fused_lm_head.py:398-457 bound: rtol 1e-3, atol 1e-4)
      k2_loss_abs_delta_on_own_logprobs
                         K2's scalar loss vs the oracle's restatement of compute_loss (loss.py:285-403) on the SAME (K2's own input) log-probs: a check
                         of the LOSS KERNEL, not of the forward — the forward is checked by `realistic` below and by tests/test_engine_gpu.py
      flips              DPPO trust-region mask decisions (loss.py:124-133) that differ between K2 and the oracle, summed over samples
      metric keys        K2's per-sample metrics == the reference's keys (loss.py:151-159), values compared too."""
    import math

    import numpy as np

    from oracle import grpo_oracle as O
    from prime_rl_amd import _lib

    out: dict = {}
    out["gather_bit_exact"] = bool(probe["embed_gather_equal"].item()) and bool(probe["label_gather_equal"].item())
    out["label_gather_max_abs"] = float(probe["label_gather_max_abs"].item())
    rows = probe["rows"]
    z = probe["logits_rows"].float() * probe["inv_t_rows"][:, None]
    lsm = torch.log_softmax(z, dim=-1)
    lp_ref = lsm.gather(1, probe["labels_rows"][:, None])[:, 0]
    ent_ref = torch.logsumexp(z, -1) - (torch.softmax(z, -1) * z).sum(-1)
    out["logprob_max_abs"] = float((probe["logprobs"][rows] - lp_ref).abs().max().item())
    out["entropy_max_abs"] = float((probe["entropy"][rows] - ent_ref).abs().max().item())
    out["rows_sampled"] = int(rows.numel())
    b = probe["batch"]
    cu = b.cu_samples.cpu().numpy()
    lens = [int(cu[i + 1] - cu[i]) for i in range(len(cu) - 1)]
    lp = probe["logprobs"].cpu().numpy().astype(np.float32)
    pad = np.float32(math.log(1.0 / dims.vocab_size))
    lp_shift = np.concatenate([O.shift_right(lp[cu[i]:cu[i + 1]], pad) for i in range(len(lens))])  # shift_tensor_right per sample (loss.py:76-91)
    loss_o, met_o, _ = O.compute_loss(lp_shift, b.inference_logprobs.cpu().numpy(), b.advantages.cpu().numpy(), b.loss_mask.cpu().numpy(), lens, O.LossConfig(),
                                      rl_scale=probe["scales"][0])
    out["loss_engine"] = float(probe["loss"].item())
    out["loss_oracle"] = float(loss_o)
    out["k2_loss_abs_delta_on_own_logprobs"] = abs(out["loss_engine"] - out["loss_oracle"])
    out["k2_loss_note"] = "K2 (pm_grpo_loss_fwd_bwd) vs oracle.compute_loss on K2's own input log-probs: a loss-kernel check; forward parity = parity.realistic"
    stats = probe["sample_stats"].cpu().numpy()
    mask = b.loss_mask.cpu().numpy()
    n_mask = np.array([mask[cu[i]:cu[i + 1]].sum() for i in range(len(lens))], dtype=np.float64)
    drop_e = stats[:, _lib.STAT["RL_N_DROP"]].astype(np.float64)
    drop_o = np.round(met_o["is_masked"].astype(np.float64) * np.maximum(n_mask, 1))
    out["flips"] = int(np.abs(drop_e - drop_o).sum())
    out["masked_tokens"] = int(drop_e.sum())
    keys_ref = ["masked_mismatch_kl", "unmasked_mismatch_kl", "is_masked", "is_masked_low", "is_masked_high", "masked_advantage_positive",
                "masked_advantage_negative"]  # default_loss_fn's metrics dict, loss.py:151-159
    keys_e = _lib.METRIC_KEYS[:len(keys_ref)]
    out["metric_keys_identical"] = (list(keys_e) == keys_ref) and (sorted(met_o.keys()) == sorted(keys_ref))
    met_e = probe["sample_metrics"].cpu().numpy()
    out["metrics_max_abs"] = float(max(np.abs(met_e[:, i] - met_o[k]).max() for i, k in enumerate(keys_ref)))
    out["gates"] = {"K2 loss vs oracle on its own log-probs <= 1e-4": out["k2_loss_abs_delta_on_own_logprobs"] <= 1e-4, "logprob atol 1e-4 rtol 1e-3": out["logprob_max_abs"] <= 1e-4 + 1e-3 * float(lp_ref.abs().max().item()),
                    "gather_bit_exact": out["gather_bit_exact"], "metric_keys_identical": out["metric_keys_identical"], "flips==0": out["flips"] == 0}
    out["source"] = "one extra untimed step of this run's engine right after the timed region (device copies taken inside that step)"
    return out


def realistic_parity(engine, dims, device, seq_len: int, n_rollouts: int = 8, n_rollouts_2l: int = 64, prompt_len: int = 256, temperature: float = 0.25,
                     stale_sigma: float = 0.3, detail_path=None) -> dict:
    """Forward + loss of the engine on rollout-like batches, against the oracle's reference-dtype restatement of the scoring forward (oracle B: the
    reference's bf16 rounding points, exact GEMMs) run with torch on the device — untimed, after the timed region.  What makes a batch rollout-like (the
    FakeDataLoader batches of the timed region are not: ids < 100, log-probs near -11.9, the trust-region mask nowhere near a decision):
      * input ids uniform over the FULL vocabulary, `prompt_len` masked prompt tokens per rollout, group-relative advantages of complete groups of 8
        (r - mean(group) = +-0.5: zero mean over the batch, as for every complete group, orchestrator/algo/grpo.py:27-45);
      * the scored tokens are the policy's own high-probability tokens: labels = argmax of the oracle's logits at T = `temperature` (a sampled token has
        p ~ 0.2-0.9 in RL rollouts; under a random-init model the next input id has p ~ 1e-5 and exp(lp) - exp(lp_inf) can never reach +-0.2);
      * a stale sampling policy: lp_inf = lp_oracle + N(0, `stale_sigma`^2), so a sizeable share of the tokens sits NEAR the DPPO mask threshold
        (loss.py:124-133) and the mask decisions depend on the forward's log-probs.
    Three legs (VERDICT r5 'next' 1):
      two_layer         2 layers at the model's width on a FIXED seeded weight state (GRPOEngine(seed=0) — independent of --steps / --warmup, so every run of
                        this script evaluates the same numbers) over `n_rollouts_2l` x seq_len tokens (64 x 2048 -> 114 688 trained tokens: the paired standard
                        error is <= 1e-4, the north-star tolerance is RESOLVABLE) — THE loss-parity gate: |loss delta to the nearer oracle (B / B_fa)| <= 1e-4 + 2 SE;
      two_layer_live    the same on the first two layers of the LIVE weights (what the timed steps left), same token count, same gate;
      full_depth        the live engine at its full depth on `n_rollouts` x seq_len tokens (14 336 trained): a NOISE-LEVEL check only (a 36-layer random-init
                        network turns every bf16 rounding into a 5e-2 per-token log-prob difference; at 14k tokens the SE of the scalar loss is 2-8e-4) —
                        gated on the per-token noise level and the mask-flip counts against the reference restatements' own, the loss delta is reported in SE units.
    Oracles / yardsticks: B = reference dtype, exact GEMMs; B_fa = B + flash-attention's bf16 P; C = torch bf16 GEMMs; D = torch bf16 GEMMs + bf16 P — the
    reference's own arithmetic on this box (cuBLAS-like GEMMs + flash-attn rounding), the like-for-like yardstick of the engine."""
    import dataclasses
    import math

    from oracle import grpo_oracle as O
    from prime_rl_amd import _lib
    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, OptimConfig, pack_micro_batches

    lcfg = O.LossConfig()
    n = seq_len
    pad = math.log(1.0 / dims.vocab_size)
    GROUP = 8

    def make_rollouts(count: int, seed: int):
        g = torch.Generator().manual_seed(seed)
        mbs = []
        for g0 in range(0, count, GROUP):
            gs = min(GROUP, count - g0)
            rewards = torch.zeros(gs)
            rewards[torch.randperm(gs, generator=g)[: gs // 2]] = 1.0
            adv_g = rewards - rewards.mean()  # ZERO mean over every complete group: the convexity term -mean(adv) * var(lp noise) / 2 cancels (profiles/r04_parity.md)
            for r in range(gs):
                mask = torch.ones(1, n, dtype=torch.bool)
                mask[0, :prompt_len] = False
                a = float(adv_g[r]) or 0.5
                mbs.append(dict(input_ids=torch.randint(0, dims.vocab_size, (1, n), generator=g), position_ids=torch.arange(n)[None], advantages=a * mask.float(),
                                inference_logprobs=torch.zeros(1, n), loss_mask=mask, temperatures=torch.full((1, n), temperature), seq_lens=torch.tensor([n]),
                                sequence_lengths=[n], env_names=["parity"] * n))
        return mbs

    def oracle_logprobs(sd, cfg, mbs, labels, mm, attn_p_bf16=False):
        """per-token log-prob of `labels` (or, labels None: the greedy tokens, returned as well) under the oracle forward; shifted right per rollout"""
        lps, labs = [], []
        with torch.no_grad():
            for i, mb in enumerate(mbs):
                ids = mb["input_ids"][0].to(device)
                h = O.qwen3_forward(sd, cfg, ids, mb["position_ids"][0].to(device), [n], mm=mm, attn_p_bf16=attn_p_bf16)
                hw = sd["lm_head.weight"] if "lm_head.weight" in sd else sd["model.embed_tokens.weight"]
                logits = (mm(h, hw) if mm is not O.exact_mm else h.float() @ hw.float().T).float() / temperature
                lab = logits.argmax(-1) if labels is None else labels[i]
                lp = torch.log_softmax(logits, -1).gather(-1, lab[:, None])[:, 0]
                lps.append(torch.cat([lp.new_full((1,), pad), lp[:-1]]).cpu())
                labs.append(lab)
                del logits, h
        return torch.cat(lps), labs

    def per_token(lp, inf, adv, mask):
        lp, inf, a = lp.double(), inf.double(), adv.double()
        lr = lp - inf
        pd = torch.exp(lp) - torch.exp(inf)
        invalid = torch.where(a > 0, pd > lcfg.dppo_mask_high, pd < -lcfg.dppo_mask_low) & mask
        keep = mask & ~invalid
        near = (torch.where(a > 0, (pd - lcfg.dppo_mask_high).abs(), (pd + lcfg.dppo_mask_low).abs()) < 0.02) & mask
        return -(keep * (lcfg.adv_tau * a) * torch.exp(lr)) + lcfg.kl_tau * (mask * lr**2), invalid, near

    torch_mm = lambda a, w, b=None: torch.nn.functional.linear(a, w, b)  # noqa: E731

    def one_depth(eng, sd, n_layers, mbs, sigmas, chunk=8):
        """Chunks of `chunk` rollouts: the four oracle forwards per rollout, ONE engine forward per chunk, the loss kernel once per sampling policy and chunk
        (scale = the trained tokens of the WHOLE leg, so the chunk losses add up to the leg's loss)."""
        cfg = O.TinyQwen3Config(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=n_layers,
                                num_attention_heads=dims.num_attention_heads, num_key_value_heads=dims.num_key_value_heads, head_dim=dims.head_dim,
                                vocab_size=dims.vocab_size, rms_norm_eps=dims.rms_norm_eps, rope_theta=dims.rope_theta, qk_norm=dims.qk_norm,
                                attention_bias=dims.attention_bias, rope_scaling=dims.rope_scaling)
        mask = torch.cat([mb["loss_mask"][0] for mb in mbs])
        adv = torch.cat([mb["advantages"][0] for mb in mbs])
        n_tr = int(mask.sum())
        lp = {k: [] for k in "bcfde"}
        infs = {name: [] for name, _ in sigmas}
        k2_loss = {name: 0.0 for name, _ in sigmas}
        k2_drop = {name: 0 for name, _ in sigmas}
        gi = torch.Generator().manual_seed(77)
        for c0 in range(0, len(mbs), chunk):
            part = mbs[c0:c0 + chunk]
            lp_b, labels = oracle_logprobs(sd, cfg, part, None, O.exact_mm)               # B: reference dtype, exact GEMMs; greedy labels
            lp_c, _ = oracle_logprobs(sd, cfg, part, labels, torch_mm)                      # C: same arithmetic, torch bf16 GEMMs
            lp_f, _ = oracle_logprobs(sd, cfg, part, labels, O.exact_mm, attn_p_bf16=True)  # B_fa: B + the ONE rounding point inside flash-attention (bf16 P into P.V)
            lp_d, _ = oracle_logprobs(sd, cfg, part, labels, torch_mm, attn_p_bf16=True)    # D: torch bf16 GEMMs + bf16 P = the reference as it runs (like-for-like yardstick)
            pb = pack_micro_batches(part, device)
            pb.labels = torch.cat(labels).to(torch.int64)  # the scored tokens: the policy's greedy tokens instead of the next input id
            ctx = eng.forward(pb)  # ONE forward of the engine per chunk; the loss kernel then runs once per sampling policy
            lp_raw = ctx["logprobs"].cpu()
            lp_e = torch.cat([torch.cat([lp_raw.new_full((1,), pad), lp_raw[i * n:(i + 1) * n - 1]]) for i in range(len(part))])
            for k, v in zip("bcfde", (lp_b, lp_c, lp_f, lp_d, lp_e)):
                lp[k].append(v)
            for name, sigma in sigmas:
                inf = lp_b + sigma * torch.randn(lp_b.shape, generator=gi)
                infs[name].append(inf)
                pb.inference_logprobs = inf.to(torch.float32).to(device)
                eng.loss(ctx, (n_tr, 1, 1))
                k2_loss[name] += float(ctx["loss"].item())
                k2_drop[name] += int(ctx["sample_stats"][:, _lib.STAT["RL_N_DROP"]].sum().item())
            del ctx, pb
        lp_b, lp_c, lp_f, lp_d, lp_e = (torch.cat(lp[k]) for k in "bcfde")
        p_lab = torch.exp(lp_b[mask])
        am = lambda x, y: float((x - y)[mask].abs().mean())  # noqa: E731
        res = {"layers": n_layers, "rollouts": len(mbs), "trained_tokens": n_tr, "mean_advantage": float(adv[mask].mean()),
               "logprob_abs_mean": am(lp_e, lp_b), "logprob_abs_mean_vs_bfa": am(lp_e, lp_f),
               "logprob_mean_shift": float((lp_e - lp_b)[mask].mean()), "logprob_mean_shift_se": float((lp_e - lp_b)[mask].std() / math.sqrt(n_tr)),
               # (labels are B's OWN argmax: every other implementation of the same arithmetic scores them lower on average — a selection effect; compare with D's)
               "yardstick_d_logprob_mean_shift": float((lp_d - lp_b)[mask].mean()), "engine_minus_d_logprob_mean_shift": float((lp_e - lp_d)[mask].mean()),
               "engine_minus_d_logprob_mean_shift_se": float((lp_e - lp_d)[mask].std() / math.sqrt(n_tr)),
               "yardstick_logprob_abs_mean": am(lp_c, lp_b), "yardstick_bfa_logprob_abs_mean": am(lp_f, lp_b), "yardstick_d_logprob_abs_mean": am(lp_d, lp_b),
               "label_prob_median": float(p_lab.median()), "label_prob_gt_0.2": float((p_lab > 0.2).float().mean())}
        # noise LEVEL, like for like: the engine (bf16 GEMMs in its own summation order + flash-style bf16 P) against B, over the reference as it
        # runs (torch bf16 GEMMs + bf16 P = D) against B.  The two single-source ratios are reported next to it.
        res["noise_ratio"] = res["logprob_abs_mean"] / max(res["yardstick_d_logprob_abs_mean"], 1e-12)
        res["noise_ratio_vs_c"] = res["logprob_abs_mean"] / max(res["yardstick_logprob_abs_mean"], 1e-12)
        res["noise_ratio_bfa_pair"] = res["logprob_abs_mean_vs_bfa"] / max(res["yardstick_bfa_logprob_abs_mean"], 1e-12)
        for name, sigma in sigmas:
            inf = torch.cat(infs[name])
            l_b, inv_b, near_b = per_token(lp_b, inf, adv, mask)
            l_e, inv_e, _ = per_token(lp_e, inf, adv, mask)
            l_c, inv_c, _ = per_token(lp_c, inf, adv, mask)
            l_f, inv_f, _ = per_token(lp_f, inf, adv, mask)
            l_d, inv_d, _ = per_token(lp_d, inf, adv, mask)
            d, dc, df, dfb, dd = (l_e - l_b)[mask], (l_c - l_b)[mask], (l_e - l_f)[mask], (l_f - l_b)[mask], (l_d - l_b)[mask]
            se = lambda x: float(x.std() / math.sqrt(n_tr))  # noqa: E731
            res[name] = {
                "lp_inf": f"lp_oracle + N(0, {sigma}^2)",
                "loss_engine": k2_loss[name], "loss_oracle": float(l_b.sum() / n_tr), "loss_delta": k2_loss[name] - float(l_b.sum() / n_tr),
                "loss_abs_delta": abs(k2_loss[name] - float(l_b.sum() / n_tr)),
                "loss_delta_se": se(d), "loss_engine_host_formula": float(l_e.sum() / n_tr),
                "yardstick_loss_delta": float(dc.sum() / n_tr), "yardstick_loss_abs_delta": abs(float(dc.sum() / n_tr)), "yardstick_loss_delta_se": se(dc),
                "yardstick_d_loss_delta": float(dd.sum() / n_tr), "yardstick_d_loss_delta_se": se(dd),
                "masked_tokens_oracle": int(inv_b.sum()), "masked_tokens_engine": k2_drop[name], "masked_tokens_engine_host_formula": int(inv_e.sum()),
                "tokens_within_0.02_of_threshold": int(near_b.sum()), "flips": int((inv_e != inv_b).sum()), "yardstick_flips": int((inv_c != inv_b).sum()),
                "yardstick_d_flips": int((inv_d != inv_b).sum()),
                # the second oracle column: B_fa = B with flash-attention's bf16 P (the engine's attention rounds P to bf16 as flash-attn does)
                "loss_oracle_bfa": float(l_f.sum() / n_tr), "loss_delta_vs_bfa": float(df.sum() / n_tr), "loss_abs_delta_vs_bfa": abs(float(df.sum() / n_tr)),
                "loss_delta_vs_bfa_se": se(df),
                "yardstick_bfa_loss_abs_delta": abs(float(dfb.sum() / n_tr)), "yardstick_bfa_loss_delta_se": se(dfb),
                "masked_tokens_oracle_bfa": int(inv_f.sum()), "masked_tokens_yardstick": int(inv_c.sum()), "flips_vs_bfa": int((inv_e != inv_f).sum()),
                "yardstick_bfa_flips": int((inv_f != inv_b).sum()),
            }
            r = res[name]
            # the gate's tolerance: north-star 1e-4 + 2 standard errors, the SE from the YARDSTICK pairs (never from the engine under test: ADVICE r4)
            r["gate_se"] = max(r["yardstick_loss_delta_se"], r["yardstick_bfa_loss_delta_se"], r["yardstick_d_loss_delta_se"])
            r["gate_tol"] = 1e-4 + 2 * r["gate_se"]
            r["nearer_oracle_abs_delta"] = min(r["loss_abs_delta"], r["loss_abs_delta_vs_bfa"])
        return res

    live_sd = engine.store.state_dict()  # HF-named views of the live bf16 parameters (no copies)
    sig = (("fresh", 0.02), ("stale", stale_sigma))
    out: dict = {"data": f"rollouts of {seq_len} tokens, ids over the full vocab, {prompt_len} masked prompt tokens, labels = greedy tokens at T={temperature}, "
                         f"lp_inf = lp_oracle + N(0, sigma^2) with sigma = 0.02 (fresh policy) and {stale_sigma} (stale policy: the DPPO mask is active), "
                         f"group-relative advantages of complete groups of {GROUP}",
                 "oracle": "oracle/grpo_oracle.py qwen3_forward on the device: B = reference bf16 rounding points + exact GEMMs; B_fa = B + flash-attention's bf16 P; "
                           "yardsticks against B: C = torch bf16 GEMMs, D = torch bf16 GEMMs + bf16 P (the reference as it runs)"}
    d2 = dataclasses.replace(dims, num_hidden_layers=min(2, dims.num_hidden_layers))
    mbs2 = make_rollouts(n_rollouts_2l, 20260924)
    eng2 = GRPOEngine(d2, device=device, optim=OptimConfig(type="sign_sgd"), loss=LossConfig(), seed=0)  # FIXED weights: the seeded init, untouched by any step
    out["two_layer"] = one_depth(eng2, eng2.store.state_dict(), d2.num_hidden_layers, mbs2, sig)
    out["two_layer"]["weights"] = "fixed: GRPOEngine(seed=0) random init (independent of --steps / --warmup)"
    keep = {k: v for k, v in live_sd.items() if not k.startswith("model.layers.") or int(k.split(".")[2]) < d2.num_hidden_layers}
    eng2.store.load_state_dict(keep)
    out["two_layer_live"] = one_depth(eng2, eng2.store.state_dict(), d2.num_hidden_layers, mbs2, sig)
    out["two_layer_live"]["weights"] = "live: first two layers + embedding + head of the weights the timed steps left"
    del eng2
    torch.cuda.empty_cache()
    out["full_depth"] = one_depth(engine, live_sd, dims.num_hidden_layers, make_rollouts(n_rollouts, 20260924), sig)
    out["full_depth"]["weights"] = "live, full depth — noise-level check only (SE of the scalar loss at this token count is 2-8e-4)"
    t2, tl, tf = out["two_layer"], out["two_layer_live"], out["full_depth"]

    def within(r):  # |loss delta to the nearer oracle (B or B_fa)| <= 1e-4 + 2 SE(yardstick pairs)
        return r["nearer_oracle_abs_delta"] <= r["gate_tol"]

    def flips_ok(r):  # mask decisions that differ from the nearer oracle: <= 1.25 x the reference-as-it-runs' own disagreement with B (+ 3 sigma of that count)
        y = max(r["yardstick_flips"], r["yardstick_bfa_flips"], r["yardstick_d_flips"])
        return min(r["flips"], r["flips_vs_bfa"]) <= 1.25 * y + 3 * math.sqrt(max(y, 1)) + 2

    for r in (tf["fresh"], tf["stale"]):
        r["loss_delta_in_gate_se_units"] = r["nearer_oracle_abs_delta"] / max(r["gate_se"], 1e-12)
    out["note"] = ("two_layer / two_layer_live resolve the north-star tolerance (>= 114k trained tokens: SE <= 1e-4 fresh) and carry the loss gates; full_depth "
                   "(14k tokens, 36 random-init layers) is a noise-level check: per-token noise and mask flips vs the reference restatements' own; "
                   "tests/test_engine_gpu.py resolves the full-depth loss on 229k tokens")
    out["gates"] = {
        "two_layer (fixed weights) fresh: |loss delta to the nearer oracle| <= 1e-4 + 2 SE": within(t2["fresh"]),
        "two_layer (fixed weights) stale: |loss delta to the nearer oracle| <= 1e-4 + 2 SE": within(t2["stale"]),
        "two_layer (live weights) fresh: |loss delta to the nearer oracle| <= 1e-4 + 2 SE": within(tl["fresh"]),
        "two_layer (live weights) stale: |loss delta to the nearer oracle| <= 1e-4 + 2 SE": within(tl["stale"]),
        "per-token noise level <= 1.25 x the reference-as-it-runs' own (D vs B), all legs": all(r["noise_ratio"] <= 1.25 for r in (t2, tl, tf)),
        "flips <= 1.25 x yardstick + 3 sigma (all legs, stale policy)": all(flips_ok(r["stale"]) for r in (t2, tl, tf)),
        "mask decisions near the threshold exist": all(r["stale"]["tokens_within_0.02_of_threshold"] > 0 for r in (t2, tl, tf)),
        "K2 masked count == host formula on the engine's log-probs": all(r[k]["masked_tokens_engine"] == r[k]["masked_tokens_engine_host_formula"]
                                                                         for r in (t2, tl, tf) for k in ("fresh", "stale")),
    }
    if detail_path is not None:  # the verbose per-leg blocks go to a side file; the JSON line keeps the gates and one summary row per leg
        try:
            Path(detail_path).parent.mkdir(parents=True, exist_ok=True)
            Path(detail_path).write_text(json.dumps(out, indent=1))
        except OSError:
            detail_path = None

    def brief(r):
        return {"trained_tokens": r["trained_tokens"], "noise_ratio": round(r["noise_ratio"], 3), "logprob_mean_shift": round(r["logprob_mean_shift"], 7),
                "logprob_mean_shift_se": round(r["logprob_mean_shift_se"], 7), "yardstick_d_logprob_mean_shift": round(r["yardstick_d_logprob_mean_shift"], 7),
                **{f"{k}_{f}": (round(r[k][f], 7) if isinstance(r[k][f], float) else r[k][f]) for k in ("fresh", "stale")
                   for f in ("loss_delta", "loss_delta_vs_bfa", "gate_se", "gate_tol", "flips", "flips_vs_bfa", "yardstick_d_flips", "masked_tokens_oracle", "masked_tokens_engine")}}

    return {"gates": out["gates"], "two_layer": brief(t2), "two_layer_live": brief(tl), "full_depth_noise_check": brief(tf), "note": out["note"],
            "detail_file": str(detail_path) if detail_path is not None else None, "_full": out}


def finalize_line(result: dict) -> dict:
    """The driver keeps only SCALAR keys of `config` and the last 8 KB of stdout: hoist every number that matters to a scalar top-level
    `config` key, write the verbose blocks (notes, shapes, per-leg parity tables) to gpurun_out/bench_detail.json, and keep the printed line small."""
    cfg = result.get("config", {})
    extra = cfg.get("extra") or {}

    def _num(v):
        return v if isinstance(v, (int, float, str, bool)) or v is None else None

    hoist = {
        "all_own_ms": _dig(extra, ("all_own", "ms_per_step")), "library_mix_ms": _dig(extra, ("library_mix", "ms_per_step")),
        "forced_collectives_ms": _dig(extra, ("forced_collectives", "ms_per_step")), "fc_delta_ms": _dig(extra, ("forced_collectives", "delta_ms_vs_headline")),
        "fc_exposed_ms": _dig(extra, ("forced_collectives", "comm", "exposed_ms")), "fc_rs_ms": _dig(extra, ("forced_collectives", "comm", "rs_ms")),
        "fc_ag_ms": _dig(extra, ("forced_collectives", "comm", "ag_ms")),
        "mb2_ms": _dig(extra, ("mb2", "ms_per_step")), "mb2_samples_per_s": _dig(extra, ("mb2", "samples_per_s")),
        "moe_step_ms": _dig(extra, ("moe_step", "ms_per_step")), "moe_step_mfu": _dig(extra, ("moe_step", "mfu_vs_2.5PF_bf16")),
        "moe_layer_ms": _dig(extra, ("moe_layer", "ms")),
        "long_context_ms": _dig(extra, ("long_context", "r1", "ms_per_step")), "long_context_mfu_ref": _dig(extra, ("long_context", "r1", "mfu_ref_formula")),
        "long_context_mfu_executed": _dig(extra, ("long_context", "r1", "mfu_executed_causal")),
        "long_context_attn_share": _dig(extra, ("long_context", "r1", "attention_share")), "long_context_peak_hbm_gb": _dig(extra, ("long_context", "r1", "peak_hbm_gb")),
        "long_context_selective_ms": _dig(extra, ("long_context", "r1_selective", "ms_per_step")), "long_context_selective_mfu_ref": _dig(extra, ("long_context", "r1_selective", "mfu_ref_formula")),
        "long_context_selective_freq2_ms": _dig(extra, ("long_context", "r1_selective_freq2", "ms_per_step")),
        "long_context_selective_freq2_mfu_ref": _dig(extra, ("long_context", "r1_selective_freq2", "mfu_ref_formula")),
        "long_context_x4_ms": _dig(extra, ("long_context", "r4", "ms_per_step")), "long_context_x4_mfu_executed": _dig(extra, ("long_context", "r4", "mfu_executed_causal")),
        "long_context_attn_fwd_frac": _dig(extra, ("long_context", "attn_fwd_frac_of_2.5PF")), "long_context_attn_bwd_frac": _dig(extra, ("long_context", "attn_bwd_frac_of_2.5PF")),
        "j1_frac": _dig(extra, ("j1_roofline", "frac")), "gemm_wgrad_frac": _dig(extra, ("gemm_roofline", "frac")),
        "weight_push_ms": _dig(extra, ("weight_push", "ms")),
    }
    par = result.get("parity") or {}
    gates = dict(par.get("gates") or {})
    rgates = dict(_dig(par, ("realistic", "gates")) or {})
    hoist["parity_gates_all_true"] = (bool(gates) and all(gates.values()) and bool(rgates) and all(rgates.values())) if par else None
    hoist["parity_kernel_gates_all_true"] = (all(gates.values()) if gates else None)
    hoist["parity_realistic_gates_all_true"] = (all(rgates.values()) if rgates else None)
    for leg in ("two_layer", "two_layer_live"):
        for pol in ("fresh", "stale"):
            hoist[f"parity_{leg}_{pol}_loss_delta"] = _dig(par, ("realistic", leg, f"{pol}_loss_delta"))
            hoist[f"parity_{leg}_{pol}_tol"] = _dig(par, ("realistic", leg, f"{pol}_gate_tol"))
    for k, v in hoist.items():
        cfg[k] = _num(v)
    # side file: everything, verbatim
    try:
        DETAIL_DIR.mkdir(parents=True, exist_ok=True)
        (DETAIL_DIR / "bench_detail.json").write_text(json.dumps(result, indent=1))
        cfg["detail_file"] = "gpurun_out/bench_detail.json"
    except OSError:
        cfg["detail_file"] = None

    def strip(d):  # the printed line: drop prose, keep numbers
        if isinstance(d, dict):
            return {k: strip(v) for k, v in d.items() if k not in ("note", "pattern", "shape", "model", "kernel", "lp_inf", "traffic_source", "k2_loss_note", "source", "data", "oracle")
                    or not isinstance(v, str) or len(v) <= 48}
        return d

    # the printed `extra` keeps what the tests read and what has no hoisted twin; everything else is the scalar keys above + the side file
    slim = {k: strip(v) for k, v in extra.items() if k in ("mb2", "weight_push", "samples_per_s_incl_serial_weight_push", "error")}
    for k, v in extra.items():
        if isinstance(v, dict) and "error" in v:
            slim[k] = v  # failures are REPORTED in the line, never hidden in the side file
    cfg["extra"] = slim
    if par:
        par = strip(par)
        rp = par.get("realistic")
        if isinstance(rp, dict) and "gates" in rp:
            keep = ("trained_tokens", "noise_ratio", "logprob_mean_shift", "yardstick_d_logprob_mean_shift", "fresh_loss_delta", "fresh_loss_delta_vs_bfa", "fresh_gate_tol",
                    "stale_loss_delta", "stale_loss_delta_vs_bfa", "stale_gate_tol", "stale_flips", "stale_yardstick_d_flips")
            par["realistic"] = {"gates": rp["gates"], "detail_file": "gpurun_out/bench_parity_realistic.json",
                                **{leg: {k: rp[leg][k] for k in keep if k in rp[leg]} for leg in ("two_layer", "two_layer_live", "full_depth_noise_check") if leg in rp}}
        result["parity"] = par
    if isinstance(result.get("roofline"), dict) and isinstance(result["roofline"].get("traffic_source"), str):
        result["roofline"]["traffic_source"] = "profiles/k1_hbm_traffic.json (recorded rocprofv3 --pmc profile of this kernel at this launch shape; not a live reading)"
    if isinstance(result.get("comm"), dict):
        result["comm"] = strip(result["comm"])
    return result


def _dig(d, keys):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


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
    ap.add_argument("--recompute", default="none", choices=["none", "layer", "selective"],
                    help="layer = block-level activation recomputation (the reference's default model.ac): keep layer inputs only, re-run each layer's forward in backward")
    ap.add_argument("--recompute-freq", type=int, default=1, help="the reference's ac.freq: recompute only every N-th layer, the others keep their activations")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (device copies taken in one extra untimed step after the timed region)")
    ap.add_argument("--no-realistic-parity", action="store_true", help="skip parity.realistic (one extra untimed step on rollout-like data vs the oracle forward)")
    ap.add_argument("--parity-rollouts", type=int, default=64, help="rollouts of the two-layer legs of parity.realistic (64 x 2048 -> 114 688 trained tokens)")
    ap.add_argument("--no-long-context", dest="long_context", action="store_false", help="skip config.extra.long_context (32k-token rollouts, recompute=layer)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements reported under config.extra (2 micro-batches/GPU line, weight push)")
    args = ap.parse_args()

    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run exactly as the driver would launch it (one rank per GPU,
        # rendezvous on 127.0.0.1 — the container hostname may not resolve)
        import socket

        import random

        port = None
        for _ in range(200):  # below the kernel's ephemeral range: a port from bind(0) can be handed to an outgoing connection before the launcher listens
            cand = random.randrange(20000, 32000)
            with socket.socket() as sk:
                sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    sk.bind(("127.0.0.1", cand))
                except OSError:
                    continue
            port = cand
            break
        if port is None:
            raise SystemExit("no free rendezvous port between 20000 and 32000")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(Path(__file__).resolve())] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-launching as `{' '.join(cmd[1:9])} bench.py ...`", file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus})")
    # PM_BENCH_DRYRUN_SHARED_GPU=1: rehearsal of the N > 1 launch on a ONE-GPU box — every rank uses device 0 and the group runs over gloo (RCCL refuses
    # two ranks on one device).  Same script, same engine, same barriers / reductions / extras; the line it prints carries config.diagnostic and is
    # not a measurement (tests/test_multirank_one_gpu.py runs it).
    dryrun = os.environ.get("PM_BENCH_DRYRUN_SHARED_GPU") == "1"
    if dryrun:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():  # preflight: every rank needs its own visible GPU
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible — one rank per GPU (PM_BENCH_DRYRUN_SHARED_GPU=1 "
                         "rehearses the launch on one GPU over gloo)")
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
                        recompute=args.recompute, recompute_freq=args.recompute_freq)
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
    if dist.is_initialized() and not (args.force_collectives and world == 1):
        # preflight: the group the timed region runs on has exactly one rank per requested GPU, and the engine shards over all of them
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
        assert engine.dp_size == world and engine.use_collectives == (world > 1), (engine.dp_size, engine.use_collectives)
        if not dryrun:
            assert dist.get_backend() == "nccl" and torch.cuda.device_count() >= world, (dist.get_backend(), torch.cuda.device_count())
    if rank == 0 and dist.is_initialized():
        print(f"[bench preflight] RCCL group up: world={dist.get_world_size()} backend={dist.get_backend()} dp={engine.dp_size} visible_gpus={torch.cuda.device_count()} "
              f"collectives={'reduce-scatter/all-gather on a side stream' if engine.use_collectives else 'none (single rank)'}", file=sys.stderr, flush=True)
    from prime_rl_amd.trainer import zero

    engine.k1_events = []
    zero.COMM_EVENTS = [] if engine.use_collectives else None
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        out = engine.step(batches[s])
        losses.append(out["loss"])
    barrier()
    elapsed = time.perf_counter() - t0
    k1_events, engine.k1_events = engine.k1_events, None
    comm_events, zero.COMM_EVENTS = zero.COMM_EVENTS, None
    # host cost of enqueuing ONE step on an EMPTY queue: an extra, untimed step after the timed region (inside it the HIP queue fills up and
    # the host blocks on back-pressure, so wall time there says nothing about the host; the first warm-up step is cold)
    th = time.perf_counter()
    engine.step(batches[-1])
    host_enqueue_empty_queue = time.perf_counter() - th
    barrier()
    # the parity probe (device copies of what the gates need: an embedding-gather compare, label gathers, clones) rides on ONE MORE untimed step of the
    # same engine on the same kind of batch — nothing that is not part of the step runs inside the timed region
    probe = None
    if rank == 0 and engine.lm_head_impl == "library" and not args.no_parity:
        engine.parity_probe = {}
    engine.step(batches[-1])
    barrier()
    probe, engine.parity_probe = engine.parity_probe, None
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples = args.micro_batches * world * args.steps
    # everything the JSON line needs from the device is read NOW, on this thread: the watchdog below must be able to print the line while
    # the main thread sits in a wedged collective (a .item() or an event query from there would queue behind it)
    k1_durs = [a.elapsed_time(b) * 1e-3 for a, b, _, _ in k1_events] if k1_events else []
    comm = {"rs_ms": 0.0, "ag_ms": 0.0, "exposed_ms": 0.0}
    for kind, e0, e1 in (comm_events or []):
        comm[f"{kind}_ms"] += e0.elapsed_time(e1)
    comm = {k: round(v / args.steps, 3) for k, v in comm.items()}
    comm["note"] = ("per step, this rank: rs / ag = busy time of the comm stream in gradient reduce-scatters / parameter all-gathers (HIP events on that stream), "
                    "exposed = time the compute stream stood still waiting for them" if engine.use_collectives else
                    "N=1 without --force-collectives: no collective is issued (the step's only exchange degenerates to local copies)")
    rccl = {"world": world, "backend": (dist.get_backend() if dist.is_initialized() else None)}
    try:
        from prime_rl_amd import _lib as _pm_lib

        ver = _pm_lib.load().pm_comm_rccl_version() if _pm_lib.load().pm_comm_available() else 0
        rccl["version"] = f"{ver // 10000}.{ver // 100 % 100}.{ver % 100}" if ver else None
    except Exception:
        rccl["version"] = None
    loss_last = float(sum(l.item() for l in losses[-1]))
    engine_lm_head, engine_gemm_own = engine.lm_head_impl, sorted(engine.gemm_own)

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
                "lm_head": {"library": "stored_logits", "fused": "fused_j1"}.get(engine_lm_head, engine_lm_head),
                "gemm_own": ",".join(engine_gemm_own) or "none",
                "rccl_world": rccl["world"], "rccl_backend": rccl["backend"], "rccl_version": rccl["version"],
                "extra": extra,
            },
        }
        result["comm"] = comm
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
                "min_launch_ms": round(min(durs) * 1e3, 4), "max_launch_ms": round(max(durs) * 1e3, 4),
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
            # (one GPU: no collective can wedge, the budget only guards against a box several times slower than measured — the extras take 75-90 s)
            if not done.wait(float(os.environ.get("PM_BENCH_EXTRA_TIMEOUT", "360" if world == 1 else "180"))):  # (headline + roofline) and EVERY rank leaves
                if rank == 0:
                    line = assemble({"error": "an extra measurement did not finish in time; headline and roofline only"})
                    line["cpu_baseline"] = None
                    print(json.dumps(finalize_line(line)), flush=True)
                else:
                    time.sleep(3.0)
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        t_extra = time.perf_counter()
        extra = measure_extras(args, engine, loader, device, world, rank, barrier)
        done.set()
        if rank == 0:
            print(f"[bench] extras took {time.perf_counter() - t_extra:.1f} s", file=sys.stderr, flush=True)

    if rank == 0:
        result = assemble(extra)
        wp = extra.get("weight_push") or {}
        if "ms" in wp:  # samples/s excluding (= value) and including the weight push, if the push were NOT overlapped
            extra["samples_per_s_incl_serial_weight_push"] = round(samples / (elapsed + args.steps * wp["ms"] * 1e-3), 4)
        if probe:
            try:
                result["parity"] = parity_block(probe, dims, args.seq_len)
            except Exception as e:  # pragma: no cover - reported, never hidden
                result["parity"] = {"error": repr(e)[:300]}
            if not args.no_realistic_parity and world == 1 and dims.head_dim == dims.kernel_head_dim:
                try:  # rollout-like data against the oracle's forward: 2 layers on fixed + live weights (resolvable loss gates), full depth (noise check)
                    rp = realistic_parity(engine, dims, device, args.seq_len, n_rollouts=min(8, args.micro_batches), n_rollouts_2l=args.parity_rollouts,
                                          detail_path=DETAIL_DIR / "bench_parity_realistic.json")
                    rp.pop("_full", None)
                    result["parity"]["realistic"] = rp
                except Exception as e:  # pragma: no cover
                    result["parity"]["realistic"] = {"error": repr(e)[:300]}
        else:
            result["parity"] = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_trainer import time_cpu_baseline

            cpu = time_cpu_baseline(dims, seq_len=args.seq_len, micro_batches_per_step=args.micro_batches)
            result["cpu_baseline"] = {
                "value": round(1.0 / cpu["seconds_per_sample"], 6),
                "unit": "samples/s",
                "cores": cpu["threads"],
                "kind": "port",
                "cpu_model": cpu["cpu_model"],
                "sample": f"oracle/cpu_trainer.py (f32 torch-CPU port of the reference step), every leg one warm-up + median of 3, on a {cpu['sample_tokens']}-token rollout "
                          f"prefix: {cpu['slice_layers']}-layer full-width slice fwd+bwd {cpu['t_slice']:.2f}s (x{dims.num_hidden_layers}/{cpu['slice_layers']} layers), full-vocab LM "
                          f"head + log-probs + loss fwd+bwd {cpu['t_head_loss']:.2f}s, both scaled linearly to {args.seq_len} tokens (ignores quadratic attention: favours the "
                          f"CPU); AdamW on {cpu['opt_params'] >> 20}Mi params {cpu['t_opt']:.2f}s (x params/{cpu['opt_params'] >> 20}Mi = {cpu['t_opt_full_model']:.1f}s per "
                          f"optimizer step, charged 1/{cpu['micro_batches_per_step']} per sample: one step per {cpu['micro_batches_per_step']} micro-batches, train.py:566-570); "
                          f"{cpu['threads']} of {cpu['cores_available']} usable cores ({cpu['cpu_model']})",
            }
        if world == 1 and not args.force_collectives and not args.no_extra and not dryrun and args.recompute == "none":
            # the N > 1 CODE PATH on one GPU, driver-visible (VERDICT r4 'next' 5): a second engine with force_collectives=True over a world_size-1 RCCL
            # group — per-bucket reduce-scatter / all-gather on the comm stream, sharded norm + AdamW, per-tile GEMM grid — timed for a few steps after
            # everything else (the headline engine is released first: two 8B engines do not fit one GPU).  Guarded by its own watchdog.
            import gc
            import threading

            headline = json.dumps(result)
            fc_done = threading.Event()

            def fc_watchdog():
                if not fc_done.wait(float(os.environ.get("PM_BENCH_FC_TIMEOUT", "240"))):
                    line = json.loads(headline)
                    line["config"]["extra"]["forced_collectives"] = {"error": "the forced-collectives leg did not finish in time"}
                    print(json.dumps(finalize_line(line)), flush=True)
                    os._exit(0)

            threading.Thread(target=fc_watchdog, daemon=True).start()
            try:
                del engine
                probe = None
                gc.collect()
                torch.cuda.empty_cache()
                result["config"]["extra"]["moe_step"] = moe_step_extra(device)
                gc.collect()
                torch.cuda.empty_cache()
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29531")
                dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)
                from prime_rl_amd.trainer import zero as _zero

                eng_fc = GRPOEngine(dims, device=device, optim=OptimConfig(), loss=LossConfig(), seed=0, force_collectives=True)
                kfc = max(2, min(args.steps, 3))
                eng_fc.step(batches[0])
                barrier()
                _zero.COMM_EVENTS = []
                t0 = time.perf_counter()
                for i in range(kfc):
                    eng_fc.step(batches[(1 + i) % len(batches)])
                barrier()
                dt = time.perf_counter() - t0
                evs, _zero.COMM_EVENTS = _zero.COMM_EVENTS, None
                cfc = {"rs_ms": 0.0, "ag_ms": 0.0, "exposed_ms": 0.0}
                for kind, e0, e1 in evs:
                    cfc[f"{kind}_ms"] += e0.elapsed_time(e1)
                result["config"]["extra"]["forced_collectives"] = {
                    "steps": kfc, "ms_per_step": round(dt / kfc * 1e3, 3), "samples_per_s": round(args.micro_batches * kfc / dt, 4),
                    "delta_ms_vs_headline": round(dt / kfc * 1e3 - elapsed / args.steps * 1e3, 3),
                    "comm": {k: round(v / kfc, 3) for k, v in cfc.items()}, "gemm_variant": getattr(eng_fc, "_gemm_variant", None),
                    "note": "the multi-GPU code path at N = 1: RCCL reduce-scatter (f32 gradients, one bucket per decoder layer) / all-gather (bf16 parameters) of a "
                            "world_size-1 group on the comm stream, norm over the reduced shard, per-tile GEMM grid; comm = busy time of the comm stream per step and the "
                            "time the compute stream stood still waiting for it.  What it cannot show is link time: at world 1 a collective is a local copy"}
                del eng_fc
            except Exception as e:  # pragma: no cover
                result["config"]["extra"]["forced_collectives"] = {"error": repr(e)[:300]}
            fc_done.set()
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
        print(json.dumps(finalize_line(result)), flush=True)


if __name__ == "__main__":
    main()
