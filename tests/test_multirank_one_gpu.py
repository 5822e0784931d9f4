"""GPU suite: the MULTI-RANK engine with the real HIP kernels on ONE MI355X — N processes share `cuda:0` and talk over a gloo group (gloo runs every
collective the engine uses on device tensors).  What the world_size-2 CPU tests cover with stand-in kernels and the forced-collectives test covers at
world 1 meets here: per-bucket reduce-scatter on the comm stream behind backward, the split embedding-bucket reduction (early dense part + row
all-gather + pm_embedding_bwd_window), sharded clip + AdamW, per-bucket parameter all-gather with gather events, Ulysses all-to-all around the
attention kernels, and cp_style "ring" on the row-window attention kernels (pm_attn_varlen_*_rows) with K / V gathered and dK / dV reduce-scattered —
compared with the single-process engine on the same global batch.  (RCCL itself refuses two ranks on one device; what an 8-GPU node adds is the
transport, not the schedule.)"""

import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LR, STEPS, MB_PER_RANK = 1e-4, 2, 2


def _free_port():
    from tests._ports import free_port

    return free_port()


def _dims(tied):
    import dataclasses

    from prime_rl_amd.trainer.engine import ModelDims

    return dataclasses.replace(ModelDims.tiny(layers=2, vocab=512), tie_word_embeddings=tied)


def _batches(dp, dp_rank, step, seq):
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=MB_PER_RANK * dp, generate_samples=True), seq_len=seq, dp_world_size=dp, dp_rank=dp_rank)
    dl.batch_counter = step
    mbs = dl.get_batch()
    g = torch.Generator().manual_seed(1000 * dp_rank + step)
    for mb in mbs:
        mb["input_ids"] = mb["input_ids"] % 512
        mb["inference_logprobs"] = -torch.rand(1, seq, generator=g) * 0.1 - 6.2  # near log(1/512)
    return mbs


def _worker(rank, world, port, out_dir, cp, cp_style, tied, seq):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prime_rl_amd.trainer.engine import GRPOEngine, OptimConfig, pack_micro_batches
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=cp, pp=1, ep=1, world_size=world)
        dp = world // cp
        kw = dict(dp_group=pd.get_group("dp"), dp_size=pd.dp_size)
        if cp > 1:
            kw.update(cp_group=pd.get_group("cp"), cp_size=cp, shard_group=pd.get_group("dp_cp"), cp_style=cp_style)
        eng = GRPOEngine(_dims(tied), device=DEV, optim=OptimConfig(lr=LR), seed=5, **kw)
        assert eng.use_collectives and eng.comm_stream is not None and eng.embed_split and eng.shard_size == world
        rec = {"loss": [], "grad_norm": []}
        for step in range(STEPS):
            mbs = _batches(dp, rank // cp, step, seq)
            passes = [pack_micro_batches(mbs[:1], DEV), pack_micro_batches(mbs[1:], DEV)]
            out = eng.step(passes)
            eng.wait_params_ready()
            torch.cuda.synchronize()
            rec["loss"].append(sum(l.item() for l in out["loss"]))
            rec["grad_norm"].append(out["grad_norm"].item())
            if step == 0:
                rec["shards"] = eng.store.grad_shard.cpu()
        assert eng._embed_bufs is not None
        rec["params"] = eng.store.params.cpu()
        torch.save(rec, os.path.join(out_dir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cp,cp_style,tied,seq", [(2, 1, "ulysses", False, 128), (2, 1, "ulysses", True, 128), (2, 2, "ulysses", False, 128),
                                                        (2, 2, "ring", False, 256),  # the row-window kernels take 128-token multiples per cp rank
                                                        (4, 2, "ring", False, 256),
                                                        # BASELINE.json configs[4]'s rollout length through both context-parallel styles: every pass is ONE
                                                        # 32 768-token sample, cp rank 1's window starts at row 16 384
                                                        (2, 2, "ulysses", False, 32768), (2, 2, "ring", False, 32768)])
def test_multi_rank_engine_on_one_gpu_equals_single_process(tmp_path, world, cp, cp_style, tied, seq):
    from prime_rl_amd.trainer.engine import GRPOEngine, OptimConfig, ParamStore, pack_micro_batches

    dp = world // cp
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), cp, cp_style, tied, seq), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(world)]
    eng = GRPOEngine(_dims(tied), device=DEV, optim=OptimConfig(lr=LR), seed=5)
    ref = {"loss": [], "grad_norm": []}
    for step in range(STEPS):
        passes = []
        for k in range(dp):
            mbs = _batches(dp, k, step, seq)
            passes += [pack_micro_batches(mbs[:1], DEV), pack_micro_batches(mbs[1:], DEV)]
        out = eng.step(passes)
        ref["loss"].append(sum(l.item() for l in out["loss"]))
        ref["grad_norm"].append(out["grad_norm"].item())
        if step == 0:
            ref["grad"] = eng.store.main_grad.cpu()
    torch.cuda.synchronize()
    for step in range(STEPS):
        got = sum(r[k * cp]["loss"][step] for k in range(dp))  # one loss per dp position (cp ranks evaluate the same full-pass loss)
        assert got == pytest.approx(ref["loss"][step], rel=2e-3 if step == 0 else 2e-2, abs=1e-4), (step, got, ref["loss"][step])
        for k in range(world):
            assert r[k]["grad_norm"][step] == pytest.approx(r[0]["grad_norm"][step], rel=1e-6)  # every rank holds the GLOBAL norm
            assert r[k]["loss"][step] == pytest.approx(r[(k // cp) * cp]["loss"][step], rel=1e-5)
        assert r[0]["grad_norm"][step] == pytest.approx(ref["grad_norm"][step], rel=5e-3 if step == 0 else 5e-2)
    # reduce-scattered shards side by side == the single-process gradient, parameter by parameter (layouts differ: buckets pad to 256 * world)
    lay = ParamStore(eng.dims, "cpu", dp_rank=0, dp_size=world)
    off1 = {name: off for b in eng.store.buckets for name, off, _ in b.params}
    p_ref = eng.store.params.cpu()
    for b in lay.buckets:
        per = b.size // world
        grad_b = torch.cat([r[k]["shards"][b.offset // world: b.offset // world + per] for k in range(world)])
        for name, off, shape in b.params:
            n = math.prod(shape)
            got, want = grad_b[off - b.offset: off - b.offset + n], ref["grad"][off1[name]: off1[name] + n]
            rel = ((got - want).norm() / (want.norm() + 1e-12)).item()
            assert rel < 2e-2, (name, rel)
            diff = (r[0]["params"][off: off + n].float() - p_ref[off1[name]: off1[name] + n].float()).abs()
            assert float(diff.max()) <= 25 * LR and float((diff > 0).float().mean()) < 0.1, (name, float(diff.max()), float((diff > 0).float().mean()))
    assert all(torch.equal(r[0]["params"], r[k]["params"]) for k in range(1, world))  # replicas in lock-step after the in-place all-gather


def _moe_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
        from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
        from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
        from prime_rl_amd.trainer.module_engine import ModuleEngine
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        cfg = Qwen3MoeConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                             num_experts=8, num_experts_per_tok=2, moe_intermediate_size=128)
        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=1, pp=1, ep=2, world_size=world)
        ep_group, replica = pd.get_group("ep"), pd.get_group("ep_replica")
        torch.manual_seed(11)
        with torch.device(DEV):
            full = Qwen3MoeForCausalLM(cfg)
        for p in full.parameters():
            torch.nn.init.normal_(p, std=0.05)
        for n_, p in full.named_parameters():
            if n_.endswith("norm.weight") or n_.endswith("layernorm.weight"):
                torch.nn.init.ones_(p)
        full = full.to(torch.bfloat16)
        with torch.device(DEV):
            local = Qwen3MoeForCausalLM(cfg, ep_group=ep_group)
        local = local.to(torch.bfloat16)
        er, le = dist.get_rank(ep_group), cfg.num_experts // 2
        local.load_state_dict({k: (v[er * le:(er + 1) * le].clone() if ".experts.w" in k else v.clone()) for k, v in full.state_dict().items()})
        inject_prime_lm_head(full, chunk_size=128)
        inject_prime_lm_head(local, chunk_size=128)
        eng = ModuleEngine(local, DEV, optim=OptimConfig(lr=1e-3, max_norm=0.5), dp_group=pd.get_group("dp"), dp_size=world, ep_group=ep_group, ep_size=2,
                           ep_replica_group=replica)
        ref = ModuleEngine(full, DEV, optim=OptimConfig(lr=1e-3, max_norm=0.5))
        for step in range(2):
            mine = _batches(world, rank, step, 128)
            out = eng.step([pack_micro_batches(mine[:1], DEV), pack_micro_batches(mine[1:], DEV)])
            want = ref.step([pack_micro_batches(_batches(world, r, step, 128)[i:i + 1], DEV) for r in range(world) for i in range(2)])
            loss = torch.stack([x.reshape(()) for x in out["loss"]]).sum()
            dist.all_reduce(loss)
            want_loss = sum(x.item() for x in want["loss"])
            assert abs(loss.item() - want_loss) < (3e-4 if step == 0 else 3e-3), (rank, step, loss.item(), want_loss)
            gn, wn = out["grad_norm"].item(), want["grad_norm"].item()
            assert abs(gn - wn) / wn < (1e-2 if step == 0 else 5e-2), (rank, step, gn, wn)
            if step == 0:  # same weights: this rank's log-probs are those of the model that holds every expert (the dispatch moves rows, not values)
                d = (torch.cat(out["logprobs"]) - torch.cat(want["logprobs"][2 * rank: 2 * rank + 2])).abs()
                assert float(d.max()) == 0.0, (rank, float(d.max()))
        sd, sd_ref = eng.state_dict(), ref.state_dict()
        assert set(sd) == set(sd_ref) and "model.layers.1.mlp.experts.7.up_proj.weight" in sd
        for k in sd:
            da = (sd[k].float() - sd_ref[k].float()).abs()
            assert float(da.max()) <= 5e-3, (rank, k, float(da.max()))  # two AdamW steps of lr 1e-3 each, a few flipped updates
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_expert_parallel_module_engine_on_one_gpu(world):
    """The MoE trainer step under expert parallelism with the real router / sort / grouped-GEMM kernels and the all-to-all dispatch between processes
    that share the GPU: ep = 2 (world 4: two ep groups, expert gradients summed over `ep_replica`), against the engine that holds every expert."""
    mp.start_processes(_moe_worker, args=(world, _free_port(), ""), nprocs=world, join=True, start_method="spawn")


def test_bench_script_multi_rank_launch_rehearsal(tmp_path):
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`) with PM_BENCH_DRYRUN_SHARED_GPU=1: both ranks share the GPU over gloo, everything else is the script as it runs on a node — warm-up,
    barrier-bracketed timed region, max-over-ranks, the extras (2-micro-batch line, the sharded weight push between the ranks), ONE JSON line from rank 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PM_BENCH_DRYRUN_SHARED_GPU="1", PM_BENCH_EXTRA_TIMEOUT="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "tiny", "--seq-len", "256", "--tokens-per-pass", "512",
           "--micro-batches", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["value"] == pytest.approx(4 * 2 * 2 / (rec["ms_per_step"] * 2 / 1e3), rel=1e-3)  # whole-job samples / max-over-ranks time
    assert "rehearsal" in rec["config"]["diagnostic"] and rec["roofline"]["frac"] > 0 and rec.get("cpu_baseline") is None
    extra = rec["config"]["extra"]
    assert "samples_per_s" in extra["mb2"] and extra["weight_push"]["bytes_sent_per_rank"] > 0, extra
    assert "RCCL group up: world=2" in out.stderr or "world=2" in out.stderr


def test_bench_script_launches_itself_when_no_launcher_is_present(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-execs itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port <free> bench.py --gpus 2 ...` (VERDICT r4 'next' 5) instead of exiting; with PM_BENCH_DRYRUN_SHARED_GPU=1 the two
    ranks share the GPU over gloo.  Same single JSON line as under the driver's own launcher; the preflight line names the world."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.update(PM_BENCH_DRYRUN_SHARED_GPU="1", PM_BENCH_EXTRA_TIMEOUT="240")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "tiny", "--seq-len", "256", "--tokens-per-pass", "512",
           "--micro-batches", "4", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "re-launching as" in out.stderr and "--nproc-per-node=2" in out.stderr, out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["config"]["rccl_world"] == 2
    assert "world=2" in out.stderr


def test_bench_script_refuses_a_world_that_does_not_match_gpus():
    """--gpus N must equal WORLD_SIZE when a launcher set it (one rank per GPU): a mismatch exits with a message instead of timing the wrong job."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--model", "tiny"], capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    assert out.returncode != 0 and "one rank per GPU" in (out.stderr + out.stdout)


@pytest.mark.parametrize("model,extra", [("tiny", []), ("tiny-moe", ["--model.ep", "2"]), ("tiny", ["--model.cp", "2", "--model.cp-style", "ring"]),
                                         ("tiny", ["--model.cp", "2", "--model.cp-style", "ulysses"])])
def test_train_cli_two_ranks_on_one_gpu(tmp_path, model, extra):
    """`torchrun --nproc-per-node 2 -m prime_rl_amd.trainer.rl.train @ rl.toml …` (the reference's launch grammar) with PRIME_TRAINER_SHARED_GPU=1: the
    whole train loop at world 2 — dense model: ZeRO-2 over dp = 2; MoE model: expert parallel over the two ranks — data-parallel FakeDataLoader shards,
    step statistics gathered over the ranks on the device, filesystem weight broadcast (the MoE state dict assembled collectively), metrics from rank 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    from safetensors.torch import load_file

    root = Path(__file__).resolve().parent.parent
    toml = tmp_path / "rl.toml"
    toml.write_text('max_steps = 2\n\n[data.fake]\nbatch_size = 8\ngenerate_samples = true\n\n[model]\nseq_len = 128\n\n[optim]\nlr = 1e-3\n')
    env = dict(os.environ, PRIME_TRAINER_SHARED_GPU="1", PYTHONPATH=str(root))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "prime_rl_amd.trainer.rl.train", "@", str(toml), "--model.name", model, "--model.debug.random-init", "--output-dir", str(tmp_path),
           "--tokens-per-pass", "256", "--monitors.file.path", "metrics.jsonl", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    recs = [json.loads(ln) for ln in (tmp_path / "metrics.jsonl").read_text().strip().splitlines()]
    assert len(recs) == 2 and all(math.isfinite(r[k]) for r in recs for k in ("loss/mean", "entropy/all/mean", "optim/grad_norm", "perf/throughput"))
    assert "Step 2 |" in out.stdout
    # the throughput counts the whole job (with cp = 2 both ranks work on the same micro-batches: one dp position)
    assert recs[0]["perf/throughput_per_gpu"] == pytest.approx(recs[0]["perf/throughput"] / 2, rel=1e-6)
    sd = load_file(str(tmp_path / "broadcasts" / "step_2" / "model.safetensors"))
    if model == "tiny-moe":
        assert "max_vio/mean" in recs[0] and sum(1 for k in sd if ".mlp.experts." in k and k.startswith("model.layers.0.")) == 8 * 3  # ALL experts, HF names
    else:
        assert "model.layers.1.self_attn.q_proj.weight" in sd and "lm_head.weight" in sd
