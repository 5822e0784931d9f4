"""CPU suite: the trainer's step statistics (prime_rl_amd/trainer/stats.py) reproduce the REFERENCE's `Tensors.compute_stats`
value for value — tests/golden/step_stats.json holds what the reference's own class (src/prime_rl/trainer/utils.py:238-281,
file-loaded by tests/golden/make_step_stats.py) returned for these inputs at world_size 1 and world_size 2: unbiased std, NaN for
an empty key and for a one-element std, sorted key union over ranks, all-rank gather before reducing."""

import json
import math
import os
import socket
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = json.loads((Path(__file__).parent / "golden" / "step_stats.json").read_text())["cases"]


def _fill(tensors, inputs):
    for key, vecs in inputs.items():
        for v in vecs:
            tensors[key].append(torch.tensor(v, dtype=torch.float32))


def _assert_equal(got: dict, expected: dict):
    assert sorted(got) == sorted(expected)
    for k, e in expected.items():
        g = got[k]
        if e is None:
            assert math.isnan(g), (k, g)
        else:
            assert g == pytest.approx(e, rel=1e-6, abs=1e-7), (k, g, e)


def test_compute_stats_world1_equals_reference():
    from prime_rl_amd.trainer.stats import Tensors

    case = GOLD["world1"]
    t = Tensors()
    _fill(t, case["inputs"][0])
    _assert_equal(t.compute_stats(), case["expected"])
    assert not t  # keys are consumed, like the reference's pop()


def test_five_stats_is_unbiased_and_nan_aware():
    from prime_rl_amd.trainer.stats import five_stats

    x = torch.tensor([1.0, 2.0, 4.0, 8.0])
    s = five_stats(x)
    assert s[2].item() == pytest.approx(x.std(unbiased=True).item()) and s[2].item() != pytest.approx(x.std(unbiased=False).item())
    assert s[1].item() == 2.0  # torch.median = LOWER median (utils.py:273)
    assert torch.isnan(five_stats(torch.empty(0))).all()
    one = five_stats(torch.tensor([3.0]))
    assert math.isnan(one[2].item()) and one[0].item() == 3.0


def _free_port():
    from tests._ports import free_port

    return free_port()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from prime_rl_amd.trainer.stats import Tensors, flexible_all_gather

    t = Tensors()
    _fill(t, GOLD["world2"]["inputs"][rank])
    got = t.compute_stats()
    fg = flexible_all_gather(torch.arange(3 + 2 * rank, dtype=torch.float32) + 10 * rank)
    torch.save({"stats": got, "gathered": fg}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_compute_stats_world2_gloo_equals_reference(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    _assert_equal(r0["stats"], GOLD["world2"]["expected"])
    _assert_equal(r1["stats"], GOLD["world2"]["expected"])  # every rank logs the GLOBAL distribution
    want = torch.cat([torch.arange(3.0), torch.arange(5.0) + 10])
    assert torch.equal(r0["gathered"], want) and torch.equal(r1["gathered"], want)


def test_collect_step_tensors_reference_semantics():
    """One packed pass -> the vectors train.py:483-539 appends: one loss per MICRO-BATCH, entropy over loss_mask with the
    shift_tensor_right pad, mismatch-KL over loss_mask & rl membership, rl metrics for every sample of a stream-less micro-batch but
    only for member samples of a micro-batch that ships rl_weights, 3 rl keys under IPO."""
    from prime_rl_amd import _lib
    from prime_rl_amd.trainer.engine import pack_micro_batches
    from prime_rl_amd.trainer.stats import Tensors, collect_step_tensors

    n = 6
    def mb(ids, mask, rlw=None, sl=(6,), env="a"):
        d = dict(input_ids=torch.tensor([ids]), position_ids=torch.arange(n)[None], advantages=torch.ones(1, n), inference_logprobs=torch.full((1, n), -1.0),
                 loss_mask=torch.tensor([mask]), temperatures=torch.ones(1, n), seq_lens=torch.tensor(list(sl)), sequence_lengths=list(sl), env_names=[env] * n)
        if rlw is not None:
            d["rl_weights"] = torch.tensor([rlw])
        return d

    mbs = [mb([1, 2, 3, 4, 5, 6], [False, True, True, False, False, False], sl=(3, 3), env="a"),           # no rl stream: both samples log
           mb([1, 2, 3, 4, 5, 6], [False, True, True, True, True, True], rlw=[1, 1, 0, 0, 0, 0], sl=(3, 3), env="b")]  # stream: sample 2 has no member
    pb = pack_micro_batches(mbs, "cpu")
    assert pb.rl_every_sample.tolist() == [True, True, False, False]
    V = 8
    lp = torch.full((12,), -1.5)
    ent = torch.arange(12, dtype=torch.float32)
    S, M = _lib.STAT, _lib.METRIC_KEYS
    stats = torch.zeros(4, _lib.PM_LOSS_NSTATS)
    stats[:, S["RL_LOSS"]] = torch.tensor([1.0, 2.0, 4.0, 8.0])
    stats[:, S["RL_N_MASK"]] = torch.tensor([2.0, 0.0, 1.0, 0.0])
    met = torch.arange(4 * _lib.PM_LOSS_NMETRICS, dtype=torch.float32).view(4, -1)
    t = Tensors()
    collect_step_tensors(t, pb, lp, ent, stats, met, (4, 1, 1), V, "default", torch.tensor([0, 0, 1, 1]), ["a", "b"], M, S)
    assert t["loss"][0].tolist() == [0.75, 3.0]  # (1+2)/4 and (4+8)/4: one entry per micro-batch
    # entropy shifted right per micro-batch: token i shows head position i-1, first token of a micro-batch shows log V
    mask = pb.loss_mask
    shifted = torch.cat([torch.tensor([math.log(V)]), ent[0:5], torch.tensor([math.log(V)]), ent[6:11]])
    assert torch.allclose(t["entropy/all"][0], shifted[mask])
    assert torch.allclose(t["entropy/a"][0], shifted[:6][mask[:6]]) and torch.allclose(t["entropy/b"][0], shifted[6:][mask[6:]])
    # mismatch-KL only over loss_mask & (rl_w != 0) once a stream exists (stream-less micro-batches materialise weight 1)
    mm = mask & (pb.rl_weights != 0)
    assert t["mismatch_kl/all"][0].numel() == int(mm.sum()) == 3
    lr = -1.5 - (-1.0)
    assert torch.allclose(t["mismatch_kl/all"][0], torch.full((3,), math.exp(lr) - lr - 1))
    # rl metric rows: samples 0,1 (stream-less micro-batch: every sample) + sample 2 (member); sample 3 is skipped
    assert t["is_masked"][0].tolist() == met[[0, 1, 2], M.index("is_masked")].tolist()
    assert set(t) >= set(M[:7]) and "nll" not in t and "ref_kl" not in t
    t2 = Tensors()
    collect_step_tensors(t2, pb, lp, ent, stats, met, (4, 1, 1), V, "ipo", None, [], M, S)
    assert "is_masked_low" not in t2 and "masked_advantage_positive" not in t2 and "is_masked" in t2
