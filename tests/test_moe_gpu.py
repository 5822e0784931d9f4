"""MoE routing / permutation / combine kernels and the MoE layer (EP half) against the oracle and the reference's golden
vectors (tests/golden/moe.npz, generated from the reference's own MoE module).  Integer work (selection, counts, sort) is bit-exact;
the combine reproduces the reference's bf16 scatter_add bit for bit; f32 paths <= 1e-6; the full layer at the bf16 GEMM bound."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import moe_oracle as M
from tests.test_moe_oracle_cpu import CASES, GOLD, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from prime_rl_amd import ops

    return ops


@pytest.mark.parametrize("name", CASES)
def test_router_forward_matches_reference_goldens(name):
    ops = _ops()
    args, params, _, x, _, bias, routed, kw = load_case(name, DEV)
    logits = torch.nn.functional.linear(x, params["router.gate.weight"])
    forced = None if routed is None else routed.to(torch.int32).contiguous()
    scores, top, idx, counts, conf = ops.moe_router_fwd(logits, args.top_k, args.score_func, args.route_norm, args.route_scale, bias, forced)
    ref = torch.sigmoid(logits.float()) if args.score_func == "sigmoid" else torch.softmax(logits.float(), dim=1)
    assert torch.allclose(scores, ref, rtol=1e-6, atol=1e-7)
    # the golden selection was made by the reference on the CPU from a CPU gate GEMM; the GPU gate GEMM may differ in the last bf16 bit, so
    # compare the selection with the oracle run on THESE logits, and the gating values with the goldens at the bf16 GEMM bound
    o_top, o_idx, o_counts, o_conf, _ = M.router(x, params["router.gate.weight"], args, bias, None if routed is None else routed)
    assert torch.equal(idx.long(), o_idx)
    assert torch.equal(counts.long(), o_counts)
    assert torch.allclose(top, o_top, rtol=1e-5, atol=1e-7)
    assert float(conf.sum()) == pytest.approx(float(o_conf), rel=1e-5)
    gold_idx = torch.from_numpy(GOLD[f"{name}.top_idx"]).to(DEV)
    same = (idx == gold_idx).all(dim=1)
    assert same.float().mean() > 0.9  # near-ties may flip between a CPU and a GPU bf16 GEMM, nothing else may
    assert torch.allclose(top[same], torch.from_numpy(GOLD[f"{name}.top_scores"]).to(DEV)[same], rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("score_func,route_norm,n,e,k", [("softmax", True, 1000, 128, 8), ("sigmoid", False, 37, 64, 1), ("sigmoid", True, 513, 200, 6),
                                                       ("softmax", False, 5, 8, 8), ("softmax", True, 300, 1024, 16)])
def test_router_forward_backward_f32_against_autograd(score_func, route_norm, n, e, k):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(n + e)
    logits = (torch.randn(n, e, generator=g) * 2).to(DEV).requires_grad_(True)
    scores, top, idx, counts, conf = ops.moe_router_fwd(logits.detach(), k, score_func, route_norm, 1.7)
    s = torch.sigmoid(logits) if score_func == "sigmoid" else torch.softmax(logits, dim=1)
    rt, ri = torch.topk(s, k, dim=1)
    assert torch.equal(idx.long(), ri)  # random f32 logits: no ties
    t = s.gather(1, ri)
    if route_norm:
        t = t / (t.sum(-1, keepdim=True) + 1e-20)
    t = t * 1.7
    assert torch.allclose(top, t.detach(), rtol=1e-5, atol=1e-7)
    assert torch.equal(counts.long(), torch.bincount(ri.reshape(-1), minlength=e))
    gt = torch.randn(n, k, generator=g).to(DEV)
    (gl,) = torch.autograd.grad(t, logits, gt)
    got = ops.moe_router_bwd(scores, idx, gt, score_func, route_norm, 1.7)
    assert torch.allclose(got, gl, rtol=1e-4, atol=1e-6)


def test_router_ties_and_forced_duplicates():
    ops = _ops()
    logits = torch.zeros(3, 16, device=DEV)  # all equal: lowest indices win, in order
    _, top, idx, counts, _ = ops.moe_router_fwd(logits, 4, "softmax", True, 1.0)
    assert idx.tolist() == [[0, 1, 2, 3]] * 3
    assert torch.allclose(top, torch.full_like(top, 0.25))
    assert counts.tolist() == [3, 3, 3, 3] + [0] * 12
    forced = torch.tensor([[5, 5], [0, 15], [7, 2]], dtype=torch.int32, device=DEV)
    scores, top, idx, counts, _ = ops.moe_router_fwd(torch.randn(3, 16, device=DEV), 2, "softmax", False, 1.0, None, forced)
    assert torch.equal(idx, forced)
    assert torch.allclose(top, scores.gather(1, forced.long()))
    assert counts[5] == 2
    got = ops.moe_router_bwd(scores, idx, torch.ones(3, 2, device=DEV), "softmax", False, 1.0)  # duplicate selection: both gradients land on expert 5
    s = scores[0]
    want = s * ((torch.arange(16, device=DEV) == 5).float() * 2 - 2 * s[5])
    assert torch.allclose(got[0], want, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,k,e", [(16384, 8, 128), (1, 1, 4), (1023, 3, 7), (1025, 2, 1000), (0, 2, 8)])
def test_sort_is_the_stable_argsort(n, k, e):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(n * 31 + e)
    idx = torch.randint(0, e, (n, k), generator=g, dtype=torch.int32).to(DEV)
    if n > 100:
        idx[: n // 3] = idx[0, 0]  # a heavily loaded expert and empty ones
    dest, slot, offsets = ops.moe_sort(idx.reshape(-1).contiguous(), e)
    order = torch.argsort(idx.reshape(-1).long(), stable=True)
    assert torch.equal(slot.long(), order)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=DEV)
    assert torch.equal(dest.long(), inv)
    counts = torch.bincount(idx.reshape(-1).long(), minlength=e)
    assert torch.equal(offsets.long(), torch.cat([counts.new_zeros(1), counts.cumsum(0)]))


@pytest.mark.parametrize("n,k,h,with_scores,with_base", [(300, 4, 256, True, False), (64, 1, 64, True, True), (1000, 8, 2048, True, False),
                                                       (50, 2, 128, False, False)])
def test_gather_combine_and_slot_grads_match_the_reference_ops(n, k, h, with_scores, with_base):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(n + h)
    e = 16
    x = torch.randn(n, h, generator=g).to(torch.bfloat16).to(DEV)
    idx = torch.randint(0, e, (n, k), generator=g, dtype=torch.int32).to(DEV)
    scores = torch.rand(n, k, generator=g).to(DEV)
    dest, slot, _ = ops.moe_sort(idx.reshape(-1).contiguous(), e)
    tok = (slot // k).long()
    # gather (+ score): moe.py:891-895
    got = ops.moe_gather_rows(x, slot, k, scores.reshape(-1) if with_scores else None)
    want = x[tok]
    if with_scores:
        want = (want.float() * scores.reshape(-1)[slot.long()][:, None]).to(torch.bfloat16)
    assert torch.equal(got, want)
    # combine: the reference's ops on the CPU (sequential bf16 scatter_add in sorted-row order) — bit for bit
    y = torch.randn(n * k, h, generator=g).to(torch.bfloat16).to(DEV)
    base = torch.randn(n, h, generator=g).to(torch.bfloat16).to(DEV) if with_base else None
    out = ops.moe_combine(y, dest, k, scores.reshape(-1) if with_scores else None, base)
    yc = y.cpu()
    if with_scores:
        yc = (yc.float() * scores.reshape(-1)[slot.long()].cpu()[:, None]).to(torch.bfloat16)
    ref = (base.cpu() if with_base else torch.zeros(n, h, dtype=torch.bfloat16)).scatter_add(0, tok.cpu().reshape(-1, 1).expand(-1, h), yc)
    assert torch.equal(out.cpu(), ref)
    # slot gradients
    a = torch.randn(n, h, generator=g).to(torch.bfloat16).to(DEV)
    rows, ds = ops.moe_slot_grads(a, y, slot, k, scores.reshape(-1) if with_scores else None, True, True)
    want_rows = a[tok] if not with_scores else (a[tok].float() * scores.reshape(-1)[slot.long()][:, None]).to(torch.bfloat16)
    assert torch.equal(rows, want_rows)
    want_ds = torch.zeros(n * k, device=DEV)
    want_ds[slot.long()] = (a[tok].float() * y.float()).sum(-1)
    assert torch.allclose(ds, want_ds, rtol=1e-4, atol=1e-3)


def test_gather_rows_negative_map_writes_zero_rows():
    ops = _ops()
    x = torch.ones(4, 64, dtype=torch.bfloat16, device=DEV)
    out = ops.moe_gather_rows(x, torch.tensor([2, -1, 0], dtype=torch.int32, device=DEV))
    assert out[1].abs().sum() == 0 and out[0].sum() == 64


def test_ep_row_map_matches_the_oracle_layout():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    ep, e = 4, 12
    counts = [torch.randint(0, 6, (e,), generator=g) for _ in range(ep)]
    counts[1][3] = 0
    plan, totals = M.ep_dispatch_layout(counts, ep)
    le = e // ep
    for d in range(ep):
        recv = torch.stack([c[d * le:(d + 1) * le] for c in counts]).to(torch.int32).to(DEV)  # [src, local expert]
        total = int(recv.sum())
        row_map, inv_map, offs = ops.moe_ep_row_map(recv.contiguous(), total)
        # received buffer: rows in (source, local expert) order; label every row by (source, position in the source's sorted buffer)
        labels = []
        for s in range(ep):
            o = torch.cat([counts[s].new_zeros(1), counts[s].cumsum(0)])
            for j in range(le):
                labels += [(s, int(o[d * le + j]) + i) for i in range(int(counts[s][d * le + j]))]
        want = [(s, first + i) for (s, first, c) in plan[d] for i in range(c)]
        got = [labels[r] for r in row_map.tolist()]
        assert got == want
        assert [inv_map.tolist()[r] for r in row_map.tolist()] == list(range(total))
        assert offs.tolist() == [0] + list(np.cumsum(totals[d]))


def _build_layer(name, use_grouped_mm):
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    args, params, grads, x, gy, bias, routed, kw = load_case(name, DEV)
    layer = MoE(MoEArgs(num_experts=args.num_experts, num_shared_experts=args.num_shared_experts, score_func=args.score_func, route_norm=args.route_norm,
                        route_scale=args.route_scale, score_before_experts=args.score_before_experts, top_k=args.top_k, use_grouped_mm=use_grouped_mm,
                        load_balance_coeff=args.load_balance_coeff), dim=kw["dim"], hidden_dim=kw["hidden"]).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        for k, v in params.items():
            dict(layer.named_parameters())[k].copy_(v)
        if bias is not None:
            layer.expert_bias = bias.float()
    return layer, x, gy, bias, routed, kw, grads


@pytest.mark.parametrize("name", CASES)
def test_moe_layer_forward_backward_against_the_reference_module(name):
    layer, x, gy, bias, routed, kw, grads = _build_layer(name, use_grouped_mm=False)
    gold_idx = torch.from_numpy(GOLD[f"{name}.top_idx"]).to(DEV)
    # route exactly as the reference did (a flipped near-tie would compare different experts, not different arithmetic)
    x = x.reshape(1, -1, kw["dim"]).clone().requires_grad_(True)
    y = layer(x, routed_experts=gold_idx.reshape(1, -1, gold_idx.shape[1]).long())
    y.backward(gy.reshape_as(y))
    gold_y = torch.from_numpy(GOLD[f"{name}.y"]).to(DEV)
    gold_gx = torch.from_numpy(GOLD[f"{name}.gx"]).to(DEV)
    scale = gold_y.abs().mean()
    assert (y.float() - gold_y).abs().max() <= 0.03 * max(1.0, float(gold_y.abs().max()))
    assert (y.float() - gold_y).abs().mean() <= 0.01 * scale
    assert (x.grad.float() - gold_gx).abs().mean() <= 0.02 * gold_gx.abs().mean()
    for k, gref in grads.items():
        got = dict(layer.named_parameters())[k].grad.float()
        gref = gref.to(DEV)
        assert (got - gref).abs().mean() <= 0.03 * gref.abs().mean() + 1e-6, k
    assert torch.equal(layer.tokens_per_expert.cpu(), torch.from_numpy(GOLD[f"{name}.tokens_per_expert"]))


def test_moe_layer_free_routing_counts_and_statistics():
    layer, x, gy, bias, routed, kw, _ = _build_layer("qwen3", use_grouped_mm=False)
    y = layer(x.reshape(1, -1, kw["dim"]))
    assert y.shape == (1, x.shape[0], kw["dim"])
    assert float(layer.tokens_per_expert.sum()) == x.shape[0] * 2
    assert 0 < float(layer.routing_confidence_sum) <= x.shape[0]


def test_grouped_mm_path_equals_the_for_loop_path():
    if not hasattr(torch, "_grouped_mm"):
        pytest.skip("this torch build has no torch._grouped_mm")
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    torch.manual_seed(0)
    dim, hidden, e, k, n = 256, 128, 8, 2, 512
    a = MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k, use_grouped_mm=True,
                    load_balance_coeff=None), dim, hidden).to(DEV).to(torch.bfloat16)
    a.init_weights(0.02)
    b = MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k, use_grouped_mm=False,
                    load_balance_coeff=None), dim, hidden).to(DEV).to(torch.bfloat16)
    b.load_state_dict(a.state_dict())
    x = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    try:
        ya = a(xa)
    except (RuntimeError, NotImplementedError) as err:  # the library grouped GEMM is optional on this stack; the loop path is the tested default
        pytest.skip(f"torch._grouped_mm unavailable on this device: {err}")
    yb = b(xb)
    assert torch.allclose(ya.float(), yb.float(), rtol=2e-2, atol=2e-3)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    # both are bf16 chains with different rounding points (tools/probes/moe_paths_vs_f32.py: each ~0.5 % mean error against f32): compare in the mean
    assert (xa.grad.float() - xb.grad.float()).abs().mean() <= 0.02 * xb.grad.float().abs().mean()
    assert (a.experts.w1.grad.float() - b.experts.w1.grad.float()).abs().mean() <= 0.02 * b.experts.w1.grad.float().abs().mean()


def _tiny_moe_model(extra=None):
    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer import model as MM
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims

    cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny-moe", seq_len=128, debug=dict(random_init=True), fused_lm_head_token_chunk_size=64,
                                                       **(extra or {}))))
    pd = get_parallel_dims(cfg.model, cfg.model.seq_len, world_size=1)
    return MM, MM.setup_model(cfg.model, pd)


def test_qwen3_moe_model_through_setup_model_forward_and_router_replay():
    """setup_model() builds Qwen3MoeForCausalLM for an MoE name; forward() returns log-probs; every parameter (router gates and expert
    matrices included) receives a gradient; replaying the recorded routing (`routed_experts`, model.py:1462-1463) reproduces the free
    forward bit for bit; the flat-buffer engine refuses the model loudly."""
    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeForCausalLM
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims

    MM, model = _tiny_moe_model()
    assert isinstance(model, Qwen3MoeForCausalLM) and len(model.moe_layers()) == 2
    n = 128
    g = torch.Generator(device="cpu").manual_seed(3)
    ids = torch.randint(0, 512, (1, n), generator=g).to(DEV)
    labels = torch.cat([ids[:, 1:], ids.new_zeros(1, 1)], 1)
    kw = dict(seq_lens=torch.tensor([70, 58], device=DEV), labels=labels, temperature=torch.ones(1, n, device=DEV))
    recorded = []
    hooks = [m.router.register_forward_hook(lambda mod, inp, out: recorded.append(out[1].clone())) for m in model.moe_layers()]
    out = MM.forward(model, ids, torch.arange(n, device=DEV)[None], **kw)
    for h in hooks:
        h.remove()
    lp = out["logprobs"]
    assert lp.shape == (1, n) and torch.isfinite(lp).all() and (lp <= 0).all()
    (-lp.mean()).backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert all(float(m.experts.w1.grad.abs().sum()) > 0 and float(m.router.gate.weight.grad.abs().sum()) > 0 for m in model.moe_layers())
    assert all(float(m.tokens_per_expert.sum()) == n * 2 for m in model.moe_layers())
    stats = MM.get_load_balance_stats(model)  # model.py:561-592: per-layer load violation and routing confidence, counters reset
    assert stats["max_vio"].shape == (2,) and torch.isfinite(stats["max_vio"]).all() and (stats["max_vio"] >= 0).all()
    assert ((stats["routing_confidence"] > 0) & (stats["routing_confidence"] <= 1.0 + 1e-5)).all()
    assert all(float(m.tokens_per_expert.sum()) == 0 for m in model.moe_layers())
    routed = torch.stack(recorded, dim=1).reshape(1, n, 2, 2).long()  # [1, N, layers, top_k]
    replay = MM.forward(model, ids, torch.arange(n, device=DEV)[None], routed_experts=routed, **kw)
    assert torch.equal(replay["logprobs"], lp)
    other = routed.clone()
    other[..., 0] = (other[..., 0] + 1) % 8
    assert not torch.equal(MM.forward(model, ids, torch.arange(n, device=DEV)[None], routed_experts=other, **kw)["logprobs"], lp)
    cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny-moe", seq_len=128, debug=dict(random_init=True))))
    with pytest.raises(NotImplementedError, match="flat-buffer engine"):
        MM.setup_engine(cfg, get_parallel_dims(cfg.model, 128, world_size=1))


def test_qwen3_moe_setup_model_switches():
    """model.moe_router_dtype (default float32: f32 gate weight and gate GEMM), model.freeze_moe_router, model.debug.force_balanced_routing."""
    MM, model = _tiny_moe_model()
    assert all(m.router.gate.weight.dtype == torch.float32 and m.router.fp32_gate for m in model.moe_layers())
    assert model.model.layers[0].mlp.experts.w1.dtype == torch.bfloat16
    _, m16 = _tiny_moe_model(dict(moe_router_dtype="bfloat16"))
    assert all(m.router.gate.weight.dtype == torch.bfloat16 and not m.router.fp32_gate for m in m16.moe_layers())
    _, frozen = _tiny_moe_model(dict(freeze_moe_router=True))
    assert all(not p.requires_grad for m in frozen.moe_layers() for p in m.router.parameters())
    from prime_rl_amd.configs.trainer import TrainerConfig
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims

    cfg = TrainerConfig.model_validate(dict(model=dict(name="tiny-moe", seq_len=128, debug=dict(random_init=True, force_balanced_routing=True))))
    bal = MM.setup_model(cfg.model, get_parallel_dims(cfg.model, 128, world_size=1))
    n = 128
    ids = torch.randint(0, 512, (1, n), device=DEV)
    out = MM.forward(bal, ids, torch.arange(n, device=DEV)[None], seq_lens=torch.tensor([n], device=DEV), labels=ids, temperature=torch.ones(1, n, device=DEV))
    out["logprobs"].sum().backward()
    for m in bal.moe_layers():  # 128 tokens x top-2 over 8 experts, round robin: exactly 32 rows per expert
        assert m.tokens_per_expert.tolist() == [32.0] * 8
    with pytest.raises(NotImplementedError, match="moe_fused_kernel"):
        _tiny_moe_model(dict(moe_fused_kernel=True))


def test_qwen3_moe_training_steps_reduce_the_loss():
    MM, model = _tiny_moe_model()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.0)
    n = 128
    g = torch.Generator(device="cpu").manual_seed(11)
    ids = torch.randint(0, 512, (1, n), generator=g).to(DEV)
    labels = torch.cat([ids[:, 1:], ids.new_zeros(1, 1)], 1)
    losses = []
    for _ in range(8):
        out = MM.forward(model, ids, torch.arange(n, device=DEV)[None], seq_lens=torch.tensor([n], device=DEV), labels=labels,
                         temperature=torch.ones(1, n, device=DEV))
        loss = -out["logprobs"][:, :-1].mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.5, losses


def test_expert_parallel_dispatch_on_the_device_equals_local_experts():
    """The dispatch / regroup / combine path of trainer/distributed/expert_parallel.py over an RCCL group (world 1: the all-to-alls are
    self-copies, every kernel and index map of the path runs) gives exactly the local grouped-experts result, forward and backward."""
    import os

    import torch.distributed as dist

    from prime_rl_amd import ops
    from prime_rl_amd.trainer.distributed import expert_parallel as EP
    from prime_rl_amd.trainer.models.layers.moe import GroupedExperts

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        torch.manual_seed(0)
        e, dim, hidden, n, k = 8, 128, 64, 200, 2
        experts = GroupedExperts(dim, hidden, e, use_grouped_mm=False).to(DEV).to(torch.bfloat16)
        experts.init_weights(0.05)
        idx = torch.randint(0, e, (n, k), dtype=torch.int32, device=DEV)
        idx[:, 0] = torch.where(idx[:, 0] == 3, 4, idx[:, 0])  # an expert with fewer rows
        dest, slot, offsets = ops.moe_sort(idx.reshape(-1).contiguous(), e)
        counts = (offsets[1:] - offsets[:-1]).contiguous()
        x = torch.randn(n, dim, device=DEV).to(torch.bfloat16)
        rows = ops.moe_gather_rows(x, slot, k)
        a = rows.clone().requires_grad_(True)
        b = rows.clone().requires_grad_(True)
        ya = EP.run_experts(experts, a, counts, dist.group.WORLD)
        gy = torch.randn_like(ya)
        ya.backward(gy)
        ga = {n_: p.grad.clone() for n_, p in experts.named_parameters()}
        experts.zero_grad()
        yb = experts(b, offsets)
        yb.backward(gy)
        assert torch.equal(ya, yb) and torch.equal(a.grad, b.grad)
        for n_, p in experts.named_parameters():
            assert torch.equal(ga[n_], p.grad), n_
    finally:
        if created:
            dist.destroy_process_group()


def test_expert_parallel_layer_with_overlapped_chunked_exchange_on_the_device(monkeypatch):
    """MoE.forward under expert parallelism (an RCCL group of one rank: the all-to-alls are asynchronous self-copies on the communicator's stream) with
    the tokens cut into 1 / 2 / 3 pieces (PM_EP_OVERLAP_CHUNKS; 2 is the default on a GPU): the chunked, overlapped exchange gives the one-piece layer's
    output and input gradient bit for bit, the weight gradients up to the bf16 sum over pieces, and equals the layer without expert parallelism."""
    import os

    import torch.distributed as dist

    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29542")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        args = MoEArgs(num_experts=8, num_shared_experts=0, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False, top_k=2,
                       use_grouped_mm=True, load_balance_coeff=None)
        dim, hidden, n = 256, 256, 777
        torch.manual_seed(0)
        plain = MoE(args, dim, hidden).to(DEV)
        for p in plain.parameters():
            torch.nn.init.normal_(p, std=0.05)
        plain = plain.to(torch.bfloat16)
        ep = MoE(args, dim, hidden, ep_group=dist.group.WORLD).to(DEV).to(torch.bfloat16)
        ep.ep_size = 2  # take the expert-parallel branch on the one-rank group (ep_size only gates the branch; the group's size does the arithmetic)
        ep.load_state_dict(plain.state_dict())
        x = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
        gy = torch.randn_like(x)
        res = {}
        for chunks in (0, 1, 2, 3):  # 0: the layer without expert parallelism
            layer = plain if chunks == 0 else ep
            monkeypatch.setenv("PM_EP_OVERLAP_CHUNKS", str(max(chunks, 1)))
            layer.zero_grad()
            xa = x.clone().requires_grad_(True)
            y = layer(xa)
            y.backward(gy)
            torch.cuda.synchronize()
            res[chunks] = (y.detach().clone(), xa.grad.clone(), {k: v.grad.float().clone() for k, v in layer.named_parameters()})
        for chunks in (1, 2, 3):
            y, gx, gw = res[chunks]
            assert torch.equal(y, res[0][0]) and torch.equal(gx, res[0][1]), chunks
            for k, v in gw.items():
                ref = res[0][2][k]
                assert (v - ref).abs().max().item() <= 2.0 ** -6 * max(ref.abs().max().item(), 1e-6), (chunks, k)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("e,n_out,k,sizes", [(4, 128, 128, [5, 0, 300, 128]), (8, 256, 384, [64, 1, 127, 129, 0, 0, 700, 3]), (3, 768, 2048, [1000, 24, 513]),
                                             (2, 128, 256, [0, 0]), (16, 128, 128, [17] * 16), (5, 512, 256, [300, 0, 257, 255, 1]),
                                             (3, 256, 768, [600, 256, 31]),
                                             # the row-tile lookup: 200 experts = four register sets of the table, 300 = the memory form (csrc/gemm.hip RowTab)
                                             (200, 256, 256, [3, 0, 300, 1] * 50), (300, 256, 256, [2, 0, 270] * 100),
                                             # imbalanced routing: the weight gradient's tile line is cut into 8 XCD pieces by COST (build_wgrad_range)
                                             (16, 256, 256, [5000, 3, 0, 17] * 4), (9, 512, 256, [0, 0, 0, 0, 0, 0, 0, 0, 4096]), (128, 256, 256, [0] * 127 + [700])])
@pytest.mark.parametrize("variant", [0, 1])  # 0: 256 x 256 tiles where the shape allows; 1: 128 x 128 tiles everywhere
def test_grouped_gemm_kernels_against_per_expert_matmul(e, n_out, k, sizes, variant):
    """pm_grouped_gemm_nt / nn / tn vs a per-expert f32 matmul of the same bf16 operands: ragged groups, empty groups, groups that are not
    multiples of the 64-row contraction step or the 128-row tile, an empty batch."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(e * 1000 + n_out)
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(e, n_out, k, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    dy = torch.randn(m, n_out, generator=g).to(torch.bfloat16).to(DEV)
    assert ops.grouped_gemm_supported(m, n_out, k)
    from prime_rl_amd import _lib

    sched = ((variant << 16) | 7) if variant else None  # PM_GROUPED_SCHED_SMALL / _BIG ride in the launch's schedule word (no library state)
    y = ops.grouped_gemm_nt(x, w, offsets, variant=sched)
    dx = ops.grouped_gemm_nn(dy, w, offsets, variant=sched)
    dw = ops.grouped_gemm_tn(dy, x, offsets, variant=sched)
    o = 0
    for i, c in enumerate(sizes):
        xe, de, we = x[o:o + c].float(), dy[o:o + c].float(), w[i].float()
        ref_y, ref_dx, ref_dw = xe @ we.T, de @ we, de.T @ xe
        for got, ref, what in ((y[o:o + c], ref_y, "nt"), (dx[o:o + c], ref_dx, "nn"), (dw[i], ref_dw, "tn")):
            err = (got.float() - ref).abs()
            tol = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * max(1e-3, float(ref.abs().mean()))  # one bf16 rounding of an exactly accumulated sum
            assert (err <= tol).all(), f"{what} expert {i} (rows {c}): max err {float(err.max())}"
        o += c


def test_grouped_gemm_two_weight_sets_equal_two_single_calls():
    """The gate|up forms: nt with (w1, w3) = the two single products side by side; nn over both = the sum of the two single row gradients
    (exactly: one f32 accumulation instead of two rounded ones, so compare with the f32 reference); tn with two outputs = two single calls."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(9)
    e, n1, n2, k = 5, 256, 128, 384
    sizes = [70, 0, 333, 128, 1]
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(e, n1, k, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    w3 = (torch.randn(e, n2, k, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    y = ops.grouped_gemm_nt(x, w1, offsets, w2=w3)
    assert torch.equal(y[:, :n1], ops.grouped_gemm_nt(x, w1, offsets)) and torch.equal(y[:, n1:], ops.grouped_gemm_nt(x, w3, offsets))
    dy = torch.randn(m, n1 + n2, generator=g).to(torch.bfloat16).to(DEV)
    dw1, dw3 = ops.grouped_gemm_tn(dy, x, offsets, n_first=n1)
    assert torch.equal(dw1, ops.grouped_gemm_tn(dy[:, :n1].contiguous(), x, offsets)) and torch.equal(dw3, ops.grouped_gemm_tn(dy[:, n1:].contiguous(), x, offsets))
    dx = ops.grouped_gemm_nn(dy, w1, offsets, w2=w3)
    o = 0
    for i, c in enumerate(sizes):
        ref = dy[o:o + c, :n1].float() @ w1[i].float() + dy[o:o + c, n1:].float() @ w3[i].float()
        err = (dx[o:o + c].float() - ref).abs()
        assert (err <= 2.0 ** -8 * ref.abs() + 2.0 ** -8 * max(1e-3, float(ref.abs().mean()))).all(), i
        o += c


@pytest.mark.parametrize("e,inter,n_dim,sizes", [(4, 256, 128, [5, 0, 300, 128]), (8, 768, 2048, [64, 1, 127, 129, 0, 0, 700, 3]), (3, 512, 384, [600, 256, 31]),
                                                 (2, 256, 256, [0, 0])])
def test_grouped_row_gradient_with_swiglu_backward_epilogue_equals_two_launches(e, inter, n_dim, sizes):
    """pm_grouped_gemm_nn_swiglu_bwd = pm_swiglu_bwd(pm_grouped_gemm_nn(dy, w_down), gate_up) bit for bit: ragged and empty groups, an empty batch, in place."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(e * 100 + inter)
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    dy = torch.randn(m, n_dim, generator=g).to(torch.bfloat16).to(DEV)
    w_down = (torch.randn(e, n_dim, inter, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    gate_up = (torch.randn(m, 2 * inter, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    assert ops.grouped_gemm_nn_swiglu_bwd_supported(m, n_dim, inter)
    got = ops.grouped_gemm_nn_swiglu_bwd(dy, w_down, offsets, gate_up)
    if m:
        ref = ops.swiglu_bwd(ops.grouped_gemm_nn(dy, w_down, offsets), gate_up)
        assert torch.equal(got, ref)
        inplace = gate_up.clone()
        ops.grouped_gemm_nn_swiglu_bwd(dy, w_down, offsets, inplace, out=inplace)
        assert torch.equal(inplace, ref)
    assert not ops.grouped_gemm_nn_swiglu_bwd_supported(m, n_dim, inter + 128)  # inter must be a multiple of 256


@pytest.mark.parametrize("e,inter,k,sizes", [(4, 128, 128, [5, 0, 300, 128]), (8, 768, 2048, [64, 1, 127, 129, 0, 0, 700, 3]), (3, 384, 256, [600, 256, 31]),
                                             (2, 256, 256, [0, 0]), (200, 128, 128, [3, 0, 300, 1] * 50)])
def test_grouped_gate_up_forward_with_swiglu_epilogue_equals_two_launches(e, inter, k, sizes):
    """pm_grouped_gemm_nt_swiglu: gate_up = pm_grouped_gemm_nt(x, w_gate, w_up) and act = pm_swiglu_fwd(gate_up), bit for bit; the two weight tensors are read
    through one descriptor, so the op is offered only when w_up lies above w_gate within 2 GiB."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(e * 10 + inter)
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    both = (torch.randn(2, e, inter, k, generator=g) * 0.1).to(torch.bfloat16).to(DEV)  # one allocation: w_up directly above w_gate
    w_gate, w_up = both[0], both[1]
    assert ops.grouped_gemm_nt_swiglu_supported(x, w_gate, w_up)
    assert not ops.grouped_gemm_nt_swiglu_supported(x, w_up, w_gate)  # the other way round: the caller keeps the two-launch form
    gate_up, act = ops.grouped_gemm_nt_swiglu(x, w_gate, w_up, offsets)
    if m:
        ref = ops.grouped_gemm_nt(x, w_gate, offsets, w2=w_up)
        assert torch.equal(gate_up, ref)
        assert torch.equal(act, ops.swiglu_fwd(ref))


def test_expert_ffn_as_one_autograd_node_equals_the_two_node_form():
    """_GroupedFFNFn (fused SwiGLU backward) vs _GroupedLinearFn(_GroupedSwiGLUFn(x)): outputs and all four gradients identical."""
    from prime_rl_amd.trainer.models.layers import moe as M

    g = torch.Generator(device="cpu").manual_seed(3)
    e, dim, hidden, sizes = 6, 256, 512, [300, 0, 129, 1, 700, 64]
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).to(DEV)  # noqa: E731
    x0, w10, w20, w30, gy = mk(m, dim), mk(e, hidden, dim, s=0.1), mk(e, dim, hidden, s=0.1), mk(e, hidden, dim, s=0.1), mk(m, dim)
    outs = []
    for fused in (True, False):
        x, w2 = (t.clone().requires_grad_(True) for t in (x0, w20))
        w13 = torch.stack([w10, w30]).requires_grad_(True)  # one allocation (w3 above w1): the fused gate|up forward is used as well
        w1, w3 = w13[0], w13[1]
        y = M._GroupedFFNFn.apply(x, w1, w2, w3, offsets) if fused else M._GroupedLinearFn.apply(M._GroupedSwiGLUFn.apply(x, w1, w3, offsets), w2, offsets)
        y.backward(gy)
        outs.append((y.detach(), x.grad, w13.grad, w2.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_grouped_weight_gradient_in_f32_written_and_accumulated_in_place():
    """pm_grouped_gemm_tn_f32: dw (and dw2) in f32 straight into caller-owned buffers, first call overwrites (garbage in), second accumulates; ragged and
    empty groups; the bf16-output kernel's result is this one rounded once."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(21)
    e, n1, n2, k = 6, 512, 256, 768
    sizes = [300, 0, 1, 255, 513, 64]
    m = sum(sizes)
    offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    dy = torch.randn(m, n1 + n2, generator=g).to(torch.bfloat16).to(DEV)
    assert ops.grouped_gemm_tn_f32_supported(n1, n1 + n2, k)
    flat = torch.full((e * (n1 + n2) * k + 64,), float("nan"), dtype=torch.float32, device=DEV)  # views into one flat buffer, like a main-grad store
    dw, dw2 = flat[:e * n1 * k].view(e, n1, k), flat[e * n1 * k:e * (n1 + n2) * k].view(e, n2, k)
    ops.grouped_gemm_tn_f32(dy, x, offsets, dw, dw2, accumulate=False)
    assert torch.isnan(flat[-64:]).all()  # nothing written past the views
    o = 0
    for i, c in enumerate(sizes):
        ref = dy[o:o + c].float().T @ x[o:o + c].float()
        got = torch.cat([dw[i], dw2[i]], 0)
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), i
        o += c
    b1, b3 = ops.grouped_gemm_tn(dy, x, offsets, n_first=n1)
    assert torch.equal(b1, dw.to(torch.bfloat16)) and torch.equal(b3, dw2.to(torch.bfloat16))
    first = flat[:e * (n1 + n2) * k].clone()
    ops.grouped_gemm_tn_f32(dy, x, offsets, dw, dw2, accumulate=True)
    assert torch.allclose(flat[:e * (n1 + n2) * k], 2 * first, rtol=1e-6, atol=0)
    single = torch.empty(e, n1, k, dtype=torch.float32, device=DEV)
    ops.grouped_gemm_tn_f32(dy[:, :n1].contiguous(), x, offsets, single)
    assert torch.equal(single, first[:e * n1 * k].view(e, n1, k))


def test_grouped_weight_gradient_leaves_its_squared_norm_in_the_tile_slots():
    """pm_grouped_gemm_tn_f32_sumsq: sum(tile_sumsq) is the squared norm of the FINAL dw and dw2 — after an overwriting launch, after an accumulating one
    (slots overwritten, not added to), with an empty expert (its tiles hold zeros / what was there) and with different rows per launch; the gradients
    themselves are those of the plain entry point bit for bit."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(22)
    e, n1, n2, k = 5, 512, 256, 768
    ws = torch.full((ops.grouped_gemm_tn_sumsq_slots(e, n1 + n2, k) + 8,), float("nan"), dtype=torch.float32, device=DEV)
    assert ws.numel() - 8 == e * ((n1 + n2) // 256) * (k // 256) * 8
    dw, dw2 = torch.empty(e, n1, k, dtype=torch.float32, device=DEV), torch.empty(e, n2, k, dtype=torch.float32, device=DEV)
    rw, rw2 = torch.empty_like(dw), torch.empty_like(dw2)
    for step, sizes in enumerate(([300, 0, 1, 255, 513], [0, 700, 64, 0, 3])):
        m = sum(sizes)
        offsets = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
        x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
        dy = torch.randn(m, n1 + n2, generator=g).to(torch.bfloat16).to(DEV)
        ops.grouped_gemm_tn_f32(dy, x, offsets, dw, dw2, accumulate=step > 0, tile_sumsq=ws[:-8])
        ops.grouped_gemm_tn_f32(dy, x, offsets, rw, rw2, accumulate=step > 0)
        assert torch.equal(dw, rw) and torch.equal(dw2, rw2)
        want = (dw.double() ** 2).sum() + (dw2.double() ** 2).sum()
        got = ws[:-8].double().sum()
        assert abs(got.item() - want.item()) <= 1e-6 * want.item(), (step, got.item(), want.item())
        assert torch.isnan(ws[-8:]).all()
    with pytest.raises(Exception):
        ops.grouped_gemm_tn_f32(dy, x, offsets, dw, dw2, tile_sumsq=ws[:16])  # too few slots


def test_grouped_gemm_rejects_unsupported_shapes_loudly():
    ops = _ops()
    from prime_rl_amd._lib import PrimeMI355XError

    assert not ops.grouped_gemm_supported(100, 96, 128)
    x = torch.zeros(8, 96, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(2, 128, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(PrimeMI355XError, match="multiples of 128"):
        ops.grouped_gemm_nt(x, w, torch.tensor([0, 4, 8], dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("n,e,dim", [(1000, 128, 256), (16384, 128, 2048), (777, 256, 384)])
def test_router_gate_weight_gradient_on_the_own_core_matches_autograd(n, e, dim):
    """_gate_linear (the router's nn.Linear with the split-K weight-gradient product of csrc/gemm.hip) vs the plain module under autograd: same logits bit
    for bit (same library product), row gradient identical, weight gradient = one bf16 rounding of the f32-accumulated g^T x."""
    from prime_rl_amd.trainer.models.layers import moe as M

    torch.manual_seed(1)
    gate = torch.nn.Linear(dim, e, bias=False).to(DEV).to(torch.bfloat16)
    x = torch.randn(n, dim, device=DEV).to(torch.bfloat16)
    g = torch.randn(n, e, device=DEV).to(torch.bfloat16)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = M._gate_linear(gate, xa)
    assert ya.grad_fn is not None and "GateLinear" in type(ya.grad_fn).__name__
    ya.backward(g)
    gwa, gate.weight.grad = gate.weight.grad.clone(), None
    yb = gate(xb)
    yb.backward(g)
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
    ref = g.float().t() @ x.float()
    tol = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * float(ref.abs().mean())
    assert ((gwa.float() - ref).abs() <= tol).all()
    assert ((gate.weight.grad.float() - ref).abs() <= 2 * tol).all()  # (the library's own rounding of the same sum)
    small = torch.nn.Linear(dim, 8, bias=False).to(DEV).to(torch.bfloat16)  # 8 experts: below the core's K-major row minimum -> the module itself
    assert "GateLinear" not in type(M._gate_linear(small, xa).grad_fn).__name__


def test_moe_layer_on_the_mfma_grouped_gemms_matches_the_loop_path():
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    torch.manual_seed(0)
    dim, hidden, e, k, n = 256, 128, 8, 2, 777
    mk = lambda grouped: MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k,
                                     use_grouped_mm=grouped, load_balance_coeff=None), dim, hidden).to(DEV).to(torch.bfloat16)
    a, b = mk(True), mk(False)
    a.init_weights(0.05)
    a.experts.grouped_impl = "hip"
    b.load_state_dict(a.state_dict())
    x = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    assert torch.allclose(ya.float(), yb.float(), rtol=2e-2, atol=2e-3)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    # the fused SwiGLU path rounds at fewer points than the eager loop (tools/probes/moe_paths_vs_f32.py: 0.46 % vs 0.50 % mean error of the
    # row gradient against an f32 evaluation): compare in the mean
    assert (xa.grad.float() - xb.grad.float()).abs().mean() <= 0.02 * xb.grad.float().abs().mean()
    for w in ("w1", "w2", "w3"):
        ga, gb = getattr(a.experts, w).grad.float(), getattr(b.experts, w).grad.float()
        assert (ga - gb).abs().mean() <= 0.02 * gb.abs().mean(), w


@pytest.mark.parametrize("n,e,k,dim,hidden,grouped", [(1, 8, 2, 128, 128, True), (3, 64, 8, 256, 128, True), (257, 4, 4, 128, 256, True), (65, 8, 3, 64, 96, True),
                                                      (40, 8, 2, 64, 32, False)])
def test_moe_layer_edge_shapes_against_the_oracle(n, e, k, dim, hidden, grouped):
    """One token, more experts than routed rows (most groups empty), every expert selected by every token (k = E), dimensions that are not
    multiples of 128 (library grouped GEMM / loop), odd k: the layer against oracle/moe_oracle.py with the same routing, forward and backward."""
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    torch.manual_seed(n * 7 + e)
    layer = MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k, use_grouped_mm=grouped,
                        load_balance_coeff=None), dim, hidden).to(DEV).to(torch.bfloat16)
    layer.init_weights(0.05)
    x = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
    try:
        y = layer(x)
    except (RuntimeError, NotImplementedError) as err:
        if grouped and (dim % 128 or hidden % 128):
            pytest.skip(f"library grouped GEMM unavailable for this shape: {err}")
        raise
    y.backward(gy)
    idx = layer.router.last_sort[0]
    args = M.MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False, top_k=k,
                     load_balance_coeff=None)
    params = {k_: v.detach().float().clone().requires_grad_(True) for k_, v in layer.state_dict().items() if v.dtype == torch.bfloat16}
    xo = x.detach().float().reshape(-1, dim).clone().requires_grad_(True)
    yo, aux = M.moe_forward(xo, params, args, routed_experts=idx.long())
    yo.backward(gy.float().reshape(-1, dim))
    assert torch.equal(aux["counts"].float().cpu(), layer.tokens_per_expert.cpu())
    tol = lambda ref: 0.02 * ref.abs().mean() + 1e-6
    assert (y.float().reshape(-1, dim) - yo).abs().mean() <= tol(yo.detach())
    assert (x.grad.float().reshape(-1, dim) - xo.grad).abs().mean() <= tol(xo.grad)
    # the gate gradient goes through the top-k normalisation, i.e. through DIFFERENCES of the bf16-rounded score gradients <gy, y_row>: with a
    # handful of tokens nothing averages that rounding out, so it is compared from 32 tokens up (the router kernels have their own f32 test)
    for name in ("experts.w1", "experts.w2", "experts.w3") + (("router.gate.weight",) if n >= 32 else ()):
        got = dict(layer.named_parameters())[name].grad.float()
        assert (got - params[name].grad).abs().mean() <= 0.03 * params[name].grad.abs().mean() + 1e-6, name
