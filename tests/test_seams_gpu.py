"""GPU suite: the drop-in seams (prime_rl_amd.trainer.rl.loss, ...models.layers.lm_head) pass the reference's own unit tests.

Each test restates one test of the reference (file:line cited) against our module with the same inputs, tolerances and
assertions; inputs live on the GPU because the kernels only run there."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _baseline(hidden, weight, labels, temperature):
    logits = (hidden @ weight.t()) / temperature.unsqueeze(-1)
    logp = torch.log_softmax(logits, dim=-1).gather(dim=-1, index=labels.unsqueeze(-1)).squeeze(-1)
    pd = torch.softmax(logits, dim=-1)
    ent = torch.logsumexp(logits, dim=-1) - (pd * logits).sum(-1)
    return logp, ent


def test_fused_lm_head_matches_full_logits_forward_and_backward():
    # reference tests/unit/train/rl/test_fused_lm_head.py:25-61 (atol 1e-5 on logprobs, entropy, grad_hidden, grad_weight)
    from prime_rl_amd.trainer.models.layers.lm_head import FusedOutputLinear

    torch.manual_seed(0)
    b, s, h, v = 2, 4, 8, 37
    temperature = torch.full((b, s), 1.7, device=DEV)
    hidden0 = torch.randn(b, s, h, device=DEV, requires_grad=True)
    labels = torch.randint(0, v, (b, s), device=DEV)
    weight0 = torch.randn(v, h, device=DEV, requires_grad=True)
    logp0, ent0 = _baseline(hidden0, weight0, labels, temperature)
    logp0.sum().backward()
    hidden1 = hidden0.detach().clone().requires_grad_(True)
    lm = FusedOutputLinear(in_features=h, out_features=v, chunk_size=3).to(DEV)
    lm.weight = torch.nn.Parameter(weight0.detach().clone())
    out = lm(hidden1, labels, temperature=temperature)
    assert out.get("logits") is None and out.get("logprobs") is not None and out.get("entropy") is not None
    out["logprobs"].sum().backward()
    torch.testing.assert_close(out["logprobs"], logp0, rtol=0, atol=1e-5)
    torch.testing.assert_close(out["entropy"], ent0.detach(), rtol=0, atol=1e-5)
    torch.testing.assert_close(hidden1.grad, hidden0.grad, rtol=0, atol=1e-5)
    torch.testing.assert_close(lm.weight.grad, weight0.grad, rtol=0, atol=1e-5)


def test_fused_lm_head_frozen_weight_and_asserts():
    # :64-89 (frozen weight -> no weight grad) and :92-105 (labels required)
    from prime_rl_amd.trainer.models.layers.lm_head import FusedOutputLinear, VanillaOutputLinear

    torch.manual_seed(0)
    b, s, h, v = 2, 4, 8, 37
    temperature = torch.full((b, s), 1.7, device=DEV)
    hidden = torch.randn(b, s, h, device=DEV, requires_grad=True)
    labels = torch.randint(0, v, (b, s), device=DEV)
    lm = FusedOutputLinear(in_features=h, out_features=v, chunk_size=3).to(DEV)
    lm.weight = torch.nn.Parameter(torch.randn(v, h, device=DEV), requires_grad=False)
    lm(hidden, labels, temperature=temperature)["logprobs"].sum().backward()
    assert lm.weight.grad is None and hidden.grad is not None
    with pytest.raises(AssertionError, match="FusedOutputLinear requires labels"):
        lm(hidden, labels=None, temperature=temperature)
    vl = VanillaOutputLinear(in_features=h, out_features=v).to(DEV)
    o = vl(hidden.detach(), labels=None, temperature=None)
    assert o.get("logits") is not None and o.get("logprobs") is None  # :108-126


def test_fused_vs_vanilla_bf16():
    # :129-161 / :398-457: fused vs vanilla post-processing, rtol 1e-3 atol 1e-4, in bf16 on the GPU, vocab crossing the chunk size
    from prime_rl_amd.trainer.models.layers.lm_head import FusedOutputLinear, VanillaOutputLinear, cast_float_and_contiguous
    from prime_rl_amd.trainer.rl.loss import compute_entropy, selective_log_softmax

    torch.manual_seed(42)
    b, s, h, v = 2, 33, 64, 20011
    temperature = torch.where(torch.rand(b, s, device=DEV) < 0.5, 0.7, 1.7)
    hidden = torch.randn(b, s, h, device=DEV).to(torch.bfloat16)
    labels = torch.randint(0, v, (b, s), device=DEV)
    weight = (torch.randn(v, h, device=DEV) * 0.3).to(torch.bfloat16)
    vl = VanillaOutputLinear(in_features=h, out_features=v).to(DEV).to(torch.bfloat16)
    vl.weight = torch.nn.Parameter(weight.clone())
    logits = cast_float_and_contiguous(vl(hidden))["logits"] / temperature.unsqueeze(-1)
    lp_v = selective_log_softmax(logits, labels)
    ent_v = compute_entropy(logits)
    fl = FusedOutputLinear(in_features=h, out_features=v, chunk_size=16).to(DEV).to(torch.bfloat16)
    fl.weight = torch.nn.Parameter(weight.clone())
    out = cast_float_and_contiguous(fl(hidden, labels=labels, temperature=temperature))
    torch.testing.assert_close(out["logprobs"], lp_v, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["entropy"], ent_v, rtol=1e-3, atol=1e-4)
    # and the vanilla helpers agree with plain torch
    torch.testing.assert_close(lp_v, torch.log_softmax(logits, -1).gather(-1, labels[..., None])[..., 0], rtol=0, atol=1e-5)


def test_selective_log_softmax_backward():
    from prime_rl_amd.trainer.rl.loss import selective_log_softmax

    torch.manual_seed(1)
    logits = (torch.randn(2, 5, 1003, device=DEV) * 2).requires_grad_(True)
    idx = torch.randint(0, 1003, (2, 5), device=DEV)
    w = torch.randn(2, 5, device=DEV)
    (selective_log_softmax(logits, idx) * w).sum().backward()
    ref = logits.detach().clone().requires_grad_(True)
    (torch.log_softmax(ref, -1).gather(-1, idx[..., None])[..., 0] * w).sum().backward()
    torch.testing.assert_close(logits.grad, ref.grad, rtol=1e-4, atol=1e-6)


def test_shift_alignment():
    # :253-298: position 0 == pad after shift_tensor_right; labels from shift_tensor_left
    import math

    from prime_rl_amd.trainer.rl.loss import shift_tensor_left, shift_tensor_right

    ids = torch.tensor([[5, 6, 7, 8]], device=DEV)
    assert shift_tensor_left(ids).tolist() == [[6, 7, 8, 0]]
    lp = torch.tensor([[0.1, 0.2, 0.3, 0.4]], device=DEV)
    out = shift_tensor_right(lp, pad_value=math.log(1 / 37))
    assert abs(out[0, 0].item() - math.log(1 / 37)) < 1e-6 and out[0, 1:].tolist() == pytest.approx([0.1, 0.2, 0.3])


# ---- loss seam: reference tests/unit/train/rl/test_loss.py --------------------------------------------------------------
def _c(*a, **k):
    from prime_rl_amd.trainer.rl.loss import compute_loss

    return compute_loss(*a, **k)


def test_grpo_loss_shapes_and_custom_plugin():
    # :12-63 (scalar loss for two packed samples) and :66-85 (CustomLossConfig import path + kwargs)
    from prime_rl_amd.trainer.rl.loss import CustomLossConfig, DefaultLossConfig, LossInputs, LossOutputs, setup_rl_loss_fn

    r = lambda n: torch.randn(n, device=DEV)
    fn = setup_rl_loss_fn(DefaultLossConfig(dppo_mask_high=10.0))
    loss, metrics = _c([r(50), r(30)], [r(50), r(30)], [r(50), r(30)], [r(50), r(30)], [torch.ones(50, dtype=torch.bool, device=DEV), torch.ones(30, dtype=torch.bool, device=DEV)],
                       None, None, None, rl_loss_fn=fn, rl_scale=1, ce_scale=1, ref_kl_scale=1)
    assert loss.shape == () and metrics["is_masked"].shape == (2,)
    cfn = setup_rl_loss_fn(CustomLossConfig(import_path="tests.test_seams_gpu._dummy_custom_loss", kwargs={"multiplier": 2.0}))
    res = cfn(LossInputs(trainer_logprobs=r(50), inference_logprobs=r(50), ref_logprobs=None, advantages=r(50), loss_mask=torch.ones(50, dtype=torch.bool, device=DEV)))
    assert isinstance(res, LossOutputs) and res.loss.shape == () and "custom_metric" in res.metrics
    loss, metrics = _c([r(50)], [r(50)], None, [r(50)], [torch.ones(50, dtype=torch.bool, device=DEV)], None, None, None, rl_loss_fn=cfn, rl_scale=1, ce_scale=1,
                       ref_kl_scale=1)
    assert loss.shape == () and "custom_metric" in metrics


def _dummy_custom_loss(inputs, multiplier: float = 1.0):
    from prime_rl_amd.trainer.rl.loss import LossOutputs

    loss = (inputs.trainer_logprobs[inputs.loss_mask].sum() * multiplier).abs()
    return LossOutputs(loss=loss, metrics={"custom_metric": torch.tensor(multiplier, device=loss.device)})


def test_ce_known_answers_and_components():
    # :88-115 (0.15), :118-144 (0.03), :147-172 (torch.equal), :175-210 (disjoint components), :213-242 (empty), :245-274 (additivity)
    from prime_rl_amd.trainer.rl.loss import DefaultLossConfig, setup_rl_loss_fn

    t = lambda x: torch.tensor(x, dtype=torch.float32, device=DEV)
    fn = setup_rl_loss_fn(DefaultLossConfig())
    kw = dict(trainer_logprobs=[t([-0.1, -0.5, -0.2])], inference_logprobs=[torch.zeros(3, device=DEV)], ref_logprobs=None, advantages=[torch.zeros(3, device=DEV)],
              loss_mask=[torch.tensor([True, False, True], device=DEV)], rl_weights=[torch.zeros(3, device=DEV)], ref_kl_weights=None, rl_loss_fn=fn, rl_scale=1,
              ref_kl_scale=1)
    loss, metrics = _c(ce_weights=[t([1.0, 0.0, 1.0])], ce_scale=2, **kw)
    assert torch.isclose(loss, t(0.15), atol=1e-6) and "nll" in metrics and "mismatch_kl" not in metrics and "is_masked" not in metrics
    loss, _ = _c(ce_weights=[t([0.1, 0.0, 0.1])], ce_scale=1, **kw)
    assert torch.isclose(loss, t(0.03), atol=1e-6)

    torch.manual_seed(0)
    base = dict(trainer_logprobs=[torch.randn(50, device=DEV)], inference_logprobs=[torch.randn(50, device=DEV)], ref_logprobs=None, advantages=[torch.randn(50, device=DEV)],
                loss_mask=[torch.rand(50, device=DEV) > 0.3], ce_weights=None, ref_kl_weights=None, rl_loss_fn=fn, rl_scale=1, ce_scale=1, ref_kl_scale=1)
    la, _ = _c(rl_weights=None, **base)
    lb, _ = _c(rl_weights=[torch.ones(50, device=DEV)], **base)
    assert torch.equal(la, lb)

    n = 12
    torch.manual_seed(1)
    rl_w = torch.zeros(n, device=DEV); rl_w[:4] = 1
    ce_w = torch.zeros(n, device=DEV); ce_w[4:8] = 1
    rk_w = torch.zeros(n, device=DEV); rk_w[8:] = 1
    loss, metrics = _c([torch.randn(n, device=DEV)], [torch.randn(n, device=DEV)], [torch.randn(n, device=DEV)], [torch.randn(n, device=DEV)],
                       [torch.ones(n, dtype=torch.bool, device=DEV)], [rl_w], [ce_w], [rk_w], rl_loss_fn=setup_rl_loss_fn(DefaultLossConfig(dppo_mask_high=10.0)), rl_scale=1,
                       ce_scale=1, ref_kl_scale=1)
    assert loss.shape == () and "nll" in metrics and "ref_kl" in metrics and "is_masked" in metrics

    lp = torch.randn(6, device=DEV, requires_grad=True)
    z = torch.zeros(6, device=DEV)
    loss, _ = _c([lp], [z], None, [z], [torch.zeros(6, dtype=torch.bool, device=DEV)], [z], [z], None, rl_loss_fn=fn, rl_scale=1, ce_scale=1, ref_kl_scale=1)
    assert torch.equal(loss, torch.zeros_like(loss))
    loss.backward()
    assert lp.grad is not None and torch.equal(lp.grad, torch.zeros_like(lp.grad))

    n = 8
    torch.manual_seed(2)
    kw = dict(trainer_logprobs=[torch.randn(n, device=DEV)], inference_logprobs=[torch.randn(n, device=DEV)], ref_logprobs=None, advantages=[torch.randn(n, device=DEV)],
              loss_mask=[torch.ones(n, dtype=torch.bool, device=DEV)], ref_kl_weights=None, rl_loss_fn=setup_rl_loss_fn(DefaultLossConfig(dppo_mask_high=10.0)), rl_scale=4,
              ce_scale=8, ref_kl_scale=1)
    cw = [torch.full((n,), 0.5, device=DEV)]
    rl_only, _ = _c(rl_weights=None, ce_weights=None, **kw)
    ce_only, _ = _c(rl_weights=[torch.zeros(n, device=DEV)], ce_weights=cw, **kw)
    both, _ = _c(rl_weights=None, ce_weights=cw, **kw)
    assert torch.isclose(both, rl_only + ce_only, atol=1e-6)


def test_compute_loss_autograd_matches_golden(golden_dir):
    from prime_rl_amd.trainer.rl.loss import DefaultLossConfig, setup_rl_loss_fn

    g = dict(np.load(golden_dir / "loss_default_weights.npz"))
    lens = g["lens"].tolist()
    t = lambda k: list(torch.tensor(g[k], device=DEV).split(lens))
    lp = torch.tensor(g["trainer_logprobs"], device=DEV, requires_grad=True)
    loss, metrics = _c(list(lp.split(lens)), t("inference_logprobs"), t("ref_logprobs"), t("advantages"), t("loss_mask"), t("rl_weights"), t("ce_weights"),
                       t("ref_kl_weights"), rl_loss_fn=setup_rl_loss_fn(DefaultLossConfig()), rl_scale=int(g["scales"][0]), ce_scale=int(g["scales"][1]),
                       ref_kl_scale=int(g["scales"][2]))
    (loss * 3.0).backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lp.grad.cpu().numpy(), 3.0 * g["grad"], rtol=1e-5, atol=1e-6)
    for k in [k for k in g if k.startswith("metric__")]:
        key = k[len("metric__"):].replace("__", "/")
        np.testing.assert_allclose(metrics[key].cpu().numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=key)


# ---- orchestrator algorithm plugin: reference tests/unit/orchestrator/test_advantage.py:173-240 ------------------------------
def _episode(rewards, n_out=None):
    from prime_rl_amd.orchestrator.algo.base import SimpleEpisode, SimpleNode, SimpleTrace

    traces = []
    for i, r in enumerate(rewards):
        k = (n_out[i] if n_out else 4)
        traces.append(SimpleTrace(f"t{i}", r, [SimpleNode(list(range(k + 2)), [False, False] + [True] * k)], num_output_tokens=k, num_total_tokens=k + 2))
    return [SimpleEpisode([t]) for t in traces]


def test_grpo_algorithm_plugin(device="cuda"):
    import asyncio

    from prime_rl_amd.orchestrator.algo import ALGORITHM_CLASSES, GRPOAlgoConfig, LengthPenaltyConfig, build_algorithm

    algo = build_algorithm(GRPOAlgoConfig(), device=device)
    eps = _episode([1.0, 0.0, 0.0, 1.0, 1.0])
    asyncio.run(algo.finalize_group(eps))
    advs = [e.traces[0].nodes[0].advantages[0] for e in eps]
    assert abs(sum(advs)) < 1e-6 and advs[0] == pytest.approx(0.4) and len(eps[0].traces[0].nodes[0].advantages) == 4
    single = _episode([0.7])
    asyncio.run(algo.finalize_group(single))
    assert single[0].traces[0].nodes[0].advantages == [0.0] * 4  # singleton group -> zero
    mx = ALGORITHM_CLASSES["max_rl"](device=device)
    eps = _episode([1.0, 0.0, 0.0, 0.0])
    asyncio.run(mx.finalize_group(eps))
    assert [e.traces[0].nodes[0].advantages[0] for e in eps] == pytest.approx([3.0, -1.0, -1.0, -1.0])
    eps = _episode([0.0, 0.0, 0.0])
    asyncio.run(mx.finalize_group(eps))
    assert [e.traces[0].nodes[0].advantages[0] for e in eps] == [0.0, 0.0, 0.0]
    # length penalty monotonicity (:198-240): equal rewards, longer outputs get lower advantages; many groups in one launch
    lp = build_algorithm(GRPOAlgoConfig(length_penalty=LengthPenaltyConfig()), device=device)
    g1, g2 = _episode([1.0, 1.0, 1.0], n_out=[10, 100, 1000]), _episode([1.0, 0.0], n_out=[5, 5])
    lp.score_groups([g1, g2])
    a = [e.traces[0].nodes[0].advantages[0] for e in g1]
    assert a[0] > a[1] > a[2] and abs(sum(a)) < 1e-5
    b = [e.traces[0].nodes[0].advantages[0] for e in g2]
    assert abs(sum(b)) < 1e-6 and b[0] > 0 > b[1]


def test_length_penalty_terms_like_the_reference(device="cuda"):
    """The remaining properties of tests/unit/orchestrator/test_advantage.py:184-240 on pm_grpo_advantage: MaxRL all-success groups
    center to zero; equal lengths/turns reduce the linear penalty to plain GRPO; the context term penalises non-completion tokens
    (total - output), the turns term higher turn counts; advantages stay zero-mean."""
    import asyncio

    from prime_rl_amd.orchestrator.algo import ALGORITHM_CLASSES, GRPOAlgoConfig, LengthPenaltyConfig, build_algorithm
    from prime_rl_amd.orchestrator.algo.base import SimpleEpisode, SimpleNode, SimpleTrace

    def group(rewards, out, total=None, turns=None):
        eps = []
        for i, r in enumerate(rewards):
            t = SimpleTrace(f"t{i}", r, [SimpleNode([1, 2, 3], [False, True, True])], num_output_tokens=out[i],
                            num_total_tokens=(total[i] if total else out[i] + 2), num_turns=(turns[i] if turns else 1))
            eps.append(SimpleEpisode([t]))
        return eps

    def run(algo, eps):
        asyncio.run(algo.finalize_group(eps))
        return [e.traces[0].nodes[0].advantages[0] for e in eps]

    assert run(ALGORITHM_CLASSES["max_rl"](device=device), group([1.0, 1.0], [5, 5])) == pytest.approx([0.0, 0.0])
    plain = run(build_algorithm(GRPOAlgoConfig(), device=device), group([1.0, 0.0, 1.0], [10, 10, 10], turns=[2, 2, 2]))
    shaped = run(build_algorithm(GRPOAlgoConfig(length_penalty=LengthPenaltyConfig()), device=device), group([1.0, 0.0, 1.0], [10, 10, 10], turns=[2, 2, 2]))
    assert shaped == pytest.approx(plain, abs=1e-6)
    ctx = LengthPenaltyConfig(num_output_tokens_weight=0.0, num_input_tokens_weight=0.25, num_turns_weight=0.0)
    a = run(build_algorithm(GRPOAlgoConfig(length_penalty=ctx), device=device), group([1.0, 1.0], [10, 10], total=[12, 112]))
    assert a[0] > a[1] and abs(sum(a)) < 1e-6
    trn = LengthPenaltyConfig(num_output_tokens_weight=0.0, num_input_tokens_weight=0.0, num_turns_weight=0.25)
    a = run(build_algorithm(GRPOAlgoConfig(length_penalty=trn), device=device), group([1.0, 1.0], [100, 100], turns=[1, 4]))
    assert a[0] > a[1] and abs(sum(a)) < 1e-6
    # exact value of the completion-only penalty: shaped_i = r_i - mean * w * out_i / max(out); advantages = shaped - mean(shaped)
    cmp_ = LengthPenaltyConfig(num_output_tokens_weight=0.25, num_input_tokens_weight=0.0, num_turns_weight=0.0)
    a = run(build_algorithm(GRPOAlgoConfig(length_penalty=cmp_), device=device), group([1.0, 1.0, 1.0], [10, 20, 30]))
    shaped_ref = [1.0 - 0.25 * x / 30 for x in (10, 20, 30)]
    mean_ref = sum(shaped_ref) / 3
    assert a == pytest.approx([s - mean_ref for s in shaped_ref], abs=1e-6)


def test_hierarchical_grpo_peer_sets(device="cuda"):
    """hierarchical_grpo.py:33-42: solver traces are centered per episode, proposer traces across the cohort, singletons get zero —
    all peer sets in one pm_grpo_advantage launch; checked against the plain restatement."""
    import asyncio

    from prime_rl_amd.orchestrator.algo import HierarchicalGRPOAlgoConfig, build_algorithm
    from prime_rl_amd.orchestrator.algo.base import SimpleAgent, SimpleEpisode, SimpleNode, SimpleTrace

    def tr(name, role, reward):
        return SimpleTrace(name, reward, [SimpleNode([1, 2, 3], [False, True, True])], agent=SimpleAgent(name=role))

    eps = [SimpleEpisode([tr("p0", "proposer", 0.2), tr("s00", "solver", 1.0), tr("s01", "solver", 0.0), tr("s02", "solver", 0.0)], id="e0"),
           SimpleEpisode([tr("p1", "proposer", 0.8), tr("s10", "solver", 1.0), tr("s11", "solver", 1.0)], id="e1"),
           SimpleEpisode([tr("p2", "proposer", 0.5), tr("s20", "solver", 0.3)], id="e2")]
    asyncio.run(build_algorithm(HierarchicalGRPOAlgoConfig(episode_agents=["solver"]), device=device).finalize_group(eps))
    got = {t.id: t.nodes[0].advantages for e in eps for t in e.traces}
    want = {"p0": 0.2 - 0.5, "p1": 0.8 - 0.5, "p2": 0.0, "s00": 1 - 1 / 3, "s01": -1 / 3, "s02": -1 / 3, "s10": 0.0, "s11": 0.0, "s20": 0.0}
    for k, v in want.items():
        assert got[k] == pytest.approx([v, v], abs=1e-6), k
