"""GPU suite (`-m gpu`): every HIP kernel, called through the C ABI, against the oracle / the golden fixtures.

Tolerances: bit-exact for the label gather and integer work; f32 kernels 1e-5 (the reference's own bound,
tests/unit/train/rl/test_fused_lm_head.py:58-61); bf16 kernels rtol 1e-3 / atol 1e-4 on log-probs (ibid. :160-161) and
bf16-resolution bounds (stated per test) on activations.
"""

import math

import numpy as np
import pytest
import torch

from oracle import grpo_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _attn_variants_build() -> bool:
    """`make VARIANTS=1` library?  (the A/B builds of the attention kernels exist only there; setting the word needs no GPU)"""
    from prime_rl_amd import _lib

    lib = _lib.load()
    ok = lib.pm_attn_debug_set_variant(1, 1) == 0
    lib.pm_attn_debug_set_variant(1, 0)
    return ok


ATTN_AB = _attn_variants_build()


@pytest.fixture(scope="module")
def ops():
    from prime_rl_amd import ops as _ops

    return _ops


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _close(got, want, rtol, atol, what):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()}/{bad.size} outside tol; max abs err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} (got {got.flat[err.argmax()]:.6g}, want {want.flat[err.argmax()]:.6g})"


def test_device_is_gfx950(ops):
    arch = ops.device_arch()
    assert arch.startswith("gfx950"), arch


def test_mfma_fragment_layout(ops):
    torch.manual_seed(0)
    a = torch.randn(16, 32).to(torch.bfloat16)
    b = (torch.randn(16, 32) + torch.arange(16)[:, None] * 0.25).to(torch.bfloat16)  # asymmetric on purpose (guide §3)
    c = ops.mfma_probe(a.to(DEV), b.to(DEV))
    _close(c, a.float() @ b.float().T, 1e-6, 1e-5, "mfma 16x16x32 layout")


# ================================================================================================ K1
def test_ds_read_tr16_semantics(ops):
    """Pins what the attention kernels assume about gfx950's `ds_read_b64_tr_b16`: every lane supplies the address of 4 contiguous
    16-bit elements; within each 16-lane group the 16 x 4 elements form a 4 x 16 block (lane p holds row p>>2, columns 4*(p&3)..+3)
    and lane i receives column i: result(i, j) = element (i & 3) of what lane 4*j + (i >> 2) of the same group addressed."""
    g = torch.Generator().manual_seed(5)
    linear = torch.arange(64, dtype=torch.int32) * 4
    scattered = (torch.randperm(1000, generator=g)[:64].to(torch.int32)) * 4  # arbitrary distinct 8-byte aligned addresses
    rows = torch.tensor([((l >> 4) * 8 + ((l & 15) >> 2)) * 128 + 4 * (l & 3) for l in range(64)], dtype=torch.int32)  # a [32][128] tile
    for idx in (linear, scattered, rows):
        got = ops.ds_read_tr16_probe(idx.to(DEV)).cpu()
        want = torch.empty(64, 4, dtype=torch.int32)
        for l in range(64):
            grp, i = l >> 4, l & 15
            for j in range(4):
                want[l, j] = idx[grp * 16 + 4 * j + (i >> 2)] + (i & 3)
        assert torch.equal(got, want), f"ds_read_b64_tr_b16 mapping differs:\nidx={idx.tolist()}\ngot={got.tolist()}"
    # the documented closed form for the linear image: lane l, element j <- lds[(l & 15) + 16 j + 64 (l >> 4)]
    got = ops.ds_read_tr16_probe(linear.to(DEV)).cpu()
    for l in (0, 5, 17, 63):
        assert got[l].tolist() == [(l & 15) + 16 * j + 64 * (l >> 4) for j in range(4)]


def test_logprob_vanilla_golden_unaligned(ops, golden_dir):
    g = dict(np.load(golden_dir / "logprob_vanilla.npz"))
    inv_t = _t(1.0 / g["temperature"], torch.float32)
    lp, ent, _ = ops.logprob_entropy_fwd(_t(g["logits"]), _t(g["labels"]), inv_t)
    _close(lp, g["logprobs"], 0, 1e-5, "logprobs")
    _close(ent, g["entropy"], 0, 1e-5, "entropy")


def test_logprob_fp32_head_golden(ops, golden_dir):
    g = dict(np.load(golden_dir / "lm_head_fp32.npz"))
    hidden = g["hidden"].reshape(-1, g["hidden"].shape[-1])
    logits = _t(hidden @ g["weight"].T)
    inv_t = _t(1.0 / g["temperature"].reshape(-1), torch.float32)
    lp, ent, logz = ops.logprob_entropy_fwd(logits, _t(g["labels"].reshape(-1)), inv_t)
    _close(lp, g["logprobs"].reshape(-1), 0, 1e-5, "logprobs")
    _close(ent, g["entropy"].reshape(-1), 0, 1e-5, "entropy")
    gl = _t(g["grad_logprobs"].reshape(-1))
    dl = ops.logprob_bwd(logits, _t(g["labels"].reshape(-1)), inv_t, logz, gl)
    dl_ref, gh_ref, gw_ref = O.fused_head_backward(hidden, g["weight"], g["labels"].reshape(-1), (1.0 / g["temperature"].reshape(-1)).astype(np.float32),
                                                   logz.cpu().numpy(), g["grad_logprobs"].reshape(-1))
    _close(dl, dl_ref, 0, 1e-5, "dlogits")
    # the two GEMMs of lm_head.py:194-197 on the kernel's dlogits reproduce the reference's gradients
    _close(dl.cpu() @ torch.from_numpy(g["weight"]), g["grad_hidden"].reshape(hidden.shape), 0, 1e-5, "grad_hidden")
    _close(dl.cpu().T @ torch.from_numpy(hidden), g["grad_weight"], 0, 1e-5, "grad_weight")


def test_logprob_bf16_head_golden(ops, golden_dir):
    g = dict(np.load(golden_dir / "lm_head_bf16.npz"))
    hidden = _t(g["hidden"][0], torch.bfloat16)
    weight = _t(g["weight"], torch.bfloat16)
    logits = hidden @ weight.T  # bf16 logits, as the reference's chunk GEMM produces them (lm_head.py:133)
    inv_t = _t(1.0 / g["temperature"][0], torch.float32)
    lp, ent, _ = ops.logprob_entropy_fwd(logits, _t(g["labels"][0]), inv_t)
    _close(lp, g["logprobs"][0], 1e-3, 1e-4, "logprobs bf16")
    _close(ent, g["entropy"][0], 1e-3, 1e-4, "entropy bf16")
    # vocab-chunked form: two chunks of the same logits give the same answer
    st = ops.ChunkedLogprobState(logits.shape[0], logits.device)
    st.update(logits[:, :8192], 0, _t(g["labels"][0]), inv_t)
    st.update(logits[:, 8192:], 8192, _t(g["labels"][0]), inv_t)
    lp2, ent2, _ = st.finalize()
    _close(lp2, lp, 0, 2e-6, "chunked == one-shot (logprobs)")
    _close(ent2, ent, 0, 2e-6, "chunked == one-shot (entropy)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n_rows", [2048, 8])
def test_logprob_full_vocab(ops, dtype, n_rows):
    """BASELINE shape (V = 151936): label gather bit-exact, values vs the oracle on a row sample, size-independent properties."""
    v = 151936
    g = torch.Generator(device=DEV).manual_seed(n_rows)
    logits = (torch.randn(n_rows, v, device=DEV, generator=g) * 4.0).to(dtype)
    labels = torch.randint(0, v, (n_rows,), device=DEV, generator=g)
    temp = torch.where(torch.rand(n_rows, device=DEV, generator=g) < 0.5, 0.7, 1.0).float()
    inv_t = 1.0 / temp
    lp, ent, logz = ops.logprob_entropy_fwd(logits, labels, inv_t)
    # bit-exact token-index gather: logprob + logz is exactly the label logit times inv_T
    tl = logits[torch.arange(n_rows, device=DEV), labels].float() * inv_t
    assert torch.equal(lp, tl - logz), "label gather is not bit-exact"
    rows = np.linspace(0, n_rows - 1, num=min(n_rows, 6)).astype(int)
    lp_o, ent_o, logz_o = O.logprob_entropy_from_logits(logits[rows].float().cpu().numpy(), labels[rows].cpu().numpy(), temp[rows].cpu().numpy())
    _close(lp[rows], lp_o, 0, 2e-5, "logprobs vs oracle")
    _close(ent[rows], ent_o, 0, 2e-5, "entropy vs oracle")
    # properties over ALL rows: agreement with an independent f32 implementation, bounds, shift invariance
    ref = torch.log_softmax(logits.float() * inv_t[:, None], dim=-1)
    _close(lp, ref[torch.arange(n_rows, device=DEV), labels], 0, 2e-5, "logprobs vs torch.log_softmax (all rows)")
    assert (lp <= 0).all() and (ent >= 0).all() and (ent <= math.log(v) + 1e-4).all()
    if dtype == torch.float32:
        lp_s, ent_s, logz_s = ops.logprob_entropy_fwd(logits + 3.0, labels, inv_t)
        _close(lp_s, lp, 0, 3e-5, "shift invariance of logprobs")
        _close(logz_s - logz, 3.0 * inv_t, 0, 3e-5, "logz shifts by c/T")


@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16),
                                          (torch.bfloat16, torch.float32)])
def test_logprob_bwd(ops, in_dt, out_dt):
    n, v = 64, 151936
    g = torch.Generator(device=DEV).manual_seed(7)
    logits = (torch.randn(n, v, device=DEV, generator=g) * 3.0).to(in_dt)
    labels = torch.randint(0, v, (n,), device=DEV, generator=g)
    inv_t = (1.0 / torch.where(torch.rand(n, device=DEV, generator=g) < 0.5, 0.7, 1.0)).float()
    grad = torch.randn(n, device=DEV, generator=g)
    grad[::5] = 0.0  # masked tokens: rows of exact zeros
    _, _, logz = ops.logprob_entropy_fwd(logits, labels, inv_t)
    dl = ops.logprob_bwd(logits, labels, inv_t, logz, grad, out_dtype=out_dt)
    assert dl.dtype == out_dt
    l = logits.float() * inv_t[:, None]
    want = (grad * inv_t)[:, None] * (torch.nn.functional.one_hot(labels, v).float() - torch.exp(l - logz[:, None]))
    tol = 1e-6 if out_dt == torch.float32 else 4e-3
    _close(dl, want, tol if out_dt == torch.bfloat16 else 1e-4, 1e-6 if out_dt == torch.float32 else 1e-6, "dlogits")
    assert not dl[::5].float().any(), "masked rows must be exactly zero"
    # sum_v dlogits = a*(1 - sum softmax) = 0; with bf16 outputs each of the 151936 terms carries up to 2^-9 relative rounding
    _close(dl.float().sum(-1), torch.zeros(n), 0, 5e-2 if out_dt == torch.bfloat16 else 1e-4, "rows of dlogits sum to zero")
    if in_dt == out_dt:  # in place
        buf = logits.clone()
        ops.logprob_bwd(buf, labels, inv_t, logz, grad, out=buf)
        assert torch.equal(buf, dl), "in-place backward differs"


# ================================================================================================ K3
def test_grpo_advantage(ops):
    rng = np.random.default_rng(0)
    sizes = [16] * 8 + [1, 3, 70, 130]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rewards = (rng.random(offs[-1]) < 0.5).astype(np.float32)
    rewards[offs[-2]:] = rng.standard_normal(130).astype(np.float32)
    adv = ops.grpo_advantage(_t(rewards), _t(offs)).cpu().numpy()
    for a, b in zip(offs[:-1], offs[1:]):
        np.testing.assert_allclose(adv[a:b], O.grpo_advantages(rewards[a:b]), rtol=0, atol=1e-6)
        assert abs(adv[a:b].sum()) < 1e-4
    assert adv[offs[8]] == 0.0  # singleton group
    from prime_rl_amd import _lib
    mx = ops.grpo_advantage(_t(np.array([1, 0, 0, 0, 0, 0, 0], dtype=np.float32)), _t(np.array([0, 4, 7], dtype=np.int32)), mode=_lib.PM_ADV_MAXRL)
    np.testing.assert_allclose(mx.cpu().numpy(), [3, -1, -1, -1, 0, 0, 0], atol=1e-6)
    out, tot, turns = (rng.integers(1, 500, offs[-1]).astype(np.float32) for _ in range(3))
    tot = tot + out
    w = (0.25, 0.1, 0.1)
    sh = ops.grpo_advantage(_t(rewards), _t(offs), num_output_tokens=_t(out), num_total_tokens=_t(tot), num_turns=_t(turns), weights=w).cpu().numpy()
    for a, b in zip(offs[:-1], offs[1:]):
        want = O.grpo_advantages(rewards[a:b], dict(output=out[a:b], total=tot[a:b], turns=turns[a:b], w_output=w[0], w_input=w[1], w_turns=w[2]))
        np.testing.assert_allclose(sh[a:b], want, rtol=0, atol=2e-6)


def test_advantage_broadcast(ops):
    radv = np.array([0.7, -0.25, 0.0], dtype=np.float32)
    roffs = np.array([0, 3, 300, 1000], dtype=np.int32)
    mask = np.random.default_rng(0).random(1000) < 0.7
    mask[:3] = [False, True, True]
    got = ops.advantage_broadcast(_t(radv), _t(roffs), _t(mask)).cpu().numpy()
    want = np.concatenate([O.broadcast_advantage(radv[r], mask[roffs[r]:roffs[r + 1]]) for r in range(3)])
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got[:3], [0.0, 0.7, 0.7])


# ================================================================================================ K2
def _loss_params(cfg: O.LossConfig, scales, shift=0, vocab=151936):
    from prime_rl_amd import _lib

    return _lib.LossParams(loss_type=_lib.PM_LOSS_IPO if cfg.type == "ipo" else _lib.PM_LOSS_DEFAULT, dppo_mask_low=cfg.dppo_mask_low,
                           dppo_mask_high=cfg.dppo_mask_high, adv_tau=cfg.adv_tau, kl_tau=cfg.kl_tau, ipo_threshold=cfg.ipo_threshold,
                           rl_scale=float(scales[0]), ce_scale=float(scales[1]), ref_kl_scale=float(scales[2]), shift=shift,
                           pad_logprob=math.log(1.0 / vocab))


@pytest.mark.parametrize("name,cfg", [("loss_default.npz", O.LossConfig()), ("loss_default_weights.npz", O.LossConfig()),
                                      ("loss_ipo.npz", O.LossConfig(type="ipo"))])
def test_grpo_loss_golden(ops, golden_dir, name, cfg):
    from prime_rl_amd import _lib

    g = dict(np.load(golden_dir / name))
    cu = _t(np.concatenate([[0], np.cumsum(g["lens"])]).astype(np.int32))
    kw = {}
    if "rl_weights" in g:
        kw = dict(ref_logprobs=_t(g["ref_logprobs"]), rl_weights=_t(g["rl_weights"]), ce_weights=_t(g["ce_weights"]), ref_kl_weights=_t(g["ref_kl_weights"]))
    loss, grad, stats, metrics = ops.grpo_loss_fwd_bwd(_t(g["trainer_logprobs"]), _t(g["inference_logprobs"]), _t(g["advantages"]), _t(g["loss_mask"]), cu,
                                                       _loss_params(cfg, g["scales"]), **kw)
    _close(loss[0], g["loss"], 1e-5, 1e-6, "loss")
    _close(grad, g["grad"], 1e-5, 1e-7, "dloss/dlogprob")
    stats, metrics = stats.cpu().numpy(), metrics.cpu().numpy()
    comp_count = {"rl": stats[:, _lib.STAT["RL_N_MASK"]], "ce": stats[:, _lib.STAT["CE_N"]], "ref_kl": stats[:, _lib.STAT["REFKL_N"]]}
    for k in [k for k in g if k.startswith("metric__")]:
        key = k[len("metric__"):].replace("__", "/")
        col = _lib.METRIC_KEYS.index(key)
        comp = "ce" if key == "nll" else ("ref_kl" if key.startswith("ref_kl") else "rl")
        keep = comp_count[comp] > 0 if ("rl_weights" in g or comp != "rl") else np.ones(len(stats), bool)  # loss.py:367-380 appends only non-empty components
        _close(metrics[keep, col], g[k], 1e-5, 1e-6, key)


def test_grpo_loss_known_answers_and_edge_cases(ops):
    lp = _t(np.array([-0.1, -0.5, -0.2], dtype=np.float32))
    z = _t(np.zeros(3, dtype=np.float32))
    mask = _t(np.array([True, False, True]))
    cu = _t(np.array([0, 3], dtype=np.int32))
    loss, grad, _, _ = ops.grpo_loss_fwd_bwd(lp, z, z, mask, cu, _loss_params(O.LossConfig(), (1, 2, 1)), rl_weights=z,
                                             ce_weights=_t(np.array([1, 0, 1], dtype=np.float32)))
    assert abs(loss.item() - 0.15) < 1e-6  # reference tests/unit/train/rl/test_loss.py:88-115
    np.testing.assert_allclose(grad.cpu().numpy(), [-0.5, 0.0, -0.5], atol=1e-7)
    loss, _, _, _ = ops.grpo_loss_fwd_bwd(lp, z, z, mask, cu, _loss_params(O.LossConfig(), (1, 1, 1)), rl_weights=z,
                                          ce_weights=_t(np.array([0.1, 0, 0.1], dtype=np.float32)))
    assert abs(loss.item() - 0.03) < 1e-6  # :118-144
    # explicit all-ones rl stream is bitwise the absent stream (:147-172)
    rng = np.random.default_rng(0)
    a, b, c = (_t(rng.standard_normal(50).astype(np.float32)) for _ in range(3))
    m = _t(rng.random(50) > 0.3)
    cu50 = _t(np.array([0, 50], dtype=np.int32))
    p = _loss_params(O.LossConfig(), (1, 1, 1))
    l1, g1, _, _ = ops.grpo_loss_fwd_bwd(a, b, c, m, cu50, p)
    l2, g2, _, _ = ops.grpo_loss_fwd_bwd(a, b, c, m, cu50, p, rl_weights=torch.ones(50, device=DEV))
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    # empty components: exact zero loss and exact zero gradient (:213-242)
    z6 = torch.zeros(6, device=DEV)
    l0, g0, _, _ = ops.grpo_loss_fwd_bwd(torch.randn(6, device=DEV), z6, z6, torch.zeros(6, dtype=torch.bool, device=DEV),
                                         _t(np.array([0, 6], dtype=np.int32)), p, rl_weights=z6, ce_weights=z6)
    assert l0.item() == 0.0 and not g0.any()


def test_grpo_loss_fused_shift_and_microbatch_concat(ops):
    """shift_tensor_right fused in the kernel (loss.py:76-91 / train.py:450-457), two micro-batches concatenated in one launch."""
    rng = np.random.default_rng(3)
    vocab = 151936
    lens_a, lens_b = [40, 24], [30, 34]
    n = 128
    raw = rng.standard_normal(n).astype(np.float32) * 0.3 - 1.0
    inf = raw + rng.standard_normal(n).astype(np.float32) * 0.2
    adv = rng.standard_normal(n).astype(np.float32)
    mask = rng.random(n) > 0.3
    pad = math.log(1.0 / vocab)
    cfg = O.LossConfig()
    shifted = np.concatenate([O.shift_right(raw[:64][None], pad)[0], O.shift_right(raw[64:][None], pad)[0]])
    scale = max(int(mask.sum()), 1)
    la, _, ga = O.compute_loss(shifted[:64], inf[:64], adv[:64], mask[:64], lens_a, cfg, scale)
    lb, _, gb = O.compute_loss(shifted[64:], inf[64:], adv[64:], mask[64:], lens_b, cfg, scale)
    want_grad = np.zeros(n, dtype=np.float32)  # un-shift: gradient of position i lands on head position i-1, per micro-batch
    want_grad[0:63] = ga[1:]
    want_grad[64:127] = gb[1:]
    cu = _t(np.array([0, 40, 64, 94, 128], dtype=np.int32))
    starts = _t(np.array([1, 0, 1, 0], dtype=np.uint8))
    loss, grad, _, _ = ops.grpo_loss_fwd_bwd(_t(raw), _t(inf), _t(adv), _t(mask), cu, _loss_params(cfg, (scale, 1, 1), shift=1, vocab=vocab),
                                             sample_starts_mb=starts)
    _close(loss[0], np.float32(la + lb), 1e-5, 1e-6, "loss of two concatenated micro-batches")
    _close(grad, want_grad, 1e-5, 1e-7, "gradient in head indexing")


# ================================================================================================ elementwise / norms
@pytest.mark.parametrize("h", [4096, 1024, 256])
def test_rmsnorm(ops, h):
    torch.manual_seed(h)
    n = 300
    x = torch.randn(n, h).to(torch.bfloat16)
    res = torch.randn(n, h).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(h)).to(torch.bfloat16)
    y, rstd, _ = ops.rmsnorm_fwd(x.to(DEV), w.to(DEV), 1e-6)
    want = O.rmsnorm(x, w, 1e-6)
    mism = (y.cpu() != want).float().mean().item()
    assert mism < 2e-3, f"rmsnorm fwd: {mism:.4%} of outputs differ from the reference rounding"  # reduction-order ulps only
    _close(y, want, 8e-3, 1e-6, "rmsnorm fwd")
    y2, _, so = ops.rmsnorm_fwd(x.to(DEV), w.to(DEV), 1e-6, residual=res.to(DEV))
    assert torch.equal(so.cpu(), x + res)
    _close(y2, O.rmsnorm(x + res, w, 1e-6), 8e-3, 1e-6, "rmsnorm fwd (fused residual)")
    # backward vs f32 autograd of the same formula
    gy = torch.randn(n, h).to(torch.bfloat16)
    dres = torch.randn(n, h).to(torch.bfloat16)
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * wf * gy.float()).sum().backward()
    gw = torch.zeros(h, device=DEV)
    gx = ops.rmsnorm_bwd(gy.to(DEV), x.to(DEV), w.to(DEV), rstd, gw, dresidual=dres.to(DEV))
    _close(gx, xf.grad + dres.float(), 8e-3, 2e-3, "rmsnorm dx (+dresidual)")
    _close(gw, wf.grad, 2e-3, 2e-3, "rmsnorm dweight")
    gw2 = gw.clone()
    ops.rmsnorm_bwd(gy.to(DEV), x.to(DEV), w.to(DEV), rstd, gw2)
    _close(gw2, 2 * wf.grad, 2e-3, 4e-3, "rmsnorm dweight accumulates")


def test_rope_table_and_qknorm_rope(ops):
    torch.manual_seed(0)
    n, hq, hkv, d = 77, 8, 2, 128
    pos = torch.cat([torch.arange(50), torch.arange(27)]) + 1000
    cos, sin = ops.rope_table(pos.to(DEV), ops.rope_inv_freq(d, 1_000_000.0, DEV))
    cos_o, sin_o = O.rope_cos_sin(pos, d, 1_000_000.0, torch.bfloat16)
    assert (cos.cpu() != cos_o[:, :64]).float().mean() < 5e-3 and (sin.cpu() != sin_o[:, :64]).float().mean() < 5e-3
    _close(cos, cos_o[:, :64], 8e-3, 1e-6, "cos table")
    qkv = torch.randn(n, (hq + 2 * hkv) * d).to(torch.bfloat16)
    qw = (1 + 0.1 * torch.randn(d)).to(torch.bfloat16)
    kw = (1 + 0.1 * torch.randn(d)).to(torch.bfloat16)
    q, k, rstd = ops.qknorm_rope_fwd(qkv.to(DEV), qw.to(DEV), kw.to(DEV), cos, sin, hq, hkv, d, 1e-6)
    cos_f = torch.cat([cos.cpu(), cos.cpu()], -1)
    sin_f = torch.cat([sin.cpu(), sin.cpu()], -1)
    q_raw = qkv[:, : hq * d].view(n, hq, d)
    k_raw = qkv[:, hq * d: (hq + hkv) * d].view(n, hkv, d)
    q_o = O.apply_rope(O.rmsnorm(q_raw, qw, 1e-6), cos_f, sin_f)
    k_o = O.apply_rope(O.rmsnorm(k_raw, kw, 1e-6), cos_f, sin_f)
    assert (q.cpu() != q_o).float().mean() < 3e-3, "q differs from the reference's bf16 rounding sequence"
    _close(q, q_o, 1.6e-2, 1e-6, "q norm+rope")
    _close(k, k_o, 1.6e-2, 1e-6, "k norm+rope")
    # backward vs f32 autograd
    gq = torch.randn(n, hq, d).to(torch.bfloat16)
    gk = torch.randn(n, hkv, d).to(torch.bfloat16)
    qr = q_raw.float().requires_grad_(True)
    kr = k_raw.float().requires_grad_(True)
    qwf = qw.float().requires_grad_(True)
    kwf = kw.float().requires_grad_(True)

    def f(x, w):
        xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        return O.apply_rope(xn, cos_f.float(), sin_f.float())

    ((f(qr, qwf) * gq.float()).sum() + (f(kr, kwf) * gk.float()).sum()).backward()
    gqkv = torch.zeros_like(qkv).to(DEV)
    gqw = torch.zeros(d, device=DEV)
    gkw = torch.zeros(d, device=DEV)
    ops.qknorm_rope_bwd(gq.to(DEV), gk.to(DEV), qkv.to(DEV), qw.to(DEV), kw.to(DEV), cos, sin, rstd, gqkv, gqw, gkw, hq, hkv, d)
    _close(gqkv[:, : hq * d].view(n, hq, d), qr.grad, 1e-2, 4e-3, "dq_raw")
    _close(gqkv[:, hq * d: (hq + hkv) * d].view(n, hkv, d), kr.grad, 1e-2, 4e-3, "dk_raw")
    assert not gqkv[:, (hq + hkv) * d:].any(), "v columns must be left untouched"
    _close(gqw, qwf.grad, 3e-3, 3e-2, "d q_norm.weight")
    _close(gkw, kwf.grad, 3e-3, 3e-2, "d k_norm.weight")


def test_swiglu(ops):
    torch.manual_seed(1)
    n, inter = 130, 1536
    gu = (torch.randn(n, 2 * inter) * 2).to(torch.bfloat16)
    act = ops.swiglu_fwd(gu.to(DEV))
    want = O.swiglu(gu[:, :inter], gu[:, inter:])
    assert (act.cpu() != want).float().mean() < 2e-3
    _close(act, want, 8e-3, 1e-6, "swiglu fwd")
    ga = torch.randn(n, inter).to(torch.bfloat16)
    gf = gu.float().requires_grad_(True)
    (torch.nn.functional.silu(gf[:, :inter]) * gf[:, inter:] * ga.float()).sum().backward()
    _close(ops.swiglu_bwd(ga.to(DEV), gu.to(DEV)), gf.grad, 8e-3, 1e-3, "swiglu bwd")


@pytest.mark.parametrize("rows,cols", [(8, 8), (64, 64), (136, 72), (1, 8), (187, 512), (2043, 4096), (2048, 4096), (16384, 12288)])
def test_transpose_2d_bit_exact(ops, rows, cols):
    """Byte work: bit-exact against torch's transpose, incl. ragged tiles, row counts that are not a multiple of 8 (unpadded token
    counts), a strided source (a column slice of a wider matrix, like v inside the fused qkv) and a strided destination."""
    g = torch.Generator().manual_seed(rows * 131 + cols)
    src = torch.randint(-32768, 32767, (rows, cols), generator=g, dtype=torch.int16).view(torch.bfloat16).to(DEV)
    out = ops.transpose_2d(src)
    assert out.shape == (cols, rows) and torch.equal(out.view(torch.int16), src.view(torch.int16).t())
    if rows % 8 == 0:
        assert torch.equal(ops.transpose_2d(out).view(torch.int16), src.view(torch.int16))  # involution
    if cols >= 16 and rows % 8 == 0:
        wide = torch.zeros(rows, cols + 24, dtype=torch.bfloat16, device=DEV)
        wide[:, 8:8 + cols] = src
        dst_wide = torch.full((cols, rows + 16), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.transpose_2d(wide[:, 8:8 + cols], out=dst_wide[:, :rows])
        assert torch.equal(dst_wide[:, :rows].view(torch.int16), src.view(torch.int16).t())
        assert bool((dst_wide[:, rows:] == 7.0).all())  # nothing written past the requested extent


def test_embedding(ops):
    torch.manual_seed(2)
    v, h, n = 1000, 256, 500
    table = torch.randn(v, h).to(torch.bfloat16)
    ids = torch.randint(0, 40, (n,))  # heavy duplication, like FakeDataLoader ids < 100
    out = ops.embedding_fwd(table.to(DEV), ids.to(DEV))
    assert torch.equal(out.cpu(), table[ids])  # integer gather: bit-exact
    go = torch.randn(n, h).to(torch.bfloat16)
    gt = torch.zeros(v, h, device=DEV)
    ops.embedding_bwd(go.to(DEV), ids.to(DEV), gt)
    want = torch.zeros(v, h).index_add_(0, ids, go.float())
    _close(gt, want, 1e-5, 1e-5, "embedding bwd")
    gt2 = gt.clone()
    ops.embedding_bwd(go.to(DEV), ids.to(DEV), gt2)
    assert torch.equal(gt2, 2 * gt), "embedding bwd is not deterministic/accumulating"


def test_rope_without_qk_norm_and_column_sum(ops):
    """use_qk_norm = False (Llama / Qwen2 attention, models/llama/modeling_llama.py:59-60): pm_qknorm_rope_* with NULL norm weights is
    apply_rotary_pos_emb alone, with the reference's bf16 rounding points (x*cos -> bf16, rotate_half(x)*sin -> bf16, sum -> bf16); its
    backward is the transposed rotation.  pm_column_sum_bf16 = the bias gradient of the q/k/v projections, also on a strided column block."""
    torch.manual_seed(11)
    n, hq, hkv, d = 333, 6, 2, 128
    qkv = torch.randn(n, (hq + 2 * hkv) * d).to(torch.bfloat16)
    pos = torch.randint(0, 5000, (n,))
    inv_freq = ops.rope_inv_freq(d, 1e6, DEV)
    cos, sin = ops.rope_table(pos.to(DEV), inv_freq)
    q, k, rstd = ops.qknorm_rope_fwd(qkv.to(DEV), None, None, cos, sin, hq, hkv, d, 1e-6)
    assert rstd is None
    c = torch.cat([cos, cos], -1).cpu()[:, None, :]
    s = torch.cat([sin, sin], -1).cpu()[:, None, :]

    def rope(x):  # bf16 in, bf16 arithmetic like the reference (rotary_emb.py:83-118 on bf16 tensors)
        rot = torch.cat([-x[..., d // 2:], x[..., : d // 2]], -1)
        return x * c + rot * s

    assert torch.equal(q.cpu(), rope(qkv[:, : hq * d].view(n, hq, d)))
    assert torch.equal(k.cpu(), rope(qkv[:, hq * d: (hq + hkv) * d].view(n, hkv, d)))
    gq = torch.randn(n, hq, d).to(torch.bfloat16)
    gk = torch.randn(n, hkv, d).to(torch.bfloat16)
    gqkv = torch.full((n, (hq + 2 * hkv) * d), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.qknorm_rope_bwd(gq.to(DEV), gk.to(DEV), None, None, None, cos, sin, None, gqkv, None, None, hq, hkv, d)
    x = qkv[:, : (hq + hkv) * d].float().requires_grad_(True)
    cf, sf = c.float(), s.float()
    y = x.view(n, hq + hkv, d)
    out = y * cf + torch.cat([-y[..., d // 2:], y[..., : d // 2]], -1) * sf
    (gx,) = torch.autograd.grad(out, x, torch.cat([gq, gk], 1).float())
    _close(gqkv[:, : (hq + hkv) * d], gx, 1e-2, 1e-2, "rope-only backward")
    assert torch.all(gqkv[:, (hq + hkv) * d:] == 7.0), "the v columns belong to attention backward"
    with pytest.raises(Exception):
        ops.qknorm_rope_fwd(qkv.to(DEV), torch.ones(d, dtype=torch.bfloat16, device=DEV), None, cos, sin, hq, hkv, d, 1e-6)
    # column sums: whole matrix, strided block, ragged row counts, accumulate semantics
    for rows in (1, 3, 255, 256, 257, 1000):
        xm = torch.randn(rows, 1280).to(torch.bfloat16).to(DEV)
        out = torch.ones(1280, device=DEV)
        ops.column_sum(xm, out)
        _close(out, 1 + xm.float().sum(0), 1e-5, 1e-4, f"column_sum rows={rows}")
        blk = xm[:, 256:768]
        out2 = torch.zeros(512, device=DEV)
        ops.column_sum(blk, out2)
        out3 = torch.zeros(512, device=DEV)
        ops.column_sum(blk, out3)
        _close(out2, blk.float().sum(0), 1e-5, 1e-4, f"column_sum strided rows={rows}")
        assert torch.equal(out2, out3), "column_sum is not deterministic"


def test_embedding_bwd_window(ops):
    """Windows of the flattened [vocab, hidden] gradient (a ZeRO shard of the embedding bucket): pieced together they are the whole-table
    scatter-add bit for bit; windows that start / end inside a row, padding ids (-1) and ids >= vocab."""
    torch.manual_seed(3)
    v, h, n = 90, 256, 700
    ids = torch.randint(0, v, (n,))
    ids[::17] = -1  # padding of the gathered id lists
    ids[5] = v + 3  # outside the table: skipped
    go = torch.randn(n, h).to(torch.bfloat16)
    keep = (ids >= 0) & (ids < v)
    whole = torch.zeros(v, h, device=DEV)
    ops.embedding_bwd(go[keep].to(DEV), ids[keep].to(DEV), whole)
    total = v * h
    for parts in (1, 3, 8, 45):  # 3 and 45 cut rows: 23040 / 3 = 7680 = 30 rows, / 45 = 512 = 2 rows; use uneven 8-aligned cuts below as well
        per = total // parts
        assert per % 8 == 0
        got = torch.zeros(total, device=DEV)
        for r in range(parts):
            ops.embedding_bwd_window(go.to(DEV), ids.to(DEV), got[r * per:(r + 1) * per], r * per, v)
        assert torch.equal(got.view(v, h), whole), parts
    cuts = [0, 8, 8 + 256 * 3 + 40, 8 + 256 * 3 + 48, total - 264, total]  # windows starting and ending mid-row, one 8-element window
    got = torch.zeros(total, device=DEV)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ops.embedding_bwd_window(go.to(DEV), ids.to(DEV), got[lo:hi], lo, v)
    assert torch.equal(got.view(v, h), whole)
    before = got.clone()
    ops.embedding_bwd_window(go.to(DEV), ids.to(DEV), got[:0], 0, v)  # empty window: nothing to do
    ops.embedding_bwd_window(go[:0].to(DEV), ids[:0].to(DEV), got, 0, v)  # no tokens
    assert torch.equal(got, before)
    with pytest.raises(Exception):
        ops.embedding_bwd_window(go.to(DEV), ids.to(DEV), got[4:12], 4, v)


# ================================================================================================ attention
def _attn_case(n, seq_lens, hq, hkv, seed):
    torch.manual_seed(seed)
    d = 128
    q = torch.randn(n, hq, d).to(torch.bfloat16)
    k = torch.randn(n, hkv, d).to(torch.bfloat16)
    v = torch.randn(n, hkv, d).to(torch.bfloat16)
    return q, k, v


@pytest.mark.parametrize("seq_lens,hq,hkv", [([200], 4, 2), ([64, 1, 100, 37, 98], 8, 2), ([1024], 4, 1), ([300, 212], 8, 8), ([5], 2, 1),
                                             # tile edges of the LDS-DMA kernels: one token, exactly one 64 / 128-row tile, one row past it, documents
                                             # ending on and straddling tile boundaries, many one- and two-token documents inside one tile
                                             ([1], 2, 2), ([64], 2, 1), ([128], 4, 2), ([129], 4, 2), ([63, 65, 128, 1], 4, 1), ([1, 2] * 45, 2, 1),
                                             ([255, 257], 8, 2),
                                             # query-head groups of 8 (Qwen3-30B-A3B: 32 q / 4 kv heads) and 16 on ragged documents
                                             ([300, 212], 8, 1), ([513, 3, 140], 16, 2), ([700], 32, 2)])
def test_attention_forward_backward(ops, seq_lens, hq, hkv):
    n, d = sum(seq_lens), 128
    q, k, v = _attn_case(n, seq_lens, hq, hkv, n)
    scale = d**-0.5
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    want_ss = np.repeat(np.cumsum([0] + seq_lens[:-1]), seq_lens)
    assert np.array_equal(ss.cpu().numpy(), want_ss)  # integer work: exact
    # v handed over as the strided v-slice of a fused qkv buffer, like the engine does
    qkv = torch.zeros(n, (hq + 2 * hkv) * d, dtype=torch.bfloat16, device=DEV)
    qkv[:, (hq + hkv) * d:] = v.reshape(n, -1).to(DEV)
    v_view = qkv[:, (hq + hkv) * d:].view(n, hkv, d)
    out, lse = ops.attn_varlen_fwd(q.to(DEV), k.to(DEV), v_view, ss, scale)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    want = O.varlen_causal_attention(qf, kf, vf, seq_lens, scale)
    _close(out, want, 1e-2, 1e-2, "attention forward")  # bf16 P and bf16 output
    # the softmax statistic handed to backward: lse[h, i] = logsumexp_j(scale * q_i . k_j) over the visible keys (f32 math on bf16 q, k)
    bounds = np.cumsum([0] + seq_lens)
    want_lse = torch.empty(hq, n)
    for h in range(hq):
        for s, e in zip(bounds[:-1], bounds[1:]):
            sc = (q[s:e, h].float() @ k[s:e, h // (hq // hkv)].float().T) * scale
            want_lse[h, s:e] = torch.logsumexp(sc.masked_fill(torch.ones(e - s, e - s, dtype=torch.bool).triu(1), float("-inf")), -1)
    _close(lse, want_lse, 1e-4, 1e-4, "attention lse")
    go = torch.randn(n, hq, d).to(torch.bfloat16)
    (want * go.float()).sum().backward()
    gq, gk, gv = ops.attn_varlen_bwd(go.to(DEV), q.to(DEV), k.to(DEV), v_view, out, lse, ss, scale)
    _close(gq, qf.grad, 2e-2, 3e-2, "dq")
    _close(gk, kf.grad, 2e-2, 3e-2, "dk")
    _close(gv, vf.grad, 2e-2, 3e-2, "dv")
    gq2, gk2, gv2 = ops.attn_varlen_bwd(go.to(DEV), q.to(DEV), k.to(DEV), v_view, out, lse, ss, scale)
    assert torch.equal(gq, gq2) and torch.equal(gk, gk2) and torch.equal(gv, gv2), "backward must be bitwise reproducible"


@pytest.mark.parametrize("seq_lens,hq,hkv,cp", [([300, 212], 8, 2, 4), ([1024], 4, 1, 2), ([100, 700, 90, 134], 4, 4, 8), ([384], 2, 2, 3), ([130, 126], 2, 1, 2),
                                                # BASELINE.json configs[4]: one 32k-token multi-turn rollout, Qwen3-8B heads, context parallel over 8 ranks
                                                ([32768], 32, 8, 8), ([20000, 12768], 32, 8, 8)])
def test_attention_row_windows(ops, seq_lens, hq, hkv, cp):
    """pm_attn_varlen_{fwd,bwd}_rows — context parallelism with gathered K / V (the reference's ring_fa3_varlen_func, ring_attn.py:99-280): every rank
    runs its contiguous q window against the K / V prefix of the pass.  With windows on multiples of 128 the tiles are the full call's tiles:
    out / lse / dq are the full call's rows BIT FOR BIT; dk / dv of the windows are partial sums whose total is the full call's dk / dv
    (each partial rounded to bf16 once, like the reference's bf16 reduce-scatter of dk / dv)."""
    n, d = sum(seq_lens), 128
    q, k, v = (t.to(DEV) for t in _attn_case(n, seq_lens, hq, hkv, 5 * n))
    scale = d**-0.5
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    go = torch.randn(n, hq, d).to(torch.bfloat16).to(DEV)
    out, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
    gq, gk, gv = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
    per = -(-n // cp)
    per = (per + 127) // 128 * 128  # window starts on multiples of 128; the last window takes what is left (possibly nothing)
    gk_sum, gv_sum = torch.zeros(n, hkv, d, device=DEV), torch.zeros(n, hkv, d, device=DEV)
    for r in range(cp):
        r0, r1 = min(r * per, n), min((r + 1) * per, n)
        if r1 == r0:
            continue
        o_r, l_r = ops.attn_varlen_fwd_rows(q[r0:r1].contiguous(), k[:r1], v[:r1], ss[:r1], scale, r0)
        assert torch.equal(o_r, out[r0:r1]) and torch.equal(l_r, lse[:, r0:r1]), (r, "forward rows differ from the full call")
        gq_r, gk_r, gv_r = ops.attn_varlen_bwd_rows(go[r0:r1].contiguous(), q[r0:r1].contiguous(), k[:r1], v[:r1], o_r, l_r, ss[:r1], scale, r0)
        assert torch.equal(gq_r, gq[r0:r1]), (r, "dq rows differ from the full call")
        assert gk_r.shape == (r1, hkv, d) and bool(torch.isfinite(gk_r.float()).all()) and bool(torch.isfinite(gv_r.float()).all())
        gk_sum[:r1] += gk_r.float()
        gv_sum[:r1] += gv_r.float()
        if r == 0:  # the first window sees only its own keys: its dk / dv ARE a full call on the prefix
            o0, l0 = ops.attn_varlen_fwd(q[:r1].contiguous(), k[:r1].contiguous(), v[:r1].contiguous(), ss[:r1], scale)
            g0 = ops.attn_varlen_bwd(go[:r1].contiguous(), q[:r1].contiguous(), k[:r1].contiguous(), v[:r1].contiguous(), o0, l0, ss[:r1], scale)
            assert torch.equal(gk_r, g0[1]) and torch.equal(gv_r, g0[2])
    _close(gk_sum, gk.float(), 2e-2, 3e-2, "sum of the windows' dk")
    _close(gv_sum, gv.float(), 2e-2, 3e-2, "sum of the windows' dv")
    with pytest.raises(Exception):
        ops.attn_varlen_fwd_rows(q[64:128].contiguous(), k[:128], v[:128], ss[:128], scale, 64)  # window start not on a multiple of 128


@pytest.mark.parametrize("seq_lens,hq,hkv", [([200], 4, 2), ([64, 1, 100, 37, 98], 8, 2), ([1024], 4, 1), ([5], 2, 1), ([128], 4, 2), ([129], 4, 2),
                                             ([63, 65, 128, 1], 4, 1), ([1, 2] * 45, 2, 1), ([255, 257], 8, 2), ([2048] * 3 + [777], 8, 2)])
@pytest.mark.skipif(not ATTN_AB, reason="the A/B builds of the dK/dV kernel exist only in a `make VARIANTS=1` library")
def test_attention_dkdv_variants_are_bit_identical(ops, seq_lens, hq, hkv):
    """The dK/dV kernel's build variants (pm_attn_debug_set_variant(1, v)): the default is attn_bwd_dkdv_p2_kernel (32 kv rows per wave, one wave per SIMD,
    hand-placed two-stage tile loop); the round-2..5 kernel (7; + 16: 8-byte epilogue stores), its fragment-batch builds (1, 2) and its compiler-scheduled
    32-kv-rows-per-wave builds (4, 5) walk the same q tiles in the same order and feed the same MFMA chains: dk / dv equal the default build bit for bit,
    full calls and row windows."""
    from prime_rl_amd import _lib

    lib = _lib.load()
    n, d = sum(seq_lens), 128
    q, k, v = (t.to(DEV) for t in _attn_case(n, seq_lens, hq, hkv, 11 * n))
    scale = d**-0.5
    ss = ops.seq_start_from_cu(torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV), n)
    go = torch.randn(n, hq, d).to(torch.bfloat16).to(DEV)
    out, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
    r0 = (n // 2) // 128 * 128

    def run():
        full = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
        if r0 == 0:
            return full
        return full + ops.attn_varlen_bwd_rows(go[r0:].contiguous(), q[r0:].contiguous(), k, v, out[r0:].contiguous(), lse[:, r0:].contiguous(), ss, scale, r0)

    try:
        base = run()
        for var in (7, 7 + 16, 1, 2, 4, 5, 4 + 16, 16):
            rc = lib.pm_attn_debug_set_variant(1, var)
            if rc == -2:
                pytest.skip("the A/B builds of the dK/dV kernel exist only in a `make VARIANTS=1` library")
            assert rc == 0
            got = run()
            assert all(torch.equal(a, b) for a, b in zip(base, got)), f"dK/dV variant {var} differs from the default build"
    finally:
        lib.pm_attn_debug_set_variant(1, 0)


@pytest.mark.parametrize("seq_lens,hq,hkv", [([200], 4, 2), ([64, 1, 100, 37, 98], 8, 2), ([1024], 4, 1), ([5], 2, 1), ([128], 4, 2), ([129], 4, 2),
                                             ([63, 65, 128, 1], 4, 1), ([1, 2] * 45, 2, 1), ([255, 257], 8, 2), ([2048] * 3 + [777], 8, 2)])
@pytest.mark.skipif(not ATTN_AB, reason="the A/B builds of the forward kernel exist only in a `make VARIANTS=1` library")
def test_attention_fwd_variants_are_bit_identical(ops, seq_lens, hq, hkv):
    """Forward build variants (pm_attn_debug_set_variant(0, v)): the default is the three-stage pipelined kernel with the hand-placed tile loop
    (attention_fwd_p3.hip.h); the round-1..5 kernel (15, 7), its 8-wave x 16-row build (47), the pipelined kernel with compiler-scheduled stages (128) and
    the placement A/B builds of the hand-placed loop (130-133: no fast path / merged reduction / in-stream DMA) do the same per-row arithmetic over the
    same kv tiles: out and lse equal the default build bit for bit, full calls and row windows."""
    from prime_rl_amd import _lib

    lib = _lib.load()
    n, d = sum(seq_lens), 128
    q, k, v = (t.to(DEV) for t in _attn_case(n, seq_lens, hq, hkv, 13 * n))
    scale = d**-0.5
    ss = ops.seq_start_from_cu(torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV), n)
    r0 = (n // 2) // 128 * 128

    def run():
        full = ops.attn_varlen_fwd(q, k, v, ss, scale)
        return full if r0 == 0 else full + ops.attn_varlen_fwd_rows(q[r0:].contiguous(), k, v, ss, scale, r0)

    try:
        base = run()
        for var in (15, 7, 47, 128, 130, 131, 132, 133):
            rc = lib.pm_attn_debug_set_variant(0, var)
            if rc == -2:
                pytest.skip("the A/B builds of the forward kernel exist only in a `make VARIANTS=1` library")
            assert rc == 0
            got = run()
            assert all(torch.equal(a, b) for a, b in zip(base, got)), f"forward variant {var} differs from the default build"
    finally:
        lib.pm_attn_debug_set_variant(0, -1)


@pytest.mark.parametrize("seq_lens,hq,hkv,window", [([300, 212], 4, 2, 64), ([1024], 4, 1, 100), ([64, 1, 100, 37, 98], 8, 2, 7), ([700], 2, 2, 1), ([200], 2, 1, 4096)])
def test_attention_sliding_window(ops, seq_lens, hq, hkv, window):
    """The optional sliding window of the reference's attention (attn.py:97-99) with the SAME kernels: `seq_start_with_window` raises each row's first
    visible key to row - window + 1; forward, lse-free check of the output and all three gradients against the windowed oracle; a window longer than
    every document is the plain causal call bit for bit."""
    n, d = sum(seq_lens), 128
    q, k, v = _attn_case(n, seq_lens, hq, hkv, 3 * n + window)
    scale = d**-0.5
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    lo = ops.seq_start_with_window(ss, window)
    out, lse = ops.attn_varlen_fwd(q.to(DEV), k.to(DEV), v.to(DEV), lo, scale)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    want = O.varlen_causal_attention(qf, kf, vf, seq_lens, scale, window=window)
    _close(out, want, 1e-2, 1e-2, "windowed attention forward")
    go = torch.randn(n, hq, d).to(torch.bfloat16)
    (want * go.float()).sum().backward()
    gq, gk, gv = ops.attn_varlen_bwd(go.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), out, lse, lo, scale)
    _close(gq, qf.grad, 2e-2, 3e-2, "windowed dq")
    _close(gk, kf.grad, 2e-2, 3e-2, "windowed dk")
    _close(gv, vf.grad, 2e-2, 3e-2, "windowed dv")
    if window >= max(seq_lens):
        plain, plain_lse = ops.attn_varlen_fwd(q.to(DEV), k.to(DEV), v.to(DEV), ss, scale)
        assert torch.equal(lo, ss) and torch.equal(out, plain) and torch.equal(lse, plain_lse)
    if window == 1:  # every query sees only itself: the output is its own value row
        g = hq // hkv
        assert torch.equal(out.cpu(), v.repeat_interleave(g, dim=1))


def test_attention_full_size_properties(ops):
    """BASELINE shape (8 x 2048-token rollouts packed in one 16 384-token row, 32 q / 8 kv heads, d = 128), where the CPU oracle would
    take minutes: size-independent properties instead.
      * document isolation + causality, bit-exact: rewriting q, k, v, dO of every token from the middle of document 2 onwards leaves
        out / lse / dq of all earlier tokens and dk / dv of documents 0 and 1 unchanged to the bit — no tile ever reads across a
        document boundary or ahead of a query;
      * one (document, head) pair of the big run equals an f32 torch computation of that pair on the same GPU;
      * backward is linear in dO (2 x dO -> 2 x gradients, exact in bf16 up to the final rounding) and bitwise reproducible."""
    torch.manual_seed(11)
    docs, hq, hkv, d = [2048] * 8, 32, 8, 128
    n, scale = sum(docs), 128**-0.5
    q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    go = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(docs)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    out, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
    gq, gk, gv = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
    assert all(bool(torch.isfinite(t.float()).all()) for t in (out, lse, gq, gk, gv))
    # --- isolation / causality
    cut = 2 * 2048 + 1024
    q2, k2, v2, go2 = (t.clone() for t in (q, k, v, go))
    for t in (q2, k2, v2, go2):
        t[cut:] = torch.randn_like(t[cut:].float()).to(torch.bfloat16)
    out2, lse2 = ops.attn_varlen_fwd(q2, k2, v2, ss, scale)
    assert torch.equal(out[:cut], out2[:cut]) and torch.equal(lse[:, :cut], lse2[:, :cut])
    gq2, gk2, gv2 = ops.attn_varlen_bwd(go2, q2, k2, v2, out2, lse2, ss, scale)
    assert torch.equal(gq[:cut], gq2[:cut])  # dq of a query depends only on keys at or before it
    assert torch.equal(gk[:4096], gk2[:4096]) and torch.equal(gv[:4096], gv2[:4096])  # documents 0 and 1 are closed
    # --- one (document, head) against f32 torch on the device
    doc, head = 5, 13
    s, e = doc * 2048, (doc + 1) * 2048
    qf, kf, vf = q[s:e, head].float().requires_grad_(True), k[s:e, head // 4].float().requires_grad_(True), v[s:e, head // 4].float().requires_grad_(True)
    sc = (qf @ kf.T) * scale
    sc = sc.masked_fill(torch.ones(2048, 2048, dtype=torch.bool, device=DEV).triu(1), float("-inf"))
    want = torch.softmax(sc, -1) @ vf
    _close(out[s:e, head], want, 1e-2, 1e-2, "full-size forward, doc 5 head 13")
    assert float((lse[head, s:e] - torch.logsumexp(sc.detach(), -1)).abs().max()) < 2e-3
    (want * go[s:e, head].float()).sum().backward()
    _close(gq[s:e, head], qf.grad, 2e-2, 3e-2, "full-size dq")
    # dk / dv of a kv head sum over its 4 query heads: compare the linear-in-dO property instead of one head's share
    gq3, gk3, gv3 = ops.attn_varlen_bwd((go.float() * 2).to(torch.bfloat16), q, k, v, out, lse, ss, scale)
    for a, b, name in ((gq3, gq, "dq"), (gk3, gk, "dk"), (gv3, gv, "dv")):
        err = (a.float() - 2 * b.float()).abs().max().item()
        assert err <= 2e-2 * max(b.float().abs().max().item(), 1.0), (name, err)
    gq4, gk4, gv4 = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
    assert torch.equal(gq, gq4) and torch.equal(gk, gk4) and torch.equal(gv, gv4)


def test_attention_long_document_and_ragged_tail(ops):
    """One 8 192-token document followed by ragged short ones (total not a multiple of any tile size): every q tile of the long
    document walks 128 KV tiles, the tail exercises partial tiles; two heads are checked against f32 torch on the device."""
    torch.manual_seed(12)
    docs, hq, hkv, d = [8192, 77, 1, 300, 5], 8, 2, 128
    n, scale = sum(docs), 128**-0.5
    q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    go = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(docs)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    out, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
    gq, gk, gv = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
    bounds = np.cumsum([0] + docs)
    for head in (0, 5):
        kvh = head // (hq // hkv)
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q[:, head], k[:, kvh], v[:, kvh]))
        want = torch.zeros(n, d, device=DEV)
        parts = []
        for s, e in zip(bounds[:-1], bounds[1:]):
            sc = (qf[s:e] @ kf[s:e].T) * scale
            sc = sc.masked_fill(torch.ones(e - s, e - s, dtype=torch.bool, device=DEV).triu(1), float("-inf"))
            parts.append(torch.softmax(sc, -1) @ vf[s:e])
        want = torch.cat(parts)
        _close(out[:, head], want, 1e-2, 1e-2, f"long-doc forward head {head}")
        (want * go[:, head].float()).sum().backward()
        _close(gq[:, head], qf.grad, 2e-2, 3e-2, f"long-doc dq head {head}")
    # dk / dv of kv head 0 = sum over its 4 query heads
    kf, vf = k[:, 0].float().requires_grad_(True), v[:, 0].float().requires_grad_(True)
    total = 0.0
    for head in range(4):
        for s, e in zip(bounds[:-1], bounds[1:]):
            sc = (q[s:e, head].float() @ kf[s:e].T) * scale
            sc = sc.masked_fill(torch.ones(e - s, e - s, dtype=torch.bool, device=DEV).triu(1), float("-inf"))
            total = total + ((torch.softmax(sc, -1) @ vf[s:e]) * go[s:e, head].float()).sum()
    total.backward()
    _close(gk[:, 0], kf.grad, 2e-2, 4e-2, "long-doc dk")
    _close(gv[:, 0], vf.grad, 2e-2, 4e-2, "long-doc dv")


def test_attention_65k_token_document(ops):
    """Twice BASELINE.json configs[4]'s rollout length in ONE document (65 536 tokens + a ragged tail document): every byte offset the kernels form for the
    last q / kv tiles lies beyond 2^24 elements per head row and the tile loops run 1 024 kv tiles.  Checked against an f32 computation on the device that
    walks the score matrix in 4 096-row blocks (the full matrix would be 17 GB per head): out, lse, dq, dk, dv of both q heads."""
    torch.manual_seed(13)
    docs, hq, hkv, d = [65536, 333], 2, 1, 128
    n, scale = sum(docs), 128**-0.5
    q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    go = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(docs)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    out, lse = ops.attn_varlen_fwd(q, k, v, ss, scale)
    gq, gk, gv = ops.attn_varlen_bwd(go, q, k, v, out, lse, ss, scale)
    want_o, want_lse, want_dq = torch.empty(n, hq, d, device=DEV), torch.empty(hq, n, device=DEV), torch.empty(n, hq, d, device=DEV)
    want_dk, want_dv = torch.zeros(n, d, device=DEV), torch.zeros(n, d, device=DEV)
    kf, vf = k[:, 0].float(), v[:, 0].float()
    bounds = np.cumsum([0] + docs)
    for s, e in zip(bounds[:-1], bounds[1:]):
        for b0 in range(s, e, 4096):
            b1 = min(b0 + 4096, e)
            rows = torch.arange(b0, b1, device=DEV)[:, None]
            hidden = torch.arange(s, b1, device=DEV)[None, :] > rows
            for h in range(hq):
                sc = ((q[b0:b1, h].float() @ kf[s:b1].T) * scale).masked_fill(hidden, float("-inf"))
                want_lse[h, b0:b1] = torch.logsumexp(sc, -1)
                p = torch.softmax(sc, -1)
                o = p @ vf[s:b1]
                want_o[b0:b1, h] = o
                dof = go[b0:b1, h].float()
                dp = dof @ vf[s:b1].T
                ds = p * (dp - (dof * o).sum(-1, keepdim=True)) * scale
                want_dq[b0:b1, h] = ds @ kf[s:b1]
                want_dk[s:b1] += ds.T @ q[b0:b1, h].float()
                want_dv[s:b1] += p.T @ dof
    _close(out, want_o, 1e-2, 1e-2, "65k forward")
    _close(lse, want_lse, 1e-4, 2e-4, "65k lse")
    _close(gq, want_dq, 2e-2, 3e-2, "65k dq")
    _close(gk[:, 0], want_dk, 2e-2, 4e-2, "65k dk")
    _close(gv[:, 0], want_dv, 2e-2, 4e-2, "65k dv")


# ================================================================================================ optimizer
def test_adamw_sumsq_clip(ops):
    torch.manual_seed(0)
    n = 100_003
    p0 = torch.randn(n)
    g = torch.randn(n) * 3
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    pad = (-n) % 4
    master = torch.cat([p0, torch.zeros(pad)]).to(DEV)
    m = torch.zeros_like(master)
    v = torch.zeros_like(master)
    grad = torch.cat([g, torch.zeros(pad)]).to(DEV)
    pb = torch.zeros(n + pad, dtype=torch.bfloat16, device=DEV)
    ssq = torch.zeros(1, device=DEV)
    ops.sumsq_accumulate(grad, ssq)
    _close(ssq[0], (g.double() ** 2).sum().float(), 1e-5, 0, "sum of squares")
    coef, norm = ops.clip_coef(ssq, 1.0, 1.0)
    want_norm = g.double().norm().item()
    assert abs(norm.item() - want_norm) / want_norm < 1e-5 and abs(coef.item() - 1.0 / (want_norm + 1e-6)) < 1e-7
    for step in range(1, 4):
        p.grad = g.clone() * coef.item()
        opt.step()
        grad_step = torch.cat([g, torch.zeros(pad)]).to(DEV)
        ops.adamw_step(master, m, v, grad_step, pb, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, grad_scale=1.0, clip=coef, zero_grad=True)
        assert not grad_step.any(), "zero_grad"
    _close(master[:n], p.detach(), 1e-6, 1e-7, "AdamW master weights")  # reference bound tests/unit/train/test_cpu_adam.py:34-134
    assert torch.equal(pb, master.to(torch.bfloat16)), "bf16 compute copy must be the RNE rounding of the updated master weights"
    mg = torch.zeros(4096 + 8, device=DEV)
    gb = torch.randn(4096 + 8).to(torch.bfloat16).to(DEV)
    ops.grad_accumulate(mg, gb)
    ops.grad_accumulate(mg, gb)
    assert torch.equal(mg, 2 * gb.float())


@pytest.mark.parametrize("momentum,nesterov,wd", [(0.9, True, 0.01), (0.9, False, 0.0), (0.0, False, 0.01), (0.5, True, 0.0)])
def test_sgd_step_matches_torch_sgd(ops, momentum, nesterov, wd):
    """pm_sgd_step against torch.optim.SGD (what the reference builds for optim.type = "sgd", trainer/optim/__init__.py:74-81): four steps with a changing
    gradient, clip coefficient applied to the gradient first, bf16 shadow = RNE of the master weights, gradient cleared."""
    torch.manual_seed(1)
    n = 70_001
    pad = (-n) % 4
    p0 = torch.randn(n)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=3e-3, momentum=momentum, nesterov=nesterov, weight_decay=wd)
    master = torch.cat([p0, torch.zeros(pad)]).to(DEV)
    buf = torch.zeros_like(master) if momentum else None
    pb = torch.zeros(n + pad, dtype=torch.bfloat16, device=DEV)
    coef = torch.tensor([0.37], device=DEV)
    for step in range(1, 5):
        g = torch.randn(n) * step
        p.grad = g * 0.37
        opt.step()
        grad = torch.cat([g, torch.zeros(pad)]).to(DEV)
        ops.sgd_step(master, buf, grad, pb, 3e-3, momentum, wd, nesterov, step, grad_scale=1.0, clip=coef, zero_grad=True)
        assert not grad.any()
    _close(master[:n], p.detach(), 2e-6, 1e-6, "SGD master weights")
    assert torch.equal(pb, master.to(torch.bfloat16))
    if momentum:
        _close(buf[:n], opt.state[p]["momentum_buffer"], 2e-6, 1e-6, "momentum buffer")
    with pytest.raises(Exception):
        ops.sgd_step(master, None, grad, pb, 1e-3, 0.9, 0.0, True, 1)  # momentum without a buffer
    with pytest.raises(Exception):
        ops.sgd_step(master, None, grad, pb, 1e-3, 0.0, 0.0, True, 1)  # Nesterov needs a momentum (torch raises as well)


def test_sign_sgd_step_matches_reference_formula(ops):
    """pm_sign_sgd_step against SignSGD.step (trainer/sign_sgd.py:34-51): p.add_(p, alpha=-lr*wd); p.add_(sign(grad), alpha=-lr); zeros stay put."""
    torch.manual_seed(2)
    n = 50_002
    pad = (-n) % 4
    p = torch.randn(n)
    master = torch.cat([p, torch.zeros(pad)]).to(DEV)
    pb = torch.zeros(n + pad, dtype=torch.bfloat16, device=DEV)
    for step in range(3):
        g = torch.randn(n)
        g[::7] = 0.0
        p.add_(p, alpha=-2e-3 * 0.05)
        p.add_(torch.sign(g), alpha=-2e-3)
        grad = torch.cat([g, torch.zeros(pad)]).to(DEV)
        ops.sign_sgd_step(master, grad, pb, 2e-3, 0.05, grad_scale=1.0, clip=torch.tensor([0.5], device=DEV), zero_grad=True)  # the clip cannot change a sign
        assert not grad.any()
    _close(master[:n], p, 1e-6, 1e-7, "SignSGD master weights")
    assert torch.equal(pb, master.to(torch.bfloat16))


# ================================================================================================ edge cases
def test_edge_cases(ops):
    """Empty and degenerate inputs (the reference's tests cover empty components, singleton groups, length-1 documents)."""
    from prime_rl_amd import _lib

    # empty micro-batch
    lp, ent, logz = ops.logprob_entropy_fwd(torch.empty(0, 128, device=DEV), torch.empty(0, dtype=torch.int64, device=DEV), torch.empty(0, device=DEV))
    assert lp.numel() == ent.numel() == logz.numel() == 0
    assert ops.grpo_advantage(torch.empty(0, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)).numel() == 0
    # vocab of one: log-prob 0, entropy 0; every alignment of a tiny odd-width row
    for v in (1, 2, 3, 5, 7, 9, 17, 33):
        logits = torch.randn(5, v, device=DEV)
        labels = torch.randint(0, v, (5,), device=DEV)
        lp, ent, _ = ops.logprob_entropy_fwd(logits, labels, torch.ones(5, device=DEV))
        ref = torch.log_softmax(logits, -1)
        _close(lp, ref[torch.arange(5), labels], 0, 1e-6, f"tiny vocab {v}")
        _close(ent, -(ref.exp() * ref).sum(-1), 0, 2e-6, f"tiny vocab entropy {v}")
    # rows that are views with a stride (a [N, V] slice of a wider buffer)
    wide = torch.randn(9, 4099, device=DEV)
    view = wide[:, 3:3 + 4001]
    labels = torch.randint(0, 4001, (9,), device=DEV)
    lp, _, _ = ops.logprob_entropy_fwd(view, labels, torch.ones(9, device=DEV))
    _close(lp, torch.log_softmax(view, -1)[torch.arange(9), labels], 0, 1e-5, "strided unaligned rows")
    # extreme logits: no overflow / NaN
    big = torch.tensor([[1e4, -1e4, 0.0, 9999.0]], device=DEV)
    lp, ent, logz = ops.logprob_entropy_fwd(big, torch.tensor([0], device=DEV), torch.ones(1, device=DEV))
    # logprob = target - logz like the reference's fused head (lm_head.py:147): at |logit| = 1e4 one f32 ulp of logz is 1e-3
    assert torch.isfinite(lp).all() and torch.isfinite(ent).all() and abs(lp.item() + math.log1p(math.exp(-1.0))) < 1e-3
    # loss: a zero-length sample between two real ones, and a fully masked micro-batch
    lpv = torch.randn(10, device=DEV)
    z = torch.zeros(10, device=DEV)
    cu = _t(np.array([0, 4, 4, 10], dtype=np.int32))
    loss, grad, stats, _ = ops.grpo_loss_fwd_bwd(lpv, lpv.clone(), z + 1.0, torch.ones(10, dtype=torch.bool, device=DEV), cu, _loss_params(O.LossConfig(), (10, 1, 1)))
    assert stats[1].abs().sum() == 0 and stats[:, _lib.STAT["RL_N_MASK"]].tolist() == [4.0, 0.0, 6.0]
    _close(loss[0], -1.0, 1e-6, 1e-6, "ratio 1, adv 1 -> loss -1")
    loss, grad, _, _ = ops.grpo_loss_fwd_bwd(lpv, z, z, torch.zeros(10, dtype=torch.bool, device=DEV), cu, _loss_params(O.LossConfig(), (1, 1, 1)))
    assert loss.item() == 0.0 and not grad.any()
    # attention on a single token and on documents of length 1
    for lens in ([1], [1, 1, 1, 2]):
        n = sum(lens)
        q, k, v = _attn_case(n, lens, 4, 2, 1)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
        ss = ops.seq_start_from_cu(cu, n)
        out, lse = ops.attn_varlen_fwd(q.to(DEV), k.to(DEV), v.to(DEV), ss, 128**-0.5)
        want = O.varlen_causal_attention(q, k, v, lens, 128**-0.5)
        _close(out, want, 1e-2, 1e-2, f"attention lens={lens}")
