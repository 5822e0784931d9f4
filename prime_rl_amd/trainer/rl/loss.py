"""GRPO loss seam — mirrors `prime_rl.trainer.rl.loss` (src/prime_rl/trainer/rl/loss.py) name for name.

Same public surface (LossInputs, LossOutputs, LossFn, selective_log_softmax, compute_entropy, shift_tensor_left/right,
default_loss_fn, ipo_loss_fn, ref_kl_loss_fn, ce_loss_fn, setup_rl_loss_fn, compute_loss) with the arithmetic done by
the HIP kernels behind the C ABI: `pm_logprob_entropy_fwd/bwd` for the log-prob gather, `pm_grpo_loss_fwd_bwd` for the
loss.  `compute_loss` runs ONE launch for the whole micro-batch when the rl loss is a built-in type; a `CustomLossConfig`
function keeps the reference's per-sample plugin loop (loss.py:345-392).  All functions are differentiable w.r.t.
`trainer_logprobs` / `logits` through torch.autograd.Function wrappers, so they drop into the reference's train loop.
"""

from __future__ import annotations

import importlib
import math
from dataclasses import dataclass, field
from typing import Any, Callable

import torch
from torch import Tensor

from ... import _lib, ops
from ..._lib import LossParams


@dataclass
class DefaultLossConfig:
    """configs/trainer.py:494-507."""

    type: str = "default"
    dppo_mask_low: float = 0.2
    dppo_mask_high: float = 0.2
    adv_tau: float = 1.0
    kl_tau: float = 1e-3


@dataclass
class IPOLossConfig:
    """configs/trainer.py:509-520."""

    type: str = "ipo"
    ipo_threshold: float = 0.1
    adv_tau: float = 1.0
    kl_tau: float = 1e-3


@dataclass
class CustomLossConfig:
    """configs/trainer.py:522-532: `import_path` of a `LossFn`, `kwargs` forwarded to it."""

    type: str = "custom"
    import_path: str = ""
    kwargs: dict = field(default_factory=dict)


LossConfig = DefaultLossConfig | IPOLossConfig | CustomLossConfig


@dataclass
class LossInputs:
    """Inputs of a per-sample loss function (loss.py:13-30); `loss_mask` already selects the component's tokens."""

    trainer_logprobs: Tensor
    inference_logprobs: Tensor
    ref_logprobs: Tensor | None
    advantages: Tensor
    loss_mask: Tensor
    loss_weights: Tensor | None = field(default=None)


@dataclass
class LossOutputs:
    loss: Tensor
    metrics: dict[str, Tensor]


LossFn = Callable[..., LossOutputs]


# ---------------------------------------------------------------------------------------------------------------------
class _SelectiveLogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d: Tensor, index: Tensor):
        inv_t = torch.ones(logits2d.shape[0], dtype=torch.float32, device=logits2d.device)
        lp, _, logz = ops.logprob_entropy_fwd(logits2d, index, inv_t)
        ctx.save_for_backward(logits2d, index, inv_t, logz)
        return lp

    @staticmethod
    def backward(ctx, grad):
        logits2d, index, inv_t, logz = ctx.saved_tensors
        return ops.logprob_bwd(logits2d, index, inv_t, logz, grad.contiguous().float()), None


def selective_log_softmax(logits: Tensor, index: Tensor) -> Tensor:
    """log_softmax(logits).gather(index) without materialising the log-softmax (loss.py:48-54). logits [b, s, v], index [b, s]."""
    b, s, v = logits.shape
    return _SelectiveLogSoftmaxFn.apply(logits.reshape(b * s, v), index.reshape(b * s).contiguous()).view(b, s)


def compute_entropy(shifted_logits: Tensor) -> Tensor:
    """logsumexp - sum softmax*logits, no gradient (loss.py:57-63)."""
    b, s, v = shifted_logits.shape
    with torch.no_grad():
        flat = shifted_logits.detach().reshape(b * s, v)
        zeros = torch.zeros(b * s, dtype=torch.int64, device=flat.device)
        _, ent, _ = ops.logprob_entropy_fwd(flat, zeros, torch.ones(b * s, dtype=torch.float32, device=flat.device))
    return ent.view(b, s)


def shift_tensor_left(t: Tensor) -> Tensor:
    """labels[i] = input_ids[i+1], last = 0 (loss.py:66-73)."""
    out = torch.zeros_like(t)
    out[:, :-1] = t[:, 1:]
    return out


def shift_tensor_right(t: Tensor, pad_value: float | None = None) -> Tensor:
    """out[i] = t[i-1], out[0] = pad_value or 0 (loss.py:76-91).  Differentiable (slice + cat)."""
    pad = torch.full((t.shape[0], 1), 0.0 if pad_value is None else pad_value, device=t.device, dtype=t.dtype)
    return torch.cat([pad, t[:, :-1]], dim=1)


# ---------------------------------------------------------------------------------------------------------------------
def _params(cfg, scales=(1.0, 1.0, 1.0)) -> LossParams:
    ipo = getattr(cfg, "type", "default") == "ipo"
    return LossParams(
        loss_type=_lib.PM_LOSS_IPO if ipo else _lib.PM_LOSS_DEFAULT, dppo_mask_low=getattr(cfg, "dppo_mask_low", 0.2),
        dppo_mask_high=getattr(cfg, "dppo_mask_high", 0.2), adv_tau=cfg.adv_tau, kl_tau=cfg.kl_tau, ipo_threshold=getattr(cfg, "ipo_threshold", 0.1),
        rl_scale=float(scales[0]), ce_scale=float(scales[1]), ref_kl_scale=float(scales[2]), shift=0, pad_logprob=0.0,
    )


class _GrpoLossFn(torch.autograd.Function):
    """One kernel launch over flat token streams split by cu_samples; returns (scaled loss, stats, metrics)."""

    @staticmethod
    def forward(ctx, trainer_lp, inference_lp, ref_lp, adv, mask, rl_w, ce_w, rk_w, cu_samples, params: LossParams):
        loss, grad, stats, metrics = ops.grpo_loss_fwd_bwd(trainer_lp.contiguous().float(), inference_lp.contiguous().float(), adv.contiguous().float(),
                                                           mask.contiguous(), cu_samples, params, ref_logprobs=ref_lp, rl_weights=rl_w, ce_weights=ce_w,
                                                           ref_kl_weights=rk_w)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(stats, metrics)
        return loss[0], stats, metrics

    @staticmethod
    def backward(ctx, g_loss, _gs, _gm):
        (grad,) = ctx.saved_tensors
        return (g_loss * grad,) + (None,) * 9


def _cat(ts):
    return None if ts is None else torch.cat([t.reshape(-1) for t in ts]).contiguous().float()


def _one_sample(inputs: LossInputs, cfg, component: str) -> LossOutputs:
    """Built-in per-sample loss functions routed through the kernel (single segment)."""
    n = inputs.trainer_logprobs.numel()
    dev = inputs.trainer_logprobs.device
    cu = torch.tensor([0, n], dtype=torch.int32, device=dev)
    w = inputs.loss_weights
    zeros = None
    if component == "rl":
        # the kernel derives membership as mask & (w != 0); an explicit component mask with w == None needs no stream
        loss, stats, met = _GrpoLossFn.apply(inputs.trainer_logprobs, inputs.inference_logprobs, None, inputs.advantages, inputs.loss_mask, w, None, None,
                                             cu, _params(cfg))
        keys = _lib.METRIC_KEYS[:3] if getattr(cfg, "type", "default") == "ipo" else _lib.METRIC_KEYS[:7]
    elif component == "ce":
        # The kernel derives a component's membership from its weight stream (w != 0); the reference's ce / ref_kl loss fns take the
        # given loss_mask as membership and the weights only as multipliers (loss.py:246-259, 199-243), so an in-mask token with weight
        # 0.0 still counts in the metric denominators.  Loss and gradient come from the weighted launch; when weights are present the
        # METRICS come from a second launch whose weight stream is 1.0 on the mask (membership == loss_mask exactly).
        zeros = torch.zeros(n, dtype=torch.float32, device=dev)
        ones_on_mask = inputs.loss_mask.to(torch.float32)
        no_rl = torch.zeros(n, dtype=torch.bool, device=dev)
        cw = ones_on_mask if w is None else torch.where(inputs.loss_mask, w.float(), zeros)
        loss, stats, met = _GrpoLossFn.apply(inputs.trainer_logprobs, zeros, None, zeros, no_rl, None, cw, None, cu, _params(DefaultLossConfig()))
        if w is not None:
            _, _, met = _GrpoLossFn.apply(inputs.trainer_logprobs.detach(), zeros, None, zeros, no_rl, None, ones_on_mask, None, cu, _params(DefaultLossConfig()))
        keys = ["nll"]
    else:
        if inputs.ref_logprobs is None:
            raise ValueError("ref_kl loss type requires ref_logprobs — use the 'opd' or 'opsd' algorithm.")
        zeros = torch.zeros(n, dtype=torch.float32, device=dev)
        ones_on_mask = inputs.loss_mask.to(torch.float32)
        no_rl = torch.zeros(n, dtype=torch.bool, device=dev)
        ref = inputs.ref_logprobs.contiguous().float()
        rw = ones_on_mask if w is None else torch.where(inputs.loss_mask, w.float(), zeros)
        loss, stats, met = _GrpoLossFn.apply(inputs.trainer_logprobs, inputs.inference_logprobs, ref, zeros, no_rl, None, None, rw, cu, _params(DefaultLossConfig()))
        if w is not None:
            _, _, met = _GrpoLossFn.apply(inputs.trainer_logprobs.detach(), inputs.inference_logprobs, ref, zeros, no_rl, None, None, ones_on_mask, cu,
                                          _params(DefaultLossConfig()))
        keys = _lib.METRIC_KEYS[8:12]
    return LossOutputs(loss=loss, metrics={k: met[0, _lib.METRIC_KEYS.index(k)] for k in keys})


def default_loss_fn(inputs: LossInputs, loss_config: DefaultLossConfig) -> LossOutputs:
    """DPPO-binary trust-region mask + squared-log-ratio KL (loss.py:109-161)."""
    return _one_sample(inputs, loss_config, "rl")


def ipo_loss_fn(inputs: LossInputs, loss_config: IPOLossConfig) -> LossOutputs:
    """Symmetric |prob diff| trust region (loss.py:164-196)."""
    return _one_sample(inputs, loss_config, "rl")


def ref_kl_loss_fn(inputs: LossInputs) -> LossOutputs:
    """Reverse KL to the reference model as the policy-gradient signal (loss.py:199-243)."""
    return _one_sample(inputs, None, "ref_kl")


def ce_loss_fn(inputs: LossInputs) -> LossOutputs:
    """Masked negative log-likelihood (loss.py:246-259)."""
    return _one_sample(inputs, None, "ce")


def import_object(path: str) -> Any:
    """`pkg.module.attr` -> object (prime_rl.utils.utils.import_object)."""
    module, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(module), attr)


def setup_rl_loss_fn(loss_config) -> LossFn:
    """Build the rl-component loss fn from `trainer.loss` (loss.py:262-282).  Built-in types carry `_pm_config` so
    compute_loss can take the single-launch path."""
    if getattr(loss_config, "type", "default") == "custom":
        custom_fn = import_object(loss_config.import_path)
        kwargs = loss_config.kwargs

        def rl_fn(inputs: LossInputs) -> LossOutputs:
            return custom_fn(inputs, **kwargs)

        return rl_fn
    if getattr(loss_config, "type", "default") == "ipo":

        def rl_fn(inputs: LossInputs) -> LossOutputs:
            return ipo_loss_fn(inputs, loss_config)
    else:

        def rl_fn(inputs: LossInputs) -> LossOutputs:
            return default_loss_fn(inputs, loss_config)

    rl_fn._pm_config = loss_config
    return rl_fn


def compute_loss(trainer_logprobs: list[Tensor], inference_logprobs: list[Tensor], ref_logprobs: list[Tensor] | None, advantages: list[Tensor],
                 loss_mask: list[Tensor], rl_weights: list[Tensor] | None, ce_weights: list[Tensor] | None, ref_kl_weights: list[Tensor] | None,
                 rl_loss_fn: LossFn, rl_scale: int, ce_scale: int, ref_kl_scale: int) -> tuple[Tensor, dict[str, Any]]:
    """Loss of a packed micro-batch = rl/rl_scale + ce/ce_scale + ref_kl/ref_kl_scale (loss.py:285-403); same signature, returns
    (scaled_loss, {metric: stacked per-sample tensor})."""
    cfg = getattr(rl_loss_fn, "_pm_config", None)
    if cfg is None:
        return _compute_loss_plugin(trainer_logprobs, inference_logprobs, ref_logprobs, advantages, loss_mask, rl_weights, ce_weights, ref_kl_weights,
                                    rl_loss_fn, rl_scale, ce_scale, ref_kl_scale)
    if ref_kl_weights is not None and ref_logprobs is None:
        raise ValueError("ref_kl loss type requires ref_logprobs — use the 'opd' or 'opsd' algorithm.")
    dev = trainer_logprobs[0].device
    lens = [t.numel() for t in trainer_logprobs]
    cu = torch.tensor([0] + lens, dtype=torch.int64).cumsum(0).to(torch.int32).to(dev)
    mask = torch.cat([m.reshape(-1) for m in loss_mask]).contiguous()
    loss, stats, met = _GrpoLossFn.apply(torch.cat([t.reshape(-1) for t in trainer_logprobs]), _cat(inference_logprobs), _cat(ref_logprobs), _cat(advantages),
                                         mask, _cat(rl_weights), _cat(ce_weights), _cat(ref_kl_weights), cu, _params(cfg, (rl_scale, ce_scale, ref_kl_scale)))
    aggregated: dict[str, Any] = {}
    S = _lib.STAT
    # a component's metrics are appended only for samples where it has members (loss.py:367-380); absent rl stream = every sample
    rl_keys = _lib.METRIC_KEYS[:3] if getattr(cfg, "type", "default") == "ipo" else _lib.METRIC_KEYS[:7]
    rl_rows = None if rl_weights is None else stats[:, S["RL_N_MASK"]] > 0
    for k in rl_keys:
        col = met[:, _lib.METRIC_KEYS.index(k)]
        col = col if rl_rows is None else col[rl_rows]
        if col.numel():
            aggregated[k] = col
    if ce_weights is not None:
        col = met[:, _lib.METRIC_KEYS.index("nll")][stats[:, S["CE_N"]] > 0]
        if col.numel():
            aggregated["nll"] = col
    if ref_kl_weights is not None:
        rows = stats[:, S["REFKL_N"]] > 0
        for k in _lib.METRIC_KEYS[8:12]:
            col = met[:, _lib.METRIC_KEYS.index(k)][rows]
            if col.numel():
                aggregated[k] = col
    return loss, aggregated


def _compute_loss_plugin(trainer_logprobs, inference_logprobs, ref_logprobs, advantages, loss_mask, rl_weights, ce_weights, ref_kl_weights, rl_loss_fn,
                         rl_scale, ce_scale, ref_kl_scale):
    """Custom-loss path: the reference's per-sample plugin contract (LossFn on LossInputs, un-normalised sum loss + metrics)."""
    n = len(trainer_logprobs)
    none = [None] * n
    ref_logprobs, rl_weights, ce_weights, ref_kl_weights = (x if x is not None else none for x in (ref_logprobs, rl_weights, ce_weights, ref_kl_weights))
    all_metrics: dict[str, list[Tensor]] = {}

    def run(fn, inputs):
        res = fn(inputs)
        for k, v in res.metrics.items():
            all_metrics.setdefault(k, []).append(v)
        return res.loss

    rl_loss = trainer_logprobs[0].sum() * 0.0  # graph anchor (loss.py:350-355)
    ce_loss = ref_kl_loss = 0.0
    for t, i, r, a, m, rw, cw, kw in zip(trainer_logprobs, inference_logprobs, ref_logprobs, advantages, loss_mask, rl_weights, ce_weights, ref_kl_weights):
        mk = lambda cm, w: LossInputs(trainer_logprobs=t, inference_logprobs=i, ref_logprobs=r, advantages=a, loss_mask=cm, loss_weights=w)
        if rw is None:
            rl_loss = rl_loss + run(rl_loss_fn, mk(m, None))
        else:
            rm = m & (rw != 0)
            if bool(rm.any()):
                rl_loss = rl_loss + run(rl_loss_fn, mk(rm, rw))
        if cw is not None and bool((cw != 0).any()):
            ce_loss = ce_loss + run(ce_loss_fn, mk(cw != 0, cw))
        if kw is not None and bool((kw != 0).any()):
            ref_kl_loss = ref_kl_loss + run(ref_kl_loss_fn, mk(kw != 0, kw))
    scaled = rl_loss / rl_scale + ce_loss / ce_scale + ref_kl_loss / ref_kl_scale
    agg = {k: (torch.stack(v) if v[0].dim() == 0 else torch.cat(v)) for k, v in all_metrics.items()}
    return scaled, agg


def plugin_loss_over_pass(logprobs: Tensor, packed, scales: tuple[int, int, int], rl_loss_fn: LossFn, vocab_size: int):
    """A custom loss plugin (`[trainer.loss] type = "custom"`, loss.py:262-282) over one packed pass of the engines: the reference's micro-batch body
    (train.py:450-480) — shift_tensor_right with log(1/V) at the head of every micro-batch, per-sample split, `compute_loss` with the plugin as the rl
    component — run under autograd on the engine's (detached) log-probs, so that d loss / d log-prob comes back as a plain tensor the engines feed into
    their own backward.  Returns (total scaled loss, grad w.r.t. the UNSHIFTED log-probs, per-micro-batch losses [n_mb], {metric: [per-micro-batch tensors]})."""
    n = logprobs.numel()
    lp = logprobs.detach().float().clone().requires_grad_(True)
    starts = packed.sample_starts_mb.tolist()
    cu = packed.cu_samples.tolist()
    pad = lp.new_full((1,), math.log(1.0 / vocab_size))
    mb_bounds = [cu[i] for i, s in enumerate(starts) if s] + [n]
    sample_of_mb: list[list[tuple[int, int]]] = []
    for i, s in enumerate(starts):
        if s:
            sample_of_mb.append([])
        sample_of_mb[-1].append((cu[i], cu[i + 1]))

    def opt(t):
        return None if t is None else t

    total = lp.sum() * 0.0
    per_mb, metrics = [], {}
    for (m0, m1), samples in zip(zip(mb_bounds[:-1], mb_bounds[1:]), sample_of_mb):
        shifted = torch.cat([pad, lp[m0:m1 - 1]])  # shift_tensor_right within the micro-batch (loss.py:76-91)

        def cut(t):
            return None if t is None else [t[a:b] for a, b in samples]

        loss, met = compute_loss([shifted[a - m0:b - m0] for a, b in samples], cut(packed.inference_logprobs), cut(opt(packed.ref_logprobs)),
                                 cut(packed.advantages), cut(packed.loss_mask), cut(opt(packed.rl_weights)), cut(opt(packed.ce_weights)),
                                 cut(opt(packed.ref_kl_weights)), rl_loss_fn=rl_loss_fn, rl_scale=scales[0], ce_scale=scales[1], ref_kl_scale=scales[2])
        total = total + loss
        per_mb.append(loss.detach().reshape(()))
        for k, v in met.items():
            metrics.setdefault(k, []).append(v.detach().reshape(-1))
    total.backward()
    return total.detach().reshape(1), lp.grad, torch.stack(per_mb), metrics
