"""oracle/grpo_oracle.py — CPU restatement of the reference's GRPO policy-update arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under `prime_rl_amd/` may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only as the checker.

Every function restates (does not copy) one piece of PrimeIntellect-ai/prime-rl and cites the file:line it follows
(paths relative to `/root/reference/src/prime_rl/`).  The restatement is vectorised over the whole packed micro-batch
(flat arrays + segment offsets) instead of the reference's per-sample Python loop, so it shares no structure with the
reference code while computing the same numbers.

Parity pin: `tests/golden/make_golden.py` file-loads the reference's own `lm_head.py` and `loss.py` inside the build
container (where `/root/reference` exists), runs them on seeded inputs and commits the outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this oracle against those fixtures and against the known-answer values of the
reference's unit tests (0.15 / 0.03 ce losses, zero-sum advantages, MaxRL [3,-1,-1,-1], ...).

numpy float32 for the integer/float streams (K1-K3); torch-CPU for the bf16 scoring forward (bf16 GEMM semantics).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

F32 = np.float32


# =====================================================================================================
# K1 — per-token log-prob gather + entropy
# =====================================================================================================
def logprob_entropy_from_logits(logits: np.ndarray, labels: np.ndarray, temperature: np.ndarray):
    """selective_log_softmax + compute_entropy on logits/T (trainer/rl/loss.py:48-63; divide at trainer/rl/train.py:441).

    logits [N, V] (any float dtype, promoted to f32), labels [N] int, temperature [N] f32.
    Returns (logprobs, entropy, logz) f32 [N].
    """
    x = np.asarray(logits, dtype=F32) / np.asarray(temperature, dtype=F32)[:, None]
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m, dtype=F32)
    s = e.sum(axis=-1, dtype=F32)
    logz = (m[:, 0] + np.log(s, dtype=F32)).astype(F32)
    lp = x[np.arange(x.shape[0]), np.asarray(labels, dtype=np.int64)] - logz
    pd = e / s[:, None]
    ent = logz - (pd * x).sum(axis=-1, dtype=F32)
    return lp.astype(F32), ent.astype(F32), logz


def fused_head_logprob_entropy(hidden: np.ndarray, weight: np.ndarray, labels: np.ndarray, inv_t: np.ndarray, vocab_chunk: int = 8192):
    """_SequenceChunkedLogProbEntropyFn.forward (trainer/models/layers/lm_head.py:89-156): vocab-chunked logits with the
    online (m, s, t) update of lm_head.py:76-86 and the branch-free label pick of lm_head.py:140-143.
    hidden [N, H], weight [V, H] f32 (callers emulate bf16 by pre-rounding).  Returns (logprobs, entropy, logz)."""
    n, v = hidden.shape[0], weight.shape[0]
    m = np.full(n, -np.inf, dtype=F32)
    s = np.zeros(n, dtype=F32)
    t = np.zeros(n, dtype=F32)
    tgt = np.zeros(n, dtype=F32)
    labels = np.asarray(labels, dtype=np.int64)
    for v0 in range(0, v, vocab_chunk):
        v1 = min(v0 + vocab_chunk, v)
        lg = (hidden.astype(F32) @ weight[v0:v1].astype(F32).T).astype(F32) * inv_t[:, None].astype(F32)
        cm = lg.max(axis=-1)
        mn = np.maximum(m, cm)
        old = np.exp(m - mn, dtype=F32)
        ce = np.exp(lg - mn[:, None], dtype=F32)
        s = s * old + ce.sum(axis=-1, dtype=F32)
        t = t * old + (ce * lg).sum(axis=-1, dtype=F32)
        m = mn
        inr = (labels >= v0) & (labels < v1)
        loc = np.clip(labels - v0, 0, v1 - v0 - 1)
        tgt = np.where(inr, lg[np.arange(n), loc], tgt).astype(F32)
    logz = (m + np.log(s, dtype=F32)).astype(F32)
    return (tgt - logz).astype(F32), (logz - t / s).astype(F32), logz


def fused_head_backward(hidden, weight, labels, inv_t, logz, grad_logprobs):
    """_SequenceChunkedLogProbEntropyFn.backward (lm_head.py:158-202), un-chunked: returns (dlogits, grad_hidden, grad_weight)."""
    lg = (hidden.astype(F32) @ weight.astype(F32).T) * inv_t[:, None]
    probs = np.exp(lg - logz[:, None], dtype=F32)
    g = np.asarray(grad_logprobs, dtype=F32)
    dl = (-g)[:, None] * probs
    dl[np.arange(dl.shape[0]), np.asarray(labels, dtype=np.int64)] += g
    dl = (dl * inv_t[:, None]).astype(F32)
    return dl, dl @ weight.astype(F32), dl.T @ hidden.astype(F32)


def shift_left(ids: np.ndarray) -> np.ndarray:
    """shift_tensor_left (trainer/rl/loss.py:66-73): labels[i] = ids[i+1], last = 0."""
    out = np.zeros_like(ids)
    out[..., :-1] = ids[..., 1:]
    return out


def shift_right(t: np.ndarray, pad: float) -> np.ndarray:
    """shift_tensor_right (trainer/rl/loss.py:76-91): out[i] = t[i-1], out[0] = pad."""
    out = np.empty_like(t)
    out[..., 1:] = t[..., :-1]
    out[..., 0] = pad
    return out


# =====================================================================================================
# K3 — advantages
# =====================================================================================================
def grpo_advantages(rewards: np.ndarray, length_penalty: dict | None = None) -> np.ndarray:
    """GRPOAlgorithm.score_group (orchestrator/algo/grpo.py:25-45) for ONE group.  length_penalty = dict(output=, total=, turns=,
    w_output=, w_input=, w_turns=) or None."""
    r = np.asarray(rewards, dtype=F32)
    if length_penalty is None:
        return (r - r.mean(dtype=F32)).astype(F32)
    out = np.asarray(length_penalty["output"], dtype=F32)
    tot = np.asarray(length_penalty["total"], dtype=F32)
    turns = np.asarray(length_penalty["turns"], dtype=F32)
    inp = tot - out
    frac = (
        F32(length_penalty["w_output"]) * (out / max(out.max(), F32(1)))
        + F32(length_penalty["w_input"]) * (inp / max(inp.max(), F32(1)))
        + F32(length_penalty["w_turns"]) * (turns / max(turns.max(), F32(1)))
    ).astype(F32)
    shaped = (r - r.mean(dtype=F32) * frac).astype(F32)
    return (shaped - shaped.mean(dtype=F32)).astype(F32)


def maxrl_advantages(rewards: np.ndarray) -> np.ndarray:
    """MaxRLAlgorithm.score_group (orchestrator/algo/max_rl.py:22-28)."""
    r = np.asarray(rewards, dtype=F32)
    mean = r.mean(dtype=F32)
    return np.zeros_like(r) if mean <= 0 else ((r - mean) / mean).astype(F32)


def broadcast_advantage(scalar: float, mask: np.ndarray) -> np.ndarray:
    """assign_advantages with a scalar (orchestrator/algo/routing.py:11-34) flattened to the per-token stream of
    TrainingSample.advantages (transports/rollouts/types.py:70-74): the rollout scalar on sampled tokens, 0 elsewhere."""
    return np.where(np.asarray(mask, dtype=bool), F32(scalar), F32(0)).astype(F32)


# =====================================================================================================
# K2 — loss
# =====================================================================================================
@dataclass
class LossConfig:
    """DefaultLossConfig / IPOLossConfig defaults (packages/prime-rl-configs/src/prime_rl/configs/trainer.py:494-520)."""

    type: str = "default"
    dppo_mask_low: float = 0.2
    dppo_mask_high: float = 0.2
    adv_tau: float = 1.0
    kl_tau: float = 1e-3
    ipo_threshold: float = 0.1


def _safe_mean(values: np.ndarray, mask: np.ndarray) -> F32:
    """_safe_mean (trainer/rl/loss.py:94-97)."""
    return F32(np.asarray(values, dtype=F32)[mask].sum(dtype=F32) / F32(max(int(mask.sum()), 1)))


def rl_loss_sample(lp, li, adv, mask, cfg: LossConfig, weights=None):
    """default_loss_fn (trainer/rl/loss.py:109-161) / ipo_loss_fn (:164-196) on one sample. Returns (sum loss, metrics, dloss/dlp)."""
    lp, li, adv = (np.asarray(a, dtype=F32) for a in (lp, li, adv))
    mask = np.asarray(mask, dtype=bool)
    lr = lp - li
    ratio = np.exp(lr, dtype=F32)
    mkl = ratio - lr - F32(1)
    pd = np.exp(lp, dtype=F32) - np.exp(li, dtype=F32)
    pos, neg = adv > 0, adv < 0
    metrics = {}
    if cfg.type == "ipo":
        is_masked = np.abs(pd) > F32(cfg.ipo_threshold)
    else:
        high, low = pd > F32(cfg.dppo_mask_high), pd < -F32(cfg.dppo_mask_low)
        is_masked = np.where(pos, high, low)
    drop, keep = mask & is_masked, mask & ~is_masked
    a_s = F32(cfg.adv_tau) * adv
    pg = keep * a_s * ratio
    kl = mask * lr**2
    ptl = -pg + F32(cfg.kl_tau) * kl
    dptl = -pg + F32(cfg.kl_tau) * mask * F32(2) * lr
    if weights is not None:
        w = np.asarray(weights, dtype=F32)
        ptl, dptl = ptl * w, dptl * w
    metrics["masked_mismatch_kl"] = _safe_mean(mkl, drop)
    metrics["unmasked_mismatch_kl"] = _safe_mean(mkl, keep)
    metrics["is_masked"] = _safe_mean(is_masked, mask)
    if cfg.type != "ipo":
        metrics["is_masked_low"] = _safe_mean(neg & low, mask)
        metrics["is_masked_high"] = _safe_mean(pos & high, mask)
        metrics["masked_advantage_positive"] = _safe_mean(pos, drop)
        metrics["masked_advantage_negative"] = _safe_mean(neg, drop)
    return F32(ptl.sum(dtype=F32)), metrics, dptl.astype(F32)


def ce_loss_sample(lp, mask, weights):
    """ce_loss_fn (trainer/rl/loss.py:246-259)."""
    lp = np.asarray(lp, dtype=F32)
    w = np.asarray(weights, dtype=F32)
    nll = -lp * w
    return F32(nll[mask].sum(dtype=F32)), {"nll": _safe_mean(-lp, mask)}, np.where(mask, -w, F32(0)).astype(F32)


def ref_kl_loss_sample(lp, li, lref, mask, weights):
    """ref_kl_loss_fn (trainer/rl/loss.py:199-243)."""
    lp, li, lref = (np.asarray(a, dtype=F32) for a in (lp, li, lref))
    lr = lp - li
    ratio = np.exp(lr, dtype=F32)
    mkl = ratio - lr - F32(1)
    pd = np.exp(lp, dtype=F32) - np.exp(li, dtype=F32)
    is_masked = pd < F32(-0.2)
    drop, keep = mask & is_masked, mask & ~is_masked
    rkl = lref - lp
    pg = keep * rkl * ratio
    ptl = (-pg + F32(1e-3) * (mask * lr**2))
    dptl = (-pg + F32(1e-3) * mask * F32(2) * lr)
    w = np.asarray(weights, dtype=F32)
    ptl, dptl = ptl * w, dptl * w
    metrics = {
        "ref_kl/masked_mismatch_kl": _safe_mean(mkl, drop),
        "ref_kl/unmasked_mismatch_kl": _safe_mean(mkl, keep),
        "ref_kl/is_masked": _safe_mean(is_masked, mask),
        "ref_kl": _safe_mean(rkl, mask),
    }
    return F32(ptl.sum(dtype=F32)), metrics, dptl.astype(F32)


def compute_loss(trainer_lp, inference_lp, advantages, loss_mask, sample_lengths, cfg: LossConfig, rl_scale=1, ce_scale=1, ref_kl_scale=1,
                 ref_lp=None, rl_weights=None, ce_weights=None, ref_kl_weights=None):
    """compute_loss (trainer/rl/loss.py:285-403) on flat arrays split by `sample_lengths`.
    Returns (scaled loss f32, metrics dict name -> f32[n_appended], dloss/d trainer_lp f32[N])."""
    n = len(trainer_lp)
    grad = np.zeros(n, dtype=F32)
    metrics: dict[str, list] = {}
    rl = F32(0)
    ce = F32(0)
    rk = F32(0)
    off = 0
    for ln in sample_lengths:
        sl = slice(off, off + ln)
        off += ln
        m = np.asarray(loss_mask[sl], dtype=bool)
        if rl_weights is None:
            l, met, d = rl_loss_sample(trainer_lp[sl], inference_lp[sl], advantages[sl], m, cfg)
            rl = F32(rl + l)
            grad[sl] += d / F32(rl_scale)
            for k, v in met.items():
                metrics.setdefault(k, []).append(v)
        else:
            w = np.asarray(rl_weights[sl], dtype=F32)
            mm = m & (w != 0)
            if mm.any():
                l, met, d = rl_loss_sample(trainer_lp[sl], inference_lp[sl], advantages[sl], mm, cfg, weights=w)
                rl = F32(rl + l)
                grad[sl] += d / F32(rl_scale)
                for k, v in met.items():
                    metrics.setdefault(k, []).append(v)
        if ce_weights is not None:
            w = np.asarray(ce_weights[sl], dtype=F32)
            mm = w != 0
            if mm.any():
                l, met, d = ce_loss_sample(trainer_lp[sl], mm, w)
                ce = F32(ce + l)
                grad[sl] += d / F32(ce_scale)
                for k, v in met.items():
                    metrics.setdefault(k, []).append(v)
        if ref_kl_weights is not None:
            if ref_lp is None:
                raise ValueError("ref_kl loss type requires ref_logprobs")  # loss.py:213-214
            w = np.asarray(ref_kl_weights[sl], dtype=F32)
            mm = w != 0
            if mm.any():
                l, met, d = ref_kl_loss_sample(trainer_lp[sl], inference_lp[sl], ref_lp[sl], mm, w)
                rk = F32(rk + l)
                grad[sl] += d / F32(ref_kl_scale)
                for k, v in met.items():
                    metrics.setdefault(k, []).append(v)
    scaled = F32(rl / F32(rl_scale) + ce / F32(ce_scale) + rk / F32(ref_kl_scale))
    return scaled, {k: np.asarray(v, dtype=F32) for k, v in metrics.items()}, grad


# =====================================================================================================
# optimizer / gradient post-processing
# =====================================================================================================
def clip_grad_norm(grads: list[np.ndarray], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ semantics used by trainer/utils.py:88-98: global L2, coef = max_norm/(norm+1e-6) clamped to 1."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    return total, coef


def adamw_step(p, m, v, g, lr, b1, b2, eps, wd, step):
    """torch.optim.AdamW single-tensor step (trainer/optim/__init__.py:101-108 builds it with lr, weight_decay, betas)."""
    p = p * F32(1 - lr * wd)
    m = m + (g - m) * F32(1 - b1)
    v = v * F32(b2) + F32(1 - b2) * g * g
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = np.sqrt(v, dtype=F32) / F32(math.sqrt(bc2)) + F32(eps)
    p = p - F32(lr / bc1) * (m / denom)
    return p.astype(F32), m.astype(F32), v.astype(F32)


# =====================================================================================================
# packed-sequence scoring forward (torch CPU; bf16 rounding points of the reference's eager bf16 ops)
# =====================================================================================================
def _torch():
    import torch

    return torch


def rmsnorm(x, weight, eps: float):
    """RMSNorm.forward torch branch (trainer/models/layers/norms.py:40-44)."""
    torch = _torch()
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_inv_freq(head_dim: int, theta: float, rope_scaling: dict | None = None):
    """ROPE_INIT_FUNCTIONS of the reference's RotaryEmbedding (rotary_emb.py:41-57): "default" (:10-26), "linear" (inv_freq / factor) and "llama3"
    (transformers `_compute_llama3_parameters`: long wavelengths / factor, a smooth band, short wavelengths untouched); attention_scaling is 1 for all."""
    import math

    torch = _torch()
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))
    kind = (rope_scaling or {}).get("rope_type", (rope_scaling or {}).get("type", "default"))
    if kind == "linear":
        return inv_freq / float(rope_scaling["factor"])
    if kind == "llama3":
        f, lo, hi, old = (float(rope_scaling[k]) for k in ("factor", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings"))
        wavelen = 2 * math.pi / inv_freq
        base = torch.where(wavelen > old / lo, inv_freq / f, inv_freq)
        smooth = (old / wavelen - lo) / (hi - lo)
        mid = ~(wavelen < old / hi) & ~(wavelen > old / lo)
        return torch.where(mid, (1 - smooth) * base / f + smooth * base, base)
    assert kind == "default", kind
    return inv_freq


def rope_cos_sin(position_ids, head_dim: int, theta: float, dtype, rope_scaling: dict | None = None):
    """RotaryEmbedding.forward (trainer/models/layers/rotary_emb.py:60-73) with the config's ROPE_INIT_FUNCTIONS entry (rope_inv_freq above)."""
    torch = _torch()
    inv_freq = rope_inv_freq(head_dim, theta, rope_scaling).to(position_ids.device)
    freqs = position_ids.to(torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rope(x, cos, sin):
    """apply_rotary_pos_emb / rotate_half (rotary_emb.py:76-118); x [N, heads, D], cos/sin [N, D]."""
    torch = _torch()
    d = x.shape[-1]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    rot = torch.cat((-x2, x1), dim=-1)
    return (x * cos[:, None, :]) + (rot * sin[:, None, :])


def varlen_causal_attention(q, k, v, seq_lens: list[int], scale: float, window: int | None = None, p_bf16: bool = False):
    """flash_attn_varlen_func(q,k,v,cu,cu,max,max,causal=True) semantics at FlashAttention._compute_attention
    (trainer/models/layers/attn.py:94-110): per-document causal softmax attention with GQA, f32 math.
    q [N,Hq,D], k/v [N,Hkv,D] -> [N,Hq,D] f32.  `window` = the optional sliding window (attn.py:97-99: `window_size=(sliding_window - 1, 0)`):
    query i sees keys i - window + 1 .. i of its document.
    `p_bf16`: model the ONE rounding point inside the flash-attention kernels the f32 restatement does not have (flash-attn 2.8.3, the reference's
    pinned dependency): the unnormalised probabilities exp(s - rowmax) are cast to the input dtype for the P.V matrix product
    (f32 accumulation) and the row sum stays f32.  Not bit-identical to the kernel (its running row max moves block by block); same rounding statistics."""
    torch = _torch()
    n, hq, d = q.shape
    hkv = k.shape[1]
    g = hq // hkv
    out = torch.zeros((n, hq, d), dtype=torch.float32, device=q.device)
    off = 0
    for ln in seq_lens:
        causal = torch.ones(ln, ln, dtype=torch.bool, device=q.device).tril()
        if window is not None:
            causal &= torch.ones(ln, ln, dtype=torch.bool, device=q.device).triu(-(window - 1))
        # heads are independent: long documents (the 32k-token rollouts of BASELINE.json configs[4]) are processed a few heads at a time so that
        # the [heads, ln, ln] f32 score block stays below ~8 GB; the arithmetic per head is the same either way
        hc = max(1, min(hq, int(8e9 // max(4 * ln * ln, 1))))
        hc = max(g, hc // g * g) if hc >= g else hc  # whole KV groups when possible (keeps the einsum shapes of the short-document path)
        for h0 in range(0, hq, hc):
            h1 = min(h0 + hc, hq)
            qs = q[off:off + ln, h0:h1].to(torch.float32)
            kv_idx = torch.arange(h0, h1, device=q.device) // g
            ks = k[off:off + ln].to(torch.float32)[:, kv_idx]
            vs = v[off:off + ln].to(torch.float32)[:, kv_idx]
            s = torch.einsum("ihd,jhd->hij", qs, ks) * scale
            s = s.masked_fill(~causal, float("-inf"))
            if p_bf16:
                e = torch.exp(s - s.amax(dim=-1, keepdim=True))
                den = e.sum(dim=-1)
                o = torch.einsum("hij,jhd->ihd", e.to(torch.bfloat16).to(torch.float32), vs)
                out[off:off + ln, h0:h1] = o / den.transpose(0, 1)[:, :, None]
            else:
                p = torch.softmax(s, dim=-1)
                out[off:off + ln, h0:h1] = torch.einsum("hij,jhd->ihd", p, vs)
        off += ln
    return out


def swiglu(gate, up):
    """MLP.forward activation (trainer/models/layers/mlp.py:27-29): silu(gate) * up in the activation dtype."""
    torch = _torch()
    return torch.nn.functional.silu(gate) * up


@dataclass
class TinyQwen3Config:
    hidden_size: int = 256
    intermediate_size: int = 512
    num_hidden_layers: int = 2
    num_attention_heads: int = 4
    num_key_value_heads: int = 2
    head_dim: int = 128
    vocab_size: int = 512
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    qk_norm: bool = True  # AttentionConfig.use_qk_norm (models/layers/attn.py:37): Qwen3 True, Llama / Qwen2 False
    attention_bias: bool = False  # AttentionConfig.attention_bias (attn.py:36,64-72): bias on q/k/v_proj — Qwen2 True
    rope_scaling: dict | None = None  # HF rope_scaling of type "linear" / "llama3" (Llama-3.x checkpoints)


def exact_mm(a, w, bias=None):
    """a @ w.T (+ bias, added to the f32 accumulator as F.linear does) with exact products and f32 accumulation, rounded ONCE to a's dtype: the GEMM every bf16 implementation of the
    reference approximates (cuBLAS / hipBLASLt / CPU differ only in accumulation order).  Used as the `mm` hook of qwen3_forward to
    take GEMM summation order out of an engine-vs-oracle comparison while keeping the reference's bf16 rounding points."""
    torch = _torch()
    y = a.to(torch.float32) @ w.to(torch.float32).T
    return (y if bias is None else y + bias.to(torch.float32)).to(a.dtype)


def qwen3_forward(params: dict, cfg, input_ids, position_ids, seq_lens: list[int], mm=None, attn_p_bf16: bool = False):
    """Qwen3Model.forward + Qwen3DecoderLayer.forward (trainer/models/qwen3/modeling_qwen3.py:65-85,155-196) with
    FlashAttention.attn_projections (trainer/models/layers/attn.py:123-160).  `params` uses HF names
    (model.layers.{i}.self_attn.q_proj.weight, ...); returns the final-norm hidden states [N, H] in the params' dtype.
    The same function is the Llama block (models/llama/modeling_llama.py:52-100: use_qk_norm=False) and HF Qwen2 (bias on q/k/v, no q/k
    norm — the class the reference loads for Qwen2.5): a `q_proj.bias` in `params` is added after the projection
    (nn.Linear, attn.py:64-72), a missing `q_norm.weight` skips the per-head norm (attn.py:143-145).
    Runs on whatever device the tensors live on; `mm(a, w[, bias]) -> a @ w.T (+ bias)` overrides the linear layers' GEMM (default: torch's
    F.linear); the three-argument form is only used for projections that have a bias."""
    torch = _torch()
    mm = mm or (lambda a, w, b=None: torch.nn.functional.linear(a, w, b))
    hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    x = params["model.embed_tokens.weight"][input_ids]
    cos, sin = rope_cos_sin(position_ids, d, cfg.rope_theta, x.dtype, getattr(cfg, "rope_scaling", None))
    n = x.shape[0]
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        res = x
        h = rmsnorm(x, params[pre + "input_layernorm.weight"], cfg.rms_norm_eps)
        def proj(name):
            b = params.get(pre + f"self_attn.{name}_proj.bias")
            wt = params[pre + f"self_attn.{name}_proj.weight"]
            return mm(h, wt) if b is None else mm(h, wt, b)  # nn.Linear: the bias joins the accumulator before the single rounding

        q, k, v = proj("q").view(n, hq, d), proj("k").view(n, hkv, d), proj("v").view(n, hkv, d)
        if pre + "self_attn.q_norm.weight" in params:
            q = rmsnorm(q, params[pre + "self_attn.q_norm.weight"], cfg.rms_norm_eps)
            k = rmsnorm(k, params[pre + "self_attn.k_norm.weight"], cfg.rms_norm_eps)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        a = varlen_causal_attention(q, k, v, seq_lens, d**-0.5, p_bf16=attn_p_bf16).to(x.dtype).reshape(n, hq * d)
        x = res + mm(a, params[pre + "self_attn.o_proj.weight"])
        res = x
        h = rmsnorm(x, params[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        act = swiglu(mm(h, params[pre + "mlp.gate_proj.weight"]), mm(h, params[pre + "mlp.up_proj.weight"]))
        x = res + mm(act, params[pre + "mlp.down_proj.weight"])
    return rmsnorm(x, params["model.norm.weight"], cfg.rms_norm_eps)


def init_qwen3_params(cfg, seed: int = 0, dtype=None, std: float = 0.02) -> dict:
    """Random-init weights of the reference's `config.debug.random_init` kind (no checkpoints exist offline): N(0, std) linears
    and embeddings, ones for norms, HF parameter names."""
    torch = _torch()
    dtype = dtype or torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    hq, hkv, d, h, inter = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size

    def w(*shape):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    p = {"model.embed_tokens.weight": w(cfg.vocab_size, h), "model.norm.weight": (1 + 0.1 * torch.randn(h, generator=g)).to(dtype),
         "lm_head.weight": w(cfg.vocab_size, h)}
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        p[pre + "self_attn.q_proj.weight"] = w(hq * d, h)
        p[pre + "self_attn.k_proj.weight"] = w(hkv * d, h)
        p[pre + "self_attn.v_proj.weight"] = w(hkv * d, h)
        p[pre + "self_attn.o_proj.weight"] = w(h, hq * d)
        if getattr(cfg, "qk_norm", True):
            p[pre + "self_attn.q_norm.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
            p[pre + "self_attn.k_norm.weight"] = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
        if getattr(cfg, "attention_bias", False):
            for nm, rows in (("q", hq * d), ("k", hkv * d), ("v", hkv * d)):
                p[pre + f"self_attn.{nm}_proj.bias"] = (torch.randn(rows, generator=g) * 0.1).to(dtype)
        p[pre + "mlp.gate_proj.weight"] = w(inter, h)
        p[pre + "mlp.up_proj.weight"] = w(inter, h)
        p[pre + "mlp.down_proj.weight"] = w(h, inter)
        p[pre + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype)
        p[pre + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype)
    return p
