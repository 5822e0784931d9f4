"""oracle/cpu_trainer.py — the reference's GRPO trainer step restated for host cores (torch CPU, f32, autograd).

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/grpo_oracle.py header): used by tests as the end-to-end checker of the
engine and by bench.py's `cpu_baseline` leg ("port").  The reference trainer itself cannot run on a CPU (unconditional
torch.cuda.* and flash-attn-only attention: src/prime_rl/trainer/rl/train.py:263,318-345, models/layers/attn.py:185-189),
so the baseline is this port of its arithmetic:

  forward  : oracle.qwen3_forward  (modeling_qwen3.py:155-196 / attn.py:123-160 restated)
  log-probs: logits/T -> log_softmax -> gather, entropy (trainer/rl/loss.py:48-63, train.py:436-443)
  shift    : shift_tensor_right with pad log(1/V) (loss.py:76-91, train.py:450-457)
  loss     : default_loss_fn summed over samples / global scale (loss.py:109-161, 285-403)
  backward : torch autograd;  clip_grad_norm_(1.0) + AdamW (trainer/utils.py:88-98, optim/__init__.py:101-108)
"""

from __future__ import annotations

import math
import time

import torch

from . import grpo_oracle as O


def default_loss_torch(lp, li, adv, mask, cfg: O.LossConfig):
    """default_loss_fn (trainer/rl/loss.py:109-161) in differentiable torch ops, whole flat batch at once."""
    lr = lp - li
    ratio = torch.exp(lr)
    pd = torch.exp(lp) - torch.exp(li)
    pos = adv > 0
    invalid = torch.where(pos, pd > cfg.dppo_mask_high, pd < -cfg.dppo_mask_low)
    keep = mask & ~invalid
    pg = keep * (cfg.adv_tau * adv) * ratio
    kl = mask * lr**2
    return (-pg + cfg.kl_tau * kl).sum()


def step_logprobs_and_loss(params: dict, cfg, micro_batches: list[dict], loss_cfg: O.LossConfig, rl_scale: int, mm=None, f32_logits: bool = False,
                           attn_p_bf16: bool = False):
    """Forward + log-probs + loss of a list of TensorMicroBatch dicts; returns (loss, [logprobs per mb], [entropy per mb]).
    Device-agnostic (runs where the tensors live).  `mm` = the GEMM hook of O.qwen3_forward; `f32_logits` keeps the head GEMM's
    output in f32 instead of rounding it to the activation dtype first (the reference rounds: lm_head.py:133 / :23-34)."""
    total = 0.0
    lps, ents = [], []
    for mb in micro_batches:
        ids = mb["input_ids"][0]
        labels = torch.cat([ids[1:], ids.new_zeros(1)])
        h = O.qwen3_forward(params, cfg, ids, mb["position_ids"][0], [int(x) for x in mb["seq_lens"]], mm=mm, attn_p_bf16=attn_p_bf16)
        if f32_logits:
            logits = (h.float() @ params["lm_head.weight"].float().T) / mb["temperatures"][0][:, None]
        else:
            logits = ((mm(h, params["lm_head.weight"]) if mm is not None else h @ params["lm_head.weight"].T)).float() / mb["temperatures"][0][:, None]
        logp_all = torch.log_softmax(logits, dim=-1)
        lp = logp_all.gather(-1, labels[:, None])[:, 0]
        with torch.no_grad():
            ent = torch.logsumexp(logits, -1) - (torch.softmax(logits, -1) * logits).sum(-1)
        v = logits.shape[-1]
        lp_s = torch.cat([lp.new_full((1,), math.log(1.0 / v)), lp[:-1]])
        ent_s = torch.cat([ent.new_full((1,), math.log(float(v))), ent[:-1]])
        total = total + default_loss_torch(lp_s, mb["inference_logprobs"][0], mb["advantages"][0], mb["loss_mask"][0], loss_cfg) / rl_scale
        lps.append(lp_s.detach())
        ents.append(ent_s)
    return total, lps, ents


def train_step(params: dict, cfg, micro_batches: list[dict], opt: torch.optim.Optimizer, loss_cfg: O.LossConfig | None = None, max_norm: float = 1.0):
    """One full CPU trainer step on leaf tensors in `params` (requires_grad=True). Returns (loss, grad_norm, logprobs, entropy)."""
    loss_cfg = loss_cfg or O.LossConfig()
    rl_scale = max(sum(int(mb["loss_mask"].sum()) for mb in micro_batches), 1)
    opt.zero_grad(set_to_none=True)
    loss, lps, ents = step_logprobs_and_loss(params, cfg, micro_batches, loss_cfg, rl_scale)
    loss.backward()
    leaves = [p for p in params.values() if p.requires_grad]
    norm = torch.nn.utils.clip_grad_norm_(leaves, max_norm)
    opt.step()
    return loss.detach(), norm, lps, ents


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _median_of_3_after_warmup(fn) -> float:
    fn()  # warm-up (page faults, thread pool, oneDNN primitive caches)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1]


def time_cpu_baseline(dims, seq_len: int = 2048, sample_tokens: int = 256, opt_params: int = 64 * 1024 * 1024, threads: int | None = None,
                      micro_batches_per_step: int = 8) -> dict:
    """Bounded CPU sample (~10-30 s of host work) of the GRPO step, f32, host cores (SURVEY.md section 8d "CPU baseline"):
      (a) a 2-LAYER slice of the model at full width, forward + backward on a `sample_tokens`-token rollout prefix  (x L/2 layers),
      (b) the full-vocab LM head + log-probs + loss forward + backward on the same tokens,
      (c) AdamW on `opt_params` parameters (x params / opt_params) — ONE optimizer step per trainer step, i.e. charged 1 / micro_batches_per_step per
          sample (train.py:566-570 runs it once after all micro-batches).
    Every leg: one warm-up run, then the median of three.  Extrapolation to one `seq_len`-token sample of the full model is linear in tokens (this
    ignores the quadratic attention term and therefore FAVOURS the CPU):
        t = ((a / 2) * L + b) * seq_len / sample_tokens + c * (P / opt_params) / micro_batches_per_step."""
    import os

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or avail
    n_slice = min(2, dims.num_hidden_layers)
    cfg = O.TinyQwen3Config(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=n_slice,
                            num_attention_heads=dims.num_attention_heads, num_key_value_heads=dims.num_key_value_heads, head_dim=dims.head_dim,
                            vocab_size=dims.vocab_size, rms_norm_eps=dims.rms_norm_eps, rope_theta=dims.rope_theta,
                            qk_norm=getattr(dims, "qk_norm", True), attention_bias=getattr(dims, "attention_bias", False))
    params = {k: v.float().requires_grad_(True) for k, v in O.init_qwen3_params(cfg, seed=0, dtype=torch.float32).items()}
    n = sample_tokens
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 100, (1, n), generator=g)
    mb = dict(input_ids=ids, position_ids=torch.arange(n)[None], advantages=torch.randn(1, n, generator=g), inference_logprobs=torch.randn(1, n, generator=g),
              temperatures=torch.ones(1, n), loss_mask=torch.ones(1, n, dtype=torch.bool), seq_lens=torch.tensor([n]))
    hidden = {}

    def slice_fb():
        h = O.qwen3_forward(params, cfg, ids[0], mb["position_ids"][0], [n])
        h.float().square().mean().backward()
        hidden["h"] = h.detach()
        for p_ in params.values():
            p_.grad = None

    # (a) the thread count is tuned: all cores is often NOT the fastest for these GEMM sizes; the best of {all, 64, 32} usable cores is reported
    t_slice, best_threads = float("inf"), threads
    for nt in sorted({threads, min(threads, 64), min(threads, 32)}, reverse=True):
        torch.set_num_threads(nt)
        dt = _median_of_3_after_warmup(slice_fb)
        if dt < t_slice:
            t_slice, best_threads = dt, nt
    threads = best_threads
    torch.set_num_threads(threads)

    # (b) LM head + log-probs + loss on detached hidden states
    def head_fb():
        hd = hidden["h"].clone().requires_grad_(True)
        labels = torch.cat([ids[0, 1:], ids.new_zeros(1)])
        logits = (hd @ params["lm_head.weight"].T).float() / mb["temperatures"][0][:, None]
        lp = torch.log_softmax(logits, dim=-1).gather(-1, labels[:, None])[:, 0]
        with torch.no_grad():
            _ = torch.logsumexp(logits, -1) - (torch.softmax(logits, -1) * logits).sum(-1)
        lp_s = torch.cat([lp.new_full((1,), math.log(1.0 / dims.vocab_size)), lp[:-1]])
        loss = default_loss_torch(lp_s, mb["inference_logprobs"][0], mb["advantages"][0], mb["loss_mask"][0], O.LossConfig()) / n
        loss.backward()
        params["lm_head.weight"].grad = None

    t_head = _median_of_3_after_warmup(head_fb)
    # (c) AdamW on a fixed-size flat parameter
    flat = torch.randn(opt_params).requires_grad_(True)
    flat.grad = torch.randn(opt_params)
    opt = torch.optim.AdamW([flat], lr=1e-6, weight_decay=0.01)
    t_opt = _median_of_3_after_warmup(opt.step)
    scale = seq_len / sample_tokens
    t_opt_full = t_opt * (dims.num_params() / opt_params)
    est = (t_slice / n_slice * dims.num_hidden_layers + t_head) * scale + t_opt_full / micro_batches_per_step
    return dict(seconds_per_sample=est, t_slice=t_slice, slice_layers=n_slice, t_head_loss=t_head, t_opt=t_opt, t_opt_full_model=t_opt_full, threads=threads,
                sample_tokens=sample_tokens, opt_params=opt_params, seq_len=seq_len, micro_batches_per_step=micro_batches_per_step, cpu_model=_cpu_model(),
                cores_available=avail)
