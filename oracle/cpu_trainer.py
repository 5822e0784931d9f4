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


def step_logprobs_and_loss(params: dict, cfg, micro_batches: list[dict], loss_cfg: O.LossConfig, rl_scale: int):
    """Forward + log-probs + loss of a list of TensorMicroBatch dicts; returns (loss, [logprobs per mb], [entropy per mb])."""
    total = 0.0
    lps, ents = [], []
    for mb in micro_batches:
        ids = mb["input_ids"][0]
        labels = torch.cat([ids[1:], ids.new_zeros(1)])
        h = O.qwen3_forward(params, cfg, ids, mb["position_ids"][0], [int(x) for x in mb["seq_lens"]])
        logits = (h @ params["lm_head.weight"].T).float() / mb["temperatures"][0][:, None]
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


def time_cpu_baseline(dims, seq_len: int = 2048, layers_timed: int = 1, threads: int | None = None) -> dict:
    """Bounded CPU sample of the Qwen3 GRPO step: `layers_timed` decoder layers + the full-vocab LM head + loss + AdamW on one
    `seq_len`-token rollout, f32, all host cores.  Extrapolated to the full depth: t = t_layers*(L/layers_timed) + t_head + t_opt*(P/P_timed)."""
    import os

    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    cfg = O.TinyQwen3Config(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=layers_timed,
                            num_attention_heads=dims.num_attention_heads, num_key_value_heads=dims.num_key_value_heads, head_dim=dims.head_dim,
                            vocab_size=dims.vocab_size, rms_norm_eps=dims.rms_norm_eps, rope_theta=dims.rope_theta)
    params = {k: v.float().requires_grad_(True) for k, v in O.init_qwen3_params(cfg, seed=0, dtype=torch.float32).items()}
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 100, (1, seq_len), generator=g)
    mb = dict(input_ids=ids, position_ids=torch.arange(seq_len)[None], advantages=torch.randn(1, seq_len, generator=g),
              inference_logprobs=torch.randn(1, seq_len, generator=g), temperatures=torch.ones(1, seq_len), loss_mask=torch.ones(1, seq_len, dtype=torch.bool),
              seq_lens=torch.tensor([seq_len]))
    layer_keys = [k for k in params if k.startswith("model.layers.")]
    n_layer_params = sum(params[k].numel() for k in layer_keys)
    n_other = sum(v.numel() for k, v in params.items() if k not in layer_keys)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-6, weight_decay=0.01)

    # (a) decoder layers only: forward+backward of the hidden-state path
    t0 = time.perf_counter()
    h = O.qwen3_forward(params, cfg, ids[0], mb["position_ids"][0], [seq_len])
    h.float().square().mean().backward()
    t_layers = time.perf_counter() - t0
    opt.zero_grad(set_to_none=True)
    # (b) full step on the slice: layers + head + loss + clip + AdamW
    t0 = time.perf_counter()
    train_step(params, cfg, [mb], opt)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    opt.step()  # (c) optimizer alone on the slice (grads still present)
    t_opt = time.perf_counter() - t0
    t_head = max(t_full - t_layers - t_opt, 0.0)
    n_total = dims.num_params()
    est = t_layers * (dims.num_hidden_layers / layers_timed) + t_head + t_opt * (n_total / (n_layer_params + n_other))
    return dict(seconds_per_sample=est, t_layers=t_layers, t_head_loss=t_head, t_opt=t_opt, threads=threads, layers_timed=layers_timed, seq_len=seq_len)
