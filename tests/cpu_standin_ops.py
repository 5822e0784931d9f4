"""CPU stand-ins for `prime_rl_amd.ops` — TEST INFRASTRUCTURE ONLY (tests/test_engine_gloo_cpu.py).

They let the UNCHANGED GRPOEngine / zero.py code (bucket bookkeeping, scale all-reduce, per-bucket reduce-scatter, sharded clip +
AdamW, in-place parameter all-gather, multi-pass accumulation) run end to end over gloo at world_size 2 in this GPU-less container,
so that the first multi-GPU run of the engine is not a debugging session ('Next round' 9).  Every function has the
signature of its `prime_rl_amd.ops` namesake and is written with plain torch on the oracle's restatements (autograd supplies the
backward passes).  Nothing here is a fallback of the product: the engine module is pointed at this file by monkeypatching inside the
test's worker processes only."""

from __future__ import annotations


import torch

from oracle import grpo_oracle as O

F32, BF16 = torch.float32, torch.bfloat16


def rope_inv_freq(head_dim, theta, device):
    return (1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(F32) / head_dim))).to(device)


def embedding_fwd(table, ids):
    return table[ids]


def embedding_bwd(grad_out, ids, grad_table_f32):
    grad_table_f32.index_add_(0, ids.reshape(-1), grad_out.float())


def embedding_bwd_window(grad_out, ids, grad_window_f32, elem_lo, vocab):
    h = grad_out.shape[1]
    ids = ids.reshape(-1)
    keep = (ids >= 0) & (ids < vocab)
    elems = (ids[keep, None] * h + torch.arange(h)[None, :]).reshape(-1) - elem_lo  # flat element index of every gradient entry, window-relative
    vals = grad_out.float()[keep].reshape(-1)
    inside = (elems >= 0) & (elems < grad_window_f32.numel())
    grad_window_f32.index_add_(0, elems[inside], vals[inside])


def rope_table(position_ids, inv_freq):
    ang = position_ids.to(F32)[:, None] * inv_freq[None, :]
    return ang.cos().to(BF16), ang.sin().to(BF16)


def seq_start_from_cu(cu_seqlens, n_tokens):
    lens = (cu_seqlens[1:] - cu_seqlens[:-1]).long()
    return torch.repeat_interleave(cu_seqlens[:-1], lens, output_size=n_tokens).to(torch.int32)


def _rms(x, w, eps):
    h = x.float()
    rstd = torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return (w.float() * (h * rstd).to(x.dtype).float()).to(x.dtype), rstd[..., 0]


def rmsnorm_fwd(x, weight, eps, residual=None):
    s = (x.float() + residual.float()).to(BF16) if residual is not None else None
    y, rstd = _rms(s if s is not None else x, weight, eps)
    return y, rstd.reshape(-1), s


@torch.enable_grad()  # also callable from inside an autograd.Function.backward (grad mode is off there)
def rmsnorm_bwd(grad_y, x, weight, rstd, grad_weight_f32, dresidual=None):
    xf = x.float().requires_grad_(True)
    wf = weight.float().requires_grad_(True)
    y = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    gx, gw = torch.autograd.grad(y, (xf, wf), grad_y.float())
    grad_weight_f32 += gw
    if dresidual is not None:
        gx = gx + dresidual.float()
    return gx.to(BF16)


def _qk(qkv, qw, kw, cos, sin, hq, hkv, d, eps):
    n = qkv.shape[0]
    q = qkv[:, : hq * d].reshape(n, hq, d)
    k = qkv[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
    c = torch.cat([cos, cos], -1).float()[:, None, :]
    s = torch.cat([sin, sin], -1).float()[:, None, :]

    def one(x, w):
        xn = x if w is None else w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))  # w None: use_qk_norm False, RoPE only
        rot = torch.cat([-xn[..., d // 2:], xn[..., : d // 2]], -1)
        return xn * c + rot * s

    return one(q, qw), one(k, kw)


def qknorm_rope_fwd(qkv, q_norm_w, k_norm_w, cos, sin, n_q_heads, n_kv_heads, head_dim, eps):
    if q_norm_w is None:
        q, k = _qk(qkv.float(), None, None, cos, sin, n_q_heads, n_kv_heads, head_dim, eps)
        return q.to(BF16), k.to(BF16), None
    q, k = _qk(qkv.float(), q_norm_w.float(), k_norm_w.float(), cos, sin, n_q_heads, n_kv_heads, head_dim, eps)
    rstd = torch.zeros(qkv.shape[0], n_q_heads + n_kv_heads)
    return q.to(BF16), k.to(BF16), rstd


@torch.enable_grad()  # also callable from inside an autograd.Function.backward (grad mode is off there)
def qknorm_rope_bwd(grad_q, grad_k, qkv, q_norm_w, k_norm_w, cos, sin, rstd, grad_qkv, gqw, gkw, n_q_heads, n_kv_heads, head_dim):
    w = (n_q_heads + n_kv_heads) * head_dim
    if q_norm_w is None:  # the rotation is linear: its transpose does not need the forward input
        x = torch.zeros(grad_qkv.shape, dtype=F32).requires_grad_(True)
        q, k = _qk(x, None, None, cos, sin, n_q_heads, n_kv_heads, head_dim, 1e-6)
        (gx,) = torch.autograd.grad((q, k), (x,), (grad_q.float().reshape(q.shape), grad_k.float().reshape(k.shape)))
        grad_qkv[:, :w] = gx[:, :w].to(BF16)
        return
    x = qkv.float().requires_grad_(True)
    qw, kw = q_norm_w.float().requires_grad_(True), k_norm_w.float().requires_grad_(True)
    q, k = _qk(x, qw, kw, cos, sin, n_q_heads, n_kv_heads, head_dim, 1e-6)
    gx, gq, gk = torch.autograd.grad((q, k), (x, qw, kw), (grad_q.float(), grad_k.float()))
    w = (n_q_heads + n_kv_heads) * head_dim
    grad_qkv[:, :w] = gx[:, :w].to(BF16)
    gqw += gq
    gkw += gk


def column_sum(x, out_f32):
    out_f32 += x.float().sum(0)


def _seq_lens(seq_start):
    starts = seq_start.long()
    n = starts.numel()
    bounds = torch.unique_consecutive(starts).tolist() + [n]
    return [b - a for a, b in zip(bounds[:-1], bounds[1:])]


def attn_varlen_fwd(q, k, v, seq_start, softmax_scale):
    o = O.varlen_causal_attention(q.float(), k.float(), v.float(), _seq_lens(seq_start), softmax_scale)
    return o.to(BF16), torch.zeros(q.shape[1], q.shape[0])


@torch.enable_grad()  # also callable from inside an autograd.Function.backward (grad mode is off there)
def attn_varlen_bwd(grad_out, q, k, v, out, lse, seq_start, softmax_scale, grad_v_out=None):
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    o = O.varlen_causal_attention(qf, kf, vf, _seq_lens(seq_start), softmax_scale)
    gq, gk, gv = torch.autograd.grad(o, (qf, kf, vf), grad_out.float())
    if grad_v_out is not None:
        grad_v_out.copy_(gv.to(BF16))
    return gq.to(BF16), gk.to(BF16), gv.to(BF16)


def attn_varlen_fwd_rows(q, k, v, seq_start, softmax_scale, q_row0):
    """q rows [q_row0, q_row0 + n_q) of the pass against its K / V prefix: the full-prefix attention with zero queries in front, rows kept."""
    n = q_row0 + q.shape[0]
    qf = torch.zeros((n,) + tuple(q.shape[1:]))
    qf[q_row0:] = q.float()
    o = O.varlen_causal_attention(qf, k[:n].float(), v[:n].float(), _seq_lens(seq_start[:n]), softmax_scale)
    return o[q_row0:].to(BF16), torch.zeros(q.shape[1], q.shape[0])


@torch.enable_grad()
def attn_varlen_bwd_rows(grad_out, q, k, v, out, lse, seq_start, softmax_scale, q_row0):
    n = q_row0 + q.shape[0]
    ql = q.float().requires_grad_(True)
    kf, vf = k[:n].float().requires_grad_(True), v[:n].float().requires_grad_(True)
    qf = torch.cat([torch.zeros((q_row0,) + tuple(q.shape[1:])), ql])
    o = O.varlen_causal_attention(qf, kf, vf, _seq_lens(seq_start[:n]), softmax_scale)
    gq, gk, gv = torch.autograd.grad(o[q_row0:], (ql, kf, vf), grad_out.float())
    return gq.to(BF16), gk.to(BF16), gv.to(BF16)


def swiglu_fwd(gate_up, out=None):
    i = gate_up.shape[1] // 2
    act = (torch.nn.functional.silu(gate_up[:, :i].float()).to(BF16).float() * gate_up[:, i:].float()).to(BF16)
    if out is not None:
        out.copy_(act)
        return out
    return act


@torch.enable_grad()  # also callable from inside an autograd.Function.backward (grad mode is off there)
def swiglu_bwd(grad_act, gate_up):
    x = gate_up.float().requires_grad_(True)
    i = x.shape[1] // 2
    act = torch.nn.functional.silu(x[:, :i]) * x[:, i:]
    (g,) = torch.autograd.grad(act, x, grad_act.float())
    return g.to(BF16)


def logprob_entropy_fwd(logits, labels, inv_temp):
    l = logits.float() * inv_temp[:, None]
    logz = torch.logsumexp(l, -1)
    lp = l.gather(-1, labels[:, None])[:, 0] - logz
    ent = logz - (torch.softmax(l, -1) * l).sum(-1)
    return lp, ent, logz


def logprob_bwd(logits, labels, inv_temp, logz, grad_logprobs, out=None, out_dtype=None, vocab_start=0):
    l = logits.float() * inv_temp[:, None]
    p = torch.exp(l - logz[:, None])
    onehot = torch.zeros_like(p).scatter_(1, labels[:, None], 1.0)
    d = (grad_logprobs * inv_temp)[:, None] * (onehot - p)
    if out is None:
        return d.to(out_dtype or logits.dtype)
    out.copy_(d.to(out.dtype))
    return out


def lmhead_logprob_fwd(hidden, weight, labels, inv_temp, round_logits_bf16=True, logits_out=None):
    return logprob_entropy_fwd((hidden.float() @ weight.float().T).to(BF16), labels, inv_temp)


def lmhead_logprob_bwd(hidden, weight, labels, inv_temp, logz, grad_logprobs, round_logits_bf16=True, out=None):
    return logprob_bwd((hidden.float() @ weight.float().T).to(BF16), labels, inv_temp, logz, grad_logprobs, out_dtype=BF16)


def grpo_loss_fwd_bwd(trainer_logprobs, inference_logprobs, advantages, loss_mask, cu_samples, params, ref_logprobs=None, rl_weights=None,
                      ce_weights=None, ref_kl_weights=None, sample_starts_mb=None):
    """default_loss_fn over the packed pass with the in-kernel shift_tensor_right (K2 with shift=1); only the rl component, which is all
    the FakeDataLoader batches of the gloo test exercise.  Per-sample stats / metrics are zeros (not under test here)."""
    assert ref_logprobs is None and rl_weights is None and ce_weights is None and ref_kl_weights is None and params.shift == 1
    lp = trainer_logprobs.clone().requires_grad_(True)
    first = cu_samples[:-1][sample_starts_mb.bool()].long() if sample_starts_mb is not None else torch.zeros(1, dtype=torch.long)
    shifted = torch.roll(lp, 1)
    pad = torch.zeros_like(shifted)
    pad[first] = 1.0
    shifted = shifted * (1 - pad) + params.pad_logprob * pad
    lr = shifted - inference_logprobs
    ratio = torch.exp(lr)
    pd = torch.exp(shifted) - torch.exp(inference_logprobs)
    invalid = torch.where(advantages > 0, pd > params.dppo_mask_high, pd < -params.dppo_mask_low)
    keep = loss_mask & ~invalid
    loss = (-(keep * (params.adv_tau * advantages) * ratio) + params.kl_tau * loss_mask * lr**2).sum() / params.rl_scale
    (g,) = torch.autograd.grad(loss, lp)
    s = cu_samples.numel() - 1
    return loss.detach().reshape(1), g, torch.zeros(s, 20), torch.zeros(s, 12)


def transpose_2d(src, out=None):
    if out is None:
        return src.t().contiguous()
    out.copy_(src.t())
    return out


def grad_accumulate(main_grad_f32, grad_bf16):
    main_grad_f32 += grad_bf16.float()


def sumsq_accumulate(x, out_scalar):
    out_scalar += x.double().pow(2).sum().float()


def clip_coef(total_sumsq, grad_scale, max_norm):
    norm = total_sumsq.sqrt() * grad_scale
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0) if max_norm > 0 else torch.ones(1)
    return coef, norm


def adamw_step(master, exp_avg, exp_avg_sq, grad, param_bf16, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, clip=None, zero_grad=True):
    g = grad * grad_scale * (clip if clip is not None else 1.0)
    master.mul_(1 - lr * weight_decay)
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (exp_avg_sq / (1 - beta2**step)).sqrt_().add_(eps)
    master.addcdiv_(exp_avg / (1 - beta1**step), denom, value=-lr)
    if param_bf16 is not None:
        param_bf16.copy_(master.to(BF16))
    if zero_grad:
        grad.zero_()


# ------------------------------------------------------------------------------------------------ MoE stand-ins (tests/test_moe_ep_gloo_cpu.py)
def moe_router_fwd(gate_logits, top_k, score_func="softmax", route_norm=False, route_scale=1.0, expert_bias=None, forced_idx=None, want_counts=True):
    scores = torch.sigmoid(gate_logits.float()) if score_func == "sigmoid" else torch.softmax(gate_logits.float(), dim=1)
    if forced_idx is not None:
        idx = forced_idx.long()
    else:
        idx = torch.topk(scores if expert_bias is None else scores + expert_bias, top_k, dim=1).indices
    top = scores.gather(1, idx)
    conf = top.sum(-1) if score_func == "softmax" else top.sum(-1) / (scores.sum(-1) + 1e-20)
    if route_norm:
        top = top / (top.sum(-1, keepdim=True) + 1e-20)
    top = top * route_scale
    counts = torch.bincount(idx.reshape(-1), minlength=gate_logits.shape[1]).to(torch.int32)
    return scores, top, idx.to(torch.int32), counts, conf


def moe_router_bwd(scores, top_idx, grad_top, score_func, route_norm, route_scale):
    idx = top_idx.long()
    s = scores.gather(1, idx)
    g = grad_top * route_scale
    if route_norm:
        tot = s.sum(-1, keepdim=True) + 1e-20
        g = (g - (g * s).sum(-1, keepdim=True) / tot) / tot
    ds = torch.zeros_like(scores).scatter_add(1, idx, g)
    if score_func == "sigmoid":
        return scores * (1 - scores) * ds
    return scores * (ds - (s * g).sum(-1, keepdim=True))


def moe_sort(top_idx, n_experts):
    flat = top_idx.reshape(-1).long()
    order = torch.argsort(flat, stable=True)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel())
    counts = torch.bincount(flat, minlength=n_experts)
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
    return inv.to(torch.int32), order.to(torch.int32), offsets


def moe_gather_rows(x, row_map, divisor=1, scale=None):
    m = row_map.long()
    out = x[(m.clamp(min=0) // divisor)]
    if scale is not None:
        out = (out.float() * scale[m.clamp(min=0)][:, None]).to(BF16)
    return torch.where((m < 0)[:, None], torch.zeros_like(out), out)


def moe_combine(y, dest_of_slot, top_k, scores=None, base=None):
    n = dest_of_slot.numel() // top_k
    rows = y[dest_of_slot.long()].float()
    if scores is not None:
        rows = (rows * scores[:, None]).to(BF16).float()
    acc = rows.reshape(n, top_k, -1).sum(1)
    if base is not None:
        acc = acc + base.float()
    return acc.to(BF16)


def moe_slot_grads(a, b, slot_of_dest, top_k, scores, want_rows, want_dscore):
    slot = slot_of_dest.long()
    at = a[slot // top_k]
    rows = dscore = None
    if want_rows:
        rows = at if scores is None else (at.float() * scores[slot][:, None]).to(BF16)
    if want_dscore:
        dscore = torch.zeros(slot.numel())
        dscore[slot] = (at.float() * b.float()).sum(-1)
    return rows, dscore


def moe_ep_row_map(recv_counts, total_rows):
    ep, le = recv_counts.shape
    c = recv_counts.long()
    src_off = torch.cat([c.new_zeros(1), c.reshape(-1).cumsum(0)])[:-1].reshape(ep, le)
    row_map = []
    for j in range(le):
        for s in range(ep):
            row_map += list(range(int(src_off[s, j]), int(src_off[s, j]) + int(c[s, j])))
    row_map = torch.tensor(row_map, dtype=torch.int64)
    inv = torch.empty_like(row_map)
    inv[row_map] = torch.arange(row_map.numel())
    tot = c.sum(0)
    offs = torch.cat([tot.new_zeros(1), tot.cumsum(0)]).to(torch.int32)
    return row_map.to(torch.int32), inv.to(torch.int32), offs


def sgd_step(master, momentum_buf, grad, param_bf16, lr, momentum, weight_decay, nesterov, step, grad_scale=1.0, clip=None, zero_grad=True):
    g = grad * grad_scale * (clip[0] if clip is not None else 1.0)
    if weight_decay:
        g = g + weight_decay * master
    if momentum:
        momentum_buf.copy_(g if step == 1 else momentum * momentum_buf + g)
        g = g + momentum * momentum_buf if nesterov else momentum_buf
    master.sub_(lr * g)
    if param_bf16 is not None:
        param_bf16.copy_(master.to(BF16))
    if zero_grad:
        grad.zero_()


def sign_sgd_step(master, grad, param_bf16, lr, weight_decay, grad_scale=1.0, clip=None, zero_grad=True):
    if weight_decay > 0:
        master.add_(master, alpha=-lr * weight_decay)
    master.add_(torch.sign(grad), alpha=-lr)
    if param_bf16 is not None:
        param_bf16.copy_(master.to(BF16))
    if zero_grad:
        grad.zero_()


def gemm_supported(x, w, out=None, out_dtype=torch.bfloat16):
    return False  # the CPU stand-in engine keeps the library (torch) GEMM calls


def gemm_shape_supported(m, n, k, x_kmajor, w_kmajor):
    return False


def gemm_sumsq_slots(m, n):
    return ((m + 255) // 256) * ((n + 255) // 256) * 8


def sumsq_segments(base, segments, partials):
    for i, (o, ln) in enumerate(segments.tolist()):
        partials[i] = base[o:o + ln].float().pow(2).sum()


def sum_partials(partials, out_scalar):
    out_scalar += partials.sum()


def gemm(x, w, out=None, out_dtype=torch.bfloat16, bias=None, accumulate=False, tile_sumsq=None, variant=None):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if out is None:
        return y.to(out_dtype)
    if accumulate:
        out.add_(y.to(out.dtype))
    else:
        out.copy_(y)
    if tile_sumsq is not None:  # the stand-in keeps the whole sum in slot 0
        tile_sumsq.zero_()
        tile_sumsq[0] = out.float().pow(2).sum()
    return out
