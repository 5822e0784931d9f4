"""Tensor-level wrappers over the C ABI (include/prime_mi355x.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic step of the hot path is a
kernel in libprime_mi355x.so.  Each function validates dtypes/contiguity, passes raw device pointers + the current
stream, and raises `PrimeMI355XError` on any non-zero return.  No function here falls back to eager PyTorch.
"""

from __future__ import annotations

import ctypes

import torch
from torch import Tensor

from . import _lib
from ._lib import LossParams, PrimeMI355XError, check

_DT = {torch.float32: _lib.PM_DTYPE_F32, torch.bfloat16: _lib.PM_DTYPE_BF16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _req(t: Tensor, dtype: torch.dtype | None, name: str, contiguous: bool = True) -> Tensor:
    if not t.is_cuda:
        raise PrimeMI355XError(f"{name}: expected a GPU tensor (prime_rl_amd kernels run on MI355X only), got {t.device}")
    if dtype is not None and t.dtype != dtype:
        raise PrimeMI355XError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise PrimeMI355XError(f"{name}: expected a contiguous tensor")
    return t


_workspaces: dict[tuple, Tensor] = {}


def workspace(nbytes: int, device: torch.device, tag: str) -> Tensor:
    """Growable per-(device, stream, tag) scratch buffer; reuse is ordered by the stream."""
    key = (device.index, _stream(), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _rows2d(logits: Tensor, name: str) -> tuple[int, int, int]:
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise PrimeMI355XError(f"{name}: expected [rows, vocab] with unit stride on the vocab axis, got {tuple(logits.shape)} / {logits.stride()}")
    return logits.shape[0], logits.shape[1], logits.stride(0)


# ------------------------------------------------------------------------------------------------ K1
def logprob_entropy_fwd(logits: Tensor, labels: Tensor, inv_temp: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """logprob / entropy / logz per row of `logits` ([N, V] f32|bf16); replaces trainer/rl/loss.py:48-63 + train.py:441."""
    lib = _lib.load()
    _req(logits, None, "logits", contiguous=False)
    if logits.dtype not in _DT:
        raise PrimeMI355XError(f"logits dtype {logits.dtype} unsupported (f32 | bf16)")
    n, v, stride = _rows2d(logits, "logits")
    _req(labels, torch.int64, "labels")
    _req(inv_temp, torch.float32, "inv_temp")
    if labels.numel() != n or inv_temp.numel() != n:
        raise PrimeMI355XError("labels / inv_temp length must equal the number of logits rows")
    out = torch.empty((3, n), dtype=torch.float32, device=logits.device)
    if n == 0:  # empty micro-batch: nothing to launch
        return out[0], out[1], out[2]
    wsb = lib.pm_logprob_workspace_bytes(n, v)
    ws = workspace(wsb, logits.device, "logprob")
    check(
        lib.pm_logprob_entropy_fwd(_p(logits), _DT[logits.dtype], n, v, stride, _p(labels), _p(inv_temp), _p(out[0]), _p(out[1]), _p(out[2]),
                                   _p(ws), ws.numel(), _stream()),
        "pm_logprob_entropy_fwd",
    )
    return out[0], out[1], out[2]


def logprob_bwd(logits: Tensor, labels: Tensor, inv_temp: Tensor, logz: Tensor, grad_logprobs: Tensor,
                out: Tensor | None = None, out_dtype: torch.dtype | None = None, vocab_start: int = 0) -> Tensor:
    """dlogits = g*inv_T*(onehot - softmax); `out` may be `logits` itself (in place) when dtypes match."""
    lib = _lib.load()
    n, v, stride = _rows2d(_req(logits, None, "logits", contiguous=False), "logits")
    if out is None:
        out = torch.empty((n, v), dtype=out_dtype or logits.dtype, device=logits.device)
    on, ov, ostride = _rows2d(_req(out, None, "dlogits", contiguous=False), "dlogits")
    if (on, ov) != (n, v):
        raise PrimeMI355XError("dlogits shape mismatch")
    _req(labels, torch.int64, "labels"); _req(inv_temp, torch.float32, "inv_temp"); _req(logz, torch.float32, "logz")
    _req(grad_logprobs, torch.float32, "grad_logprobs")
    if vocab_start == 0:
        rc = lib.pm_logprob_bwd(_p(logits), _DT[logits.dtype], n, v, stride, _p(labels), _p(inv_temp), _p(logz), _p(grad_logprobs), _p(out),
                                _DT[out.dtype], ostride, _stream())
    else:
        rc = lib.pm_logprob_chunk_bwd(_p(logits), _DT[logits.dtype], n, v, stride, vocab_start, _p(labels), _p(inv_temp), _p(logz),
                                      _p(grad_logprobs), _p(out), _DT[out.dtype], ostride, _stream())
    check(rc, "pm_logprob_bwd")
    return out


class ChunkedLogprobState:
    """Online (m, s, t, target) state across vocab chunks (trainer/models/layers/lm_head.py:118-148)."""

    def __init__(self, n_rows: int, device: torch.device):
        self.n = n_rows
        self.buf = torch.empty((4, n_rows), dtype=torch.float32, device=device)
        lib = _lib.load()
        check(lib.pm_logprob_chunk_init(_p(self.buf[0]), _p(self.buf[1]), _p(self.buf[2]), _p(self.buf[3]), n_rows, _stream()), "pm_logprob_chunk_init")

    def update(self, logits_chunk: Tensor, vocab_start: int, labels: Tensor, inv_temp: Tensor) -> None:
        lib = _lib.load()
        n, v, stride = _rows2d(_req(logits_chunk, None, "logits_chunk", contiguous=False), "logits_chunk")
        b = self.buf
        check(
            lib.pm_logprob_chunk_update(_p(logits_chunk), _DT[logits_chunk.dtype], n, v, stride, vocab_start, _p(labels), _p(inv_temp), _p(b[0]),
                                        _p(b[1]), _p(b[2]), _p(b[3]), _stream()),
            "pm_logprob_chunk_update",
        )

    def finalize(self) -> tuple[Tensor, Tensor, Tensor]:
        lib = _lib.load()
        out = torch.empty((3, self.n), dtype=torch.float32, device=self.buf.device)
        b = self.buf
        check(lib.pm_logprob_chunk_finalize(_p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), self.n, _p(out[0]), _p(out[1]), _p(out[2]), _stream()),
              "pm_logprob_chunk_finalize")
        return out[0], out[1], out[2]


# ------------------------------------------------------------------------------------------------ J1
def lmhead_logprob_fwd(hidden: Tensor, weight: Tensor, labels: Tensor, inv_temp: Tensor, round_logits_bf16: bool = True,
                       logits_out: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor]:
    """Fused LM head forward (pm_lmhead_logprob_fwd): hidden [N, H] bf16, weight [V, H] bf16 -> (logprobs, entropy, logz) f32 [N];
    the logits only ever exist as MFMA accumulator tiles.  `logits_out` ([N, >=V] bf16, unit inner stride) optionally receives them."""
    lib = _lib.load()
    _req(hidden, torch.bfloat16, "hidden"); _req(weight, torch.bfloat16, "weight")
    _req(labels, torch.int64, "labels"); _req(inv_temp, torch.float32, "inv_temp")
    if hidden.dim() != 2 or weight.dim() != 2 or hidden.shape[1] != weight.shape[1]:
        raise PrimeMI355XError(f"lmhead: hidden {tuple(hidden.shape)} / weight {tuple(weight.shape)} mismatch")
    n, h = hidden.shape
    v = weight.shape[0]
    if labels.numel() != n or inv_temp.numel() != n:
        raise PrimeMI355XError("labels / inv_temp length must equal the number of hidden rows")
    out = torch.empty((3, n), dtype=torch.float32, device=hidden.device)
    if n == 0:
        return out[0], out[1], out[2]
    ld = 0
    if logits_out is not None:
        _req(logits_out, torch.bfloat16, "logits_out", contiguous=False)
        if logits_out.dim() != 2 or logits_out.shape[0] != n or logits_out.shape[1] != v or logits_out.stride(1) != 1:
            raise PrimeMI355XError("logits_out must be [N, V] bf16 with unit inner stride")
        ld = logits_out.stride(0)
    ws = workspace(lib.pm_lmhead_workspace_bytes(n, v), hidden.device, "lmhead")
    check(lib.pm_lmhead_logprob_fwd(_p(hidden), _p(weight), n, h, v, _p(labels), _p(inv_temp), 1 if round_logits_bf16 else 0, _p(out[0]), _p(out[1]),
                                    _p(out[2]), _p(logits_out), ld, _p(ws), ws.numel(), _stream()), "pm_lmhead_logprob_fwd")
    return out[0], out[1], out[2]


def lmhead_logprob_bwd(hidden: Tensor, weight: Tensor, labels: Tensor, inv_temp: Tensor, logz: Tensor, grad_logprobs: Tensor,
                       round_logits_bf16: bool = True, out: Tensor | None = None) -> Tensor:
    """dlogits [N, V] bf16 = g * inv_T * (onehot(label) - softmax), logits recomputed on the MFMA units (pm_lmhead_logprob_bwd)."""
    lib = _lib.load()
    _req(hidden, torch.bfloat16, "hidden"); _req(weight, torch.bfloat16, "weight")
    for t, nm in ((labels, "labels"),):
        _req(t, torch.int64, nm)
    for t, nm in ((inv_temp, "inv_temp"), (logz, "logz"), (grad_logprobs, "grad_logprobs")):
        _req(t, torch.float32, nm)
    n, h = hidden.shape
    v = weight.shape[0]
    if out is None:
        out = torch.empty((n, (v + 7) // 8 * 8), dtype=torch.bfloat16, device=hidden.device)[:, :v]
    _req(out, torch.bfloat16, "dlogits", contiguous=False)
    if out.shape != (n, v) or out.stride(1) != 1:
        raise PrimeMI355XError("dlogits must be [N, V] bf16 with unit inner stride")
    if n:
        check(lib.pm_lmhead_logprob_bwd(_p(hidden), _p(weight), n, h, v, _p(labels), _p(inv_temp), _p(logz), _p(grad_logprobs),
                                        1 if round_logits_bf16 else 0, _p(out), out.stride(0), _stream()), "pm_lmhead_logprob_bwd")
    return out


# ------------------------------------------------------------------------------------------------ K3
def grpo_advantage(rewards: Tensor, group_offsets: Tensor, mode: int = _lib.PM_ADV_GRPO, num_output_tokens: Tensor | None = None,
                   num_total_tokens: Tensor | None = None, num_turns: Tensor | None = None,
                   weights: tuple[float, float, float] = (0.0, 0.0, 0.0)) -> Tensor:
    lib = _lib.load()
    _req(rewards, torch.float32, "rewards"); _req(group_offsets, torch.int32, "group_offsets")
    for t, nm in ((num_output_tokens, "num_output_tokens"), (num_total_tokens, "num_total_tokens"), (num_turns, "num_turns")):
        if t is not None:
            _req(t, torch.float32, nm)
    adv = torch.empty_like(rewards)
    if rewards.numel() == 0 or group_offsets.numel() < 2:
        return adv
    check(
        lib.pm_grpo_advantage(_p(rewards), _p(group_offsets), group_offsets.numel() - 1, mode, _p(num_output_tokens), _p(num_total_tokens),
                              _p(num_turns), weights[0], weights[1], weights[2], _p(adv), _stream()),
        "pm_grpo_advantage",
    )
    return adv


def advantage_broadcast(rollout_adv: Tensor, rollout_offsets: Tensor, token_mask: Tensor) -> Tensor:
    lib = _lib.load()
    _req(rollout_adv, torch.float32, "rollout_adv"); _req(rollout_offsets, torch.int32, "rollout_offsets")
    mask_u8 = _req(token_mask, None, "token_mask")
    if mask_u8.dtype == torch.bool:
        mask_u8 = mask_u8.view(torch.uint8)
    out = torch.empty(mask_u8.numel(), dtype=torch.float32, device=rollout_adv.device)
    check(lib.pm_advantage_broadcast(_p(rollout_adv), _p(rollout_offsets), rollout_adv.numel(), _p(mask_u8), _p(out), out.numel(), _stream()),
          "pm_advantage_broadcast")
    return out


# ------------------------------------------------------------------------------------------------ K2
def grpo_loss_fwd_bwd(trainer_logprobs: Tensor, inference_logprobs: Tensor, advantages: Tensor, loss_mask: Tensor, cu_samples: Tensor,
                      params: LossParams, ref_logprobs: Tensor | None = None, rl_weights: Tensor | None = None,
                      ce_weights: Tensor | None = None, ref_kl_weights: Tensor | None = None,
                      sample_starts_mb: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """Returns (loss[1], grad_logprobs[N], sample_stats[S,20], sample_metrics[S,12]); see pm_grpo_loss_fwd_bwd."""
    lib = _lib.load()
    n = trainer_logprobs.numel()
    for t, nm in ((trainer_logprobs, "trainer_logprobs"), (inference_logprobs, "inference_logprobs"), (advantages, "advantages")):
        _req(t, torch.float32, nm)
        if t.numel() != n:
            raise PrimeMI355XError(f"{nm}: length {t.numel()} != {n}")
    for t, nm in ((ref_logprobs, "ref_logprobs"), (rl_weights, "rl_weights"), (ce_weights, "ce_weights"), (ref_kl_weights, "ref_kl_weights")):
        if t is not None:
            _req(t, torch.float32, nm)
            if t.numel() != n:
                raise PrimeMI355XError(f"{nm}: length {t.numel()} != {n}")
    mask_u8 = _req(loss_mask, None, "loss_mask")
    if mask_u8.dtype == torch.bool:
        mask_u8 = mask_u8.view(torch.uint8)
    elif mask_u8.dtype != torch.uint8:
        raise PrimeMI355XError("loss_mask must be bool or uint8")
    _req(cu_samples, torch.int32, "cu_samples")
    s = cu_samples.numel() - 1
    if sample_starts_mb is not None:
        _req(sample_starts_mb, torch.uint8, "sample_starts_mb")
    dev = trainer_logprobs.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty(n, dtype=torch.float32, device=dev)
    stats = torch.empty((s, _lib.PM_LOSS_NSTATS), dtype=torch.float32, device=dev)
    metrics = torch.empty((s, _lib.PM_LOSS_NMETRICS), dtype=torch.float32, device=dev)
    check(
        lib.pm_grpo_loss_fwd_bwd(_p(trainer_logprobs), _p(inference_logprobs), _p(ref_logprobs), _p(advantages), _p(mask_u8), _p(rl_weights),
                                 _p(ce_weights), _p(ref_kl_weights), _p(cu_samples), _p(sample_starts_mb), s, n, ctypes.byref(params), _p(loss),
                                 _p(grad), _p(stats), _p(metrics), _stream()),
        "pm_grpo_loss_fwd_bwd",
    )
    return loss, grad, stats, metrics


# ------------------------------------------------------------------------------------------------ model ops
def embedding_fwd(table: Tensor, ids: Tensor) -> Tensor:
    lib = _lib.load()
    _req(table, torch.bfloat16, "embedding table"); _req(ids, torch.int64, "ids")
    n, (v, h) = ids.numel(), table.shape
    out = torch.empty((n, h), dtype=torch.bfloat16, device=table.device)
    check(lib.pm_embedding_fwd(_p(table), _p(ids), n, h, v, _p(out), _stream()), "pm_embedding_fwd")
    return out


def embedding_bwd(grad_out: Tensor, ids: Tensor, grad_table_f32: Tensor) -> None:
    """grad_table_f32[id] += rows of grad_out; deterministic (sorted run reduction)."""
    lib = _lib.load()
    _req(grad_out, torch.bfloat16, "grad_out"); _req(ids, torch.int64, "ids"); _req(grad_table_f32, torch.float32, "grad_table")
    ids_sorted, order = torch.sort(ids.reshape(-1), stable=True)  # plumbing: index sort only
    n, h = grad_out.shape
    check(lib.pm_embedding_bwd(_p(grad_out), _p(ids_sorted), _p(order), n, h, grad_table_f32.shape[0], _p(grad_table_f32), _stream()), "pm_embedding_bwd")


def embedding_bwd_window(grad_out: Tensor, ids: Tensor, grad_window_f32: Tensor, elem_lo: int, vocab: int) -> None:
    """grad_window_f32 (elements [elem_lo, elem_lo + numel) of the flattened [vocab, hidden] gradient) += rows of grad_out whose id falls in the
    window; ids < 0 are padding.  Deterministic (sorted run reduction, rows summed in the order they appear in `ids`)."""
    lib = _lib.load()
    _req(grad_out, torch.bfloat16, "grad_out"); _req(ids, torch.int64, "ids"); _req(grad_window_f32, torch.float32, "grad_window")
    ids_sorted, order = torch.sort(ids.reshape(-1), stable=True)  # plumbing: index sort only
    n, h = grad_out.shape
    check(lib.pm_embedding_bwd_window(_p(grad_out), _p(ids_sorted), _p(order), n, h, vocab, elem_lo, grad_window_f32.numel(), _p(grad_window_f32),
                                      _stream()), "pm_embedding_bwd_window")


def rmsnorm_fwd(x: Tensor, weight: Tensor, eps: float, residual: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor | None]:
    """Returns (y, rstd, x_plus_residual | None)."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x"); _req(weight, torch.bfloat16, "weight")
    h = x.shape[-1]
    n = x.numel() // h
    y = torch.empty_like(x)
    rstd = torch.empty(n, dtype=torch.float32, device=x.device)
    so = None
    if residual is not None:
        _req(residual, torch.bfloat16, "residual")
        so = torch.empty_like(x)
    check(lib.pm_rmsnorm_fwd(_p(x), _p(residual), _p(weight), n, h, eps, _p(y), _p(so), _p(rstd), _stream()), "pm_rmsnorm_fwd")
    return y, rstd, so


def rmsnorm_bwd(grad_y: Tensor, x: Tensor, weight: Tensor, rstd: Tensor, grad_weight_f32: Tensor, dresidual: Tensor | None = None) -> Tensor:
    lib = _lib.load()
    _req(grad_y, torch.bfloat16, "grad_y"); _req(x, torch.bfloat16, "x"); _req(weight, torch.bfloat16, "weight")
    _req(rstd, torch.float32, "rstd"); _req(grad_weight_f32, torch.float32, "grad_weight")
    if dresidual is not None:
        _req(dresidual, torch.bfloat16, "dresidual")
    h = x.shape[-1]
    n = x.numel() // h
    gx = torch.empty_like(x)
    wsb = lib.pm_rmsnorm_bwd_workspace_bytes(n, h)
    ws = workspace(wsb, x.device, "rmsnorm_bwd")
    check(lib.pm_rmsnorm_bwd(_p(grad_y), _p(x), _p(weight), _p(rstd), _p(dresidual), n, h, _p(gx), _p(grad_weight_f32), _p(ws), ws.numel(), _stream()),
          "pm_rmsnorm_bwd")
    return gx


def rope_inv_freq(head_dim: int, theta: float, device) -> Tensor:
    """inv_freq exactly as _compute_default_rope_parameters builds it (trainer/models/layers/rotary_emb.py:24): host f32, once per model."""
    return (1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))).to(device)


def rope_table(position_ids: Tensor, inv_freq: Tensor) -> tuple[Tensor, Tensor]:
    lib = _lib.load()
    _req(position_ids, torch.int64, "position_ids"); _req(inv_freq, torch.float32, "inv_freq")
    n, half = position_ids.numel(), inv_freq.numel()
    cs = torch.empty((2, n, half), dtype=torch.bfloat16, device=position_ids.device)
    check(lib.pm_rope_table(_p(position_ids), n, _p(inv_freq), half, _p(cs[0]), _p(cs[1]), _stream()), "pm_rope_table")
    return cs[0], cs[1]


def _check_qknorm_operands(q_norm_w: Tensor, k_norm_w: Tensor, cos: Tensor, sin: Tensor, n: int, head_dim: int) -> None:
    """The kernels read these through raw bf16 pointers: an f32 cos/sin table (a model that is not bf16) or a mis-shaped norm weight
    would be read as garbage without any error."""
    if (q_norm_w is None) != (k_norm_w is None):
        raise PrimeMI355XError("q_norm_w and k_norm_w go together (both None = use_qk_norm False: RoPE only)")
    for t, nm in ((q_norm_w, "q_norm_w"), (k_norm_w, "k_norm_w")):
        if t is None:
            continue
        _req(t, torch.bfloat16, nm)
        if t.numel() != head_dim:
            raise PrimeMI355XError(f"{nm}: expected {head_dim} elements, got {t.numel()}")
    for t, nm in ((cos, "cos"), (sin, "sin")):
        _req(t, torch.bfloat16, nm)
        if t.numel() != n * (head_dim // 2):
            raise PrimeMI355XError(f"{nm}: expected a [{n}, {head_dim // 2}] bf16 half-table (pm_rope_table layout), got {tuple(t.shape)}")


def qknorm_rope_fwd(qkv: Tensor, q_norm_w: Tensor | None, k_norm_w: Tensor | None, cos: Tensor, sin: Tensor, n_q_heads: int, n_kv_heads: int,
                    head_dim: int, eps: float) -> tuple[Tensor, Tensor, Tensor | None]:
    """(q, k, rstd).  q_norm_w = k_norm_w = None: no per-head norm (Llama / Qwen2 attention) — RoPE only, rstd is None."""
    lib = _lib.load()
    _req(qkv, torch.bfloat16, "qkv")
    n = qkv.shape[0]
    if qkv.shape[1] != (n_q_heads + 2 * n_kv_heads) * head_dim:
        raise PrimeMI355XError("qkv: wrong width")
    _check_qknorm_operands(q_norm_w, k_norm_w, cos, sin, n, head_dim)
    q = torch.empty((n, n_q_heads, head_dim), dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty((n, n_kv_heads, head_dim), dtype=torch.bfloat16, device=qkv.device)
    rstd = None if q_norm_w is None else torch.empty((n, n_q_heads + n_kv_heads), dtype=torch.float32, device=qkv.device)
    check(
        lib.pm_qknorm_rope_fwd(_p(qkv), _p(q_norm_w), _p(k_norm_w), _p(cos), _p(sin), n, n_q_heads, n_kv_heads, head_dim, eps, _p(q), _p(k),
                               _p(rstd), _stream()),
        "pm_qknorm_rope_fwd",
    )
    return q, k, rstd


def qknorm_rope_bwd(grad_q: Tensor, grad_k: Tensor, qkv: Tensor | None, q_norm_w: Tensor | None, k_norm_w: Tensor | None, cos: Tensor, sin: Tensor,
                    rstd: Tensor | None, grad_qkv: Tensor, grad_q_norm_w_f32: Tensor | None, grad_k_norm_w_f32: Tensor | None, n_q_heads: int,
                    n_kv_heads: int, head_dim: int) -> None:
    """Writes the q|k columns of grad_qkv (the v columns are written by attention backward).  Without norm weights (RoPE only) qkv, rstd
    and the weight gradients are not used."""
    lib = _lib.load()
    for t, nm in ((grad_q, "grad_q"), (grad_k, "grad_k"), (grad_qkv, "grad_qkv")):
        _req(t, torch.bfloat16, nm)
    n = grad_qkv.shape[0]
    if grad_qkv.shape[1] != (n_q_heads + 2 * n_kv_heads) * head_dim or grad_q.numel() != n * n_q_heads * head_dim or grad_k.numel() != n * n_kv_heads * head_dim:
        raise PrimeMI355XError("qknorm_rope_bwd: grad_q / grad_k / grad_qkv shapes do not agree")
    _check_qknorm_operands(q_norm_w, k_norm_w, cos, sin, n, head_dim)
    if q_norm_w is None:
        check(lib.pm_qknorm_rope_bwd(_p(grad_q), _p(grad_k), None, None, None, _p(cos), _p(sin), None, n, n_q_heads, n_kv_heads, head_dim, _p(grad_qkv),
                                     None, None, None, 0, _stream()), "pm_qknorm_rope_bwd")
        return
    _req(qkv, torch.bfloat16, "qkv")
    _req(rstd, torch.float32, "rstd")
    wsb = lib.pm_qknorm_rope_bwd_workspace_bytes(n, n_q_heads, n_kv_heads)
    ws = workspace(wsb, qkv.device, "qknorm_bwd")
    check(
        lib.pm_qknorm_rope_bwd(_p(grad_q), _p(grad_k), _p(qkv), _p(q_norm_w), _p(k_norm_w), _p(cos), _p(sin), _p(rstd), n, n_q_heads, n_kv_heads,
                               head_dim, _p(grad_qkv), _p(grad_q_norm_w_f32), _p(grad_k_norm_w_f32), _p(ws), ws.numel(), _stream()),
        "pm_qknorm_rope_bwd",
    )


def column_sum(x: Tensor, out_f32: Tensor) -> None:
    """out_f32[c] += sum_r x[r, c] for a bf16 matrix whose rows may be strided (a column block of a wider buffer): the bias gradient of a
    linear layer.  Deterministic."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x", contiguous=False); _req(out_f32, torch.float32, "out")
    if x.dim() != 2 or x.stride(1) != 1 or out_f32.numel() != x.shape[1]:
        raise PrimeMI355XError(f"column_sum: x must be [rows, cols] with unit column stride and out must hold cols floats, got {tuple(x.shape)} / {tuple(out_f32.shape)}")
    n, c = x.shape
    wsb = lib.pm_column_sum_workspace_bytes(c)
    ws = workspace(wsb, x.device, "column_sum")
    check(lib.pm_column_sum_bf16(_p(x), n, c, x.stride(0) if n > 1 else c, _p(out_f32), _p(ws), ws.numel(), _stream()), "pm_column_sum_bf16")


def seq_start_from_cu(cu_seqlens: Tensor, n_tokens: int) -> Tensor:
    lib = _lib.load()
    _req(cu_seqlens, torch.int32, "cu_seqlens")
    out = torch.empty(n_tokens, dtype=torch.int32, device=cu_seqlens.device)
    check(lib.pm_seq_start_from_cu(_p(cu_seqlens), cu_seqlens.numel() - 1, n_tokens, _p(out), _stream()), "pm_seq_start_from_cu")
    return out


def seq_start_with_window(seq_start: Tensor, window: int) -> Tensor:
    """First visible key per query row under a sliding window of `window` keys (the reference passes `window_size=(sliding_window - 1, 0)` to
    flash-attn, models/layers/attn.py:94-100): max(document start, row - window + 1).  The attention kernels only ever read `seq_start[row]` as
    "lowest visible key of this row" (monotone in the row), so the windowed mask needs no kernel of its own — hand them this array instead."""
    _req(seq_start, torch.int32, "seq_start")
    if window < 1:
        raise PrimeMI355XError(f"sliding window must be >= 1, got {window}")
    rows = torch.arange(seq_start.numel(), dtype=torch.int32, device=seq_start.device)
    return torch.maximum(seq_start, rows - (window - 1)).contiguous()


def attn_varlen_fwd(q: Tensor, k: Tensor, v: Tensor, seq_start: Tensor, softmax_scale: float) -> tuple[Tensor, Tensor]:
    """q [N,Hq,128], k [N,Hkv,128] contiguous; v [N,Hkv,128] may be a row-strided view (e.g. the v slice of qkv)."""
    lib = _lib.load()
    _req(q, torch.bfloat16, "q"); _req(k, torch.bfloat16, "k"); _req(v, torch.bfloat16, "v", contiguous=False)
    _req(seq_start, torch.int32, "seq_start")
    n, hq, d = q.shape
    hkv = k.shape[1]
    if v.stride(2) != 1 or v.stride(1) != d:
        raise PrimeMI355XError("v: heads must be contiguous within a row")
    out = torch.empty_like(q)
    lse = torch.empty((hq, n), dtype=torch.float32, device=q.device)
    wsb = lib.pm_attn_fwd_workspace_bytes(n, hkv, d)
    ws = workspace(wsb, q.device, "attn_fwd")
    check(
        lib.pm_attn_varlen_fwd(_p(q), _p(k), _p(v), v.stride(0), _p(seq_start), n, hq, hkv, d, softmax_scale, _p(out), _p(lse), _p(ws), ws.numel(),
                               _stream()),
        "pm_attn_varlen_fwd",
    )
    return out, lse


def attn_varlen_bwd(grad_out: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor, seq_start: Tensor, softmax_scale: float,
                    grad_v_out: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor]:
    lib = _lib.load()
    for t, nm in ((grad_out, "grad_out"), (q, "q"), (k, "k"), (out, "out")):
        _req(t, torch.bfloat16, nm)
    _req(v, torch.bfloat16, "v", contiguous=False)
    n, hq, d = q.shape
    hkv = k.shape[1]
    gq = torch.empty_like(q)
    gk = torch.empty_like(k)
    gv = grad_v_out if grad_v_out is not None else torch.empty((n, hkv, d), dtype=torch.bfloat16, device=q.device)
    if gv.stride(2) != 1 or gv.stride(1) != d:
        raise PrimeMI355XError("grad_v: heads must be contiguous within a row")
    wsb = lib.pm_attn_bwd_workspace_bytes(n, hq, hkv, d)
    ws = workspace(wsb, q.device, "attn_bwd")
    check(
        lib.pm_attn_varlen_bwd(_p(grad_out), _p(q), _p(k), _p(v), v.stride(0), _p(out), _p(lse), _p(seq_start), n, hq, hkv, d, softmax_scale,
                               _p(gq), _p(gk), _p(gv), gv.stride(0), _p(ws), ws.numel(), _stream()),
        "pm_attn_varlen_bwd",
    )
    return gq, gk, gv


def attn_varlen_fwd_rows(q: Tensor, k: Tensor, v: Tensor, seq_start: Tensor, softmax_scale: float, q_row0: int) -> tuple[Tensor, Tensor]:
    """The q rows [q_row0, q_row0 + n_q) of a pass against its K / V rows [0, q_row0 + n_q): q [n_q,Hq,128]; k, v hold at least q_row0 + n_q rows
    (v may be row-strided), seq_start the document starts of those rows (pass-global indices).  Returns (out [n_q,Hq,128], lse [Hq,n_q])."""
    lib = _lib.load()
    _req(q, torch.bfloat16, "q"); _req(k, torch.bfloat16, "k"); _req(v, torch.bfloat16, "v", contiguous=False)
    _req(seq_start, torch.int32, "seq_start")
    nq, hq, d = q.shape
    hkv = k.shape[1]
    n = q_row0 + nq
    if k.shape[0] < n or v.shape[0] < n or seq_start.numel() < n:
        raise PrimeMI355XError(f"attn_varlen_fwd_rows: k / v / seq_start must hold the {n} rows up to the end of the q window")
    if v.stride(2) != 1 or v.stride(1) != d:
        raise PrimeMI355XError("v: heads must be contiguous within a row")
    out = torch.empty_like(q)
    lse = torch.empty((hq, nq), dtype=torch.float32, device=q.device)
    check(lib.pm_attn_varlen_fwd_rows(_p(q), _p(k), _p(v), v.stride(0), _p(seq_start), q_row0, nq, hq, hkv, d, softmax_scale, _p(out), _p(lse), _stream()),
          "pm_attn_varlen_fwd_rows")
    return out, lse


def attn_varlen_bwd_rows(grad_out: Tensor, q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor, seq_start: Tensor, softmax_scale: float,
                         q_row0: int) -> tuple[Tensor, Tensor, Tensor]:
    """Backward of attn_varlen_fwd_rows: (grad_q [n_q,...], grad_k, grad_v [q_row0 + n_q, Hkv, 128]) — grad_k / grad_v are this q window's
    contribution to every kv row of the pass prefix (the caller sums them over the ranks that hold the other windows)."""
    lib = _lib.load()
    for t, nm in ((grad_out, "grad_out"), (q, "q"), (k, "k"), (out, "out")):
        _req(t, torch.bfloat16, nm)
    _req(v, torch.bfloat16, "v", contiguous=False)
    nq, hq, d = q.shape
    hkv = k.shape[1]
    n = q_row0 + nq
    if k.shape[0] < n or v.shape[0] < n or seq_start.numel() < n:
        raise PrimeMI355XError(f"attn_varlen_bwd_rows: k / v / seq_start must hold the {n} rows up to the end of the q window")
    gq = torch.empty_like(q)
    gk = torch.empty((n, hkv, d), dtype=torch.bfloat16, device=q.device)
    gv = torch.empty((n, hkv, d), dtype=torch.bfloat16, device=q.device)
    wsb = lib.pm_attn_bwd_workspace_bytes(nq, hq, hkv, d)
    ws = workspace(wsb, q.device, "attn_bwd")
    check(lib.pm_attn_varlen_bwd_rows(_p(grad_out), _p(q), _p(k), _p(v), v.stride(0), _p(out), _p(lse), _p(seq_start), q_row0, nq, hq, hkv, d, softmax_scale,
                                      _p(gq), _p(gk), _p(gv), gv.stride(0), _p(ws), ws.numel(), _stream()), "pm_attn_varlen_bwd_rows")
    return gq, gk, gv


def transpose_2d(src: Tensor, out: Tensor | None = None) -> Tensor:
    """out[c, r] = src[r, c] for a row-major bf16/f16 matrix (rows may be strided: `src.stride(1) == 1`)."""
    lib = _lib.load()
    if src.dtype not in (torch.bfloat16, torch.float16) or src.dim() != 2 or src.stride(1) != 1 or not src.is_cuda:
        raise ValueError(f"transpose_2d: need a 2-D 16-bit GPU tensor with unit inner stride, got {src.dtype} {tuple(src.shape)} strides {src.stride()}")
    rows, cols = src.shape
    if out is None:  # leading dimension rounded up to 8 elements so every output row starts 16-byte aligned for any `rows`
        out = torch.empty((cols, (rows + 7) // 8 * 8), dtype=src.dtype, device=src.device)[:, :rows]
    elif out.shape != (cols, rows) or out.dtype != src.dtype or out.stride(1) != 1:
        raise ValueError("transpose_2d: out must be [cols, rows] of the same dtype with unit inner stride")
    if rows == 0 or cols == 0:
        return out
    check(lib.pm_transpose_2d(_p(src), rows, cols, src.stride(0), _p(out), out.stride(0), _stream()), "pm_transpose_2d")
    return out


def swiglu_fwd(gate_up: Tensor, out: Tensor | None = None) -> Tensor:
    lib = _lib.load()
    _req(gate_up, torch.bfloat16, "gate_up")
    n, two_i = gate_up.shape
    if out is not None:
        _req(out, torch.bfloat16, "out")
        if tuple(out.shape) != (n, two_i // 2):
            raise PrimeMI355XError(f"swiglu_fwd: out must be [{n}, {two_i // 2}]")
    act = out if out is not None else torch.empty((n, two_i // 2), dtype=torch.bfloat16, device=gate_up.device)
    check(lib.pm_swiglu_fwd(_p(gate_up), n, two_i // 2, _p(act), _stream()), "pm_swiglu_fwd")
    return act


def swiglu_bwd(grad_act: Tensor, gate_up: Tensor, out: Tensor | None = None) -> Tensor:
    lib = _lib.load()
    _req(grad_act, torch.bfloat16, "grad_act"); _req(gate_up, torch.bfloat16, "gate_up")
    n, two_i = gate_up.shape
    g = torch.empty_like(gate_up) if out is None else out
    if out is not None:
        _req(out, torch.bfloat16, "out")
        if tuple(out.shape) != tuple(gate_up.shape):
            raise PrimeMI355XError(f"swiglu_bwd: out {tuple(out.shape)} != gate_up {tuple(gate_up.shape)}")
    check(lib.pm_swiglu_bwd(_p(grad_act), _p(gate_up), n, two_i // 2, _p(g), _stream()), "pm_swiglu_bwd")
    return g


# ------------------------------------------------------------------------------------------------ optimizer
def grad_accumulate(main_grad_f32: Tensor, grad_bf16: Tensor) -> None:
    lib = _lib.load()
    _req(main_grad_f32, torch.float32, "main_grad"); _req(grad_bf16, torch.bfloat16, "grad")
    check(lib.pm_grad_accumulate(_p(main_grad_f32), _p(grad_bf16), main_grad_f32.numel(), _stream()), "pm_grad_accumulate")


def sumsq_accumulate(x: Tensor, out_scalar: Tensor) -> None:
    """out_scalar[0] += sum(x^2)."""
    lib = _lib.load()
    _req(x, torch.float32, "x"); _req(out_scalar, torch.float32, "out")
    ws = workspace(lib.pm_sumsq_workspace_bytes(x.numel()), x.device, "sumsq")
    check(lib.pm_sumsq(_p(x), x.numel(), _p(out_scalar), _p(ws), ws.numel(), _stream()), "pm_sumsq")


def clip_coef(total_sumsq: Tensor, grad_scale: float, max_norm: float) -> tuple[Tensor, Tensor]:
    lib = _lib.load()
    out = torch.empty(2, dtype=torch.float32, device=total_sumsq.device)
    check(lib.pm_clip_coef(_p(total_sumsq), grad_scale, max_norm, _p(out[0:1]), _p(out[1:2]), _stream()), "pm_clip_coef")
    return out[0:1], out[1:2]


def adamw_step(master: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, grad: Tensor, param_bf16: Tensor | None, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int, grad_scale: float = 1.0, clip: Tensor | None = None, zero_grad: bool = True) -> None:
    lib = _lib.load()
    for t, nm in ((master, "master"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (grad, "grad")):
        _req(t, torch.float32, nm)
    if param_bf16 is not None:
        _req(param_bf16, torch.bfloat16, "param_bf16")
    check(
        lib.pm_adamw_step(_p(master), _p(exp_avg), _p(exp_avg_sq), _p(grad), _p(param_bf16), master.numel(), lr, beta1, beta2, eps, weight_decay,
                          step, grad_scale, _p(clip), 1 if zero_grad else 0, _stream()),
        "pm_adamw_step",
    )


def ds_read_tr16_probe(elem_index: Tensor) -> Tensor:
    """int32[64] element indices (multiples of 4) -> int32[64, 4] values returned by `ds_read_b64_tr_b16` over the image lds[i] = i."""
    lib = _lib.load()
    _req(elem_index, torch.int32, "elem_index")
    out = torch.empty((64, 4), dtype=torch.int16, device=elem_index.device)
    check(lib.pm_ds_read_tr16_probe(_p(elem_index), _p(out), _stream()), "pm_ds_read_tr16_probe")
    return out.to(torch.int32) & 0xFFFF


def mfma_probe(a: Tensor, b: Tensor) -> Tensor:
    lib = _lib.load()
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
    c = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    check(lib.pm_mfma_probe_16x16x32(_p(a), _p(b), _p(c), _stream()), "pm_mfma_probe_16x16x32")
    return c


def device_arch() -> str:
    lib = _lib.load()
    buf = ctypes.create_string_buffer(128)
    check(lib.pm_device_arch(buf, 128), "pm_device_arch")
    return buf.value.decode()


# ------------------------------------------------------------------------------------------------ MoE (row f4)
def moe_router_fwd(gate_logits: Tensor, top_k: int, score_func: str = "softmax", route_norm: bool = False, route_scale: float = 1.0,
                   expert_bias: Tensor | None = None, forced_idx: Tensor | None = None, want_counts: bool = True):
    """TokenChoiceTopKRouter.forward after the gate GEMM (pm_moe_router_fwd).  gate_logits [N, E] bf16 | f32 ->
    (scores [N,E] f32, top_scores [N,k] f32, top_idx [N,k] i32, counts [E] i32 | None, confidence [N] f32).
    `want_counts=False` skips the in-kernel histogram (contended atomics); the counts are then the differences of `moe_sort`'s offsets."""
    lib = _lib.load()
    _req(gate_logits, None, "gate_logits")
    if gate_logits.dtype not in (torch.bfloat16, torch.float32) or gate_logits.dim() != 2:
        raise PrimeMI355XError(f"gate_logits must be [N, E] bf16 or f32, got {gate_logits.dtype} {tuple(gate_logits.shape)}")
    if score_func not in ("softmax", "sigmoid"):
        raise PrimeMI355XError(f"unknown score function {score_func!r}")
    n, e = gate_logits.shape
    dev = gate_logits.device
    if expert_bias is not None:
        _req(expert_bias, torch.float32, "expert_bias")
    if forced_idx is not None:
        _req(forced_idx, torch.int32, "forced_idx")
        if tuple(forced_idx.shape) != (n, top_k):
            raise PrimeMI355XError(f"forced_idx must be [N, top_k], got {tuple(forced_idx.shape)}")
    scores = torch.empty((n, e), dtype=torch.float32, device=dev)
    top_scores = torch.empty((n, top_k), dtype=torch.float32, device=dev)
    top_idx = torch.empty((n, top_k), dtype=torch.int32, device=dev)
    counts = torch.empty(e, dtype=torch.int32, device=dev) if want_counts else None
    conf = torch.empty(n, dtype=torch.float32, device=dev)
    dt = _lib.PM_DTYPE_BF16 if gate_logits.dtype == torch.bfloat16 else _lib.PM_DTYPE_F32
    check(lib.pm_moe_router_fwd(_p(gate_logits), dt, _p(expert_bias), _p(forced_idx), n, e, top_k, 1 if score_func == "sigmoid" else 0,
                                1 if route_norm else 0, float(route_scale), _p(scores), _p(top_scores), _p(top_idx), _p(counts), _p(conf), _stream()),
          "pm_moe_router_fwd")
    return scores, top_scores, top_idx, counts, conf


def moe_router_bwd(scores: Tensor, top_idx: Tensor, grad_top_scores: Tensor, score_func: str, route_norm: bool, route_scale: float) -> Tensor:
    """d top_scores [N,k] -> d gate_logits [N,E] f32 (pm_moe_router_bwd)."""
    lib = _lib.load()
    _req(scores, torch.float32, "scores"); _req(top_idx, torch.int32, "top_idx"); _req(grad_top_scores, torch.float32, "grad_top_scores")
    n, e = scores.shape
    k = top_idx.shape[1]
    out = torch.empty_like(scores)
    check(lib.pm_moe_router_bwd(_p(scores), _p(top_idx), _p(grad_top_scores), n, e, k, 1 if score_func == "sigmoid" else 0, 1 if route_norm else 0,
                                float(route_scale), _p(out), _stream()), "pm_moe_router_bwd")
    return out


def moe_sort(top_idx: Tensor, n_experts: int):
    """Stable sort of the flattened (token, slot) pairs by expert (pm_moe_sort) -> (dest_of_slot [N*k], slot_of_dest [N*k], offsets [E+1]), i32."""
    lib = _lib.load()
    _req(top_idx, torch.int32, "top_idx")
    m = top_idx.numel()
    dev = top_idx.device
    dest = torch.empty(m, dtype=torch.int32, device=dev)
    slot = torch.empty(m, dtype=torch.int32, device=dev)
    offsets = torch.empty(n_experts + 1, dtype=torch.int32, device=dev)
    ws = workspace(lib.pm_moe_sort_workspace_bytes(m, n_experts), dev, "moe_sort")
    check(lib.pm_moe_sort(_p(top_idx), m, n_experts, _p(dest), _p(slot), _p(offsets), _p(ws), ws.numel(), _stream()), "pm_moe_sort")
    return dest, slot, offsets


def moe_gather_rows(x: Tensor, row_map: Tensor, divisor: int = 1, scale: Tensor | None = None) -> Tensor:
    """out[r] = x[row_map[r] // divisor] (* scale[row_map[r]]), bf16 rows (pm_moe_gather_rows); row_map[r] < 0 -> zero row."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x"); _req(row_map, torch.int32, "row_map")
    if scale is not None:
        _req(scale, torch.float32, "scale")
    h = x.shape[-1]
    out = torch.empty((row_map.numel(), h), dtype=torch.bfloat16, device=x.device)
    check(lib.pm_moe_gather_rows(_p(x), _p(row_map), divisor, _p(scale), row_map.numel(), h, _p(out), _stream()), "pm_moe_gather_rows")
    return out


def moe_combine(y: Tensor, dest_of_slot: Tensor, top_k: int, scores: Tensor | None = None, base: Tensor | None = None) -> Tensor:
    """out[t] = base[t] + sum_j bf16(y[dest_of_slot[t*k+j]] * scores[t*k+j]) accumulated in f32 in sorted-row order, rounded once (pm_moe_combine)."""
    lib = _lib.load()
    _req(y, torch.bfloat16, "y"); _req(dest_of_slot, torch.int32, "dest_of_slot")
    if scores is not None:
        _req(scores, torch.float32, "scores")
    if base is not None:
        _req(base, torch.bfloat16, "base")
    n = dest_of_slot.numel() // top_k
    h = y.shape[-1]
    out = torch.empty((n, h), dtype=torch.bfloat16, device=y.device)
    check(lib.pm_moe_combine(_p(y), _p(dest_of_slot), _p(scores), _p(base), n, top_k, h, _p(out), _stream()), "pm_moe_combine")
    return out


def moe_slot_grads(a: Tensor, b: Tensor | None, slot_of_dest: Tensor, top_k: int, scores: Tensor | None, want_rows: bool, want_dscore: bool):
    """Per sorted row r (slot s, token t = s // k): rows[r] = bf16(a[t] * scores[s]), dscore[s] = <a[t], b[r]> (pm_moe_slot_grads)."""
    lib = _lib.load()
    _req(a, torch.bfloat16, "a"); _req(slot_of_dest, torch.int32, "slot_of_dest")
    m, h = slot_of_dest.numel(), a.shape[-1]
    rows = torch.empty((m, h), dtype=torch.bfloat16, device=a.device) if want_rows else None
    dscore = torch.zeros(m, dtype=torch.float32, device=a.device) if want_dscore else None
    if want_dscore:
        _req(b, torch.bfloat16, "b")
    check(lib.pm_moe_slot_grads(_p(a), _p(b), _p(slot_of_dest), top_k, _p(scores), m, h, _p(rows), _p(dscore), _stream()), "pm_moe_slot_grads")
    return rows, dscore


def moe_ep_row_map(recv_counts: Tensor, total_rows: int):
    """recv_counts [ep, n_local] i32 (rows received per (source rank, local expert)) -> (row_map, inv_map [total_rows], local_offsets [n_local+1])."""
    lib = _lib.load()
    _req(recv_counts, torch.int32, "recv_counts")
    ep, le = recv_counts.shape
    dev = recv_counts.device
    row_map = torch.empty(total_rows, dtype=torch.int32, device=dev)
    inv_map = torch.empty(total_rows, dtype=torch.int32, device=dev)
    offs = torch.empty(le + 1, dtype=torch.int32, device=dev)
    check(lib.pm_moe_ep_row_map(_p(recv_counts), ep, le, _p(row_map), _p(inv_map), _p(offs), _stream()), "pm_moe_ep_row_map")
    return row_map, inv_map, offs


def grouped_gemm_supported(n_rows: int, n_out: int, k_dim: int) -> bool:
    return bool(_lib.load().pm_grouped_gemm_supported(n_rows, n_out, k_dim))


def _gg_check(rows: Tensor, w: Tensor, offsets: Tensor, what: str):
    _req(rows, torch.bfloat16, what); _req(w, torch.bfloat16, "w"); _req(offsets, torch.int32, "offsets")
    if w.dim() != 3 or rows.dim() != 2 or offsets.numel() != w.shape[0] + 1:
        raise PrimeMI355XError(f"grouped GEMM: rows {tuple(rows.shape)}, w {tuple(w.shape)}, offsets {tuple(offsets.shape)}")


def _gg_second(w: Tensor, w2: Tensor | None):
    if w2 is None:
        return 0
    _req(w2, torch.bfloat16, "w2")
    if w2.dim() != 3 or w2.shape[0] != w.shape[0] or w2.shape[2] != w.shape[2]:
        raise PrimeMI355XError(f"grouped GEMM: second weight set {tuple(w2.shape)} does not match {tuple(w.shape)}")
    return w2.shape[1]


def grouped_gemm_nt(x: Tensor, w: Tensor, offsets: Tensor, w2: Tensor | None = None, variant: int | None = None) -> Tensor:
    """y[m] = x[m] @ [w | w2][e(m)].T over expert-sorted rows: x [M, K], w [E, N1, K] (, w2 [E, N2, K]), offsets [E+1] i32 -> y [M, N1 + N2]."""
    lib = _lib.load()
    _gg_check(x, w, offsets, "x")
    e, n, k = w.shape
    n2 = _gg_second(w, w2)
    if x.shape[1] != k:
        raise PrimeMI355XError(f"grouped_gemm_nt: x {tuple(x.shape)} vs w {tuple(w.shape)}")
    y = torch.empty((x.shape[0], n + n2), dtype=torch.bfloat16, device=x.device)
    check(lib.pm_grouped_gemm_nt(_p(x), _p(w), _p(w2), _p(offsets), e, x.shape[0], n, n + n2, k, _p(y), _sched(variant), _stream()), "pm_grouped_gemm_nt")
    return y


def grouped_gemm_nn(dy: Tensor, w: Tensor, offsets: Tensor, w2: Tensor | None = None, variant: int | None = None) -> Tensor:
    """dx[m] = dy[m] @ [w ; w2][e(m)]: dy [M, N1 + N2], w [E, N1, K] (, w2 [E, N2, K]) -> dx [M, K]."""
    lib = _lib.load()
    _gg_check(dy, w, offsets, "dy")
    e, n, k = w.shape
    n2 = _gg_second(w, w2)
    if dy.shape[1] != n + n2:
        raise PrimeMI355XError(f"grouped_gemm_nn: dy {tuple(dy.shape)} vs w {tuple(w.shape)} (+ {n2})")
    dx = torch.empty((dy.shape[0], k), dtype=torch.bfloat16, device=dy.device)
    check(lib.pm_grouped_gemm_nn(_p(dy), _p(w), _p(w2), _p(offsets), e, dy.shape[0], n, n + n2, k, _p(dx), _sched(variant), _stream()), "pm_grouped_gemm_nn")
    return dx


def grouped_gemm_nt_swiglu_supported(x: Tensor, w_gate: Tensor, w_up: Tensor) -> bool:
    """Can `grouped_gemm_nt_swiglu` take these operands?  Shapes (inter % 128, K % 128) and the placement of the two weight tensors: one buffer descriptor
    reads both, so w_up must start above w_gate, a whole number of weight rows away, within 2 GiB."""
    if w_gate.dim() != 3 or tuple(w_gate.shape) != tuple(w_up.shape) or x.dim() != 2 or x.shape[1] != w_gate.shape[2]:
        return False
    if not (x.is_cuda and x.dtype == w_gate.dtype == w_up.dtype == torch.bfloat16 and x.is_contiguous() and w_gate.is_contiguous() and w_up.is_contiguous()):
        return False
    delta = w_up.data_ptr() - w_gate.data_ptr()
    row_bytes = w_gate.shape[2] * 2  # the tile's descriptor is sized in whole weight rows: a distance that is not a multiple of the row pitch would cut the
    if delta <= 0 or delta % row_bytes or delta + 128 * row_bytes >= 2**31 - 4096:  # K tail of every tile's last up row short
        return False
    return bool(_lib.load().pm_grouped_gemm_nt_swiglu_supported(x.shape[0], w_gate.shape[1], w_gate.shape[2]))


def grouped_gemm_nt_swiglu(x: Tensor, w_gate: Tensor, w_up: Tensor, offsets: Tensor, variant: int | None = None) -> tuple[Tensor, Tensor]:
    """(gate_up [M, 2 inter], act [M, inter]) = (x[m] @ [w_gate | w_up][e(m)].T, silu(gate) * up) in one launch — equal to grouped_gemm_nt(x, w_gate, offsets,
    w2=w_up) and swiglu_fwd of it, bit for bit."""
    lib = _lib.load()
    _gg_check(x, w_gate, offsets, "x")
    if not grouped_gemm_nt_swiglu_supported(x, w_gate, w_up):
        raise PrimeMI355XError(f"grouped_gemm_nt_swiglu: unsupported operands x {tuple(x.shape)}, w {tuple(w_gate.shape)} / {tuple(w_up.shape)} "
                               f"(w_up - w_gate = {w_up.data_ptr() - w_gate.data_ptr()} bytes)")
    e, inter, k = w_gate.shape
    gate_up = torch.empty((x.shape[0], 2 * inter), dtype=torch.bfloat16, device=x.device)
    act = torch.empty((x.shape[0], inter), dtype=torch.bfloat16, device=x.device)
    check(lib.pm_grouped_gemm_nt_swiglu(_p(x), _p(w_gate), _p(w_up), _p(offsets), e, x.shape[0], inter, k, _p(gate_up), _p(act), _sched(variant), _stream()), "pm_grouped_gemm_nt_swiglu")
    return gate_up, act


def grouped_gemm_nn_swiglu_bwd_supported(n_rows: int, n_dim: int, inter: int) -> bool:
    return bool(_lib.load().pm_grouped_gemm_nn_swiglu_bwd_supported(n_rows, n_dim, inter))


def grouped_gemm_nn_swiglu_bwd(dy: Tensor, w_down: Tensor, offsets: Tensor, gate_up: Tensor, out: Tensor | None = None, variant: int | None = None) -> Tensor:
    """grad_gate_up[m] = swiglu_bwd(dy[m] @ w_down[e(m)], gate_up[m]): dy [M, N], w_down [E, N, inter], gate_up [M, 2 inter] -> [M, 2 inter] — equals
    swiglu_bwd(grouped_gemm_nn(dy, w_down, offsets), gate_up) bit for bit; `out` may be `gate_up`."""
    lib = _lib.load()
    _gg_check(dy, w_down, offsets, "dy")
    e, n, inter = w_down.shape
    _req(gate_up, torch.bfloat16, "gate_up")
    if dy.shape[1] != n or tuple(gate_up.shape) != (dy.shape[0], 2 * inter):
        raise PrimeMI355XError(f"grouped_gemm_nn_swiglu_bwd: dy {tuple(dy.shape)}, w_down {tuple(w_down.shape)}, gate_up {tuple(gate_up.shape)}")
    if out is None:
        out = torch.empty_like(gate_up)
    _req(out, torch.bfloat16, "out")
    if tuple(out.shape) != tuple(gate_up.shape):
        raise PrimeMI355XError(f"grouped_gemm_nn_swiglu_bwd: out {tuple(out.shape)} != gate_up {tuple(gate_up.shape)}")
    check(lib.pm_grouped_gemm_nn_swiglu_bwd(_p(dy), _p(w_down), _p(offsets), e, dy.shape[0], n, inter, _p(gate_up), _p(out), _sched(variant), _stream()), "pm_grouped_gemm_nn_swiglu_bwd")
    return out


def grouped_gemm_tn(dy: Tensor, x: Tensor, offsets: Tensor, n_first: int | None = None, variant: int | None = None):
    """dw[e] = dy[group e].T @ x[group e]: dy [M, N], x [M, K] -> dw [E, N, K]; with n_first: (dw [E, n_first, K], dw2 [E, N - n_first, K])."""
    lib = _lib.load()
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(offsets, torch.int32, "offsets")
    e = offsets.numel() - 1
    if dy.shape[0] != x.shape[0]:
        raise PrimeMI355XError(f"grouped_gemm_tn: dy {tuple(dy.shape)} vs x {tuple(x.shape)}")
    n, k = dy.shape[1], x.shape[1]
    n1 = n if n_first is None else n_first
    dw = torch.empty((e, n1, k), dtype=torch.bfloat16, device=x.device)
    dw2 = torch.empty((e, n - n1, k), dtype=torch.bfloat16, device=x.device) if n1 < n else None
    check(lib.pm_grouped_gemm_tn(_p(dy), _p(x), _p(offsets), e, x.shape[0], n1, n, k, _p(dw), _p(dw2), _sched(variant), _stream()), "pm_grouped_gemm_tn")
    return dw if n_first is None else (dw, dw2)


def grouped_gemm_tn_f32_supported(n_first: int, n: int, k: int) -> bool:
    return n_first % 256 == 0 and n % 256 == 0 and k % 256 == 0


def grouped_gemm_tn_f32(dy: Tensor, x: Tensor, offsets: Tensor, dw: Tensor, dw2: Tensor | None = None, accumulate: bool = False, variant: int | None = None,
                        tile_sumsq: Tensor | None = None) -> None:
    """dw[e] (+)= dy[group e, :n_first].T @ x[group e] and dw2[e] (+)= dy[group e, n_first:].T @ x[group e] in f32, written straight into the given
    (main-gradient) buffers: dw [E, n_first, K], dw2 [E, N - n_first, K] or None.  `tile_sumsq` (f32, >= grouped_gemm_tn_sumsq_slots(E, N, K) elements):
    every wave also writes the sum of squares of the FINAL values it stored — after the last accumulating launch of a step `tile_sumsq.sum()` is the
    squared norm of dw and dw2 (the gradient-norm pass fused into the weight-gradient epilogue)."""
    lib = _lib.load()
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(offsets, torch.int32, "offsets"); _req(dw, torch.float32, "dw")
    e = offsets.numel() - 1
    n, k = dy.shape[1], x.shape[1]
    n1 = dw.shape[1]
    if dy.shape[0] != x.shape[0] or tuple(dw.shape) != (e, n1, k) or (n1 < n) != (dw2 is not None):
        raise PrimeMI355XError(f"grouped_gemm_tn_f32: dy {tuple(dy.shape)}, x {tuple(x.shape)}, dw {tuple(dw.shape)}")
    if dw2 is not None:
        _req(dw2, torch.float32, "dw2")
        if tuple(dw2.shape) != (e, n - n1, k):
            raise PrimeMI355XError(f"grouped_gemm_tn_f32: dw2 {tuple(dw2.shape)} != {(e, n - n1, k)}")
    if tile_sumsq is None:
        check(lib.pm_grouped_gemm_tn_f32(_p(dy), _p(x), _p(offsets), e, x.shape[0], n1, n, k, _p(dw), _p(dw2), int(accumulate), _sched(variant), _stream()), "pm_grouped_gemm_tn_f32")
        return
    _req(tile_sumsq, torch.float32, "tile_sumsq")
    check(lib.pm_grouped_gemm_tn_f32_sumsq(_p(dy), _p(x), _p(offsets), e, x.shape[0], n1, n, k, _p(dw), _p(dw2), int(accumulate), _p(tile_sumsq), tile_sumsq.numel(),
                                           _sched(variant), _stream()), "pm_grouped_gemm_tn_f32_sumsq")


def grouped_gemm_tn_sumsq_slots(n_experts: int, n: int, k: int) -> int:
    return int(_lib.load().pm_grouped_gemm_tn_sumsq_slots(n_experts, n, k))


def sgd_step(master: Tensor, momentum_buf: Tensor | None, grad: Tensor, param_bf16: Tensor | None, lr: float, momentum: float, weight_decay: float,
             nesterov: bool, step: int, grad_scale: float = 1.0, clip: Tensor | None = None, zero_grad: bool = True) -> None:
    """torch.optim.SGD(momentum, nesterov, L2 weight decay) on flat f32 buffers (pm_sgd_step)."""
    lib = _lib.load()
    for t, nm in ((master, "master"), (grad, "grad")) + (((momentum_buf, "momentum_buf"),) if momentum_buf is not None else ()):
        _req(t, torch.float32, nm)
    if param_bf16 is not None:
        _req(param_bf16, torch.bfloat16, "param_bf16")
    check(lib.pm_sgd_step(_p(master), _p(momentum_buf), _p(grad), _p(param_bf16), master.numel(), lr, momentum, weight_decay, 1 if nesterov else 0, step,
                          grad_scale, _p(clip), 1 if zero_grad else 0, _stream()), "pm_sgd_step")


def sign_sgd_step(master: Tensor, grad: Tensor, param_bf16: Tensor | None, lr: float, weight_decay: float, grad_scale: float = 1.0,
                  clip: Tensor | None = None, zero_grad: bool = True) -> None:
    """SignSGD (trainer/sign_sgd.py) on flat f32 buffers (pm_sign_sgd_step)."""
    lib = _lib.load()
    _req(master, torch.float32, "master"); _req(grad, torch.float32, "grad")
    if param_bf16 is not None:
        _req(param_bf16, torch.bfloat16, "param_bf16")
    check(lib.pm_sign_sgd_step(_p(master), _p(grad), _p(param_bf16), master.numel(), lr, weight_decay, grad_scale, _p(clip), 1 if zero_grad else 0, _stream()),
          "pm_sign_sgd_step")


# ------------------------------------------------------------------------------------------------ dense MFMA GEMM (csrc/gemm.hip)
def _gemm_operand(t: Tensor, name: str) -> tuple[int, bool, int, int]:
    """(row stride in elements, stored K-major?, rows, k) of a 2-D bf16 operand given as the LOGICAL [rows, K] matrix: unit stride along K =
    K-contiguous storage; unit stride along the rows (a `.t()` view of a [K, rows] tensor) = K-major storage."""
    _req(t, torch.bfloat16, name, contiguous=False)
    if t.dim() != 2:
        raise PrimeMI355XError(f"{name}: expected a 2-D operand, got {tuple(t.shape)}")
    rows, k = t.shape
    if t.stride(1) == 1:
        return t.stride(0), False, rows, k
    if t.stride(0) == 1:
        return t.stride(1), True, rows, k
    raise PrimeMI355XError(f"{name}: one of the two axes must have unit stride, got strides {t.stride()}")


def gemm_supported(x: Tensor, w: Tensor, out: Tensor | None = None, out_dtype: torch.dtype = torch.bfloat16) -> bool:
    """Can `gemm(x, w, out=out)` run on the hand-written MFMA core (shape / layout / alignment of the operands AND of the result)?  `out` None: the
    result `gemm` would allocate itself ([M, N] contiguous of `out_dtype`) — its rows must still start 16-byte aligned (N % 8 for bf16, N % 4 for f32)."""
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
        return False
    try:
        ldx, xk, m, k = _gemm_operand(x, "x")
        ldw, wk, n, _ = _gemm_operand(w, "w")
    except PrimeMI355XError:
        return False
    if ldx % 8 or ldw % 8 or x.data_ptr() % 16 or w.data_ptr() % 16:
        return False
    if out is not None:
        if out.dim() != 2 or tuple(out.shape) != (m, n) or out.stride(1) != 1 or out.dtype not in (torch.bfloat16, torch.float32):
            return False
        ldo, esz, base = out.stride(0), out.element_size(), out.data_ptr()
    else:
        ldo, esz, base = n, (2 if out_dtype == torch.bfloat16 else 4), 0
    if ldo % 4 or (ldo * esz) % 16 or base % 16:  # pm_gemm_bf16: ldo a multiple of 4 elements, every output row 16-byte aligned
        return False
    return bool(_lib.load().pm_gemm_supported(m, n, k, int(xk), int(wk)))


def gemm_shape_supported(m: int, n: int, k: int, x_kmajor: bool, w_kmajor: bool) -> bool:
    """pm_gemm_supported for contiguous operands of these logical sizes (K-major = stored [K, rows]); 16-byte alignment of the bases is the caller's."""
    ldx, ldw = (m if x_kmajor else k), (n if w_kmajor else k)
    return ldx % 8 == 0 and ldw % 8 == 0 and n % 4 == 0 and bool(_lib.load().pm_gemm_supported(m, n, k, int(x_kmajor), int(w_kmajor)))


GEMM_SCHED_DEFAULT = -1  # include/prime_mi355x.h PM_GEMM_SCHED_*: the schedule word is an ARGUMENT of every launch (the library holds no schedule state)
GEMM_SCHED_PERSISTENT, GEMM_SCHED_NO_SPLITK, GEMM_SCHED_NT_OUT, GEMM_SCHED_SWEEP_R3 = 4, 32, 512, 1024
GROUPED_SCHED_SMALL, GROUPED_SCHED_BIG = 1 << 16, 2 << 16


_default_sched = GEMM_SCHED_DEFAULT  # what a launch with `variant=None` passes; moved only by gemm_set_variant (tests / tools)


def _sched(variant: int | None) -> int:
    return _default_sched if variant is None else int(variant)


def gemm(x: Tensor, w: Tensor, out: Tensor | None = None, out_dtype: torch.dtype = torch.bfloat16, bias: Tensor | None = None, accumulate: bool = False,
         tile_sumsq: Tensor | None = None, variant: int | None = None) -> Tensor:
    """out[M, N] (+)= x[M, K] @ w[N, K]^T (+ bias) on the MFMA core — F.linear's product for any of the three layouts the trainer step has:
    forward `gemm(x, W)`, dgrad `gemm(dy, W.t())` (W.t() = the [in, out] view of the stored [out, in] weight), wgrad
    `gemm(dy.t(), x.t(), out=grad, out_dtype=f32, accumulate=...)`.  Operands are passed as LOGICAL [rows, K] matrices; their strides say how
    they are stored (no copies are made).  `tile_sumsq` (f32, >= gemm_sumsq_slots(M, N) elements; f32 `out` only): every wave also writes the
    sum of squares of the final values it stored — the gradient-norm pass fused into the weight-gradient epilogue (sum the slots with sum_partials)."""
    lib = _lib.load()
    ldx, xk, m, k = _gemm_operand(x, "x")
    ldw, wk, n, k2 = _gemm_operand(w, "w")
    if k != k2:
        raise PrimeMI355XError(f"gemm: contraction mismatch {tuple(x.shape)} vs {tuple(w.shape)}")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=x.device)
    else:
        _req(out, None, "out", contiguous=False)
        if out.dtype not in _DT or tuple(out.shape) != (m, n) or out.stride(1) != 1:
            raise PrimeMI355XError(f"gemm: out must be [{m}, {n}] f32 | bf16 with unit stride along N, got {tuple(out.shape)} {out.dtype} {out.stride()}")
    if bias is not None:
        _req(bias, torch.bfloat16, "bias")
        if bias.numel() != n:
            raise PrimeMI355XError("gemm: bias length != N")
    if tile_sumsq is not None:
        _req(tile_sumsq, torch.float32, "tile_sumsq")
        if out.dtype != torch.float32 or bias is not None:
            raise PrimeMI355XError("gemm: tile_sumsq needs an f32 output and no bias")
    # `variant`: the schedule word of THIS launch (PM_GEMM_SCHED_*; None = the library default) — an argument of the C call, not process state
    if out.dtype == torch.float32 and bias is None:  # the f32 (weight-gradient) form: a workspace lets a half-empty last round of tiles split along K
        ws_bytes = lib.pm_gemm_workspace_bytes(m, n, k)
        ws = workspace(ws_bytes, x.device, "gemm_splitk") if ws_bytes else None
        check(lib.pm_gemm_bf16_ws(_p(x), ldx, int(xk), _p(w), ldw, int(wk), _p(out), out.stride(0), _DT[out.dtype], None, int(accumulate), m, n, k,
                                  _p(tile_sumsq), tile_sumsq.numel() if tile_sumsq is not None else 0, _p(ws), ws_bytes, _sched(variant), _stream()), "pm_gemm_bf16_ws")
        return out
    check(lib.pm_gemm_bf16(_p(x), ldx, int(xk), _p(w), ldw, int(wk), _p(out), out.stride(0), _DT[out.dtype], _p(bias), int(accumulate), m, n, k, _sched(variant),
                           _stream()), "pm_gemm_bf16")
    return out


def gemm_swiglu_supported(x: Tensor, w_gate_up: Tensor) -> bool:
    """Can `gemm_swiglu` take these operands (K-contiguous x [M, K] and w [2 inter, K], inter % 128 == 0, K % 128 == 0)?"""
    if x.dim() != 2 or w_gate_up.dim() != 2 or x.dtype != torch.bfloat16 or w_gate_up.dtype != torch.bfloat16 or x.device.type != "cuda":
        return False
    if x.stride(1) != 1 or w_gate_up.stride(1) != 1 or x.shape[1] != w_gate_up.shape[1] or w_gate_up.shape[0] % 2:
        return False
    if x.stride(0) % 8 or w_gate_up.stride(0) % 8 or x.data_ptr() % 16 or w_gate_up.data_ptr() % 16:
        return False
    return bool(_lib.load().pm_gemm_swiglu_supported(x.shape[0], w_gate_up.shape[0] // 2, x.shape[1]))


def gemm_swiglu(x: Tensor, w_gate_up: Tensor, gate_up: Tensor | None = None, act: Tensor | None = None, variant: int | None = None) -> tuple[Tensor, Tensor]:
    """(gate_up [M, 2 inter], act [M, inter]) = (x @ w^T, silu(gate) * up): the MLP's fused gate|up projection with the SwiGLU activation in the GEMM
    epilogue (mlp.py:27-29) — `gate_up` equals gemm(x, w) and `act` equals swiglu_fwd(gate_up) bit for bit; gate_up is not read back from HBM."""
    _req(x, torch.bfloat16, "x", contiguous=False); _req(w_gate_up, torch.bfloat16, "w_gate_up", contiguous=False)
    if not gemm_swiglu_supported(x, w_gate_up):
        raise PrimeMI355XError(f"gemm_swiglu: unsupported operands x {tuple(x.shape)} {x.stride()} w {tuple(w_gate_up.shape)} {w_gate_up.stride()}")
    m, k = x.shape
    inter = w_gate_up.shape[0] // 2
    if gate_up is None:
        gate_up = torch.empty((m, 2 * inter), dtype=torch.bfloat16, device=x.device)
    if act is None:
        act = torch.empty((m, inter), dtype=torch.bfloat16, device=x.device)
    for t, cols, nm in ((gate_up, 2 * inter, "gate_up"), (act, inter, "act")):
        _req(t, torch.bfloat16, nm, contiguous=False)
        if tuple(t.shape) != (m, cols) or t.stride(1) != 1 or t.stride(0) % 8:
            raise PrimeMI355XError(f"gemm_swiglu: {nm} must be [{m}, {cols}] bf16 with unit column stride and a row stride that is a multiple of 8")
    check(_lib.load().pm_gemm_swiglu_bf16(_p(x), x.stride(0), _p(w_gate_up), w_gate_up.stride(0), _p(gate_up), gate_up.stride(0), _p(act), act.stride(0), m, inter, k,
                                          _sched(variant), _stream()), "pm_gemm_swiglu_bf16")
    return gate_up, act


def gemm_swiglu_bwd_supported(dy: Tensor, w_down: Tensor) -> bool:
    """Can `gemm_swiglu_bwd` take these operands (K-contiguous dy [M, K], the stored w_down [K, inter] with unit column stride, K % 128 == 0, inter % 8 == 0)?"""
    if dy.dim() != 2 or w_down.dim() != 2 or dy.dtype != torch.bfloat16 or w_down.dtype != torch.bfloat16 or dy.device.type != "cuda":
        return False
    if dy.stride(1) != 1 or w_down.stride(1) != 1 or dy.shape[1] != w_down.shape[0]:
        return False
    if dy.stride(0) % 8 or w_down.stride(0) % 8 or dy.data_ptr() % 16 or w_down.data_ptr() % 16:
        return False
    return bool(_lib.load().pm_gemm_swiglu_bwd_supported(dy.shape[0], w_down.shape[1], dy.shape[1]))


def gemm_swiglu_bwd(dy: Tensor, w_down: Tensor, gate_up: Tensor, out: Tensor | None = None, variant: int | None = None) -> Tensor:
    """grad_gate_up [M, 2 inter] = swiglu_bwd(dy @ w_down, gate_up): the down projection's row gradient with the SwiGLU backward in the GEMM epilogue — equals
    swiglu_bwd(gemm(dy, w_down.t()), gate_up) bit for bit; the gradient of the activation is never written.  `out` may be `gate_up` (in place)."""
    _req(dy, torch.bfloat16, "dy", contiguous=False); _req(w_down, torch.bfloat16, "w_down", contiguous=False); _req(gate_up, torch.bfloat16, "gate_up", contiguous=False)
    if not gemm_swiglu_bwd_supported(dy, w_down):
        raise PrimeMI355XError(f"gemm_swiglu_bwd: unsupported operands dy {tuple(dy.shape)} {dy.stride()} w_down {tuple(w_down.shape)} {w_down.stride()}")
    m, k = dy.shape
    inter = w_down.shape[1]
    if out is None:
        out = torch.empty((m, 2 * inter), dtype=torch.bfloat16, device=dy.device)
    for t, nm in ((gate_up, "gate_up"), (out, "out")):
        _req(t, torch.bfloat16, nm, contiguous=False)
        if tuple(t.shape) != (m, 2 * inter) or t.stride(1) != 1 or t.stride(0) % 8:
            raise PrimeMI355XError(f"gemm_swiglu_bwd: {nm} must be [{m}, {2 * inter}] bf16 with unit column stride and a row stride that is a multiple of 8")
    check(_lib.load().pm_gemm_swiglu_bwd_bf16(_p(dy), dy.stride(0), _p(w_down), w_down.stride(0), _p(gate_up), gate_up.stride(0), _p(out), out.stride(0), m, inter, k,
                                              _sched(variant), _stream()), "pm_gemm_swiglu_bwd_bf16")
    return out


def gemm_sumsq_slots(m: int, n: int) -> int:
    return int(_lib.load().pm_gemm_sumsq_slots(m, n))


def sumsq_segments(base: Tensor, segments: Tensor, partials: Tensor) -> None:
    """partials[i] = sum(base[off_i : off_i + len_i]^2) for the int64 [n, 2] (offset, length) rows of `segments` (on the device)."""
    _req(base, torch.float32, "base"); _req(partials, torch.float32, "partials"); _req(segments, torch.int64, "segments")
    if segments.dim() != 2 or segments.shape[1] != 2 or partials.numel() < segments.shape[0]:
        raise PrimeMI355XError("sumsq_segments: segments must be [n, 2] and partials hold n floats")
    check(_lib.load().pm_sumsq_segments(_p(base), _p(segments), segments.shape[0], _p(partials), _stream()), "pm_sumsq_segments")


def sum_partials(partials: Tensor, out_scalar: Tensor) -> None:
    """out_scalar[0] += sum(partials), in a fixed order."""
    _req(partials, torch.float32, "partials"); _req(out_scalar, torch.float32, "out")
    check(_lib.load().pm_sum_partials(_p(partials), partials.numel(), _p(out_scalar), _stream()), "pm_sum_partials")


def gemm_set_variant(v: int) -> None:
    """Tests / tools: the schedule word that launches WITHOUT an explicit `variant=` pass from now on (7 or -1 = the library default).  A Python-side default
    only — the word still travels as the `sched` argument of every C call; the library holds no schedule state.  Words that select a build of a
    `make VARIANTS=1` library raise here on the product library, so that callers can skip."""
    global _default_sched
    lib = _lib.load()
    if v >= 0 and (v & ~(1 | 2 | 4 | 32 | 512 | 1024)):
        rc = lib.pm_gemm_debug_set_variant(0)  # accepted only by a VARIANTS=1 library
        if rc != 0:
            raise PrimeMI355XError(f"gemm_set_variant({v}): this schedule exists only in a `make VARIANTS=1` build")
        lib.pm_gemm_debug_set_variant(-1)
    _default_sched = GEMM_SCHED_DEFAULT if v in (7, -1) else int(v)
