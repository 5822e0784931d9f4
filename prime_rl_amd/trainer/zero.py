"""Data-parallel gradient/optimizer sharding over RCCL (xGMI) — the MI355X re-design of the reference's FSDP2 wrapping
(src/prime_rl/trainer/model.py:823-960) and of the gradient post-processing in trainer/rl/train.py:557-570.

Reference (ZeRO-3): per decoder block all-gather bf16 params in forward and again in backward, reduce-scatter f32 grads,
`reshard_after_forward=True`; then `scale_gradients_(x dp)` undoes FSDP's mean because the loss is already divided by
the global token count.

Here (ZeRO-2 on 288 GB parts): bf16 params stay replicated; per-bucket (= one decoder layer) `reduce_scatter(SUM)` of the
f32 gradient as soon as the layer's backward is done, on a dedicated stream; clip + AdamW on the rank's shard; in-place
`all_gather` of the updated bf16 shard per bucket, waited on only by the next forward of that layer.  SUM (not mean)
is exactly `mean x dp`.  Functions are device-agnostic so the world_size-2 gloo tests exercise the same code on CPU; the
arithmetic (`kernels.sumsq_accumulate / clip_coef / adamw_step`) is `prime_rl_amd.ops` on the GPU.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def _on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda


# bench.py sets this to a list to get HIP-event timings of every collective ON THE COMM STREAM ("rs" / "ag": busy time of the stream) and of every
# wait of the compute stream for it ("exposed": time the compute stream stood still for a collective) — so that a multi-GPU run explains itself.
COMM_EVENTS: list | None = None


def _timed(kind: str, stream, fn) -> None:
    """Run fn() with timing events recorded on `stream` around it when COMM_EVENTS is collecting."""
    if COMM_EVENTS is None:
        fn()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    fn()
    e1.record(stream)
    COMM_EVENTS.append((kind, e0, e1))


def reduce_scatter_bucket(store, bucket, group, comm_stream=None) -> None:
    """grad_shard[bucket shard] = sum over ranks of main_grad[bucket][rank's chunk]; async on `comm_stream` when on GPU."""
    _, local = store.shard_slices(bucket)
    src = store.main_grad[bucket.offset:bucket.offset + bucket.size]
    dst = store.grad_shard[local]
    if comm_stream is not None and _on_gpu(src):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            _timed("rs", comm_stream, lambda: dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=group))
    else:
        dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=group)


def apply_update(kernels, optim, master, exp_avg, exp_avg_sq, grad, params_bf16, step: int, coef, zero_grad: bool) -> None:
    """One fused optimizer launch on a flat f32 shard — `optim.type` selects the reference's optimizer (trainer/optim/__init__.py:74-95):
    "adamw" (default), "sgd" (torch.optim.SGD with momentum / Nesterov; the momentum buffer lives in `exp_avg`), "sign_sgd" (stateless)."""
    kind = getattr(optim, "type", "adamw")
    if kind == "adamw":
        kernels.adamw_step(master, exp_avg, exp_avg_sq, grad, params_bf16, optim.lr, optim.betas1, optim.betas2, optim.eps, optim.weight_decay, step, 1.0,
                           coef, zero_grad=zero_grad)
    elif kind == "sgd":
        kernels.sgd_step(master, exp_avg if optim.momentum else None, grad, params_bf16, optim.lr, optim.momentum, optim.weight_decay, optim.nesterov, step,
                         1.0, coef, zero_grad=zero_grad)
    elif kind == "sign_sgd":
        kernels.sign_sgd_step(master, grad, params_bf16, optim.lr, optim.weight_decay, 1.0, coef, zero_grad=zero_grad)
    else:
        raise ValueError(f"optim.type = {kind!r}: the fused on-GPU optimizers are adamw, sgd and sign_sgd (Muon stays with the reference)")


def sharded_optimizer_step(store, optim, step: int, group, kernels, comm_stream=None, gather_events: dict | None = None, scratch: torch.Tensor | None = None,
                           local_sumsq=None):
    """Global-norm clip + AdamW on this rank's shard + in-place all-gather of the bf16 parameters.
    `local_sumsq(out)`: optional replacement of the norm pass over the local gradient (single-rank engines only).
    Returns the global gradient norm (1-element tensor)."""
    dp = store.dp_size
    sumsq = scratch if scratch is not None else torch.zeros(1, dtype=torch.float32, device=store.main_grad.device)
    sumsq.zero_()
    gpu = _on_gpu(store.main_grad)
    collectives = getattr(store, "use_collectives", dp > 1)  # True at dp == 1 only when forced (single-GPU check of this path)
    if collectives and gpu and comm_stream is not None:
        _timed("exposed", torch.cuda.current_stream(), lambda: torch.cuda.current_stream().wait_stream(comm_stream))  # every reduce-scatter has landed
    if local_sumsq is not None and not collectives:  # the engine already holds the pieces of the norm (weight-gradient epilogue partials): no re-read
        local_sumsq(sumsq)
    else:
        kernels.sumsq_accumulate(store.grad_shard, sumsq)
    if collectives:
        dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=group)
    coef, norm = kernels.clip_coef(sumsq, 1.0, optim.max_norm if optim.max_norm else 0.0)
    if not collectives:
        apply_update(kernels, optim, store.master, store.exp_avg, store.exp_avg_sq, store.grad_shard, store.params, step, coef, zero_grad=False)
        return norm
    for b in store.buckets:
        full, local = store.shard_slices(b)
        apply_update(kernels, optim, store.master[local], store.exp_avg[local] if store.exp_avg.numel() else None,
                     store.exp_avg_sq[local] if store.exp_avg_sq.numel() else None, store.grad_shard[local], store.params[full], step, coef, zero_grad=False)
        dst = store.params[b.offset:b.offset + b.size]
        if gpu and comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                _timed("ag", comm_stream, lambda dst=dst, full=full: dist.all_gather_into_tensor(dst, store.params[full], group=group))
                gev = torch.cuda.Event()
                gev.record(comm_stream)
            if gather_events is not None:
                gather_events[b.name] = gev  # the next forward waits per bucket: the all-gather overlaps with it
        else:
            dist.all_gather_into_tensor(dst, store.params[full], group=group)
    return norm


def all_reduce_counts(local_counts: tuple[int, int, int], group, device) -> tuple[int, int, int]:
    """The i64[3] token-count all-reduce of train.py:318-323; returns max(count, 1) per component."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tuple(max(int(c), 1) for c in local_counts)
    t = torch.tensor(local_counts, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tuple(max(int(c), 1) for c in t.tolist())
