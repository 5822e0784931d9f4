"""LM-head seam — mirrors `prime_rl.trainer.models.layers.lm_head` (src/prime_rl/trainer/models/layers/lm_head.py).

Same names and call contract: `PrimeLmOutput` (a TypedDict so pytree hooks see tensors, lm_head.py:14-20),
`FusedOutputLinear(in, out, chunk_size)(hidden[b,s,h], labels[b,s], temperature=[b,s]) -> {logprobs, entropy}`,
`VanillaOutputLinear`, `inject_prime_lm_head(model, chunk_size)`.  The per-chunk epilogue of the reference
(~10 eager passes over a 64 MiB f32 chunk, lm_head.py:118-148 and :184-191) is one HIP kernel per chunk:
`pm_logprob_chunk_update` forward, `pm_logprob_chunk_bwd` backward (in place on the chunk).  The chunk GEMMs stay library
calls.  Vocab chunks default to 32768 columns: a [8192 x 32768] bf16 chunk is 512 MiB... so the token chunk is what bounds
memory; with 2k-16k tokens the chunk lives in HBM/Infinity Cache between the GEMM and the epilogue.
"""

from __future__ import annotations

import os
import types
from typing import TypedDict

import torch
import torch.nn as nn
from torch import Tensor

from .... import ops

VOCAB_CHUNK = 8192  # the reference's vocab_chunk_size = min(vocab, 8192) (lm_head.py:112)


class PrimeLmOutput(TypedDict, total=False):
    logits: Tensor | None
    logprobs: Tensor | None
    entropy: Tensor | None
    loss: Tensor | None


def cast_float_and_contiguous(output: PrimeLmOutput) -> PrimeLmOutput:
    """lm_head.py:23-34."""

    def f(t):
        return t.float().contiguous() if t is not None else None

    return PrimeLmOutput(logits=f(output.get("logits")), logprobs=f(output.get("logprobs")), entropy=f(output.get("entropy")), loss=output.get("loss"))


class _SequenceChunkedLogProbEntropyFn(torch.autograd.Function):
    """Token-chunk x vocab-chunk log-prob / entropy without materialising [N, V] (lm_head.py:89-202)."""

    @staticmethod
    def forward(ctx, hidden: Tensor, weight: Tensor, labels: Tensor, inv_temperature: Tensor, chunk_size: int):
        assert hidden.dim() == 2, f"expected hidden [N,H], got {tuple(hidden.shape)}"
        assert weight.dim() == 2, f"expected weight [V,H], got {tuple(weight.shape)}"
        assert labels.dim() == 1, f"expected labels [N], got {tuple(labels.shape)}"
        assert inv_temperature.dim() == 1, f"expected inv_temperature [N], got {tuple(inv_temperature.shape)}"
        assert hidden.shape[0] == labels.shape[0], "hidden/labels N mismatch"
        assert hidden.shape[1] == weight.shape[1], "hidden/weight H mismatch"
        assert hidden.shape[0] == inv_temperature.shape[0], "hidden/inv_temperature N mismatch"
        assert chunk_size > 0
        n, vocab = hidden.shape[0], weight.shape[0]
        vchunk = min(vocab, VOCAB_CHUNK)
        dev = hidden.device
        logprobs = torch.empty(n, dtype=torch.float32, device=dev)
        entropy = torch.empty_like(logprobs)
        logz = torch.empty_like(logprobs)
        labels = labels.contiguous()
        inv_t = inv_temperature.contiguous().float()
        # bf16 operands (the trainer's case): J1 — GEMM tile + log-prob epilogue in ONE kernel per token chunk, logits stay in the MFMA
        # accumulators (csrc/lmhead.hip).  Other dtypes (the reference's fp32 unit-test case): library GEMM per vocab chunk + the
        # online-state kernels below.
        fused = hidden.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and hidden.shape[1] % 64 == 0
        ctx.fused = fused
        for start in range(0, n, chunk_size if not fused else n):
            if fused:
                for s0 in range(0, n, chunk_size):
                    s1 = min(s0 + chunk_size, n)
                    lp, en, lz = ops.lmhead_logprob_fwd(hidden[s0:s1].contiguous(), weight.contiguous(), labels[s0:s1], inv_t[s0:s1])
                    logprobs[s0:s1], entropy[s0:s1], logz[s0:s1] = lp, en, lz
                break
            end = min(start + chunk_size, n)
            state = ops.ChunkedLogprobState(end - start, dev)
            lab, it = labels[start:end], inv_t[start:end]
            for v0 in range(0, vocab, vchunk):
                v1 = min(v0 + vchunk, vocab)
                logits = hidden[start:end] @ weight[v0:v1].t()
                state.update(logits, v0, lab, it)
            lp, en, lz = state.finalize()
            logprobs[start:end], entropy[start:end], logz[start:end] = lp, en, lz
        ctx.set_materialize_grads(False)  # unused entropy output -> grad None, no sync (lm_head.py:150-152)
        ctx.save_for_backward(hidden, weight, labels, inv_t, logz)
        ctx.chunk_size = chunk_size
        return logprobs, entropy

    @staticmethod
    def backward(ctx, grad_logprobs: Tensor, grad_entropy: Tensor | None):
        assert grad_entropy is None, "Backward through entropy is not implemented in FusedOutputLinear"
        assert grad_logprobs is not None, "FusedOutputLinear backward requires logprobs gradients"
        hidden, weight, labels, inv_t, logz = ctx.saved_tensors
        chunk_size = ctx.chunk_size
        n, vocab = hidden.shape[0], weight.shape[0]
        vchunk = min(vocab, VOCAB_CHUNK)
        needs_hidden, needs_weight = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_hidden = torch.zeros_like(hidden) if needs_hidden else None
        grad_weight = None  # allocated on first use (the bf16 form accumulates in f32 on the own core instead)
        g = grad_logprobs.contiguous().float()
        gw_f32 = None  # the fused (bf16) form accumulates the weight gradient in f32 on the own core and rounds once at the end
        for start in range(0, n, chunk_size):
            end = min(start + chunk_size, n)
            h = hidden[start:end]
            if ctx.fused:  # J1 backward: logits recomputed on the MFMA units, dlogits written once (bf16), then the two gradient GEMMs
                dl = ops.lmhead_logprob_bwd(h.contiguous(), weight.contiguous(), labels[start:end], inv_t[start:end], logz[start:end], g[start:end])
                if needs_hidden:  # the two gradient GEMMs on the hand-written core where it applies (the reference: eager matmuls, lm_head.py:184-191)
                    if ops.gemm_supported(dl, weight.t(), out=grad_hidden[start:end]):
                        ops.gemm(dl, weight.t(), out=grad_hidden[start:end])
                    else:
                        torch.mm(dl, weight, out=grad_hidden[start:end])
                if needs_weight:
                    if gw_f32 is None and ops.gemm_supported(dl.t(), h.t()):
                        gw_f32 = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
                    if gw_f32 is not None:
                        ops.gemm(dl.t(), h.t(), out=gw_f32, accumulate=True)
                    else:
                        grad_weight = torch.zeros_like(weight) if grad_weight is None else grad_weight
                        grad_weight.addmm_(dl.t(), h)
                continue
            for v0 in range(0, vocab, vchunk):
                v1 = min(v0 + vchunk, vocab)
                w = weight[v0:v1]
                logits = h @ w.t()
                dl = ops.logprob_bwd(logits, labels[start:end], inv_t[start:end], logz[start:end], g[start:end], out=logits, vocab_start=v0)
                if needs_hidden:
                    grad_hidden[start:end].addmm_(dl, w)
                if needs_weight:
                    grad_weight = torch.zeros_like(weight) if grad_weight is None else grad_weight
                    grad_weight[v0:v1].addmm_(dl.t(), h)
        if gw_f32 is not None:
            grad_weight = gw_f32.to(weight.dtype)
        return grad_hidden, grad_weight, None, None, None


class _StoredLogitsLogProbEntropyFn(torch.autograd.Function):
    """The same function with the bf16 logits KEPT in HBM between forward and backward (the flat engine's default, `lm_head_impl="library"`): on a
    288 GB part 2 bytes x N x V is cheap (5 GB for 16 384 tokens of a 152k vocabulary), and it removes the recomputation GEMM of the chunked form
    from the backward — forward = one GEMM (the MFMA core of csrc/gemm.hip, else the library) + the log-prob / entropy row kernel (K1), backward =
    K1's backward IN PLACE on the stored logits + the two gradient GEMMs.  The weight gradient goes straight into an f32 `main_grad` when the
    trainer engine hung one on the weight (layers/moe.py `_main_grads`).  Values: identical to the chunked form wherever that rounds logits to bf16."""

    @staticmethod
    def forward(ctx, hidden: Tensor, weight: Tensor, labels: Tensor, inv_temperature: Tensor, chunk_size: int):
        n, vocab = hidden.shape[0], weight.shape[0]
        dev = hidden.device
        logprobs = torch.empty(n, dtype=torch.float32, device=dev)
        entropy = torch.empty_like(logprobs)
        logz = torch.empty_like(logprobs)
        labels = labels.contiguous()
        inv_t = inv_temperature.contiguous().float()
        hidden, weight_c = hidden.contiguous(), weight.contiguous()
        logits = torch.empty(n, vocab, dtype=torch.bfloat16, device=dev)
        own = ops.gemm_supported(hidden[:1], weight_c, out=logits[:1])  # the result rows too: a vocabulary that is not a multiple of 8 falls back to the library
        for s0 in range(0, n, chunk_size):
            s1 = min(s0 + chunk_size, n)
            if own:
                ops.gemm(hidden[s0:s1], weight_c, out=logits[s0:s1])
            else:
                torch.mm(hidden[s0:s1], weight_c.t(), out=logits[s0:s1])
            lp, en, lz = ops.logprob_entropy_fwd(logits[s0:s1], labels[s0:s1], inv_t[s0:s1])
            logprobs[s0:s1], entropy[s0:s1], logz[s0:s1] = lp, en, lz
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(hidden, weight_c, labels, inv_t, logz, logits)
        ctx.chunk_size = chunk_size
        ctx.logits_consumed = False
        mg, st = getattr(weight, "main_grad", None), getattr(weight, "main_grad_written", None)
        ctx.main_grad = (mg, st) if (mg is not None and st is not None) else None
        return logprobs, entropy

    @staticmethod
    def backward(ctx, grad_logprobs: Tensor, grad_entropy: Tensor | None):
        assert grad_entropy is None, "Backward through entropy is not implemented in FusedOutputLinear"
        assert grad_logprobs is not None, "FusedOutputLinear backward requires logprobs gradients"
        hidden, weight, labels, inv_t, logz, logits = ctx.saved_tensors
        if ctx.logits_consumed:  # the backward overwrites the stored logits in place with dlogits: a second pass (retain_graph) would read garbage
            raise RuntimeError("FusedOutputLinear (stored logits): backward ran twice on the same forward — the stored logits were consumed in place; "
                               "set PM_MODULE_LM_HEAD=chunked for a re-entrant backward")
        ctx.logits_consumed = True
        n = hidden.shape[0]
        needs_hidden, needs_weight = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_hidden = torch.empty_like(hidden) if needs_hidden else None
        g = grad_logprobs.contiguous().float()
        direct = ctx.main_grad if (needs_weight and ctx.main_grad is not None and ops.gemm_supported(logits[:128].t(), hidden[:128].t())) else None
        grad_weight = torch.zeros_like(weight) if (needs_weight and direct is None) else None
        for s0 in range(0, n, ctx.chunk_size):
            s1 = min(s0 + ctx.chunk_size, n)
            dl = ops.logprob_bwd(logits[s0:s1], labels[s0:s1], inv_t[s0:s1], logz[s0:s1], g[s0:s1], out=logits[s0:s1])
            if needs_hidden:
                if ops.gemm_supported(dl, weight.t(), out=grad_hidden[s0:s1]):
                    ops.gemm(dl, weight.t(), out=grad_hidden[s0:s1])
                else:
                    torch.mm(dl, weight, out=grad_hidden[s0:s1])
            if direct is not None:
                ops.gemm(dl.t(), hidden[s0:s1].t(), out=direct[0], accumulate=direct[1][0])
                direct[1][0] = True
            elif needs_weight:
                grad_weight.addmm_(dl.t(), hidden[s0:s1])
        return grad_hidden, grad_weight, None, None, None


class FusedOutputLinear(torch.nn.Linear):
    def __init__(self, in_features: int, out_features: int, chunk_size: int):
        super().__init__(in_features, out_features, bias=False)
        self.chunk_size = chunk_size

    def forward(self, hidden_states: Tensor, labels: Tensor | None = None, temperature: Tensor | None = None) -> PrimeLmOutput:
        assert labels is not None, "FusedOutputLinear requires labels for chunked logprob computation"
        assert temperature is not None, "FusedOutputLinear requires per-token temperatures"
        b, s, h = hidden_states.shape
        hidden_states = hidden_states.reshape(b * s, h).contiguous()
        labels = labels.reshape(b * s).contiguous()
        inv_t = 1.0 / temperature.reshape(b * s).contiguous()
        # "stored" (default where it applies: bf16 on the GPU, logits within PM_LM_HEAD_LOGITS_GIB, default 24): keep the logits for the backward instead of
        # recomputing them; "chunked": the reference's memory-saving form (lm_head.py:89-202) on the fused J1 kernel
        # (forward-only passes — torch.no_grad(), or nothing that requires a gradient — never need the [N, V] logits afterwards: the chunked form keeps
        # one chunk at a time instead of up to 24 GiB)
        wants_grad = torch.is_grad_enabled() and (hidden_states.requires_grad or self.weight.requires_grad)
        stored = (wants_grad and os.environ.get("PM_MODULE_LM_HEAD", "stored") == "stored" and hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16
                  and self.weight.dtype == torch.bfloat16 and 2 * b * s * self.weight.shape[0] <= float(os.environ.get("PM_LM_HEAD_LOGITS_GIB", "24")) * 2**30)
        fn = _StoredLogitsLogProbEntropyFn if stored else _SequenceChunkedLogProbEntropyFn
        logprobs, entropy = fn.apply(hidden_states, self.weight, labels, inv_t, self.chunk_size)
        return PrimeLmOutput(logprobs=logprobs.reshape(b, s), entropy=entropy.reshape(b, s))


class VanillaOutputLinear(torch.nn.Linear):
    def __init__(self, in_features: int, out_features: int):
        super().__init__(in_features, out_features, bias=False)

    def forward(self, hidden_states: Tensor, labels: Tensor | None = None, temperature: Tensor | None = None) -> PrimeLmOutput:
        return PrimeLmOutput(logits=super().forward(hidden_states))  # temperature scaling happens in the train loop (train.py:436-443)


def inject_prime_lm_head(model: nn.Module, chunk_size: int | None = None) -> None:
    """Replace `model.lm_head` by a PrimeRL head sharing the same weight Parameter and rebind `model.forward` so labels and
    temperature reach it (lm_head.py:205-291).  Gemma soft-capping heads are outside this path."""
    assert hasattr(model, "model"), f"model doesnt have backbone in model.model:\n{model}"
    assert isinstance(model.model, nn.Module), f"model.model is not a nn.Module: {type(model.model)}\n{model}"
    assert hasattr(model, "lm_head"), f"model doesnt have lm_head in model.lm_head:\n{model}"
    assert isinstance(model.lm_head, nn.Linear), f"model.lm_head is not a nn.Linear: {type(model.lm_head)}\n{model}"
    assert not hasattr(model.lm_head, "bias") or model.lm_head.bias is None, f"model.lm_head.bias is not supported: {model.lm_head}\n{model}"
    # The reference dispatches Gemma-style models (final_logit_softcapping) to inject_gemma_lm_head (lm_head.py:214-222): that head is
    # outside this path, and the plain fused head would silently produce wrong log-probs / entropy for them — refuse loudly.
    cfg = getattr(model, "config", None)
    for c in (cfg, getattr(cfg, "text_config", None)):
        if c is not None and getattr(c, "final_logit_softcapping", None):
            raise NotImplementedError("models with final_logit_softcapping (Gemma) need the soft-capping LM head, which is outside the MI355X "
                                      "GRPO hot path (SURVEY.md section 2 row 12)")
    old = model.lm_head
    if isinstance(chunk_size, int):
        model.lm_head = FusedOutputLinear(in_features=old.in_features, out_features=old.out_features, chunk_size=chunk_size)
    else:
        model.lm_head = VanillaOutputLinear(in_features=old.in_features, out_features=old.out_features)
    model.lm_head.weight = old.weight
    del old

    def new_forward(self, input_ids=None, position_ids=None, inputs_embeds=None, labels=None, logits_to_keep: int = 0, temperature=None, **kwargs):
        if position_ids is None and kwargs.get("pixel_values") is None:
            ref = input_ids if input_ids is not None else inputs_embeds
            position_ids = torch.arange(1, ref.shape[1] + 1, device=ref.device).unsqueeze(0)
        mk = {"input_ids": input_ids, "position_ids": position_ids, **kwargs}
        if inputs_embeds is not None:
            mk["inputs_embeds"] = inputs_embeds
        hs = self.model(**mk).last_hidden_state
        sl = slice(-logits_to_keep, None) if isinstance(logits_to_keep, int) and logits_to_keep > 0 else slice(None)
        return self.lm_head(hs[:, sl, :], labels[:, sl] if labels is not None else None, temperature=temperature[:, sl] if temperature is not None else None)

    model.forward = types.MethodType(new_forward, model)
