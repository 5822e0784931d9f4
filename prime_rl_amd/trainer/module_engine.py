"""The trainer step for nn.Module models — the Qwen3-MoE family under expert parallelism (BASELINE.json configs[3]).

`GRPOEngine` (trainer/engine.py) is a flat-buffer schedule written for the dense decoder; models that run through `setup_model()` /
`forward()` (trainer/model.py — the reference's own seams, src/prime_rl/trainer/model.py:1314,1431) get the same step here, with the
same result contract as `GRPOEngine.step`, so `train(config)`, the step statistics and the filesystem weight broadcast are shared:

  scale all-reduce -> per pass: model forward (HIP layer seams, fused LM head -> log-probs + entropy) -> the fused GRPO loss kernel
  (loss, d loss / d log-prob, per-sample metrics: the K2 launch the dense engine uses) -> autograd backward seeded with that gradient ->
  f32 gradient accumulation -> gradient reduction -> global-norm clip -> fused AdamW on f32 master weights -> bf16 parameters.

What the reference does here (train.py:337-480 micro-batch loop, FSDP2 reduce-scatter with `reduce_dtype=float32`, `scale_gradients_`,
`clip_grad_norm_` with the expert-parallel mesh, torch AdamW on f32 master parameters; MoE specifics in model.py:823-960 and
distributed/expert_parallel.py) maps as follows on a 288 GB part:

  * replicated parameters (attention, norms, router gates, embeddings, LM head): one f32 all-reduce(SUM) of their contiguous gradient
    region over every trainer rank.  SUM, not mean: the loss is divided by the GLOBAL token count (train.py:307-323, 557-558).
  * expert parameters: each rank owns E / ep experts and has already seen the tokens every rank of its ep group routed to them (the
    dispatch all-to-all), so their gradients are complete within the ep group; they are summed over the ranks that hold the SAME
    experts in other ep groups (`ep_replica`), nothing when ep == world.
  * the global gradient norm counts replicated parameters once and expert parameters once per ep position:
    sum_sq(replicated) + all-reduce over the ep group of sum_sq(local experts).
  * master weights / Adam moments: flat f32, replicated for the replicated parameters (1.5 B of Qwen3-30B-A3B's 30.5 B), local for the
    experts — the experts are what is big, and expert parallelism already shards them.

Parameters are re-pointed at one flat bf16 buffer (f32 parameters such as an f32 router gate at the f32 master buffer itself), so the
optimizer is ONE fused `pm_adamw_step` launch and gradient reduction is one collective per class."""

from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .. import _lib, ops
from .._lib import LossParams, PrimeMI355XError
from . import zero
from .engine import LossConfig, OptimConfig, PackedBatch
from .models.layers.moe import GroupedExperts


@dataclass
class ModuleDims:
    """What train() reads from `engine.dims`: vocabulary size and the reference's flop model (trainer/perf.py:110-215) for any HF-style config."""

    config: object
    name: str = "module"

    @property
    def vocab_size(self) -> int:
        return self.config.vocab_size

    def active_mm_params(self) -> int:
        c = self.config
        h, L = c.hidden_size, c.num_hidden_layers
        attn = L * (2 * h * c.num_attention_heads * c.head_dim + 2 * h * c.num_key_value_heads * c.head_dim)
        k = getattr(c, "num_experts_per_tok", None)
        if k:  # every layer counted sparse, as the reference does once `num_experts_per_tok` exists (perf.py:148-150)
            moe_i = getattr(c, "moe_intermediate_size", None) or c.intermediate_size
            mlp = L * (k * 3 * moe_i * h + c.num_experts * h)
        else:
            mlp = L * 3 * c.intermediate_size * h
        return attn + mlp + c.vocab_size * h

    def flops_per_token(self, seq_len: int) -> int:
        c = self.config
        return 6 * self.active_mm_params() + 12 * c.num_hidden_layers * c.num_attention_heads * (c.hidden_size // c.num_attention_heads) * seq_len


def _align(n: int, a: int = 128) -> int:
    return (n + a - 1) // a * a


class ModuleEngine:
    def __init__(self, model: nn.Module, device, optim: OptimConfig | None = None, loss: LossConfig | None = None, dp_group=None, dp_size: int = 1,
                 ep_group=None, ep_size: int = 1, ep_replica_group=None, name: str = "module", router_replay: bool = False):
        self.model, self.device = model, torch.device(device)
        self.optim, self.loss_cfg = optim or OptimConfig(), loss or LossConfig()
        self.dp_group, self.dp_size = dp_group, dp_size
        self.ep_group, self.ep_size, self.ep_replica_group = ep_group, ep_size, ep_replica_group
        if ep_size > 1 and dp_size % ep_size:
            raise PrimeMI355XError(f"ep = {ep_size} must divide the {dp_size} trainer ranks")
        if dp_size > ep_size > 1 and ep_replica_group is None:
            raise PrimeMI355XError("ranks that hold the same experts in different ep groups need the `ep_replica` group to sum their gradients")
        self.dims = ModuleDims(model.config, name)
        # `enable_router_replay` (train.py:349-360): the experts the inference engine picked are replayed instead of the trainer's own top-k;
        # payloads that arrive while it is off are ignored, a missing payload while it is on is an error
        self.router_replay = router_replay
        self.step_count = 0
        self.cp_size = 1
        # -- flat storage: replicated parameters first, expert parameters after them ---------------------------------------------------
        expert_ids = {id(p) for m in model.modules() if isinstance(m, GroupedExperts) for p in m.parameters(recurse=False)} if ep_size > 1 else set()
        seen, replicated, experts = set(), [], []
        for pname, p in model.named_parameters():
            if id(p) in seen or not p.requires_grad:
                continue
            seen.add(id(p))
            (experts if id(p) in expert_ids else replicated).append((pname, p))
        self.entries: list[tuple[str, nn.Parameter, int]] = []
        cursor = 0
        for pname, p in replicated:
            self.entries.append((pname, p, cursor))
            cursor = _align(cursor + p.numel())
        self.n_replicated = cursor = _align(cursor, 256)  # [0, n_replicated): one all-reduce over every trainer rank
        for pname, p in experts:
            self.entries.append((pname, p, cursor))
            cursor = _align(cursor + p.numel())
        self.total = _align(cursor, 256)
        dev = self.device
        self.params_bf16 = torch.zeros(self.total, dtype=torch.bfloat16, device=dev)
        self.master = torch.zeros(self.total, dtype=torch.float32, device=dev)
        n_m, n_v = {"adamw": (self.total, self.total), "sgd": (self.total, 0), "sign_sgd": (0, 0)}[self.optim.type]
        self.exp_avg = torch.zeros(n_m, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_v, dtype=torch.float32, device=dev)
        self.main_grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self._grad_views: list[Tensor] = []
        with torch.no_grad():
            for pname, p, off in self.entries:
                n = p.numel()
                if p.dtype not in (torch.bfloat16, torch.float32):
                    raise PrimeMI355XError(f"{pname}: parameters are bf16 (compute) or f32 (e.g. an f32 router gate), got {p.dtype}")
                self.master[off:off + n].copy_(p.detach().reshape(-1).float())
                if p.dtype == torch.bfloat16:
                    self.params_bf16[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = self.params_bf16[off:off + n].view(p.shape)
                else:  # an f32 parameter IS its master copy; its bf16 shadow is scratch for the fused kernel
                    p.data = self.master[off:off + n].view(p.shape)
                self._grad_views.append(self.main_grad[off:off + n].view(p.shape))
        # Expert parameters (97 % of an MoE model's weights) and the LM head get their gradient written / accumulated in f32 by the weight-gradient GEMM itself
        # (layers/moe.py `_main_grads`, layers/lm_head.py `_StoredLogitsLogProbEntropyFn`): no bf16 `.grad` tensor and no bf16 -> f32 accumulation pass over them.  PM_MOE_MAIN_GRAD=0 is the A/B switch.
        self._direct: list[nn.Parameter] = []
        if self.device.type == "cuda" and os.environ.get("PM_MOE_MAIN_GRAD", "1") != "0":
            from .models.layers.lm_head import FusedOutputLinear

            from .models.layers.attn import HipVarlenAttention
            from .models.layers.mlp import MLP
            from .models.layers.moe import TokenChoiceTopKRouter

            direct_ids = {id(p) for m in model.modules() if isinstance(m, GroupedExperts) for p in m.parameters(recurse=False)}
            direct_ids |= {id(m.weight) for m in model.modules() if isinstance(m, FusedOutputLinear)}  # the LM head's weight gradient GEMM (stored-logits form)
            # the dense projections of the layer seams run on the hand-written core as autograd nodes (layers/linear.py) whose weight-gradient GEMM writes /
            # accumulates in f32 here as well: q / k / v / o_proj, gate / up / down_proj, the router gate — where the core takes the shape (rows and columns
            # multiples of 8, >= 128); anything else keeps the `.grad` path below
            for m in model.modules():
                lins = []
                if isinstance(m, HipVarlenAttention):
                    lins = [m.q_proj, m.k_proj, m.v_proj, m.o_proj]
                elif isinstance(m, MLP):
                    lins = [m.gate_proj, m.up_proj, m.down_proj]
                elif isinstance(m, TokenChoiceTopKRouter) and not m.fp32_gate:
                    lins = [m.gate]
                for lin in lins:
                    o, i = lin.weight.shape
                    if o % 128 == 0 and i % 128 == 0:
                        direct_ids.add(id(lin.weight))
            for (pname, p, off), gv in zip(self.entries, self._grad_views):
                if id(p) in direct_ids and p.dtype == torch.bfloat16:
                    p.main_grad, p.main_grad_written = gv, [False]
                    self._direct.append(p)
        # The optimizer launch zeroes the gradient it has consumed (zero_grad fused into the update) — except over DIRECT parameters: their first
        # weight-gradient launch of a step OVERWRITES the slice, so a zero written there (4 of the update's 34 bytes per parameter, over 97 % of an MoE
        # model) is never read.  Spans of consecutive entries of one kind, in flat-buffer order: (start, end, zero_grad).
        direct_ids = {id(p) for p in self._direct}
        self._update_spans: list[tuple[int, int, bool]] = []
        for i, (pname, p, off) in enumerate(self.entries):
            end = self.entries[i + 1][2] if i + 1 < len(self.entries) else self.total
            zero = id(p) not in direct_ids
            if self._update_spans and self._update_spans[-1][2] == zero:
                self._update_spans[-1] = (self._update_spans[-1][0], end, zero)
            else:
                self._update_spans.append((off if self._update_spans else 0, end, zero))
        self._sumsq = torch.zeros(2, dtype=torch.float32, device=dev)
        self._param_readers: list = []
        # The gradient norm of the expert parameters out of their weight-gradient epilogues (ops.grouped_gemm_tn_f32(tile_sumsq=): every launch leaves the sum
        # of squares of the FINAL values of every output tile in a small workspace) instead of a pass over 97 % of the gradient buffer — what the flat
        # engine's `fused_sumsq` is to its dense weight gradients.  Only where the norm is taken over what the launches wrote: a gradient that is summed over
        # ranks afterwards (dp > 1 replicas of the same experts) has no such partials.  PM_MOE_FUSED_NORM=0 is the A/B switch.
        self._fused_norm: list[tuple[Tensor, list, int, list[tuple[int, int]]]] = []  # (slots, [valid this step], partition 0 / 1, covered [a, b) ranges)
        self._norm_rest: list[tuple[int, int, int]] = []  # (a, b, partition): what the pass still reads when every launch's slots are valid
        if self._direct and os.environ.get("PM_MOE_FUSED_NORM", "1") != "0":
            offs = {id(p): off for _, p, off in self.entries}
            direct_ids = {id(p) for p in self._direct}
            sizes = []
            for m in model.modules():
                if not isinstance(m, GroupedExperts) or not all(id(q) in direct_ids for q in (m.w1, m.w2, m.w3)):
                    continue
                e, inter, hidden = m.w1.shape
                part = 1 if offs[id(m.w1)] >= self.n_replicated else 0
                summed_later = self.dp_size > 1 and (part == 0 or self.ep_size == 1 or self.dp_size > self.ep_size)
                if summed_later or not (ops.grouped_gemm_tn_f32_supported(inter, 2 * inter, hidden) and ops.grouped_gemm_tn_f32_supported(hidden, hidden, inter)):
                    continue
                sizes.append((m.w1, [m.w1, m.w3], ops.grouped_gemm_tn_sumsq_slots(e, 2 * inter, hidden), part))
                sizes.append((m.w2, [m.w2], ops.grouped_gemm_tn_sumsq_slots(e, hidden, inter), part))
            if sizes:
                self._fs_ws = torch.zeros(sum(n for _, _, n, _ in sizes), dtype=torch.float32, device=dev)
                cur, covered = 0, []
                for first, ps, n, part in sizes:
                    rng = [(offs[id(q)], offs[id(q)] + q.numel()) for q in ps]
                    first.main_grad_sumsq = (self._fs_ws[cur:cur + n], [False])
                    self._fused_norm.append((self._fs_ws[cur:cur + n], first.main_grad_sumsq[1], part, rng))
                    covered += rng
                    cur += n
                covered.sort()
                pos = 0
                for a, b in covered + [(self.total, self.total)]:  # the complement of the covered ranges, cut at the partition boundary
                    for lo, hi, part in ((pos, min(a, self.n_replicated), 0), (max(pos, self.n_replicated), a, 1)):
                        if hi > lo:
                            self._norm_rest.append((lo, hi, part))
                    pos = max(pos, b)

    # -- GRPOEngine-compatible surface used by train() and the weight broadcast -----------------------------------------------------
    def wait_params_ready(self) -> None:
        return None

    def add_param_reader(self, event) -> None:
        """Completion event of an asynchronous reader of the parameters (the weight push stream): the next optimizer step waits for it."""
        self._param_readers.append(event)

    @property
    def collective_state_dict(self) -> bool:
        """True when state_dict() runs collectives (every rank must call it, not only the rank that writes the checkpoint)."""
        return self.ep_size > 1

    def state_dict(self) -> dict[str, Tensor]:
        """HF-named bf16 tensors of the WHOLE model (what the inference engine loads, transports/weights): expert shards are all-gathered
        over the ep group, MoE names converted like the reference's `revert_weight_conversion` (transports/weights/nccl.py:87-98)."""
        sd = {k: v.detach() for k, v in self.model.state_dict().items()}
        sd = {k: (v if v.dtype == torch.bfloat16 or not v.is_floating_point() else v.to(torch.bfloat16)) for k, v in sd.items()}
        if self.ep_size > 1:
            for key in [k for k in sd if ".mlp.experts.w" in k]:
                local = sd[key].contiguous()
                full = torch.empty((local.shape[0] * self.ep_size, *local.shape[1:]), dtype=local.dtype, device=local.device)
                dist.all_gather_into_tensor(full, local, group=self.ep_group)
                sd[key] = full
        if getattr(self.model.config, "num_experts", None):
            from .models.qwen3_moe.converting_qwen3_moe import prime_to_hf

            sd = prime_to_hf(sd)
        return sd

    # -- the sharded ("named") weight push: every trainer rank sends what it holds, nothing is gathered first --------------------------------
    def push_plan(self, n_trainers: int) -> list[dict]:
        """Who sends what in the multi-source push of an nn.Module engine (transports/weights `protocol="sharded"`): a list of groups (the non-layer
        tensors, then one group per decoder layer — the order a layer-wise loader wants), each a list of messages `{src, shape, names | experts}`.
        Replicated tensors travel whole, round-robin over the trainer ranks (every rank holds them; every rank sends 1 / T of the bytes over its own
        links); the stacked expert tensors of an expert-parallel model are sent BY THE RANK THAT OWNS THEM (ep position e of the first ep group), one
        message per stacked tensor, and the receiver cuts the per-expert HF tensors out of it — no all-gather of the experts over the ep group and no
        single-source funnel (the reference: full_tensor() on every parameter, then rank 0 broadcasts everything, transports/weights/nccl.py:33-84).
        Computed from names and shapes alone, so every rank (and the manifest the master writes) agrees without communication."""
        from .models.qwen3_moe.converting_qwen3_moe import _PROJ

        if n_trainers != self.dp_size:
            raise PrimeMI355XError(f"push_plan: {n_trainers} trainer ranks in the push group, {self.dp_size} in the engine")
        groups: dict[str, list[dict]] = {}
        rr = 0
        for key, t in self.model.state_dict().items():
            gname = ".".join(key.split(".")[:3]) if key.startswith("model.layers.") else ""
            msgs = groups.setdefault(gname, [])
            hit = next(((wn, hf) for wn, hf in _PROJ if key.endswith(f".mlp.experts.{wn}")), None)
            if hit is not None and self.ep_size > 1:
                n_local = t.shape[0]
                for e in range(self.ep_size):  # ep groups are blocks of `ep` consecutive trainer ranks: position e of the first block sends shard e
                    msgs.append(dict(src=e, shape=list(t.shape), experts=dict(base=key[: -len(hit[0])], proj=hit[1], first=e * n_local), key=key))
                continue
            if hit is not None:
                msgs.append(dict(src=rr % n_trainers, shape=list(t.shape), experts=dict(base=key[: -len(hit[0])], proj=hit[1], first=0), key=key))
            else:
                name = key[: -len("router.gate.weight")] + "gate.weight" if key.endswith(".mlp.router.gate.weight") else key
                msgs.append(dict(src=rr % n_trainers, shape=list(t.shape), names=[name], key=key))
            rr += 1
        order = [""] + sorted((g for g in groups if g), key=lambda g: int(g.split(".")[2]))
        return [dict(name=g or "non_layer", messages=groups[g]) for g in order if g in groups]

    def push_tensors(self, plan: list[dict], rank: int) -> list[list[Tensor | None]]:
        """The bf16 tensors this rank sends, aligned with `plan` (None where another rank is the source)."""
        sd = self.model.state_dict()
        out = []
        for g in plan:
            row = []
            for m in g["messages"]:
                if m["src"] != rank:
                    row.append(None)
                    continue
                t = sd[m["key"]].detach()
                if not t.is_floating_point():  # state_dict() keeps integer buffers as they are; the sharded push announces ONE dtype (bf16) per message
                    raise PrimeMI355XError(f"push_tensors: {m['key']} is {t.dtype}; the sharded weight push carries floating tensors as bf16 only")
                row.append((t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)).contiguous())
            out.append(row)
        return out

    def global_scales(self, local_counts: tuple[int, int, int]) -> tuple[int, int, int]:
        if self.dp_size == 1:
            return tuple(max(int(c), 1) for c in local_counts)
        return zero.all_reduce_counts(local_counts, self.dp_group, self.device)

    # -- one pass -------------------------------------------------------------------------------------------------------------------
    def forward_loss(self, b: PackedBatch, scales: tuple[int, int, int]) -> dict:
        n = b.n_tokens
        seq_lens = (b.cu_seqlens[1:] - b.cu_seqlens[:-1]).to(torch.int64)
        kw = {}
        if self.router_replay:
            if b.routed_experts is None:
                raise PrimeMI355XError("enable_router_replay needs routed_experts in the rollouts: set `enable_return_routed_experts=True` in the inference "
                                       "config or pass `--enable-return-routed-experts` to the vLLM server (train.py:353-357)")
            kw["routed_experts"] = b.routed_experts.unsqueeze(0)
        out = self.model(input_ids=b.input_ids.view(1, n), position_ids=b.position_ids.view(1, n), labels=b.labels.view(1, n),
                         temperature=(1.0 / b.inv_temp).view(1, n), seq_lens=seq_lens, **kw)
        if not isinstance(out, dict) or "logprobs" not in out:
            raise PrimeMI355XError("the model must return log-probs and entropy (inject_prime_lm_head with the fused head; trainer/model.py:setup_model does)")
        logprobs = out["logprobs"].reshape(n).float()
        entropy = out["entropy"].reshape(n).float().detach()
        c = self.loss_cfg
        if c.type == "custom":  # LossFn plugin: per-sample Python under autograd on the detached log-probs (trainer/rl/loss.py)
            from .rl.loss import plugin_loss_over_pass

            loss, grad, per_mb, metrics = plugin_loss_over_pass(logprobs, b, scales, c.custom_fn, self.dims.vocab_size)
            return dict(loss=loss, grad_logprobs=grad, sample_stats=None, sample_metrics=dict(loss_per_micro_batch=per_mb, metrics=metrics),
                        logprobs=logprobs, entropy=entropy)
        params = LossParams(loss_type=_lib.PM_LOSS_IPO if c.type == "ipo" else _lib.PM_LOSS_DEFAULT, dppo_mask_low=c.dppo_mask_low,
                            dppo_mask_high=c.dppo_mask_high, adv_tau=c.adv_tau, kl_tau=c.kl_tau, ipo_threshold=c.ipo_threshold,
                            rl_scale=float(max(scales[0], 1)), ce_scale=float(max(scales[1], 1)), ref_kl_scale=float(max(scales[2], 1)), shift=1,
                            pad_logprob=math.log(1.0 / self.dims.vocab_size))
        loss, grad, stats, metrics = ops.grpo_loss_fwd_bwd(logprobs.detach().contiguous(), b.inference_logprobs, b.advantages, b.loss_mask, b.cu_samples,
                                                           params, ref_logprobs=b.ref_logprobs, rl_weights=b.rl_weights, ce_weights=b.ce_weights,
                                                           ref_kl_weights=b.ref_kl_weights, sample_starts_mb=b.sample_starts_mb)
        return dict(loss=loss, grad_logprobs=grad, sample_stats=stats, sample_metrics=metrics, logprobs=logprobs, entropy=entropy)

    def backward(self, ctx: dict) -> None:
        """d loss / d parameters through the autograd graph, seeded with the loss kernel's d loss / d log-prob; gradients land in f32."""
        ctx["logprobs"].backward(ctx["grad_logprobs"])
        with torch.no_grad():
            for (pname, p, off), gv in zip(self.entries, self._grad_views):
                if p.grad is None:
                    continue
                st = getattr(p, "main_grad_written", None)
                if st is not None and not st[0]:  # a direct parameter whose gradient came through autograd after all (a shape the f32-writing GEMM does not
                    gv.zero_()                    # take): its slice is not zeroed by the optimizer launch (_update_spans) and still holds the consumed gradient
                    st[0] = True
                if p.grad.dtype == torch.bfloat16 and p.numel() % 8 == 0:
                    ops.grad_accumulate(gv.view(-1), p.grad.contiguous().view(-1))
                else:
                    gv.add_(p.grad)
                p.grad = None

    # -- optimizer ------------------------------------------------------------------------------------------------------------------
    def optimizer_step(self) -> Tensor:
        self.step_count += 1
        for ev in self._param_readers:  # an in-flight weight push still reads the parameters this step is about to overwrite
            if self.device.type == "cuda":
                torch.cuda.current_stream().wait_event(ev)
        self._param_readers.clear()
        nr, g = self.n_replicated, self.main_grad
        for p in self._direct:  # a direct parameter no backward pass of this step wrote (unused / frozen path): its slice still holds the previous step's gradient
            if not p.main_grad_written[0]:
                p.main_grad.zero_()
        if self.dp_size > 1:
            if nr:
                dist.all_reduce(g[:nr], op=dist.ReduceOp.SUM, group=self.dp_group)
            if self.total > nr and (self.ep_size == 1 or self.dp_size > self.ep_size):
                dist.all_reduce(g[nr:], op=dist.ReduceOp.SUM, group=self.ep_replica_group if self.ep_size > 1 else self.dp_group)
        self._sumsq.zero_()
        if self._fused_norm and all(valid[0] for _, valid, _, _ in self._fused_norm):
            # every expert weight-gradient launch of this step left its tiles' sums of squares: read those (a few MB) and pass only over the rest
            for a, b, part in self._norm_rest:
                ops.sumsq_accumulate(g[a:b], self._sumsq[part:part + 1])
            for part in (0, 1):
                sl = [ws for ws, _, pt, _ in self._fused_norm if pt == part]
                if sl:  # the slots of one partition are one contiguous stretch of the workspace
                    lo = sl[0].storage_offset()
                    self._sumsq[part:part + 1] += self._fs_ws[lo:lo + sum(w.numel() for w in sl)].sum()
        else:
            if nr:
                ops.sumsq_accumulate(g[:nr], self._sumsq[0:1])
            if self.total > nr:
                ops.sumsq_accumulate(g[nr:], self._sumsq[1:2])
        for _, valid, _, _ in self._fused_norm:
            valid[0] = False
        if self.total > nr and self.ep_size > 1:
            dist.all_reduce(self._sumsq[1:2], op=dist.ReduceOp.SUM, group=self.ep_group)
        total = self._sumsq.sum().reshape(1)
        o = self.optim
        coef, norm = ops.clip_coef(total, 1.0, o.max_norm if o.max_norm else 0.0)
        for a, b, zero_grad in self._update_spans:
            zero.apply_update(ops, o, self.master[a:b], self.exp_avg[a:b] if self.exp_avg.numel() else None, self.exp_avg_sq[a:b] if self.exp_avg_sq.numel() else None,
                              g[a:b], self.params_bf16[a:b], self.step_count, coef, zero_grad=zero_grad)
        for p in self._direct:
            p.main_grad_written[0] = False  # consumed: the next step's first weight-gradient launch overwrites the slice (it is NOT zeroed, see _update_spans)
        return norm

    def step(self, batch: PackedBatch | list[PackedBatch]) -> dict:
        passes = batch if isinstance(batch, list) else [batch]
        counts = tuple(sum(p.local_counts[k] for p in passes) for k in range(3))
        scales = self.global_scales(counts)
        out = {"loss": [], "entropy": [], "logprobs": [], "sample_metrics": [], "sample_stats": [], "load_balance": []}
        moe = bool(getattr(self.model.config, "num_experts", None))
        for p in passes:
            ctx = self.forward_loss(p, scales)
            if moe:  # per-layer load violation and routing confidence of this pass, counters reset (train.py:531-535, model.py:561-592); the reference
                # reads them per micro-batch — a pass is `tokens_per_pass // seq_len` micro-batches routed together
                from .model import get_load_balance_stats

                out["load_balance"].append(get_load_balance_stats(self.model))
            self.backward(ctx)
            out["loss"].append(ctx["loss"])
            out["entropy"].append(ctx["entropy"])
            out["logprobs"].append(ctx["logprobs"].detach())
            out["sample_metrics"].append(ctx["sample_metrics"])
            out["sample_stats"].append(ctx["sample_stats"])
            del ctx
        out["grad_norm"] = self.optimizer_step()
        out["scales"] = scales
        return out
