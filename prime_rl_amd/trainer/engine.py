"""The MI355X-native GRPO policy-update engine: one process per GPU, flat parameter/gradient buffers, explicit forward
and backward schedules over the C-ABI kernels (no autograd graph, no tracing compiler), RCCL collectives on a side stream.

What it replaces in the reference (src/prime_rl/): the body of the training step in trainer/rl/train.py:307-570 —
scale all-reduce (:307-323), the micro-batch loop (:337-480: forward() of trainer/model.py:1431-1482 through
Qwen3ForCausalLM, the LM-head log-probs, shift_tensor_right, compute_loss, loss.backward()), scale_gradients_ (:557-558),
clip_grad_norm_ (:561-565) and optimizer.step() (:566-570) — and the FSDP2 sharding of trainer/model.py:823-960.

MI355X-first choices (DESIGN.md has the numbers):
  * 288 GB HBM: no activation checkpointing, no activation/optimizer offload, no per-layer parameter resharding.
    All micro-batches of a step are concatenated into ONE packed pass (varlen attention + segmented loss keep the
    reference's per-micro-batch semantics), so every weight-gradient GEMM reduces over the whole step in f32 inside the
    MFMA accumulators and gradients are written once.
  * ZeRO-2 style data parallelism instead of ZeRO-3: bf16 parameters are replicated (16 GB of 288 GB for Qwen3-8B), f32
    gradients are reduce-scattered per layer bucket while backward is still running, master weights + Adam moments are
    sharded, updated bf16 shards are all-gathered in place.  One reduce-scatter + one all-gather of the model per step
    instead of two all-gathers per layer per micro-batch (xGMI is per-link bound: fewer, larger collectives).
  * Every weight-gradient GEMM runs on the hand-written MFMA core of libprime_mi355x.so (csrc/gemm.hip: token-major operands read as they
    are, f32 accumulation straight into the gradient buffer, the gradient norm out of the epilogue); the forward / dgrad products are
    hipBLASLt calls by default because the library measures 5-10 % faster there (`PM_GEMM_OWN=all` puts them on the own core as well:
    profiles/r03_gemm.md, r04_gemm.md); everything else on the path is a kernel in libprime_mi355x.so.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import torch
import torch.distributed as dist
from torch import Tensor

from .. import _lib, ops
from ..utils.rope import scaled_inv_freq
from . import ulysses, zero
from .._lib import LossParams, PrimeMI355XError


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class ModelDims:
    """Dense Qwen3 / Llama / Qwen2 dimensions (HF config names).  `qk_norm` / `attention_bias` are AttentionConfig.use_qk_norm /
    .attention_bias of the reference (models/layers/attn.py:29-40): Qwen3 = per-head q/k RMSNorm, no bias; Llama = neither
    (models/llama/modeling_llama.py:52-62); Qwen2 / Qwen2.5 = bias on the q/k/v projections, no q/k norm (the reference runs it through
    HF's class)."""

    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    tie_word_embeddings: bool = False
    name: str = "custom"
    qk_norm: bool = True
    attention_bias: bool = False
    rope_scaling: dict | None = None  # HF `rope_scaling` / `rope_parameters` of type "linear" or "llama3" (utils/rope.py); None = default table

    @classmethod
    def qwen2_5_1_5b(cls) -> "ModelDims":  # public HF config of Qwen/Qwen2.5-1.5B(-Instruct) (BASELINE.json configs[1])
        return cls(151936, 1536, 8960, 28, 12, 2, 128, tie_word_embeddings=True, name="Qwen2.5-1.5B", qk_norm=False, attention_bias=True)

    @classmethod
    def qwen2_5_0_5b(cls) -> "ModelDims":  # Qwen/Qwen2.5-0.5B(-Instruct): BASELINE.json configs[0]; 64-wide heads (run padded to 128)
        return cls(151936, 896, 4864, 24, 14, 2, 64, tie_word_embeddings=True, name="Qwen2.5-0.5B", qk_norm=False, attention_bias=True)

    @classmethod
    def qwen3_8b(cls) -> "ModelDims":  # public HF config
        return cls(151936, 4096, 12288, 36, 32, 8, 128, name="Qwen3-8B")

    @classmethod
    def qwen3_0_6b(cls) -> "ModelDims":
        return cls(151936, 1024, 3072, 28, 16, 8, 128, tie_word_embeddings=True, name="Qwen3-0.6B")

    @classmethod
    def tiny(cls, layers: int = 2, vocab: int = 512) -> "ModelDims":
        return cls(vocab, 256, 512, layers, 4, 2, 128, name="tiny")

    @property
    def kernel_head_dim(self) -> int:
        """Head width the attention / RoPE kernels run at.  They are built for 128-wide heads; a 64-wide head (Qwen2.5-0.5B of BASELINE.json
        configs[0], Llama-3.2-1B) runs ZERO-PADDED to 128 — exact: its two rotary halves [0,32) / [32,64) sit at [0,32) / [64,96) of the padded head,
        the other 64 positions hold zero weights (zero q, k, v columns and zero o_proj inputs contribute nothing and receive zero gradients)."""
        return 128 if self.head_dim == 64 else self.head_dim

    @property
    def head_padded(self) -> bool:
        return self.kernel_head_dim != self.head_dim

    @property
    def q_dim(self) -> int:  # width of the q block of the fused projection AS STORED (padded heads included)
        return self.num_attention_heads * self.kernel_head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.kernel_head_dim

    def num_params(self) -> int:
        h, i, v = self.hidden_size, self.intermediate_size, self.vocab_size
        q, kv = self.num_attention_heads * self.head_dim, self.num_key_value_heads * self.head_dim
        per_layer = 2 * h + (q + 2 * kv) * h + h * q + 3 * i * h
        per_layer += 2 * self.head_dim if self.qk_norm else 0
        per_layer += q + 2 * kv if self.attention_bias else 0
        return v * h * (1 if self.tie_word_embeddings else 2) + self.num_hidden_layers * per_layer + h

    def flops_per_token(self, seq_len: int) -> int:
        """PerfCounter._get_num_flop_per_token (trainer/perf.py:177-213): 6 * active matmul params + 12 * L * H * q * T with
        q = hidden_size // num_attention_heads — the reference's definition, which is NOT head_dim for models such as Qwen3-0.6B
        (1024 // 16 = 64 vs head_dim 128); pinned by the reference's known answers (tests/unit/train/test_perf.py:8-35)."""
        return 6 * self.active_mm_params() + 12 * self.num_hidden_layers * self.num_attention_heads * (self.hidden_size // self.num_attention_heads) * seq_len

    def active_mm_params(self) -> int:
        """PerfCounter.get_active_mm_params (trainer/perf.py:120-175), dense case: q/k/v/o + gate/up/down per layer + LM head."""
        h, i = self.hidden_size, self.intermediate_size
        q, kv = self.num_attention_heads * self.head_dim, self.num_key_value_heads * self.head_dim  # the model's own widths, not the padded storage
        return self.num_hidden_layers * ((q + 2 * kv) * h + h * q + 3 * i * h) + self.vocab_size * h


@dataclass
class OptimConfig:
    """AdamWConfig defaults (packages/prime-rl-configs/src/prime_rl/configs/trainer.py:415-443); `type` "sgd" / "sign_sgd" select the reference's other
    elementwise optimizers (SGDConfig :426-433: momentum 0.9, Nesterov; SignSGDConfig :459-460)."""

    lr: float = 1e-6
    weight_decay: float = 0.01
    betas1: float = 0.9
    betas2: float = 0.999
    eps: float = 1e-8
    max_norm: float = 1.0
    type: str = "adamw"
    momentum: float = 0.9
    nesterov: bool = True


@dataclass
class LossConfig:
    """DefaultLossConfig / IPOLossConfig (configs/trainer.py:494-520)."""

    type: str = "default"
    dppo_mask_low: float = 0.2
    dppo_mask_high: float = 0.2
    adv_tau: float = 1.0
    kl_tau: float = 1e-3
    ipo_threshold: float = 0.1
    # type "custom": a LossFn plugin (loss.py:13-45, selected by import_path + kwargs, configs/trainer.py:522-532) evaluated per sample under autograd
    # on the engine's log-probs (trainer/rl/loss.py: plugin_loss_over_pass); everything else of the step stays the engine's
    custom_fn: object = None


@dataclass
class PackedBatch:
    """All micro-batches of one rank for one step, concatenated along the token axis (device tensors)."""

    input_ids: Tensor  # i64 [n]
    position_ids: Tensor  # i64 [n]
    labels: Tensor  # i64 [n]   shift_tensor_left per micro-batch (loss.py:66-73)
    inv_temp: Tensor  # f32 [n]  1/temperature (lm_head.py:54)
    advantages: Tensor  # f32 [n]
    inference_logprobs: Tensor  # f32 [n]
    loss_mask: Tensor  # bool [n]
    cu_seqlens: Tensor  # i32 [docs+1]     attention documents (MicroBatch.seq_lens)
    cu_samples: Tensor  # i32 [samples+1]  loss samples (MicroBatch.sequence_lengths)
    sample_starts_mb: Tensor  # u8 [samples]
    n_micro_batches: int
    ref_logprobs: Tensor | None = None
    rl_weights: Tensor | None = None
    ce_weights: Tensor | None = None
    ref_kl_weights: Tensor | None = None
    local_counts: tuple[int, int, int] = (0, 0, 0)  # rl / ce / ref_kl member tokens on this rank (train.py:307-317)
    # bool [samples] | None: samples of micro-batches that shipped NO rl stream while another micro-batch of the pass did — the
    # reference runs the rl loss fn (and logs its metrics) for every sample of such a micro-batch (loss.py:367-371)
    rl_every_sample: Tensor | None = None
    routed_experts: Tensor | None = None  # i32 [n, layers, top_k]: the inference engine's expert choices (router replay, train.py:349-360)

    @property
    def n_tokens(self) -> int:
        return self.input_ids.numel()


def pack_micro_batches(micro_batches: list[dict], device: torch.device | str) -> PackedBatch:
    """Concatenate TensorMicroBatch dicts (trainer/rl/data.py:18-52 layout, CPU tensors [1, seq]) into one PackedBatch.
    Token counts per loss component are taken on the host tensors exactly like train.py:307-317."""
    if not micro_batches:
        raise ValueError("empty step")

    def cat(key, dtype):
        return torch.cat([mb[key].reshape(-1).to(dtype) for mb in micro_batches])

    ids = cat("input_ids", torch.int64)
    labels = torch.cat([torch.cat([mb["input_ids"].reshape(-1)[1:], mb["input_ids"].new_zeros(1)]).to(torch.int64) for mb in micro_batches])
    temps = cat("temperatures", torch.float32)
    mask = cat("loss_mask", torch.bool)

    def opt(key):
        present = [mb.get(key) is not None for mb in micro_batches]
        if not any(present):
            return None
        parts = []
        for mb in micro_batches:
            t = mb.get(key)
            if t is None:  # a stream materialises as soon as one packed micro-batch carries it (types.py:96-99)
                n = mb["input_ids"].numel()
                fill = 1.0 if key == "rl_weights" else 0.0
                t = torch.full((n,), fill, dtype=torch.float32)
            parts.append(t.reshape(-1).to(torch.float32))
        return torch.cat(parts)

    rl_w, ce_w, rk_w, ref = opt("rl_weights"), opt("ce_weights"), opt("ref_kl_weights"), opt("ref_logprobs")
    n_rl = int((mask & (rl_w != 0)).sum()) if rl_w is not None else int(mask.sum())
    n_ce = int((ce_w != 0).sum()) if ce_w is not None else 0
    n_rk = int((rk_w != 0).sum()) if rk_w is not None else 0
    seq_lens = torch.cat([mb["seq_lens"].reshape(-1).to(torch.int64) for mb in micro_batches])
    sample_lens, starts, every = [], [], []
    for mb in micro_batches:
        sl = list(mb["sequence_lengths"])
        sample_lens += sl
        starts += [1] + [0] * (len(sl) - 1)
        every += [mb.get("rl_weights") is None] * len(sl)
    cu_seq = torch.zeros(seq_lens.numel() + 1, dtype=torch.int32)
    cu_seq[1:] = seq_lens.cumsum(0)
    cu_smp = torch.zeros(len(sample_lens) + 1, dtype=torch.int32)
    cu_smp[1:] = torch.tensor(sample_lens, dtype=torch.int64).cumsum(0)
    if int(cu_seq[-1]) != ids.numel() or int(cu_smp[-1]) != ids.numel():
        raise ValueError("seq_lens / sequence_lengths must sum to the packed length (trainer/batch.py:_assert_token_arrays_aligned)")

    def dev(t):
        return None if t is None else t.to(device, non_blocking=True)

    routed = [mb.get("routed_experts") for mb in micro_batches]
    if any(r is not None for r in routed) and not all(r is not None for r in routed):
        raise ValueError("router replay: every micro-batch of a pass carries routed_experts, or none does")
    routed = torch.cat([r.reshape((-1,) + tuple(r.shape[-2:])).to(torch.int32) for r in routed]) if routed[0] is not None else None
    if routed is not None and routed.shape[0] != ids.numel():
        raise ValueError(f"routed_experts rows ({routed.shape[0]}) != tokens ({ids.numel()})")

    return PackedBatch(
        input_ids=dev(ids), position_ids=dev(cat("position_ids", torch.int64)), labels=dev(labels), inv_temp=dev(1.0 / temps),
        advantages=dev(cat("advantages", torch.float32)), inference_logprobs=dev(cat("inference_logprobs", torch.float32)), loss_mask=dev(mask),
        cu_seqlens=dev(cu_seq), cu_samples=dev(cu_smp), sample_starts_mb=dev(torch.tensor(starts, dtype=torch.uint8)),
        n_micro_batches=len(micro_batches), ref_logprobs=dev(ref), rl_weights=dev(rl_w), ce_weights=dev(ce_w), ref_kl_weights=dev(rk_w),
        local_counts=(n_rl, n_ce, n_rk), rl_every_sample=dev(torch.tensor(every, dtype=torch.bool)) if (rl_w is not None and any(every)) else None,
        routed_experts=dev(routed),
    )


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class _Bucket:
    name: str
    offset: int  # element offset in the flat buffers
    size: int  # padded size (multiple of 256 * dp)
    params: list[tuple[str, int, tuple[int, ...]]] = field(default_factory=list)  # (name, offset within flat, shape)


def _align(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class ParamStore:
    """Flat bf16 parameters + flat f32 gradients, bucketed per decoder layer; f32 master/Adam state sharded over dp.

    Fused storage: q|k|v projections are one [(Hq+2Hkv)*D, H] matrix and gate|up one [2I, H] matrix (one GEMM each);
    `state_dict()` exposes HF-named views (q_proj.weight, k_proj.weight, ...) so checkpoints and the weight push keep the
    reference's names (trainer/models/qwen3/modeling_qwen3.py parameter names)."""

    def __init__(self, dims: ModelDims, device, dp_rank: int = 0, dp_size: int = 1, force_collectives: bool = False, optimizer: str = "adamw"):
        self.dims, self.device, self.dp_rank, self.dp_size = dims, torch.device(device), dp_rank, dp_size
        d = dims
        buckets: list[_Bucket] = []
        cursor = 0

        def new_bucket(name, plist):
            nonlocal cursor
            b = _Bucket(name, cursor, 0)
            off = cursor
            for pname, shape in plist:
                b.params.append((pname, off, shape))
                off = _align(off + math.prod(shape), 128)
            b.size = _align(off - cursor, 256 * dp_size)
            cursor += b.size
            buckets.append(b)

        new_bucket("embed", [("model.embed_tokens.weight", (d.vocab_size, d.hidden_size))])
        for i in range(d.num_hidden_layers):
            p = f"model.layers.{i}."
            plist = [(p + "input_layernorm.weight", (d.hidden_size,)), (p + "self_attn.qkv_proj.weight", (d.q_dim + 2 * d.kv_dim, d.hidden_size))]
            if d.attention_bias:
                plist.append((p + "self_attn.qkv_proj.bias", (d.q_dim + 2 * d.kv_dim,)))
            if d.qk_norm:
                plist += [(p + "self_attn.q_norm.weight", (d.head_dim,)), (p + "self_attn.k_norm.weight", (d.head_dim,))]
            plist += [
                (p + "self_attn.o_proj.weight", (d.hidden_size, d.q_dim)),
                (p + "post_attention_layernorm.weight", (d.hidden_size,)),
                (p + "mlp.gate_up_proj.weight", (2 * d.intermediate_size, d.hidden_size)),
                (p + "mlp.down_proj.weight", (d.hidden_size, d.intermediate_size)),
            ]
            new_bucket(f"layer{i}", plist)
        head = [("model.norm.weight", (d.hidden_size,))]
        if not d.tie_word_embeddings:
            head.append(("lm_head.weight", (d.vocab_size, d.hidden_size)))
        new_bucket("head", head)
        self.buckets = buckets
        self.total = cursor
        self.shard_total = cursor // dp_size
        self.params = torch.empty(self.total, dtype=torch.bfloat16, device=self.device)
        self.main_grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.master = torch.empty(self.shard_total, dtype=torch.float32, device=self.device)
        # optimizer state per parameter: AdamW two moments, SGD one momentum buffer (kept in exp_avg), SignSGD none
        n_m, n_v = {"adamw": (self.shard_total, self.shard_total), "sgd": (self.shard_total, 0), "sign_sgd": (0, 0)}[optimizer]
        self.exp_avg = torch.zeros(n_m, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n_v, dtype=torch.float32, device=self.device)
        # use_collectives: True whenever the reduce-scatter / all-gather path runs (dp > 1, or forced on one rank so the RCCL +
        # side-stream plumbing can be exercised on a single GPU: the collectives then degenerate to device copies)
        self.use_collectives = dp_size > 1 or force_collectives
        self.grad_shard = self.main_grad if not self.use_collectives else torch.empty(self.shard_total, dtype=torch.float32, device=self.device)
        self.version = 0  # bumped whenever the bf16 parameters change (derived copies, e.g. the dgrad-oriented weights, key on it)
        self.w: dict[str, Tensor] = {}
        self.g: dict[str, Tensor] = {}
        for b in buckets:
            for pname, off, shape in b.params:
                n = math.prod(shape)
                self.w[pname] = self.params[off:off + n].view(shape)
                self.g[pname] = self.main_grad[off:off + n].view(shape)

    # -- init ------------------------------------------------------------------------------------------------------
    def random_init(self, seed: int = 0, std: float = 0.02) -> None:
        """`config.debug.random_init`-style init (no checkpoints offline): N(0, std) matrices and biases, ones for norms; identical on every rank."""
        gen = torch.Generator(device=self.device).manual_seed(seed)
        self.params.zero_()
        for name, w in self.w.items():
            if w.dim() == 1 and not name.endswith(".bias"):
                w.fill_(1.0)
            else:
                chunk = 1 << 26  # generate in f32 chunks, round to bf16
                flat = w.view(-1)
                for s in range(0, flat.numel(), chunk):
                    e = min(s + chunk, flat.numel())
                    flat[s:e] = (torch.randn(e - s, device=self.device, generator=gen) * std).to(torch.bfloat16)
        if self.dims.head_padded:  # the padding positions of every head hold zeros and stay zero (no gradient ever reaches them)
            self.load_state_dict({k: v.clone() for k, v in self.state_dict().items()})
            return
        self.sync_master_from_params()

    # -- 64-wide heads stored zero-padded to the kernels' 128 (ModelDims.kernel_head_dim) --------------------------------------------
    def _head_index(self, n_heads: int) -> Tensor:
        """Positions of the model's own head dimensions inside the padded storage: rotary half 0 -> [0, d/2), half 1 -> [64, 64 + d/2) of every head."""
        d, kd = self.dims.head_dim, self.dims.kernel_head_dim
        one = torch.cat([torch.arange(d // 2), kd // 2 + torch.arange(d // 2)])
        return (torch.arange(n_heads)[:, None] * kd + one[None, :]).reshape(-1).to(self.device)

    def _pad_heads(self, t: Tensor, n_heads: int, dim: int) -> Tensor:
        if not self.dims.head_padded:
            return t
        shape = list(t.shape)
        shape[dim] = n_heads * self.dims.kernel_head_dim
        out = torch.zeros(shape, dtype=t.dtype, device=self.device)
        out.index_copy_(dim, self._head_index(n_heads), t.to(self.device))
        return out

    def _unpad_heads(self, t: Tensor, n_heads: int, dim: int) -> Tensor:
        return t.index_select(dim, self._head_index(n_heads)) if self.dims.head_padded else t

    def load_state_dict(self, sd: dict[str, Tensor]) -> None:
        """HF-named tensors (q_proj/k_proj/v_proj, gate_proj/up_proj separate) -> fused flat storage."""
        d = self.dims
        if d.head_padded:
            sd = dict(sd)
            for i in range(d.num_hidden_layers):
                pre = f"model.layers.{i}.self_attn."
                for nm, heads in (("q", d.num_attention_heads), ("k", d.num_key_value_heads), ("v", d.num_key_value_heads)):
                    for kind in ("weight", "bias"):
                        if pre + f"{nm}_proj.{kind}" in sd:
                            sd[pre + f"{nm}_proj.{kind}"] = self._pad_heads(sd[pre + f"{nm}_proj.{kind}"], heads, 0)
                if pre + "o_proj.weight" in sd:
                    sd[pre + "o_proj.weight"] = self._pad_heads(sd[pre + "o_proj.weight"], d.num_attention_heads, 1)
        for i in range(d.num_hidden_layers):
            p = f"model.layers.{i}."
            sd = dict(sd)
            if p + "self_attn.q_proj.weight" in sd:
                sd[p + "self_attn.qkv_proj.weight"] = torch.cat([sd.pop(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
            if p + "self_attn.q_proj.bias" in sd:
                sd[p + "self_attn.qkv_proj.bias"] = torch.cat([sd.pop(p + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
            if p + "mlp.gate_proj.weight" in sd:
                sd[p + "mlp.gate_up_proj.weight"] = torch.cat([sd.pop(p + "mlp.gate_proj.weight"), sd.pop(p + "mlp.up_proj.weight")], 0)
        if d.tie_word_embeddings:
            sd.pop("lm_head.weight", None)
        missing = set(self.w) - set(sd)
        if missing:
            raise KeyError(f"missing parameters: {sorted(missing)[:4]}...")
        self.params.zero_()
        for name, w in self.w.items():
            w.copy_(sd[name].to(torch.bfloat16))
        self.sync_master_from_params()

    def state_dict(self) -> dict[str, Tensor]:
        """HF-named views of the bf16 parameters (no copies; models with padded heads get de-padded copies of the attention projections)."""
        return self._hf_named(self.w)

    def grad_state_dict(self) -> dict[str, Tensor]:
        """The f32 gradients under the same HF names."""
        return self._hf_named(self.g)

    def _hf_named(self, flat: dict[str, Tensor]) -> dict[str, Tensor]:
        d, out = self.dims, {}
        for name, w in flat.items():
            if name.endswith("qkv_proj.weight") or name.endswith("qkv_proj.bias"):
                kind = name.rsplit(".", 1)[1]
                base = name[: -len("qkv_proj." + kind)]
                out[base + "q_proj." + kind] = w[: d.q_dim]
                out[base + "k_proj." + kind] = w[d.q_dim: d.q_dim + d.kv_dim]
                out[base + "v_proj." + kind] = w[d.q_dim + d.kv_dim:]
            elif name.endswith("gate_up_proj.weight"):
                base = name[: -len("gate_up_proj.weight")]
                out[base + "gate_proj.weight"] = w[: d.intermediate_size]
                out[base + "up_proj.weight"] = w[d.intermediate_size:]
            else:
                out[name] = w
        if d.head_padded:
            for name in list(out):
                if ".self_attn." not in name or "norm" in name:
                    continue
                heads = d.num_attention_heads if (".q_proj." in name or ".o_proj." in name) else d.num_key_value_heads
                out[name] = self._unpad_heads(out[name], heads, 1 if ".o_proj." in name else 0)
        if d.tie_word_embeddings:
            out["lm_head.weight"] = flat["model.embed_tokens.weight"]
        return out

    def shard_slices(self, b: _Bucket) -> tuple[slice, slice]:
        """(slice of this rank's shard inside the full flat buffers, slice inside the shard-sized buffers) for bucket b."""
        per = b.size // self.dp_size
        full = slice(b.offset + self.dp_rank * per, b.offset + (self.dp_rank + 1) * per)
        local = slice(b.offset // self.dp_size, b.offset // self.dp_size + per)
        return full, local

    def sync_master_from_params(self) -> None:
        self.version += 1  # every path that rewrites the bf16 parameters ends here or in the optimizer step
        for b in self.buckets:
            full, local = self.shard_slices(b)
            self.master[local].copy_(self.params[full])

    def head_weight(self) -> Tensor:
        return self.w["model.embed_tokens.weight"] if self.dims.tie_word_embeddings else self.w["lm_head.weight"]

    def head_grad(self) -> Tensor:
        return self.g["model.embed_tokens.weight"] if self.dims.tie_word_embeddings else self.g["lm_head.weight"]


# ---------------------------------------------------------------------------------------------------------------------
class GRPOEngine:
    """forward -> log-probs -> GRPO loss -> backward -> (reduce-scatter) -> clip -> AdamW -> (all-gather) for one rank."""

    def __init__(self, dims: ModelDims, device="cuda:0", optim: OptimConfig | None = None, loss: LossConfig | None = None,
                 dp_group: dist.ProcessGroup | None = None, seed: int = 0, lm_head_token_chunk: int = 16384, recompute_act: bool = False,
                 force_collectives: bool = False, gemm_layouts: bool = True, gemm_dtype: str = "bf16", dp_size: int | None = None,
                 cp_group: dist.ProcessGroup | None = None, cp_size: int = 1, lm_head_impl: str | None = None,
                 shard_group: dist.ProcessGroup | None = None, cp_style: str = "ulysses", recompute: str | None = None, recompute_freq: int | None = None,
                 fused_sumsq: bool = True,
                 embed_split: bool = True, fused_swiglu_bwd: bool = True, fused_swiglu_fwd: bool = True):
        self.dims, self.device = dims, torch.device(device)
        if gemm_dtype not in ("bf16", "f32"):
            raise PrimeMI355XError(f"gemm_dtype must be 'bf16' or 'f32', got {gemm_dtype!r}")
        # "f32" = the parity-measurement mode (tests/test_engine_gpu.py): every library GEMM runs on f32 copies of its operands
        # (exact products, f32 accumulation, no bf16 rounding inside the GEMM) and the LM head hands f32 logits to K1, so that what is
        # left between the engine and an f32 reference is the hand-written kernels alone.  Same kernels, same schedule, same buffers.
        self.gemm_f32 = gemm_dtype == "f32"
        self.optim, self.loss_cfg = optim or OptimConfig(), loss or LossConfig()
        self.dp_group = dp_group
        # `dp_size` given (trainer/model.py:setup_engine passes ParallelDims.dp_size): the data-parallel group is exactly dp_group.  Not given:
        # the whole default group is the dp group — only right when every rank of it is a trainer rank, so a caller that shares its default
        # group with other ranks MUST pass dp_group / dp_size explicitly.
        if dp_size is not None:
            self.dp_size = dp_size
            if dp_size > 1 and dist.get_world_size(dp_group) != dp_size:
                raise PrimeMI355XError(f"dp_size {dp_size} != size of dp_group {dist.get_world_size(dp_group)}")
        else:
            self.dp_size = dist.get_world_size(dp_group) if (dp_group is not None or dist.is_initialized()) else 1
        self.dp_rank = dist.get_rank(dp_group) if self.dp_size > 1 else 0
        # Context parallelism (Ulysses, trainer/ulysses.py): the cp ranks of one dp position see the SAME packed pass, run the token-local
        # layers on 1/cp of its tokens and exchange heads <-> sequence around attention.  Gradients are partial sums over token shards,
        # so the ZeRO shard group (reduce-scatter / all-gather / norm all-reduce) is the dp x cp group; token counts and step statistics
        # are per pass, so they reduce over the dp group only.
        # Two styles, the reference's `model.cp_style` (configs/trainer.py:98): "ulysses" = heads <-> sequence all-to-all around the unchanged
        # kernel (every rank attends the full pass for 1/cp of the heads; balanced under the causal mask); "ring" = what the reference's
        # ring_attn.py actually does (llama3-style): all-gather K / V over the cp group, every rank attends ITS q rows against the K / V prefix of
        # the pass (pm_attn_varlen_*_rows), dK / dV reduce-scattered back — no head-count constraint, later ranks do more attention work.
        self.cp_group, self.cp_size = cp_group, cp_size
        self.cp_rank = dist.get_rank(cp_group) if cp_size > 1 else 0
        if cp_style not in ("ulysses", "ring"):
            raise PrimeMI355XError(f"cp_style must be 'ulysses' or 'ring', got {cp_style!r}")
        self.cp_style = cp_style
        if cp_size > 1:
            if cp_style == "ulysses" and (dims.num_attention_heads % cp_size or (dims.num_key_value_heads % cp_size and cp_size % dims.num_key_value_heads)):
                raise PrimeMI355XError(f"cp_size {cp_size} must divide the {dims.num_attention_heads} query heads and divide or be divided by the "
                                       f"{dims.num_key_value_heads} KV heads (ulysses_attn.py:27-31)")
            self.shard_group = shard_group
            self.shard_size = self.dp_size * cp_size
            if dist.get_world_size(shard_group) != self.shard_size:
                raise PrimeMI355XError(f"shard_group must hold the dp x cp = {self.shard_size} trainer ranks, got {dist.get_world_size(shard_group)}")
            self.shard_rank = dist.get_rank(shard_group)
        else:
            self.shard_group, self.shard_size, self.shard_rank = self.dp_group, self.dp_size, self.dp_rank
        if dims.kernel_head_dim != 128:
            raise PrimeMI355XError(f"head_dim {dims.head_dim}: the attention kernels are built for 128-wide heads (64-wide heads run zero-padded to 128)")
        if dims.head_padded and dims.qk_norm:
            raise PrimeMI355XError("padded heads + per-head q/k RMSNorm: the norm would average over the padding (no such model family exists)")
        if force_collectives and not dist.is_initialized():
            raise PrimeMI355XError("force_collectives needs an initialised process group (a world_size-1 nccl group is enough)")
        if self.optim.type not in ("adamw", "sgd", "sign_sgd"):
            raise PrimeMI355XError(f"optim.type = {self.optim.type!r}: the fused on-GPU optimizers are adamw, sgd and sign_sgd")
        self.store = ParamStore(dims, self.device, self.shard_rank, self.shard_size, force_collectives=force_collectives, optimizer=self.optim.type)
        self.use_collectives = self.store.use_collectives
        self.store.random_init(seed)
        self.inv_freq = scaled_inv_freq(dims.head_dim, dims.rope_theta, dims.rope_scaling, self.device)  # == ops.rope_inv_freq without scaling
        if dims.head_padded:  # frequencies of the padded positions: 0 (cos 1, sin 0 — they only ever rotate zeros)
            self.inv_freq = torch.cat([self.inv_freq, torch.zeros(dims.kernel_head_dim // 2 - self.inv_freq.numel(), device=self.device)])
        self.lm_head_token_chunk = lm_head_token_chunk
        self.recompute_act = recompute_act
        # "layer": block-level activation recomputation (the reference's default `model.ac`, trainer/model.py:1188-1194): forward keeps each decoder
        # layer's INPUT only (8 KB per token and layer instead of ~130 KB), backward re-runs the layer's forward first.  One extra forward of compute;
        # bit-identical gradients (every kernel on the path is deterministic and the stored input is the already-rounded residual sum).  Off by
        # default: 288 GB holds the activations of 16k-token passes; a 32k-token pass on one GPU needs it.
        # "selective" (the reference's `ac.mode = "selective"` with targets norm / attn_proj / mlp, models/layers/checkpointing.py:120-147: the projections, norms and
        # the MLP are re-run, the attention core is NOT): like "layer", but the attention output, its lse and the residual stream behind the output projection
        # (268 + 4 + 268 MB per layer and 32k-token pass — 288 GB hold them for every layer) are kept, so the backward re-runs everything but
        # `pm_attn_varlen_fwd` and the output projection; bit-identical gradients, one attention forward per layer less (a quarter of a 32k-token step's
        # attention time).
        self.recompute = recompute or os.environ.get("PM_RECOMPUTE", "none")
        if self.recompute not in ("none", "layer", "selective"):
            raise PrimeMI355XError(f"recompute must be 'none', 'layer' or 'selective', got {self.recompute!r}")
        # the reference's `ac.freq` (trainer/model.py:1151: `layer_id % freq != 0` -> not checkpointed): only every freq-th layer is recomputed, the others keep
        # their activations (~120 KB per token and layer) — at one 32 768-token rollout freq = 2 still fits the 288 GB and halves the recompute
        self.recompute_freq = int(recompute_freq if recompute_freq is not None else os.environ.get("PM_RECOMPUTE_FREQ", "1"))
        if self.recompute_freq < 1:
            raise PrimeMI355XError(f"recompute_freq must be >= 1, got {self.recompute_freq}")
        self.step_count = 0
        # a dedicated stream for the collectives — on the GPU; the CPU flavour (tests/test_engine_gloo_cpu.py: the SAME engine code over gloo with
        # stand-in kernels) issues them synchronously
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.use_collectives and self.device.type == "cuda") else None
        self._gather_events: dict[str, torch.cuda.Event] = {}
        # The embedding bucket's gradient is complete only when the LAST pass's backward ends, so its dense reduce-scatter (vocab x hidden f32,
        # 2.5 GB for Qwen3-8B) would trail the step with nothing to hide behind.  With sharding on, step() splits it: the dense part (earlier
        # passes; the tied head gradient) is reduce-scattered while the last pass still runs, and the last pass's own embedding gradient —
        # n_tokens rows — is all-gathered as (ids, bf16 rows) and added by the rank that owns the rows (ops.embedding_bwd_window).
        # `embed_split=False` keeps the single trailing reduce-scatter (A/B switch of the tests).
        self.embed_split = self.use_collectives and embed_split
        self._embed_plan: dict | None = None
        self._embed_bufs: dict | None = None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._wgrad_direct = self._probe_wgrad_direct()
        # library-dgrad configurations (PM_GEMM_OWN without "dgrad") keep K-contiguous [in, out] copies of the projection weights for hipBLASLt's dgrad GEMMs
        # (`gemm_layouts=False`: the plain layout, A/B switch of the tests); the default configuration has neither the copies nor the library
        self.gemm_layouts = gemm_layouts and not self.gemm_f32
        # Which of the step's GEMM classes run on the hand-written MFMA core (csrc/gemm.hip, ops.gemm) instead of the library: a subset of
        # {"fwd", "dgrad", "wgrad"} ("all" / "none" accepted).  Default since round 5: ALL of them — no hipBLASLt kernel is left in the step.  The core reads
        # every operand where it lies (dgrad: the stored [out, in] weight K-major; wgrad: dy and x token-major, f32 accumulate into the gradient buffer), so
        # there are no [in, out] weight copies (13.9 GB for Qwen3-8B), no pm_transpose_2d launches, and the gate|up forward carries the SwiGLU activation in
        # its epilogue (ops.gemm_swiglu).  Measured in the step (profiles/r05_gemm.md): all 629.5 ms, `wgrad` (library forward + dgrad on weight copies,
        # the default of rounds 3-4) 632.6 ms on the same box.  `PM_GEMM_OWN=wgrad | none | fwd,wgrad ...` remain as documented configurations.
        sel = os.environ.get("PM_GEMM_OWN", "all")
        self.gemm_own = {"all": {"fwd", "dgrad", "wgrad"}, "none": set()}.get(sel, set(filter(None, sel.split(","))))
        if self.gemm_own - {"fwd", "dgrad", "wgrad"}:
            raise PrimeMI355XError(f"PM_GEMM_OWN: unknown GEMM class in {sel!r} (fwd, dgrad, wgrad, all, none)")
        if self.gemm_f32 or self.device.type != "cuda":
            self.gemm_own = set()
        if "dgrad" in self.gemm_own:  # the core reads the stored [out, in] weight K-major: no [in, out] copies
            self.gemm_layouts = False
        if self.gemm_own:
            # Schedule of the own core: one PERSISTENT workgroup per CU walking its tiles (+0-2 %) when this rank's GPU runs nothing beside the step; one
            # workgroup per tile when collectives run on the comm stream — a persistent grid of exactly 256 workgroups that each need a whole CU (160 KiB of
            # LDS) would be cut into two rounds by a handful of CUs held by RCCL kernels, a per-tile grid only loses those CUs' share.
            # The schedule word is an ARGUMENT of every launch (include/prime_mi355x.h `sched`): this engine passes ITS word (`_own_gemm`); the library keeps none.
            self._gemm_variant = int(os.environ.get("PM_GEMM_VARIANT", "3" if self.use_collectives else "7"))  # (the env override is the in-step A/B hook: + 32 = no split-K)
        # The gradient-norm pass fused into the weight-gradient epilogues (ops.gemm(tile_sumsq=...)): on the LAST backward pass of a step every wgrad GEMM
        # leaves per-tile sums of squares of the final f32 gradient; the optimizer then adds those up together with the few small tensors in between
        # (norm weights, biases, the embedding) instead of re-reading the whole 4-byte-per-parameter buffer (5.3 ms per Qwen3-8B step).  Only without
        # collectives: under ZeRO-2 the norm is taken over the REDUCED shard.  `fused_sumsq=False` is the A/B switch of the tests.
        # dgrad per STEP: the library's dgrad GEMMs are 6-11 % faster than the own core but need the K-contiguous weight copies refreshed after every parameter
        # update (145 transposes = 5.7 ms for Qwen3-8B, whatever the batch); below ~7k tokens per step the copies cost more than the library gains, and the own
        # core reads the stored weights as they are (the reference benchmark's 2 x 2048-token step: -2.5 ms).  (Only configurations with a library dgrad.)
        self.dgrad_own_below = 7168 if (self.gemm_own and self.gemm_layouts) else 0
        self._dgrad_own_now = "dgrad" in self.gemm_own
        self._fused_sumsq_wanted = fused_sumsq
        # the SwiGLU backward in the epilogue of the down projection's row-gradient GEMM (ops.gemm_swiglu_bwd) where that GEMM runs on the own core;
        # `fused_swiglu_bwd=False` is the A/B switch of the tests (GEMM, then pm_swiglu_bwd: identical results)
        self.fused_swiglu_bwd = fused_swiglu_bwd
        self.fused_swiglu_fwd = fused_swiglu_fwd  # A/B switch: the SwiGLU activation in the gate|up product's epilogue vs GEMM + pm_swiglu_fwd (bit-identical)
        self.fused_sumsq = "wgrad" in self.gemm_own and not self.use_collectives and fused_sumsq
        self._fs_plan: dict | None = None
        self._fs_covered: set[str] = set()
        self._wt: dict[str, Tensor] = {}
        self._wt_version: dict[int, int] = {}
        if self.gemm_layouts:  # allocated up front (one slab) so the caching allocator sees the same request stream from the first step on
            mats = [(f"model.layers.{i}.{s}", self.store.w[f"model.layers.{i}.{s}"].shape) for i in range(dims.num_hidden_layers) for s in self._DGRAD_MATRICES]
            if dims.vocab_size % 8 == 0:  # rows of the [H, V] copy must start 16-byte aligned
                mats.append(("head", self.store.head_weight().shape))
            slab = torch.empty(sum(o * k for _, (o, k) in mats), dtype=torch.bfloat16, device=self.device)
            cursor = 0
            for nm, (o, k) in mats:
                self._wt[nm] = slab[cursor:cursor + o * k].view(k, o)
                cursor += o * k
        # LM head: "library" = hipBLASLt logits [N, V] bf16 in HBM -> K1 (kept for backward, overwritten in place by dlogits);
        # "fused" = J1 (csrc/lmhead.hip): GEMM tile + log-prob epilogue in one kernel, no logits in HBM, backward recomputes them on the
        # MFMA units per token chunk.  The default is whichever measured faster for the step (profiles/: the library GEMM runs at
        # ~1.45 PFLOP/s, J1 at ~0.95, and the fused form pays a second GEMM in backward) — "fused" trades ~25 ms/step for 5 GB at 16k tokens.
        self.lm_head_impl = lm_head_impl or os.environ.get("PM_LM_HEAD", "library")
        if self.lm_head_impl not in ("library", "fused"):
            raise PrimeMI355XError(f"lm_head_impl must be 'library' or 'fused', got {self.lm_head_impl!r}")
        if self.lm_head_impl == "fused" and (self.gemm_f32 or dims.hidden_size % 64):
            raise PrimeMI355XError("the fused LM head is a bf16 MFMA kernel (hidden size a multiple of 64)")
        self.k1_events: list | None = None
        self._emb_zero_early = self._emb_zeroed = False  # step(): the embedding gradient's zero fill placed between the LM-head product and the log-prob kernel
        # bench.py's parity block (SURVEY §8d "parity gates in the same harness"): when set to a dict, the NEXT forward/loss leaves device-side copies
        # of what the gates need in it (no host sync inside the step): the embedding-gather and label-gather bit checks over ALL rows, 64 sampled
        # logits rows with their labels / temperatures, the per-token log-probs / entropy / logz, the scalar loss and the K2 statistics.
        self.parity_probe: dict | None = None
        self._param_readers: list = []  # events of asynchronous readers of store.params (the weight push): waited for before AdamW rewrites them

    # -- GEMM helpers (library calls: hipBLASLt via torch) ------------------------------------------------------------
    def _probe_wgrad_direct(self) -> bool:
        """Can the weight-gradient GEMM write/accumulate straight into the f32 gradient buffer (bf16 x bf16 -> f32)?"""
        try:
            a = torch.randn(64, 32, device=self.device).to(torch.bfloat16)
            b = torch.randn(64, 48, device=self.device).to(torch.bfloat16)
            out = torch.zeros(32, 48, dtype=torch.float32, device=self.device)
            torch.mm(a.t(), b, out_dtype=torch.float32, out=out)
            ref = a.float().t() @ b.float()
            ok = torch.allclose(out, ref, rtol=1e-3, atol=1e-3)
            torch.addmm(out, a.t(), b, out_dtype=torch.float32, out=out)
            ok = ok and torch.allclose(out, 2 * ref, rtol=1e-3, atol=2e-3)
            return bool(ok)
        except Exception:
            return False

    def set_gemm_own(self, classes) -> None:
        """Re-route the step's GEMM classes at run time ("all", "none", or an iterable of fwd / dgrad / wgrad): what PM_GEMM_OWN selects at construction.
        bench.py uses it to time the library mix of rounds 3-4 next to the headline on the same engine; a library dgrad builds its [in, out] weight
        copies lazily on first use (no up-front slab)."""
        own = {"all": {"fwd", "dgrad", "wgrad"}, "none": set()}.get(classes, None) if isinstance(classes, str) else set(classes)
        if own is None:
            own = set(filter(None, classes.split(",")))
        if own - {"fwd", "dgrad", "wgrad"}:
            raise PrimeMI355XError(f"set_gemm_own: unknown GEMM class in {classes!r} (fwd, dgrad, wgrad, all, none)")
        if self.gemm_f32 or self.device.type != "cuda":
            own = set()
        self.gemm_own = own
        self.gemm_layouts = "dgrad" not in own and not self.gemm_f32
        self.dgrad_own_below = 7168 if (own and self.gemm_layouts) else 0
        self._dgrad_own_now = "dgrad" in own
        self.fused_sumsq = "wgrad" in own and not self.use_collectives and self._fused_sumsq_wanted
        self._fs_covered.clear()
        if not self.gemm_layouts:
            self._wt.clear()
            self._wt_version.clear()
        elif "head" not in self._wt and self.dims.vocab_size % 8 == 0:  # the layer copies appear lazily in _dgrad; the head's is used only if it exists
            self._wt["head"] = torch.empty((self.dims.hidden_size, self.dims.vocab_size), dtype=torch.bfloat16, device=self.device)
        if own and not hasattr(self, "_gemm_variant"):
            self._gemm_variant = int(os.environ.get("PM_GEMM_VARIANT", "3" if self.use_collectives else "7"))

    def _own_gemm(self, *args, **kwargs) -> Tensor:
        """ops.gemm on the hand-written core under THIS engine's schedule word: the `sched` argument of the C call, so that no other caller in the process —
        the nn.Module LM head, the grouped MoE products, a second engine, another thread — inherits or disturbs it (the library holds no schedule state)."""
        return ops.gemm(*args, variant=self._gemm_variant, **kwargs)

    def _fwd_mm(self, x: Tensor, w: Tensor, out_f32: bool = False, bias: Tensor | None = None) -> Tensor:
        """y[n,out] = x[n,in] @ w[out,in]^T (+ bias) — the forward projection GEMM (library call; the bias rides in the GEMM epilogue, added
        to the f32 accumulator before the one rounding to bf16, as F.linear does)."""
        if self.gemm_f32:
            y = torch.mm(x.float(), w.float().t())
            if bias is not None:
                y += bias.float()
            return y if out_f32 else y.to(torch.bfloat16)
        if "fwd" in self.gemm_own and ops.gemm_supported(x, w, out_dtype=torch.float32 if out_f32 else torch.bfloat16):
            return self._own_gemm(x, w, out_dtype=torch.float32 if out_f32 else torch.bfloat16, bias=bias)
        if bias is not None:
            return torch.addmm(bias, x, w.t())
        return torch.mm(x, w.t(), out_dtype=torch.float32) if out_f32 else torch.mm(x, w.t())

    def _wgrad(self, dy: Tensor, x: Tensor, gview: Tensor, accumulate: bool, k_contiguous_x: bool = False, final_name: str | None = None) -> None:
        """gview[out,in] (+)= dy[n,out]^T @ x[n,in] with f32 accumulation.

        `k_contiguous_x`: first re-lay x as [in, n] (pm_transpose_2d) so the library GEMM reads it along the reduction dimension —
        measured on MI355X (tools/microbench.py gemm) 15 % faster for the two MLP weight gradients, more than the transpose costs;
        for the attention projections it is a wash, so those keep the plain form."""
        if self.gemm_f32:
            prod = torch.mm(dy.float().t(), x.float())
            if accumulate:
                gview.add_(prod)
            else:
                gview.copy_(prod)
            return
        if "wgrad" in self.gemm_own and ops.gemm_supported(dy.t(), x.t()):
            # dw[out, in] (+)= sum_tok dy[tok, out] x[tok, in]: both operands K-major (tokens are their rows), f32 accumulate in the epilogue
            # `final_name`: this call writes the FINAL gradient of that parameter for the step -> its squared norm comes out of the epilogue
            slot = self._fs_slot(final_name) if (final_name is not None and self.fused_sumsq) else None
            self._own_gemm(dy.t(), x.t(), out=gview, accumulate=accumulate, tile_sumsq=slot)
            if slot is not None:
                self._fs_covered.add(final_name)
            return
        if self._wgrad_direct:
            xop = ops.transpose_2d(x).t() if (k_contiguous_x and self.gemm_layouts) else x
            if accumulate:
                torch.addmm(gview, dy.t(), xop, out_dtype=torch.float32, out=gview)
            else:
                torch.mm(dy.t(), xop, out_dtype=torch.float32, out=gview)
        else:
            tmp = torch.mm(dy.t(), x)
            if not accumulate:
                gview.zero_()
            ops.grad_accumulate(gview.view(-1), tmp.view(-1))

    _FS_SEGMENT = 1 << 20  # elements per workgroup of pm_sumsq_segments

    def _fs_build(self) -> dict:
        """Slots of the per-tile partials of every GEMM weight, and the (offset, length) segments of everything else in the flat gradient buffer."""
        d, st = self.dims, self.store
        slots: dict[str, tuple[int, int]] = {}
        cursor = 0
        rest: list[tuple[int, int]] = []
        for b in st.buckets:
            for pname, off, shape in b.params:
                is_matrix = len(shape) == 2 and pname != "model.embed_tokens.weight" and not (pname == "lm_head.weight" and d.tie_word_embeddings)
                if is_matrix:
                    n = ops.gemm_sumsq_slots(shape[0], shape[1])
                    slots[pname] = (cursor, n)
                    cursor += n
                else:
                    rest.append((off, math.prod(shape)))
        big = [(o, ln) for o, ln in rest if ln >= 4 * self._FS_SEGMENT]  # the embedding: the streaming two-stage kernel (pm_sumsq) on its range
        segs = [(o + c, min(self._FS_SEGMENT, ln - c)) for o, ln in rest if ln < 4 * self._FS_SEGMENT for c in range(0, ln, self._FS_SEGMENT)]
        return {
            "slots": slots,
            "big": big,
            "tiles": torch.zeros(max(cursor, 1), dtype=torch.float32, device=self.device),  # slots of tiles that do not exist stay zero forever
            "segments": torch.tensor(segs, dtype=torch.int64, device=self.device).reshape(-1, 2),
            "seg_partials": torch.zeros(max(len(segs), 1), dtype=torch.float32, device=self.device),
        }

    def _fs_slot(self, name: str) -> Tensor | None:
        if self._fs_plan is None:
            self._fs_plan = self._fs_build()
        ent = self._fs_plan["slots"].get(name)
        if ent is None:
            return None
        return self._fs_plan["tiles"][ent[0]:ent[0] + ent[1]]

    def _fs_norm(self, out_scalar: Tensor) -> None:
        """out_scalar[0] += sum(main_grad^2), from the epilogue partials + the segments the GEMMs do not cover (+ any matrix whose last wgrad
        call did not run on the own core this step: re-read like before)."""
        plan = self._fs_plan if self._fs_plan is not None else self._fs_build()
        self._fs_plan = plan
        missing = [nm for nm in plan["slots"] if nm not in self._fs_covered]
        if missing:  # e.g. a shape the own core does not take: zero its slots and read that gradient the old way
            for nm in missing:
                o, n = plan["slots"][nm]
                plan["tiles"][o:o + n].zero_()
                ops.sumsq_accumulate(self.store.g[nm].view(-1), out_scalar)
        ops.sum_partials(plan["tiles"], out_scalar)
        for o, ln in plan["big"]:
            ops.sumsq_accumulate(self.store.main_grad[o:o + ln], out_scalar)
        if plan["segments"].shape[0]:
            ops.sumsq_segments(self.store.main_grad, plan["segments"], plan["seg_partials"])
            ops.sum_partials(plan["seg_partials"][:plan["segments"].shape[0]], out_scalar)
        self._fs_covered.clear()

    _DGRAD_MATRICES = ("self_attn.qkv_proj.weight", "self_attn.o_proj.weight", "mlp.gate_up_proj.weight", "mlp.down_proj.weight")

    def _dgrad(self, dy: Tensor, layer: int, suffix: str) -> Tensor:
        """dx[n,in] = dy[n,out] @ W[out,in].  The library runs this GEMM 10-25 % faster when the weight is K-contiguous, i.e. stored
        [in,out] (tools/microbench.py gemm), while the forward wants [out,in]: with 288 GB of HBM both layouts are kept, the [in,out]
        copy of a layer being refreshed (4 x pm_transpose_2d, ~0.2 ms) the first time its backward runs after a parameter update."""
        name = f"model.layers.{layer}.{suffix}"
        if self.gemm_f32:
            return torch.mm(dy.float(), self.store.w[name].float()).to(torch.bfloat16)
        if self._dgrad_own_now and ops.gemm_supported(dy, self.store.w[name].t()):
            return self._own_gemm(dy, self.store.w[name].t())  # the [out, in] weight read K-major
        if not self.gemm_layouts:
            return torch.mm(dy, self.store.w[name])
        if self._wt_version.get(layer) != self.store.version:
            for s in self._DGRAD_MATRICES:
                nm = f"model.layers.{layer}.{s}"
                self._wt[nm] = ops.transpose_2d(self.store.w[nm], out=self._wt.get(nm))
            self._wt_version[layer] = self.store.version
        return torch.mm(dy, self._wt[name].t())

    # -- forward ----------------------------------------------------------------------------------------------------
    def forward(self, b: PackedBatch) -> dict:
        d, w = self.dims, self.store.w
        n_full = b.n_tokens
        cp = self.cp_size
        # context parallel: this rank runs the token-local layers on chunk cp_rank of the pass (utils/cp.py:97-120); attention and the
        # loss see the full pass
        tok = (lambda t: t) if cp == 1 else (lambda t: ulysses.shard_for_cp(t, self.cp_rank, cp))
        ids, labels, inv_temp = tok(b.input_ids), tok(b.labels), tok(b.inv_temp)
        n = ids.numel()
        ctx: dict = {"batch": b, "layers": [], "labels": labels, "inv_temp": inv_temp, "ids": ids}
        self._wait_gather("embed")
        x = ops.embedding_fwd(w["model.embed_tokens.weight"], ids)
        probe = self.parity_probe if (self.parity_probe is not None and "loss" not in self.parity_probe and cp == 1) else None
        if probe is not None:  # token-index gather, all rows, bit for bit (one device bool; read after the timed region)
            probe["embed_gather_equal"] = (x == w["model.embed_tokens.weight"][ids]).all()
        cos, sin = ops.rope_table(tok(b.position_ids), self.inv_freq)
        seq_start = ops.seq_start_from_cu(b.cu_seqlens, n_full)
        ctx.update(cos=cos, sin=sin, seq_start=seq_start)
        scale = d.head_dim**-0.5
        pending = None  # mlp output whose residual add is fused into the next RMSNorm
        for i in range(d.num_hidden_layers):
            self._wait_gather(f"layer{i}")
            L, x, pending = self._layer_forward(i, x, pending, cos, sin, seq_start, scale, n)
            if self.recompute == "none" or i % self.recompute_freq != 0:
                ctx["layers"].append(L)
            elif self.recompute == "layer" or (self.recompute == "selective" and cp > 1):  # (under context parallelism the attention state is the exchanged tensors: re-run it)
                ctx["layers"].append({"x": L["x"], "recompute": True})
            elif self.recompute == "selective":
                ctx["layers"].append({"x": L["x"], "recompute": True, "o": L["o"], "lse": L["lse"], "x2": L["x2"]})
            else:
                ctx["layers"].append(L)
        self._wait_gather("head")
        hf, rstd_f, x_last = ops.rmsnorm_fwd(pending, w["model.norm.weight"], d.rms_norm_eps, residual=x)
        ctx.update(x_last=x_last, rstd_f=rstd_f, hf=hf)
        # LM head: materialised bf16 logits per token chunk (kept for backward; 288 GB makes the recompute GEMM unnecessary)
        wh = self.store.head_weight()
        logprobs = torch.empty(n, dtype=torch.float32, device=self.device)
        entropy = torch.empty_like(logprobs)
        logz = torch.empty_like(logprobs)
        chunks = []
        for s in range(0, n, self.lm_head_token_chunk):
            e = min(s + self.lm_head_token_chunk, n)
            if self.lm_head_impl == "fused":
                logprobs[s:e], entropy[s:e], logz[s:e] = ops.lmhead_logprob_fwd(hf[s:e], wh, labels[s:e], inv_temp[s:e])
                chunks.append((s, e, None))
                continue
            logits = self._fwd_mm(hf[s:e], wh, out_f32=self.gemm_f32)
            if self._emb_zero_early and s == 0:
                # The step's one zero fill of the embedding gradient (2.5 GB) goes HERE instead of in front of the embedding backward: the GEMM above leaves
                # its last 256 MiB of logits dirty in the Infinity Cache, and their write-back costs less under this fill's write stream than under the
                # log-prob kernel's read stream (tools/microbench.py k1aftergemm, "fill placement": fill + K1 1.121 -> 1.103 ms, K1 0.81 -> 0.83 of 8 TB/s).
                self.store.g["model.embed_tokens.weight"].zero_()
                self._emb_zero_early, self._emb_zeroed = False, True
            if self.k1_events is not None:  # HIP events on the launch stream around the log-prob kernel (bench.py roofline)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            lp, en, lz = ops.logprob_entropy_fwd(logits, labels[s:e], inv_temp[s:e])
            if self.k1_events is not None:
                ev1.record()
                self.k1_events.append((ev0, ev1, e - s, logits.element_size()))
            logprobs[s:e], entropy[s:e], logz[s:e] = lp, en, lz
            if probe is not None and s == 0 and not self.gemm_f32:
                # label gather over the vocab, every row of the chunk: K1's log-prob must be EXACTLY (logit[label] * inv_T) - logz in f32
                picked = logits.gather(1, labels[s:e, None])[:, 0].float() * inv_temp[s:e]
                probe["label_gather_equal"] = (lp == picked - lz).all()
                probe["label_gather_max_abs"] = (lp - (picked - lz)).abs().max()
                rows = torch.linspace(0, e - s - 1, steps=min(64, e - s), device=self.device).long()
                probe.update(rows=rows, logits_rows=logits[rows].clone(), labels_rows=labels[s:e][rows].clone(), inv_t_rows=inv_temp[s:e][rows].clone())
            chunks.append((s, e, logits))
        if cp > 1:  # the loss (and the shift_tensor_right inside it) runs over the FULL pass: gather_for_cp (utils/cp.py:129-139, train.py:446-448)
            ctx["logprobs_local"] = logprobs
            logprobs = ulysses.all_gather_tokens(logprobs, cp, self.cp_group)
            entropy = ulysses.all_gather_tokens(entropy, cp, self.cp_group)
        ctx.update(logits_chunks=chunks, logprobs=logprobs, entropy=entropy, logz=logz)
        if probe is not None:
            probe.update(logprobs=logprobs.clone(), entropy=entropy.clone(), logz=logz.clone())
        return ctx

    def _layer_forward(self, i: int, x: Tensor, pending: Tensor | None, cos: Tensor, sin: Tensor, seq_start: Tensor, scale: float, n: int,
                       kept_attn: tuple[Tensor, Tensor] | None = None, want_out: bool = True, kept_x2: Tensor | None = None):
        """One decoder layer: (saved activations, residual stream after attention, mlp output whose residual add the NEXT norm fuses).  `pending` = the
        previous layer's mlp output (None for layer 0 and when re-running a layer from its saved, already-summed input)."""
        d, w = self.dims, self.store.w
        p = f"model.layers.{i}."
        if pending is None:
            h1, rstd1, _ = ops.rmsnorm_fwd(x, w[p + "input_layernorm.weight"], d.rms_norm_eps)
        else:
            h1, rstd1, x = ops.rmsnorm_fwd(pending, w[p + "input_layernorm.weight"], d.rms_norm_eps, residual=x)
        qkv = self._fwd_mm(h1, w[p + "self_attn.qkv_proj.weight"], bias=w.get(p + "self_attn.qkv_proj.bias"))
        q, k, rstd_qk = ops.qknorm_rope_fwd(qkv, w.get(p + "self_attn.q_norm.weight"), w.get(p + "self_attn.k_norm.weight"), cos, sin,
                                             d.num_attention_heads, d.num_key_value_heads, d.kernel_head_dim, d.rms_norm_eps)
        v = qkv[:, d.q_dim + d.kv_dim:].view(n, d.num_key_value_heads, d.kernel_head_dim)
        if kept_attn is not None:  # recompute = "selective": the attention output of the forward pass was kept
            (o, lse), attn_saved = kept_attn, None
        elif self.cp_size == 1:
            o, lse = ops.attn_varlen_fwd(q, k, v, seq_start, scale)
            attn_saved = None
        else:  # Ulysses: heads <-> sequence all-to-all around the unchanged varlen kernel (full pass, 1/cp of the heads); "ring": gathered K / V
            o, attn_saved = (self._ulysses_attn_fwd if self.cp_style == "ulysses" else self._gathered_kv_attn_fwd)(q, k, v, seq_start, scale)
            lse = None
        if kept_x2 is not None:  # recompute = "selective": the residual stream after attention was kept too (the already-rounded sum): no output projection
            h2, rstd2, _ = ops.rmsnorm_fwd(kept_x2, w[p + "post_attention_layernorm.weight"], d.rms_norm_eps)
            x2 = kept_x2
        else:
            attn_out = self._fwd_mm(o.view(n, d.q_dim), w[p + "self_attn.o_proj.weight"])
            h2, rstd2, x2 = ops.rmsnorm_fwd(attn_out, w[p + "post_attention_layernorm.weight"], d.rms_norm_eps, residual=x)
        wgu = w[p + "mlp.gate_up_proj.weight"]
        if self.fused_swiglu_fwd and "fwd" in self.gemm_own and not self.gemm_f32 and ops.gemm_swiglu_supported(h2, wgu):
            # the own core writes act = silu(gate) * up from the accumulators of the gate|up product (a tile = 128 features' gate AND up columns): the
            # stand-alone SwiGLU pass (read 2 I, write I bytes x 2 per token) disappears; gu is still written once, for the backward
            gu, act = ops.gemm_swiglu(h2, wgu, variant=self._gemm_variant)
        else:
            gu = self._fwd_mm(h2, wgu)
            act = ops.swiglu_fwd(gu)
        # (a layer re-run in the backward needs its activations, not its output: the down projection — a quarter of the layer's forward GEMM work — is skipped)
        pending = self._fwd_mm(act, w[p + "mlp.down_proj.weight"]) if want_out else None
        saved = dict(x=x, rstd1=rstd1, h1=h1, qkv=qkv, q=q, k=k, rstd_qk=rstd_qk, o=o, lse=lse, x2=x2, rstd2=rstd2, h2=h2, gu=gu,
                     act=None if self.recompute_act else act, attn_saved=attn_saved)
        return saved, x2, pending

    def _ulysses_attn_fwd(self, q: Tensor, k: Tensor, v: Tensor, seq_start: Tensor, scale: float):
        cp, grp, hkv = self.cp_size, self.cp_group, self.dims.num_key_value_heads
        v = v.contiguous()
        if hkv < cp:
            k, v = ulysses.replicate_kv_heads(k, cp), ulysses.replicate_kv_heads(v, cp)
        qh, kh, vh = ulysses.seq_to_head(q, cp, grp), ulysses.seq_to_head(k, cp, grp), ulysses.seq_to_head(v, cp, grp)
        oh, lse = ops.attn_varlen_fwd(qh, kh, vh, seq_start, scale)
        return ulysses.head_to_seq(oh, cp, grp), (qh, kh, vh, oh, lse)

    def _ulysses_attn_bwd(self, do: Tensor, saved, seq_start: Tensor, scale: float, dv_view: Tensor):
        cp, grp, hkv = self.cp_size, self.cp_group, self.dims.num_key_value_heads
        qh, kh, vh, oh, lse = saved
        doh = ulysses.seq_to_head(do.contiguous(), cp, grp)
        dqh, dkh, dvh = ops.attn_varlen_bwd(doh, qh, kh, vh, oh, lse, seq_start, scale)
        dq, dk, dv = ulysses.head_to_seq(dqh, cp, grp), ulysses.head_to_seq(dkh, cp, grp), ulysses.head_to_seq(dvh, cp, grp)
        if hkv < cp:
            dk, dv = ulysses.reduce_kv_replicas(dk, hkv), ulysses.reduce_kv_replicas(dv, hkv)
        dv_view.copy_(dv)
        return dq, dk

    def _gathered_kv_attn_fwd(self, q: Tensor, k: Tensor, v: Tensor, seq_start: Tensor, scale: float):
        """cp_style "ring": K / V of the whole pass gathered over the cp group; this rank's q rows against the prefix that precedes and contains them."""
        cp, grp, n_loc = self.cp_size, self.cp_group, q.shape[0]
        if n_loc % 128:
            raise PrimeMI355XError(f"cp_style='ring': {n_loc} tokens per cp rank — the row-window attention kernels need a multiple of 128 "
                                   "(pad the pass to cp x 128 tokens: data.pad_to_multiple_of)")
        r1 = (self.cp_rank + 1) * n_loc
        kf = ulysses.all_gather_tokens(k, cp, grp)[:r1]
        vf = ulysses.all_gather_tokens(v.contiguous(), cp, grp)[:r1]
        o, lse = ops.attn_varlen_fwd_rows(q, kf, vf, seq_start[:r1], scale, r1 - n_loc)
        return o, (q, kf, vf, o, lse)

    def _gathered_kv_attn_bwd(self, do: Tensor, saved, seq_start: Tensor, scale: float, dv_view: Tensor):
        """Returns (dq, finish): `finish()` -> dk of this rank's rows, with dv written to `dv_view`.  The dK / dV exchange (every rank's contribution to the kv
        rows of the ranks before it) runs on the comm stream as ONE ragged all-to-all per tensor — only the prefix chunks a peer owns are sent, nothing is
        zero-padded — so the caller can put the o-projection's weight-gradient GEMM between the attention kernels and `finish()`."""
        cp, grp = self.cp_size, self.cp_group
        q, kf, vf, o, lse = saved
        n_loc, r1 = q.shape[0], kf.shape[0]
        dq, dk_part, dv_part = ops.attn_varlen_bwd_rows(do.contiguous(), q, kf, vf, o, lse, seq_start[:r1], scale, r1 - n_loc)
        done = None
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                rk = ulysses.exchange_prefix_chunks(dk_part, self.cp_rank, cp, n_loc, grp)
                rv = ulysses.exchange_prefix_chunks(dv_part, self.cp_rank, cp, n_loc, grp)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
        else:
            rk = ulysses.exchange_prefix_chunks(dk_part, self.cp_rank, cp, n_loc, grp)
            rv = ulysses.exchange_prefix_chunks(dv_part, self.cp_rank, cp, n_loc, grp)

        def finish():
            if done is not None:
                torch.cuda.current_stream().wait_event(done)  # (dk_part / dv_part / rk / rv stay referenced by this closure until here)
                # rk / rv were allocated under the comm stream and are read (and, for one chunk, returned as dk) on the main stream from here on: tell the
                # caching allocator, or a later comm-stream allocation could be handed their blocks while main-stream kernels still read them
                for t in (rk, rv):
                    t.record_stream(torch.cuda.current_stream())
            dv_view.copy_(ulysses.sum_prefix_chunks(rv, n_loc))
            return ulysses.sum_prefix_chunks(rk, n_loc)

        return dq, finish

    # -- loss -------------------------------------------------------------------------------------------------------
    def loss(self, ctx: dict, scales: tuple[int, int, int]) -> dict:
        b: PackedBatch = ctx["batch"]
        c = self.loss_cfg
        if c.type == "custom":
            from .rl.loss import plugin_loss_over_pass

            loss, grad, per_mb, metrics = plugin_loss_over_pass(ctx["logprobs"], b, scales, c.custom_fn, self.dims.vocab_size)
            ctx.update(loss=loss, grad_logprobs=grad, sample_stats=None, sample_metrics=dict(loss_per_micro_batch=per_mb, metrics=metrics))
            return ctx
        params = LossParams(loss_type=_lib.PM_LOSS_IPO if c.type == "ipo" else _lib.PM_LOSS_DEFAULT, dppo_mask_low=c.dppo_mask_low,
                            dppo_mask_high=c.dppo_mask_high, adv_tau=c.adv_tau, kl_tau=c.kl_tau, ipo_threshold=c.ipo_threshold,
                            rl_scale=float(max(scales[0], 1)), ce_scale=float(max(scales[1], 1)), ref_kl_scale=float(max(scales[2], 1)), shift=1,
                            pad_logprob=math.log(1.0 / self.dims.vocab_size))
        loss, grad, stats, metrics = ops.grpo_loss_fwd_bwd(ctx["logprobs"], b.inference_logprobs, b.advantages, b.loss_mask, b.cu_samples, params,
                                                           ref_logprobs=b.ref_logprobs, rl_weights=b.rl_weights, ce_weights=b.ce_weights,
                                                           ref_kl_weights=b.ref_kl_weights, sample_starts_mb=b.sample_starts_mb)
        ctx.update(loss=loss, grad_logprobs=grad, sample_stats=stats, sample_metrics=metrics)
        if self.parity_probe is not None and "logprobs" in self.parity_probe and "loss" not in self.parity_probe:
            self.parity_probe.update(loss=loss.clone(), sample_stats=stats.clone(), sample_metrics=metrics.clone(), scales=scales, batch=b)
        return ctx

    # -- backward ---------------------------------------------------------------------------------------------------
    def backward(self, ctx: dict, accumulate: bool, last_pass: bool = True) -> None:
        d, w, g = self.dims, self.store.w, self.store.g
        b: PackedBatch = ctx["batch"]
        labels, inv_temp, ids = ctx["labels"], ctx["inv_temp"], ctx["ids"]
        n = ids.numel()
        grad_lp = ctx["grad_logprobs"] if self.cp_size == 1 else ulysses.shard_for_cp(ctx["grad_logprobs"], self.cp_rank, self.cp_size)
        wh, gh = self.store.head_weight(), self.store.head_grad()
        hf = ctx["hf"]
        dhf = torch.empty_like(hf)
        acc_head = accumulate
        head_t = "head" in self._wt and not self._dgrad_own_now  # K-contiguous [H, V] copy of the head weight for the library's dgrad GEMM (same reasoning as _dgrad)
        if head_t and self._wt_version.get("head") != self.store.version:
            ops.transpose_2d(wh, out=self._wt["head"])
            self._wt_version["head"] = self.store.version
        fused_chunk = 8192  # tokens per recompute + dlogits pass of the fused head (2.5 GB of bf16 dlogits at the Qwen3 vocab)
        todo = []
        for s, e, logits in ctx["logits_chunks"]:
            if logits is None:
                todo += [(s0, min(s0 + fused_chunk, e), None) for s0 in range(s, e, fused_chunk)]
            else:
                todo.append((s, e, logits))
        for ci, (s, e, logits) in enumerate(todo):
            if logits is None:
                dl = ops.lmhead_logprob_bwd(hf[s:e], wh, labels[s:e], inv_temp[s:e], ctx["logz"][s:e], grad_lp[s:e])
            else:
                dl = ops.logprob_bwd(logits, labels[s:e], inv_temp[s:e], ctx["logz"][s:e], grad_lp[s:e], out=logits)
            if self.gemm_f32:
                dhf[s:e] = torch.mm(dl, wh.float()).to(torch.bfloat16)
            elif self._dgrad_own_now and ops.gemm_supported(dl, wh.t()):
                self._own_gemm(dl, wh.t(), out=dhf[s:e])
            elif head_t:
                torch.mm(dl, self._wt["head"].t(), out=dhf[s:e])
            else:
                torch.mm(dl, wh, out=dhf[s:e])
            self._wgrad(dl, hf[s:e], gh, acc_head, k_contiguous_x=True,
                        final_name="lm_head.weight" if (last_pass and ci == len(todo) - 1 and not d.tie_word_embeddings) else None)
            acc_head = True
        ctx["logits_chunks"] = None
        # tied embeddings: the embedding backward below must accumulate on top of the head gradient
        emb_accumulate = accumulate or d.tie_word_embeddings
        dpend = ops.rmsnorm_bwd(dhf, ctx["x_last"], w["model.norm.weight"], ctx["rstd_f"], self._norm_grad(g["model.norm.weight"], accumulate))
        self._bucket_done("head", last_pass)
        split = last_pass and self._embed_plan is not None
        if split and d.tie_word_embeddings:  # the tied head gradient is the last dense write into the embedding bucket
            self._bucket_done("embed", True)
        # dpend = gradient w.r.t. (x + mlp_out) of the last layer = gradient of both the residual stream and the mlp output
        dx = dpend
        scale = d.head_dim**-0.5
        for i in reversed(range(d.num_hidden_layers)):
            p = f"model.layers.{i}."
            L = ctx["layers"][i]
            if L.get("recompute"):  # block-level recomputation: rebuild this layer's activations from its saved input (and, "selective", its kept attention output)
                L, _, _ = self._layer_forward(i, L["x"], None, ctx["cos"], ctx["sin"], ctx["seq_start"], scale, n, kept_attn=(L["o"], L["lse"]) if "o" in L else None,
                                              want_out=False, kept_x2=L.get("x2"))
            act = L["act"] if L["act"] is not None else ops.swiglu_fwd(L["gu"])
            w_down = w[p + "mlp.down_proj.weight"]
            if self.fused_swiglu_bwd and self._dgrad_own_now and not self.gemm_f32 and ops.gemm_swiglu_bwd_supported(dx, w_down):
                # the SwiGLU backward rides in the row-gradient GEMM's epilogue: d_act never reaches HBM, [d_gate | d_up] overwrite the saved gate | up
                dgu = ops.gemm_swiglu_bwd(dx, w_down, L["gu"], out=L["gu"], variant=self._gemm_variant)
            else:
                dgu = ops.swiglu_bwd(self._dgrad(dx, i, "mlp.down_proj.weight"), L["gu"])
            self._wgrad(dx, act, g[p + "mlp.down_proj.weight"], accumulate, k_contiguous_x=True,
                        final_name=p + "mlp.down_proj.weight" if last_pass else None)
            del act
            dh2 = self._dgrad(dgu, i, "mlp.gate_up_proj.weight")
            self._wgrad(dgu, L["h2"], g[p + "mlp.gate_up_proj.weight"], accumulate, k_contiguous_x=True,
                        final_name=p + "mlp.gate_up_proj.weight" if last_pass else None)
            dx2 = ops.rmsnorm_bwd(dh2, L["x2"], w[p + "post_attention_layernorm.weight"], L["rstd2"],
                                  self._norm_grad(g[p + "post_attention_layernorm.weight"], accumulate), dresidual=dx)
            do = self._dgrad(dx2, i, "self_attn.o_proj.weight").view(n, d.num_attention_heads, d.kernel_head_dim)
            o_wgrad = lambda: self._wgrad(dx2, L["o"].view(n, d.q_dim), g[p + "self_attn.o_proj.weight"], accumulate,  # noqa: E731
                                          final_name=p + "self_attn.o_proj.weight" if last_pass else None)
            qkv = L["qkv"]
            v = qkv[:, d.q_dim + d.kv_dim:].view(n, d.num_key_value_heads, d.kernel_head_dim)
            dqkv = torch.empty_like(qkv)
            dv_view = dqkv[:, d.q_dim + d.kv_dim:].view(n, d.num_key_value_heads, d.kernel_head_dim)
            if self.cp_size == 1:
                o_wgrad()
                dq, dk, _ = ops.attn_varlen_bwd(do, L["q"], L["k"], v, L["o"], L["lse"], ctx["seq_start"], scale, grad_v_out=dv_view)
            elif self.cp_style == "ulysses":
                o_wgrad()
                dq, dk = self._ulysses_attn_bwd(do, L["attn_saved"], ctx["seq_start"], scale, dv_view)
            else:  # gathered K / V: the dK / dV exchange runs on the comm stream under the o-projection's weight-gradient GEMM
                dq, finish = self._gathered_kv_attn_bwd(do, L["attn_saved"], ctx["seq_start"], scale, dv_view)
                o_wgrad()
                dk = finish()
            if d.qk_norm:
                ops.qknorm_rope_bwd(dq, dk, qkv, w[p + "self_attn.q_norm.weight"], w[p + "self_attn.k_norm.weight"], ctx["cos"], ctx["sin"],
                                    L["rstd_qk"], dqkv, self._norm_grad(g[p + "self_attn.q_norm.weight"], accumulate),
                                    self._norm_grad(g[p + "self_attn.k_norm.weight"], accumulate), d.num_attention_heads, d.num_key_value_heads,
                                    d.kernel_head_dim)
            else:
                ops.qknorm_rope_bwd(dq, dk, None, None, None, ctx["cos"], ctx["sin"], None, dqkv, None, None, d.num_attention_heads,
                                    d.num_key_value_heads, d.kernel_head_dim)
            if d.attention_bias:  # grad_bias = column sums of the projection's output gradient
                ops.column_sum(dqkv, self._norm_grad(g[p + "self_attn.qkv_proj.bias"], accumulate))
            dh1 = self._dgrad(dqkv, i, "self_attn.qkv_proj.weight")
            self._wgrad(dqkv, L["h1"], g[p + "self_attn.qkv_proj.weight"], accumulate, final_name=p + "self_attn.qkv_proj.weight" if last_pass else None)
            dx = ops.rmsnorm_bwd(dh1, L["x"], w[p + "input_layernorm.weight"], L["rstd1"],
                                 self._norm_grad(g[p + "input_layernorm.weight"], accumulate), dresidual=dx2)
            ctx["layers"][i] = None
            self._bucket_done(f"layer{i}", last_pass)
        if split:
            self._embed_rows_to_owners(dx, ids)
            return
        if not emb_accumulate and not self._emb_zeroed:
            g["model.embed_tokens.weight"].zero_()
        self._emb_zeroed = False
        ops.embedding_bwd(dx, ids, g["model.embed_tokens.weight"])
        self._bucket_done("embed", last_pass)

    @staticmethod
    def _norm_grad(gview: Tensor, accumulate: bool) -> Tensor:
        if not accumulate:
            gview.zero_()  # the norm-weight kernels accumulate (+=); tiny vectors
        return gview

    # -- data-parallel plumbing (RCCL on a side stream; trainer/zero.py) -------------------------------------------------
    def _bucket_done(self, name: str, last_pass: bool) -> None:
        if not self.use_collectives or not last_pass:
            return
        b = next(bk for bk in self.store.buckets if bk.name == name)
        zero.reduce_scatter_bucket(self.store, b, self.shard_group, self.comm_stream)

    def _plan_embed_split(self, passes: list) -> dict | None:
        """Exchange (tokens of the last pass, number of passes) over the shard group: the row all-gather needs one capacity on every rank, and the
        early dense reduce-scatter is a collective, so every rank must take the same branch even when pass counts differ."""
        if not self.embed_split:
            return None
        n_last = passes[-1].n_tokens // self.cp_size if self.cp_size > 1 else passes[-1].n_tokens
        mine = torch.tensor([n_last, len(passes)], dtype=torch.int64, device=self.device)
        every = torch.empty(self.shard_size * 2, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(every, mine, group=self.shard_group)
        every = every.view(self.shard_size, 2).tolist()
        return {"capacity": max(r[0] for r in every), "dense": self.dims.tie_word_embeddings or any(r[1] > 1 for r in every), "passes": len(passes)}

    def _embed_dense_early(self) -> None:
        """Untied embeddings, just before the last pass: reduce-scatter what the earlier passes accumulated.  A rank with a single pass joins the
        collective with zeros when some other rank has earlier passes."""
        plan = self._embed_plan
        if plan is None or self.dims.tie_word_embeddings or not plan["dense"]:
            return
        if plan["passes"] == 1:
            self.store.g["model.embed_tokens.weight"].zero_()
        self._bucket_done("embed", True)

    def _embed_rows_to_owners(self, dx: Tensor, ids: Tensor) -> None:
        """grad_shard[embedding shard] (+)= the last pass's embedding gradient of EVERY rank, restricted to this rank's window of the table."""
        plan, st, d = self._embed_plan, self.store, self.dims
        cap, world, n = plan["capacity"], self.shard_size, ids.numel()
        bufs = self._embed_bufs
        if bufs is None or bufs["cap"] < cap:
            bufs = self._embed_bufs = {
                "cap": cap,
                "ids": torch.empty(cap, dtype=torch.int64, device=self.device),
                "rows": torch.empty(cap, d.hidden_size, dtype=torch.bfloat16, device=self.device),
                "all_ids": torch.empty(world * cap, dtype=torch.int64, device=self.device),
                "all_rows": torch.empty(world * cap, d.hidden_size, dtype=torch.bfloat16, device=self.device),
            }
        cap = bufs["cap"]
        bufs["ids"][:n].copy_(ids.reshape(-1))
        bufs["ids"][n:].fill_(-1)  # padding up to the common capacity: skipped by the kernel (the rows behind it are never read)
        bufs["rows"][:n].copy_(dx)
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                dist.all_gather_into_tensor(bufs["all_ids"], bufs["ids"], group=self.shard_group)
                dist.all_gather_into_tensor(bufs["all_rows"], bufs["rows"], group=self.shard_group)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            torch.cuda.current_stream().wait_event(done)  # also orders this after the early dense reduce-scatter (same stream)
        else:
            dist.all_gather_into_tensor(bufs["all_ids"], bufs["ids"], group=self.shard_group)
            dist.all_gather_into_tensor(bufs["all_rows"], bufs["rows"], group=self.shard_group)
        bucket = st.buckets[0]
        assert bucket.name == "embed" and bucket.params[0][1] == bucket.offset
        _, local = st.shard_slices(bucket)
        window = st.grad_shard[local]
        if not plan["dense"]:
            window.zero_()
        ops.embedding_bwd_window(bufs["all_rows"], bufs["all_ids"], window, self.shard_rank * window.numel(), d.vocab_size)

    def wait_params_ready(self) -> None:
        """Make the current stream wait for every in-flight parameter all-gather (issued on the comm stream by the last optimizer
        step): anything that reads `store.params` outside forward() — the weight push, a checkpoint — must call this first."""
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._gather_events.clear()

    def add_param_reader(self, event) -> None:
        """Register the completion event of an asynchronous reader of `store.params` (transports/weights: the push stream)."""
        self._param_readers.append(event)

    def _wait_gather(self, name: str) -> None:
        ev = self._gather_events.pop(name, None)
        if ev is not None and self.device.type == "cuda":
            zero._timed("exposed", torch.cuda.current_stream(), lambda: torch.cuda.current_stream().wait_event(ev))

    # -- optimizer ---------------------------------------------------------------------------------------------------
    def optimizer_step(self) -> Tensor:
        """scale_gradients_ + clip_grad_norm_ + AdamW (train.py:557-570) on this rank's shard; returns the global grad norm (device scalar).

        The loss is already divided by the GLOBAL token counts, so summed (not averaged) gradients across ranks are the
        global per-token mean: the reference's `FSDP mean` followed by `scale_gradients_(x dp)` (train.py:557-558) is the
        same sum, here done in one reduce-scatter(SUM)."""
        self.step_count += 1
        self.store.version += 1  # parameters change: derived weight layouts are refreshed on next use
        for ev in self._param_readers:  # an in-flight weight push still reads the parameters this step is about to overwrite
            if self.device.type != "cuda":
                break
            torch.cuda.current_stream().wait_event(ev)
            if self.comm_stream is not None:
                self.comm_stream.wait_event(ev)
        self._param_readers.clear()
        return zero.sharded_optimizer_step(self.store, self.optim, self.step_count, self.shard_group, ops, self.comm_stream, self._gather_events, self._sumsq,
                                           local_sumsq=self._fs_norm if self.fused_sumsq else None)

    # -- one full trainer step -------------------------------------------------------------------------------------------
    def global_scales(self, local_counts: tuple[int, int, int]) -> tuple[int, int, int]:
        """dist.all_reduce of the i64[3] token counts (train.py:318-323) over the DATA-parallel group: the cp ranks of one dp position hold
        the same pass, whose tokens count once (the reference reduces over dp x cp and lets the cp-fold count cancel against the cp-fold
        gradient of its gathered log-probs, utils/cp.py:129-132 — same quotient)."""
        if self.dp_size == 1:
            return tuple(max(int(c), 1) for c in local_counts)
        return zero.all_reduce_counts(local_counts, self.dp_group, self.device)

    def step(self, batch: PackedBatch | list[PackedBatch]) -> dict:
        """One optimizer step over one or more packed passes (a list accumulates gradients across passes)."""
        passes = batch if isinstance(batch, list) else [batch]
        self._dgrad_own_now = "dgrad" in self.gemm_own or sum(p.n_tokens for p in passes) // self.cp_size < self.dgrad_own_below
        counts = tuple(sum(p.local_counts[k] for p in passes) for k in range(3))
        scales = self.global_scales(counts)
        out = {"loss": [], "entropy": [], "logprobs": [], "sample_metrics": [], "sample_stats": []}
        self._embed_plan = self._plan_embed_split(passes)
        for idx, p in enumerate(passes):
            if idx == len(passes) - 1:
                self._embed_dense_early()
            # first pass of a step whose embedding gradient is a plain local buffer: its zero fill is placed inside forward() (see there)
            self._emb_zero_early = idx == 0 and self._embed_plan is None and not self.dims.tie_word_embeddings and self.lm_head_impl != "fused"
            ctx = self.forward(p)
            self._emb_zero_early = False
            self.loss(ctx, scales)
            out["loss"].append(ctx["loss"])
            out["entropy"].append(ctx["entropy"])
            out["logprobs"].append(ctx["logprobs"])
            out["sample_metrics"].append(ctx["sample_metrics"])
            out["sample_stats"].append(ctx["sample_stats"])
            self.backward(ctx, accumulate=idx > 0, last_pass=idx == len(passes) - 1)
            del ctx
        self._embed_plan = None
        out["grad_norm"] = self.optimizer_step()
        out["scales"] = scales
        return out
