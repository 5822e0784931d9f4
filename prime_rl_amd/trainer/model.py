"""Engine-API seams of the reference's `prime_rl.trainer.model` kept verbatim (SURVEY.md §8(b) "Engine API kept verbatim"):

    setup_model(config: ModelConfig, parallel_dims: ParallelDims, loading_from_checkpoint_later=False) -> nn.Module   model.py:1314-1428
    forward(model, input_ids, position_ids, *, seq_lens, labels, temperature, ...) -> PrimeLmOutput                  model.py:1431-1482
    get_model(config, device, dtype) -> nn.Module                                                                    model.py:715-800

plus `setup_engine(config: TrainerConfig, parallel_dims) -> GRPOEngine`, the flat-buffer product engine `train()` drives.

`setup_model` returns the torch.autograd flavour of the MI355X path: the `Qwen3ForCausalLM` mirror whose layers call the HIP
kernels through the C ABI (trainer/models/), with `inject_prime_lm_head(model, chunk_size)` applied exactly where the reference
applies it (model.py:1348-1352).  What the reference does between those lines and that has no MI355X counterpart is deliberately
absent and says so: activation checkpointing / offloading, torch.compile, FSDP2 wrapping and CPU offload exist to fit 80 GB parts
 — on 288 GB the sharding lives in GRPOEngine (ZeRO-2 over RCCL, trainer/zero.py), not in module wrappers.
Weights: offline there is no HF hub; `config.name` may be a LOCAL directory holding `config.json` (+ `*.safetensors`), a known model
name with `config.debug.random_init = true`, or `loading_from_checkpoint_later = True` (weights arrive through load_state_dict)."""

from __future__ import annotations

import json
import os
from pathlib import Path

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .engine import GRPOEngine, LossConfig, ModelDims, OptimConfig
from .models.layers.lm_head import PrimeLmOutput, cast_float_and_contiguous, inject_prime_lm_head
from .models.llama.modeling_llama import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
from .models.qwen3.modeling_qwen3 import Qwen3Config, Qwen3ForCausalLM
from .models.qwen3_moe.modeling_qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
from .parallel_dims import ParallelDims

DTYPE_MAP = {"bfloat16": torch.bfloat16, "float32": torch.float32}

_QWEN2 = dict(qk_norm=False, attention_bias=True)
# public HF configs of the dense families the reference registers as custom PrimeRL models (trainer/models/__init__.py:41-59)
KNOWN_MODELS: dict[str, dict] = {
    "qwen3-8b": dict(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8),
    "qwen3-4b": dict(vocab_size=151936, hidden_size=2560, intermediate_size=9728, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8,
                     tie_word_embeddings=True),
    "qwen3-1.7b": dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
                       tie_word_embeddings=True),
    "qwen3-0.6b": dict(vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
                       tie_word_embeddings=True),
    "tiny": dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2),
    # Qwen2.5 (HF Qwen2: q/k/v bias, no q/k norm) — BASELINE.json configs[1] is Qwen2.5-1.5B-Instruct; only the head_dim-128 sizes run here
    "qwen2.5-0.5b": dict(vocab_size=151936, hidden_size=896, intermediate_size=4864, num_hidden_layers=24, num_attention_heads=14, num_key_value_heads=2,
                         head_dim=64, tie_word_embeddings=True, **_QWEN2),  # BASELINE.json configs[0]; 64-wide heads run zero-padded to 128 in the engine
    "qwen2.5-1.5b": dict(vocab_size=151936, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2,
                         tie_word_embeddings=True, **_QWEN2),
    "qwen2.5-3b": dict(vocab_size=151936, hidden_size=2048, intermediate_size=11008, num_hidden_layers=36, num_attention_heads=16, num_key_value_heads=2,
                       tie_word_embeddings=True, **_QWEN2),
    "qwen2.5-7b": dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                       **_QWEN2),
    "tiny-qwen2": dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       tie_word_embeddings=True, **_QWEN2),
    "tiny-llama": dict(vocab_size=512, hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       qk_norm=False, rms_norm_eps=1e-5, rope_theta=500_000.0),
}
_LLAMA3 = dict(qk_norm=False, rms_norm_eps=1e-5, rope_theta=500_000.0,
               rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192))
KNOWN_MODELS["llama-3.1-8b"] = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                                    num_key_value_heads=8, **_LLAMA3)
KNOWN_MODELS["llama-3.2-3b"] = dict(vocab_size=128256, hidden_size=3072, intermediate_size=8192, num_hidden_layers=28, num_attention_heads=24,
                                    num_key_value_heads=8, tie_word_embeddings=True,
                                    **{**_LLAMA3, "rope_scaling": dict(_LLAMA3["rope_scaling"], factor=32.0)})
for _k in ("llama-3.1-8b", "llama-3.2-3b"):
    KNOWN_MODELS[_k + "-instruct"] = KNOWN_MODELS[_k]
KNOWN_MODELS["llama-3.2-1b"] = dict(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32,
                                    num_key_value_heads=8, head_dim=64, tie_word_embeddings=True,
                                    **{**_LLAMA3, "rope_scaling": dict(_LLAMA3["rope_scaling"], factor=32.0)})
KNOWN_MODELS["llama-3.2-1b-instruct"] = KNOWN_MODELS["llama-3.2-1b"]
for _k in ("qwen2.5-0.5b", "qwen2.5-1.5b", "qwen2.5-3b", "qwen2.5-7b"):
    KNOWN_MODELS[_k + "-instruct"] = KNOWN_MODELS[_k]
# MoE families (models/qwen3_moe): the dense fields above plus the expert fields of configuration_qwen3_moe.py:150-205
KNOWN_MOE_MODELS: dict[str, dict] = {
    "qwen3-30b-a3b": dict(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=48, num_attention_heads=32,
                          num_key_value_heads=4, num_experts=128, num_experts_per_tok=8, moe_intermediate_size=768, norm_topk_prob=True),
    "tiny-moe": dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                     num_experts=8, num_experts_per_tok=2, moe_intermediate_size=128, norm_topk_prob=True),
}
_MOE_KEYS = ("num_experts", "num_experts_per_tok", "moe_intermediate_size", "norm_topk_prob", "decoder_sparse_step", "mlp_only_layers")


def resolve_moe_config(config) -> Qwen3MoeConfig | None:
    """ModelConfig -> Qwen3MoeConfig when `name` is a Qwen3-MoE model (local HF directory with model_type qwen3_moe, or a known name), else None."""
    p = Path(config.name)
    if (p / "config.json").exists():
        hf = json.loads((p / "config.json").read_text())
        if hf.get("model_type") != "qwen3_moe":
            return None
        fields = {k: hf[k] for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads") if k in hf}
        fields["num_key_value_heads"] = hf.get("num_key_value_heads", hf["num_attention_heads"])
        fields["head_dim"] = hf.get("head_dim", hf["hidden_size"] // hf["num_attention_heads"])
        fields["rms_norm_eps"] = hf.get("rms_norm_eps", 1e-6)
        fields["rope_theta"] = hf.get("rope_theta", (hf.get("rope_parameters") or {}).get("rope_theta", 1_000_000.0))
        fields["tie_word_embeddings"] = hf.get("tie_word_embeddings", False)
        fields.update({k: hf[k] for k in _MOE_KEYS if hf.get(k) is not None})
    else:
        key = config.name.split("/")[-1].lower()
        if key not in KNOWN_MOE_MODELS:
            return None
        fields = dict(KNOWN_MOE_MODELS[key])
    if config.debug.num_layers is not None:
        fields["num_hidden_layers"] = min(config.debug.num_layers, fields["num_hidden_layers"])
    return Qwen3MoeConfig(**fields)


def resolve_model_dims(config) -> ModelDims:
    """ModelConfig -> ModelDims: `name` is a local HF directory (config.json) or a known dense Qwen3 name ("Qwen/Qwen3-8B",
    "PrimeIntellect/Qwen3-0.6B", "qwen3-8b", ...); `debug.num_layers` truncates the stack (configs/trainer.py:131-136)."""
    name = config.name
    p = Path(name)
    if (p / "config.json").exists():
        hf = json.loads((p / "config.json").read_text())
        family = hf.get("model_type")
        if family not in ("qwen3", "llama", "qwen2"):
            raise NotImplementedError(f"model_type {family!r}: the MI355X scoring forward is built for the dense Qwen3, Llama and Qwen2 decoders")
        scaling = dict(hf.get("rope_scaling") or (hf.get("rope_parameters") or {}))
        scaling.pop("rope_theta", None)
        from ..utils.rope import SUPPORTED, rope_type_of

        if rope_type_of(scaling) not in SUPPORTED:
            raise NotImplementedError(f"rope_scaling {rope_type_of(scaling)!r}: pm_rope_table takes default, linear and llama3 frequencies (utils/rope.py); "
                                      "yarn / longrope / dynamic rescale cos and sin or depend on the sequence length")
        if rope_type_of(scaling) == "llama3":
            scaling.setdefault("original_max_position_embeddings", hf.get("max_position_embeddings"))
        if hf.get("mlp_bias") or (family != "qwen2" and hf.get("attention_bias")):
            raise NotImplementedError("mlp_bias / o_proj bias are not on the MI355X path (q/k/v bias is: model_type qwen2)")
        fields = dict(vocab_size=hf["vocab_size"], hidden_size=hf["hidden_size"], intermediate_size=hf["intermediate_size"],
                      num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=hf["num_attention_heads"],
                      num_key_value_heads=hf.get("num_key_value_heads", hf["num_attention_heads"]),
                      head_dim=hf.get("head_dim", hf["hidden_size"] // hf["num_attention_heads"]), rms_norm_eps=hf.get("rms_norm_eps", 1e-6),
                      rope_theta=hf.get("rope_theta", (hf.get("rope_parameters") or {}).get("rope_theta", 1_000_000.0)),
                      tie_word_embeddings=hf.get("tie_word_embeddings", False), qk_norm=family == "qwen3", attention_bias=family == "qwen2",
                      rope_scaling=scaling if rope_type_of(scaling) != "default" else None)
        label = p.name
    else:
        key = name.split("/")[-1].lower()
        if key not in KNOWN_MODELS:
            raise ValueError(f"model.name = {name!r}: not a local HF directory and not one of {sorted(KNOWN_MODELS)} (no hub access here)")
        fields, label = dict(KNOWN_MODELS[key]), name.split("/")[-1]
    if config.debug.num_layers is not None:
        fields["num_hidden_layers"] = min(config.debug.num_layers, fields["num_hidden_layers"])
    return ModelDims(name=label, **fields)


def _local_safetensors(name: str) -> dict[str, Tensor] | None:
    p = Path(name)
    files = sorted(p.glob("*.safetensors")) if p.is_dir() else []
    if not files:
        return None
    from safetensors.torch import load_file

    sd: dict[str, Tensor] = {}
    for f in files:
        sd.update(load_file(str(f)))
    return sd


def resolve_auto_attn(config) -> None:
    """model.py:1294-1311 picks a flash-attention generation from the CUDA capability; on gfx950 there is one implementation."""
    if config.attn in ("auto", "flash_attention_2", "flash_attention_3", "flash_attention_4"):
        config.attn = "hip_varlen"


def apply_moe_options(model: nn.Module, config) -> None:
    """The MoE switches of setup_model (model.py:1358-1375): `moe_router_dtype = "float32"` (the default: f32 gate weight and gate GEMM,
    moe.py:674-676 / model.py:863-872), `freeze_moe_router` (model.py:395-420), `debug.force_balanced_routing` (model.py:522-540).  The fused
    Triton MoE kernel has no counterpart here: the expert path already runs on the grouped MFMA kernels."""
    if getattr(config, "moe_fused_kernel", False):
        raise NotImplementedError("model.moe_fused_kernel selects the reference's Triton kernel; the experts here always run on pm_grouped_gemm_*")
    for mlp in model.moe_layers():
        if config.moe_router_dtype == "float32":
            mlp.router.gate.float()
            mlp.router.fp32_gate = True
        if config.freeze_moe_router:
            for p in mlp.router.parameters():
                p.requires_grad = False
        if config.debug.force_balanced_routing:
            mlp.router.force_balanced = True


def get_load_balance_stats(model: nn.Module, reset_stats: bool = True, try_to_avoid_padding_experts: bool = True) -> dict[str, Tensor | None]:
    """model.py:561-592: per MoE layer, the maximal load violation `(max tokens_per_expert - mean) / mean` (optionally without the top_k most loaded
    experts, which absorb padding) and the routing confidence `routing_confidence_sum / routed tokens`; resets the layer counters."""
    per_layer_max_vio, per_layer_conf = [], []
    for block in model.model.layers:
        mlp = getattr(block, "mlp", None)
        if mlp is None or not hasattr(mlp, "tokens_per_expert"):
            continue
        tpe = mlp.tokens_per_expert
        routed = tpe.sum() / mlp.router.top_k
        if try_to_avoid_padding_experts:
            tpe = tpe.sort(dim=0, descending=True).values[mlp.router.top_k:]
        balanced = tpe.mean()
        per_layer_max_vio.append(((tpe.max() - balanced) / balanced).detach())
        per_layer_conf.append((mlp.routing_confidence_sum / routed).detach())
        if reset_stats:
            mlp.tokens_per_expert.zero_()
            mlp.routing_confidence_sum.zero_()
    if not per_layer_max_vio:
        return {"max_vio": None, "routing_confidence": None}
    return {"max_vio": torch.stack(per_layer_max_vio), "routing_confidence": torch.stack(per_layer_conf)}


def get_model(config, device: torch.device | str = "cuda", dtype: torch.dtype = torch.bfloat16, ep_group=None) -> nn.Module:
    """The custom PrimeRL class for the config's family (model.py:715-800, `impl` "custom"/"auto"); HF-impl models are out of scope."""
    if config.impl == "hf":
        raise NotImplementedError("model.impl = 'hf' (transformers modelling code + flash-attn) is outside the MI355X path; use 'custom' or 'auto'")
    moe_cfg = resolve_moe_config(config)
    if moe_cfg is not None:
        with torch.device(device):
            model = Qwen3MoeForCausalLM(moe_cfg, ep_group=ep_group)
        return model.to(dtype)
    if ep_group is not None:
        raise ValueError(f"model.ep > 1 needs an MoE model; {config.name!r} is dense")
    d = resolve_model_dims(config)
    # family from the two attention switches: Qwen3 (q/k norm), Llama (neither; the reference's custom class), Qwen2 (q/k/v bias; HF's class in the
    # reference, models/__init__.py:41-59 does not register Qwen2Config)
    cfg_cls, model_cls = (Qwen3Config, Qwen3ForCausalLM) if d.qk_norm else ((Qwen2Config, Qwen2ForCausalLM) if d.attention_bias else (LlamaConfig, LlamaForCausalLM))
    hf_cfg = cfg_cls(vocab_size=d.vocab_size, hidden_size=d.hidden_size, intermediate_size=d.intermediate_size, num_hidden_layers=d.num_hidden_layers,
                     num_attention_heads=d.num_attention_heads, num_key_value_heads=d.num_key_value_heads, head_dim=d.head_dim,
                     rms_norm_eps=d.rms_norm_eps, rope_theta=d.rope_theta, tie_word_embeddings=d.tie_word_embeddings, use_qk_norm=d.qk_norm,
                     attention_bias=d.attention_bias, rope_scaling=d.rope_scaling)
    with torch.device(device):
        model = model_cls(hf_cfg)
    return model.to(dtype)


def setup_model(config, parallel_dims: ParallelDims, loading_from_checkpoint_later: bool = False) -> nn.Module:
    """model.py:1314-1428 for the dense data-parallel path.  Compute dtype is bf16 (the reference's FSDP MixedPrecisionPolicy param
    dtype, model.py:823-960); `optimization_dtype` (f32 master weights) is the engine's / optimizer's business, not the module's."""
    resolve_auto_attn(config)
    ep_group = parallel_dims.get_group("ep") if parallel_dims.ep_enabled else None
    if getattr(config, "ep_comm_backend", "torch") != "torch":
        raise NotImplementedError("model.ep_comm_backend = 'deepep': expert dispatch here is RCCL all-to-all (trainer/distributed/expert_parallel.py)")
    for name in ("lora", "quantization", "vlm"):
        if getattr(config, name, None) is not None:
            raise NotImplementedError(f"model.{name} is outside the MI355X GRPO hot path (SURVEY.md section 2)")
    model = get_model(config, device="cuda", dtype=torch.bfloat16, ep_group=ep_group)
    if isinstance(model, Qwen3MoeForCausalLM):
        apply_moe_options(model, config)
    chunk = config.fused_lm_head_token_chunk_size if isinstance(config.fused_lm_head_token_chunk_size, int) else None
    inject_prime_lm_head(model, chunk_size=chunk)
    if loading_from_checkpoint_later:
        return model
    sd = _local_safetensors(config.name)
    if sd is not None:
        if model.config.tie_word_embeddings:
            sd.pop("lm_head.weight", None)
        if isinstance(model, Qwen3MoeForCausalLM):  # HF per-expert names -> stacked w1 / w2 / w3, this rank's experts only under EP
            from .models.qwen3_moe.converting_qwen3_moe import hf_to_prime

            n_exp = model.config.num_experts
            ep = dist.get_world_size(ep_group) if ep_group is not None else 1
            first = (dist.get_rank(ep_group) if ep_group is not None else 0) * (n_exp // ep)
            sd = hf_to_prime(sd, n_exp, range(first, first + n_exp // ep))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        missing = [k for k in missing if not (model.config.tie_word_embeddings and k == "lm_head.weight")]
        if missing or unexpected:
            raise KeyError(f"checkpoint / model mismatch: missing {missing[:4]}, unexpected {list(unexpected)[:4]}")
    elif config.debug.random_init:
        gen = torch.Generator(device="cuda").manual_seed(0)
        with torch.no_grad():
            for pname, p in model.named_parameters():
                if p.dim() == 1 and not pname.endswith(".bias"):
                    p.fill_(1.0)
                else:
                    p.copy_((torch.randn(p.shape, device="cuda", generator=gen) * 0.02).to(p.dtype))
    else:
        raise FileNotFoundError(f"no weights for {config.name!r}: point model.name at a local HF directory with *.safetensors, or set "
                                "model.debug.random_init = true (there is no hub access in this environment)")
    return model


def forward(model: nn.Module, input_ids: Tensor, position_ids: Tensor, *, seq_lens: Tensor, labels: Tensor | None = None,
            temperature: Tensor | None = None, routed_experts: Tensor | None = None, mm_kwargs: dict | None = None,
            mm_token_type_ids: Tensor | None = None, seq_lens_are_pre_shard: bool = False) -> PrimeLmOutput:
    """model.py:1431-1482: build the kwargs, run the backbone + head, return float / contiguous outputs."""
    if mm_kwargs or mm_token_type_ids is not None:
        raise NotImplementedError("multimodal inputs are outside the GRPO path built here (SURVEY.md section 2 row 12)")
    kw = {}
    if routed_experts is not None:  # router replay (model.py:1462-1463): only MoE models take it
        if not isinstance(model, Qwen3MoeForCausalLM):
            raise ValueError("routed_experts given to a dense model")
        kw["routed_experts"] = routed_experts
    out = model(input_ids=input_ids, position_ids=position_ids, labels=labels, temperature=temperature, seq_lens=seq_lens,
                seq_lens_are_pre_shard=seq_lens_are_pre_shard, **kw)
    if isinstance(out, dict):
        return cast_float_and_contiguous(out)
    return cast_float_and_contiguous(PrimeLmOutput(logits=out.logits))


def loss_from_config(lc) -> LossConfig:
    """LossConfig union (default | ipo | custom: configs/trainer.py:494-532) -> the engines' LossConfig; a custom plugin is imported here
    (`import_path`, called with `kwargs`: setup_rl_loss_fn, loss.py:262-282)."""
    if lc.type == "custom":
        from .rl.loss import setup_rl_loss_fn

        return LossConfig(type="custom", custom_fn=setup_rl_loss_fn(lc))
    return LossConfig(type=lc.type, dppo_mask_low=getattr(lc, "dppo_mask_low", 0.2), dppo_mask_high=getattr(lc, "dppo_mask_high", 0.2), adv_tau=lc.adv_tau,
                      kl_tau=lc.kl_tau, ipo_threshold=getattr(lc, "ipo_threshold", 0.1))


def optim_from_config(o) -> OptimConfig:
    """OptimizerConfig (adamw | sgd | sign_sgd: configs/trainer.py) -> the engine's OptimConfig; Muon is refused by validate_supported()."""
    return OptimConfig(lr=o.lr, weight_decay=o.weight_decay, betas1=getattr(o, "betas1", 0.9), betas2=getattr(o, "betas2", 0.999), eps=getattr(o, "eps", 1e-8),
                       max_norm=o.max_norm if o.max_norm is not None else 0.0, type=o.type, momentum=getattr(o, "momentum", 0.0),
                       nesterov=getattr(o, "nesterov", False))


def setup_module_engine(config, parallel_dims: ParallelDims, device: torch.device | str = "cuda"):
    """TrainerConfig -> the nn.Module flavour of the trainer step (trainer/module_engine.py) on the model `setup_model` builds: the path of
    the MoE families and of `model.ep > 1` (BASELINE.json configs[3]: Qwen3-30B-A3B, expert parallel over the node)."""
    from .module_engine import ModuleEngine

    config.validate_supported()
    if parallel_dims.cp_enabled:
        raise NotImplementedError("context parallelism runs in the flat-buffer engine (dense models); MoE models take model.cp = 1")
    model = setup_model(config.model, parallel_dims)
    o, lc = config.optim, config.loss
    optim = optim_from_config(o)
    loss = loss_from_config(lc)
    ep = parallel_dims.ep if parallel_dims.ep_enabled else 1
    return ModuleEngine(model, device, optim=optim, loss=loss, dp_group=parallel_dims.get_group("dp"), dp_size=parallel_dims.dp_size,
                        ep_group=parallel_dims.get_group("ep") if ep > 1 else None, ep_size=ep,
                        ep_replica_group=parallel_dims.get_group("ep_replica") if ep > 1 else None, name=config.model.name.split("/")[-1],
                        router_replay=bool(config.enable_router_replay))


def recompute_from_config(m) -> dict:
    """`model.ac` (configs/trainer.py ActivationCheckpointConfig; the reference applies it in trainer/model.py:1141-1185) -> the engine's recompute arguments.
    `full` = block-level recomputation, `selective` = everything of a layer but the attention core (the engine keeps its output and lse: the reference's
    norm / attn_proj / mlp targets taken together), `freq` = every freq-th layer only.  PM_RECOMPUTE / PM_RECOMPUTE_FREQ override (288 GB hold the activations
    of 16k-token passes: `PM_RECOMPUTE=none` is what bench.py's headline runs)."""
    import os

    # A `model.ac` the user wrote is honoured; the schema's DEFAULT (full checkpointing of every layer — sized for 80 GB parts) is not applied: 288 GB hold the
    # activations of the passes the default configs make, and recomputation costs a forward per layer for nothing.
    ac = getattr(m, "ac", None) if "ac" in getattr(m, "model_fields_set", ()) else None
    mode = {"full": "layer", "selective": "selective"}[ac.mode] if ac is not None else "none"
    return {"recompute": os.environ.get("PM_RECOMPUTE", mode), "recompute_freq": int(os.environ.get("PM_RECOMPUTE_FREQ", ac.freq if ac is not None else 1))}


def uses_module_engine(config, parallel_dims: ParallelDims) -> bool:
    return resolve_moe_config(config.model) is not None or parallel_dims.ep_enabled


def setup_engine(config, parallel_dims: ParallelDims, device: torch.device | str = "cuda") -> GRPOEngine:
    """TrainerConfig -> the flat-buffer GRPO engine: ZeRO-2 over the `dp` group, (Ulysses over the `cp` group when cp > 1), fused AdamW
    with the config's optimizer fields, the config's loss.  Weights as in `setup_model`."""
    config.validate_supported()
    m = config.model
    resolve_auto_attn(m)
    if resolve_moe_config(m) is not None or parallel_dims.ep_enabled:
        raise NotImplementedError("the flat-buffer engine is the dense path; MoE models (and model.ep > 1) run through setup_module_engine() "
                                  "(trainer/module_engine.py) on setup_model() / forward()")
    dims = resolve_model_dims(m)
    if config.enable_router_replay:
        raise NotImplementedError("enable_router_replay: dense models have no router (MoE models take it through setup_module_engine)")
    o = config.optim
    optim = optim_from_config(o)
    loss = loss_from_config(config.loss)
    engine = GRPOEngine(dims, device=device, optim=optim, loss=loss, dp_group=parallel_dims.get_group("dp"), seed=config.seed,
                        dp_size=parallel_dims.dp_size, cp_group=parallel_dims.get_group("cp"), cp_size=parallel_dims.cp,
                        shard_group=parallel_dims.get_group("dp_cp"), cp_style=m.cp_style, **recompute_from_config(m))
    sd = _local_safetensors(m.name)
    if sd is not None:
        engine.store.load_state_dict({k: v.to(device) for k, v in sd.items()})
    elif not m.debug.random_init:
        raise FileNotFoundError(f"no weights for {m.name!r}: point model.name at a local HF directory with *.safetensors, or set "
                                "model.debug.random_init = true (there is no hub access in this environment)")
    return engine
