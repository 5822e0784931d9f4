"""`TrainerConfig` — the reference's RL-trainer config surface (packages/prime-rl-configs/src/prime_rl/configs/trainer.py:22-700,
configs/shared.py:145-238) accepted key for key, so a reference TOML (`configs/debug/fake/rl.toml`, the `[trainer]` tables of the
`rl` launcher) or its `--model.seq_len 2048`-style CLI drives `prime_rl_amd.trainer.rl.train.train(config)` unchanged.

Every key the reference's trainer accepts on the dense data-parallel path parses here with the same name, type and default.  Keys fall in three
classes (the `Field(description=...)`s say which):
  * honoured: model.{name, seq_len, attn, impl, debug.*, fused_lm_head_token_chunk_size, optimization_dtype, reduce_dtype, dp_replicate},
    data.fake.*, loss.*, optim.* (adamw), scheduler.*, weight_broadcast.*, rollout_transport.*, output_dir, max_steps, matmul_precision,
    dist_timeout_seconds;
  * accepted and inert BY DESIGN on 288 GB parts: model.{ac, ac_offloading, compile, optim_cpu_offload,
    fsdp_cpu_offload, reshard_after_forward, full_offload}, gc — the memory-saving machinery the MI355X engine does not need;
  * rejected with NotImplementedError by `validate_supported()` because they select subsystems outside SURVEY §8
    (lora, quantization, vlm, the fused Triton MoE kernel, non-AdamW optimizers, token export, nixl broadcast).

The reference builds these classes on `pydantic_config.BaseConfig` (absent here); plain pydantic gives the same parsing rules.
"""

from __future__ import annotations

import sys
from pathlib import Path
from typing import Annotated, Any, ClassVar, Literal, Union

from pydantic import BaseModel, ConfigDict, Field, model_validator

AttnImplementation = Literal["flash_attention_2", "flash_attention_3", "flash_attention_4", "auto", "hip_varlen"]


class BaseConfig(BaseModel):
    """Unknown keys are errors (as in the reference); the string "None" means null (the reference's TOMLs write `compile = "None"`)."""

    model_config = ConfigDict(extra="forbid", validate_assignment=True, arbitrary_types_allowed=True)
    _FLAG_TABLES: ClassVar[frozenset] = frozenset({"compile", "ac", "ac_offloading", "gc"})

    @model_validator(mode="before")
    @classmethod
    def _none_strings(cls, data: Any) -> Any:
        if isinstance(data, dict):
            out = {}
            for k, v in data.items():
                if isinstance(v, str) and v == "None":
                    v = None
                elif v is True and k in cls._FLAG_TABLES:  # `--model.compile` / `--model.ac` as bare flags = "enable with defaults"
                    v = {}
                out[k] = v
            return out
        return data


# -- model ---------------------------------------------------------------------------------------------------------------
class GCConfig(BaseConfig):
    interval: int = Field(50, ge=1)


class ActivationCheckpointConfig(BaseConfig):
    mode: Literal["full", "selective"] = "full"
    freq: int = Field(1, ge=1)
    targets: list[str] = ["norm"]


class ActivationOffloadingConfig(BaseConfig):
    pin_memory: bool = True
    max_inflight_activations: int = Field(5, ge=1)


class CompileConfig(BaseConfig):
    fullgraph: bool = False


class DebugModelConfig(BaseConfig):
    num_layers: int | None = None
    random_init: bool = False
    force_balanced_routing: bool = False


class ModelConfig(BaseConfig):
    """configs/trainer.py:178-262 + BaseModelConfig (configs/shared.py:145-153)."""

    name: str = "Qwen/Qwen3-0.6B"
    trust_remote_code: bool = False
    vlm: dict | None = None
    conversion_dir: Path | None = None
    seq_len: int = 2048
    attn: AttnImplementation = "auto"
    compile: CompileConfig | None = CompileConfig()
    ac: ActivationCheckpointConfig | None = ActivationCheckpointConfig()
    ac_offloading: ActivationOffloadingConfig | None = ActivationOffloadingConfig()
    fsdp_cpu_offload: bool = False
    optim_cpu_offload: bool = True
    full_offload: dict | None = None
    reshard_after_forward: bool = True
    dp_replicate: int = 1
    ep: int | Literal["auto"] = "auto"
    ep_comm_backend: Literal["torch", "deepep"] = "torch"
    deepep_num_sms: int = Field(20, ge=1)
    deepep_token_chunk_size: int | None = Field(None, ge=1)
    cp: int = 1
    cp_style: Literal["ring", "ulysses"] = "ring"
    impl: Literal["hf", "custom", "auto"] = "auto"
    optimization_dtype: Literal["bfloat16", "float32"] = "float32"
    reduce_dtype: Literal["bfloat16", "float32"] = "float32"
    moe_router_dtype: Literal["bfloat16", "float32"] = "float32"
    moe_use_grouped_mm: bool = True
    moe_fused_kernel: bool = False
    quantization: dict | None = None
    index_cache: dict | None = None
    freeze_moe_router: bool = False
    lora: dict | None = None
    debug: DebugModelConfig = DebugModelConfig()
    fused_lm_head_token_chunk_size: int | Literal["disabled"] = 8192

    @model_validator(mode="after")
    def _cp_needs_custom_impl(self):  # configs/trainer.py:286-295
        if self.cp > 1 and self.impl not in ("custom", "auto"):
            raise ValueError("Context parallelism requires model.impl='custom' or 'auto' (resolved to a custom PrimeRL implementation)")
        return self


class TokenizerConfig(BaseConfig):
    name: str | None = None
    trust_remote_code: bool | None = None
    chat_template: str | None = None


# -- scheduler / optimizer -------------------------------------------------------------------------------------------------
class ConstantSchedulerConfig(BaseConfig):
    type: Literal["constant"] = "constant"


class LinearSchedulerConfig(BaseConfig):
    type: Literal["linear"] = "linear"
    warmup_steps: int = Field(10, ge=0)
    decay_steps: int = Field(10, ge=0)
    min_lr: float = Field(0.0, ge=0)


class CosineSchedulerConfig(BaseConfig):
    type: Literal["cosine"] = "cosine"
    warmup_steps: int = Field(10, ge=0)
    min_lr: float = Field(0.0, ge=0)


SchedulerConfig = Annotated[Union[ConstantSchedulerConfig, LinearSchedulerConfig, CosineSchedulerConfig], Field(discriminator="type")]


class BaseOptimizerConfig(BaseConfig):
    lr: float = Field(1e-6, ge=0)
    weight_decay: float = Field(0.01, ge=0)
    max_norm: float | None = Field(1.0, ge=0)


class AdamWConfig(BaseOptimizerConfig):
    type: Literal["adamw"] = "adamw"
    betas1: float = Field(0.9, ge=0)
    betas2: float = Field(0.999, ge=0)
    eps: float = 1e-8  # torch.optim.AdamW's default; the reference never overrides it (optim/__init__.py:101-108)


class SGDConfig(BaseOptimizerConfig):
    type: Literal["sgd"] = "sgd"
    nesterov: bool = True
    momentum: float = 0.9


class MuonConfig(BaseOptimizerConfig):
    type: Literal["muon"] = "muon"
    mu: float = Field(0.95, ge=0)
    betas1: float = 0.9
    betas2: float = 0.999


class SignSGDConfig(BaseOptimizerConfig):
    type: Literal["sign_sgd"] = "sign_sgd"


OptimizerConfig = Annotated[Union[AdamWConfig, SGDConfig, MuonConfig, SignSGDConfig], Field(discriminator="type")]


class CheckpointConfig(BaseConfig):
    output_dir: Path | None = None
    interval: int | None = Field(None, ge=1)
    keep_last: int | None = Field(None, ge=1)
    keep_interval: int | None = Field(None, ge=1)
    skip_progress: bool = False
    skip_scheduler: bool = False
    skip_dataloader: bool = False
    skip_optimizer: bool = False


# -- loss --------------------------------------------------------------------------------------------------------------------
class DefaultLossConfig(BaseConfig):
    type: Literal["default"] = "default"
    dppo_mask_low: float = Field(0.2, ge=0)
    dppo_mask_high: float = Field(0.2, ge=0)
    adv_tau: float = Field(1.0, ge=0)
    kl_tau: float = Field(1e-3, ge=0)


class IPOLossConfig(BaseConfig):
    type: Literal["ipo"] = "ipo"
    ipo_threshold: float = Field(0.1, ge=0)
    adv_tau: float = Field(1.0, ge=0)
    kl_tau: float = Field(1e-3, ge=0)


class CustomLossConfig(BaseConfig):
    type: Literal["custom"] = "custom"
    import_path: str
    kwargs: dict[str, Any] = Field(default_factory=dict)


LossConfig = Annotated[Union[DefaultLossConfig, IPOLossConfig, CustomLossConfig], Field(discriminator="type")]


# -- data / transports -------------------------------------------------------------------------------------------------------
class FakeDataLoaderConfig(BaseConfig):
    batch_size: int = Field(2, ge=1)
    generate_samples: bool = False


class DataLoaderConfig(BaseConfig):
    fake: FakeDataLoaderConfig | None = None


class FileSystemWeightBroadcastConfig(BaseConfig):
    type: Literal["filesystem"] = "filesystem"


class NCCLWeightBroadcastConfig(BaseConfig):
    type: Literal["nccl"] = "nccl"
    host: str = "localhost"
    port: int = 29501
    timeout: int = 1200
    inference_world_size: int = 1
    quantize_in_weight_transfer: bool = False


class NIXLWeightBroadcastConfig(BaseConfig):
    type: Literal["nixl"] = "nixl"
    host: str = "localhost"
    port: int = 8001
    timeout: int = 1200
    inference_world_size: int = 1
    session_id: str = "default"


WeightBroadcastConfig = Annotated[Union[FileSystemWeightBroadcastConfig, NCCLWeightBroadcastConfig, NIXLWeightBroadcastConfig],
                                  Field(discriminator="type")]


class FileSystemTransportConfig(BaseConfig):
    type: Literal["filesystem"] = "filesystem"
    format: Literal["msgpack", "binary"] = "msgpack"  # extension: "binary" = raw-array step blobs + pinned read-ahead


class ZMQTransportConfig(BaseConfig):
    type: Literal["zmq"] = "zmq"
    host: str = "localhost"
    port: int = 5555
    hwm: int = 10


TransportConfig = Annotated[Union[FileSystemTransportConfig, ZMQTransportConfig], Field(discriminator="type")]


class TrainerLogConfig(BaseConfig):
    level: str = "info"
    json_logging: bool = False
    file: bool = True


class TrainerConfig(BaseConfig):
    """configs/trainer.py:597-661."""

    model: ModelConfig = ModelConfig()
    tokenizer: TokenizerConfig = TokenizerConfig()
    data: DataLoaderConfig = DataLoaderConfig()
    loss: LossConfig = DefaultLossConfig()
    optim: OptimizerConfig = AdamWConfig()
    scheduler: SchedulerConfig = ConstantSchedulerConfig()
    ckpt: CheckpointConfig | None = None
    resume: dict | None = None
    weight_broadcast: WeightBroadcastConfig | None = FileSystemWeightBroadcastConfig()
    rollout_transport: TransportConfig = ZMQTransportConfig()
    log: TrainerLogConfig = TrainerLogConfig()
    monitors: dict = Field(default_factory=dict)
    output_dir: Path = Path("outputs")
    matmul_precision: Literal["highest", "high", "medium"] = "high"
    max_steps: int | None = None
    enable_router_replay: bool = False
    memory_profiler_path: Path | None = None
    gc: GCConfig | None = GCConfig()
    trace_path: Path | None = None
    dist_timeout_seconds: int = 3600
    heartbeat: dict | None = None
    metrics_server: dict | None = None
    enable_token_export: bool = False
    env_vars: dict[str, str] = Field(default_factory=dict)
    # MI355X engine knob (not in the reference): micro-batches are concatenated into packed passes of at most this many tokens
    tokens_per_pass: int = Field(16384, ge=1)
    seed: int = 0

    _DEFAULT_TAGS: ClassVar[dict] = {"optim": "adamw", "loss": "default", "scheduler": "constant", "weight_broadcast": "filesystem", "rollout_transport": "zmq"}

    @model_validator(mode="before")
    @classmethod
    def _default_union_tags(cls, data: Any) -> Any:
        """`--optim.lr 3e-6` alone keeps the default member of the tagged union (AdamW), as with the reference's config loader."""
        if isinstance(data, dict):
            data = dict(data)
            for key, tag in cls._DEFAULT_TAGS.items():
                if isinstance(data.get(key), dict) and "type" not in data[key]:
                    data[key] = dict(data[key], type=tag)
        return data

    @model_validator(mode="after")
    def _scheduler_fits_max_steps(self):  # validate_scheduler, configs/trainer.py:391-412
        s = self.scheduler
        if isinstance(s, LinearSchedulerConfig):
            if s.warmup_steps == 0 and s.decay_steps == 0:
                raise ValueError("Linear scheduler requires warmup_steps > 0 or decay_steps > 0 (use the constant scheduler instead)")
            if s.decay_steps > 0:
                if self.max_steps is None:
                    raise ValueError("Must specify max_steps when using a linear scheduler with decay_steps > 0")
                if s.warmup_steps + s.decay_steps > self.max_steps:
                    raise ValueError(f"warmup_steps ({s.warmup_steps}) + decay_steps ({s.decay_steps}) must not exceed max_steps ({self.max_steps})")
        if isinstance(s, CosineSchedulerConfig):
            if self.max_steps is None:
                raise ValueError("Must specify max_steps when using a cosine scheduler")
            if s.warmup_steps >= self.max_steps:
                raise ValueError(f"warmup_steps ({s.warmup_steps}) must be less than max_steps ({self.max_steps})")
        return self

    def validate_supported(self) -> None:
        """Raise for keys that select subsystems outside the GRPO policy-update path.  `model.ep > 1`, `freeze_moe_router`
        and `debug.force_balanced_routing` are served by the nn.Module trainer step (trainer/module_engine.py) for the MoE families."""
        m = self.model
        bad = []
        if m.lora is not None:
            bad.append("model.lora (LoRA adapters)")
        if m.quantization is not None:
            bad.append("model.quantization (FP8 / MXFP8 training)")
        if m.vlm is not None:
            bad.append("model.vlm (vision-language models)")
        if m.moe_fused_kernel:
            bad.append("model.moe_fused_kernel (the fused Triton MoE kernel; expert GEMMs here are the grouped MFMA kernels)")
        if self.optim.type not in ("adamw", "sgd", "sign_sgd"):
            bad.append(f"optim.type = {self.optim.type!r} (the fused on-GPU optimizers are adamw, sgd and sign_sgd)")
        if self.enable_token_export:
            bad.append("enable_token_export")
        if self.weight_broadcast is not None and self.weight_broadcast.type == "nixl":
            bad.append("weight_broadcast.type = 'nixl'")
        # keys the reference's train() acts on but this train loop does not build: refusing them by name beats silently writing no checkpoint
        if self.ckpt is not None:
            bad.append("ckpt (checkpoint saving / loading: control plane, stays with the reference)")
        if self.resume is not None:
            bad.append("resume")
        for key in ("heartbeat", "metrics_server", "trace_path", "memory_profiler_path"):
            if getattr(self, key) is not None:
                bad.append(f"{key} (monitoring / profiling hooks of the reference's train loop)")
        if bad:
            raise NotImplementedError("outside the MI355X GRPO hot path: " + "; ".join(bad))

    def validate_for_train_loop(self) -> None:
        """What only `train(config)` needs, checked BEFORE the engine is built: the default rollout transport of the reference (ZMQ) is networking and
        stays with the reference — a live data loader reads the filesystem transport here."""
        self.validate_supported()
        if self.rollout_transport.type == "zmq" and self.data.fake is None:
            raise NotImplementedError("outside the MI355X GRPO hot path: rollout_transport.type = 'zmq' with a live data loader (networking stays with the "
                                      "reference: use rollout_transport.type = 'filesystem', or data.fake for synthetic batches)")


# -- loading -------------------------------------------------------------------------------------------------------------------
def _set_dotted(tree: dict, dotted: str, value: Any) -> None:
    node = tree
    parts = dotted.split(".")
    for p in parts[:-1]:
        node = node.setdefault(p.replace("-", "_"), {})
    node[parts[-1].replace("-", "_")] = value


def _parse_scalar(text: str) -> Any:
    import tomli

    try:
        return tomli.loads(f"v = {text}")["v"]
    except tomli.TOMLDecodeError:
        return text  # bare strings: --model.name Qwen/Qwen3-8B


def load_toml(path: str | Path) -> dict:
    import tomli

    with open(path, "rb") as f:
        return tomli.load(f)


def _merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def parse_argv(argv: list[str] | None = None, cls=TrainerConfig):
    """The reference's launcher grammar (`uv run trainer @ cfg.toml --model.seq-len 4096 --data.fake.batch-size 8`): any number of
    `@ file.toml` (later files override earlier ones), then `--dotted.key value` overrides (`--flag` alone = true)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    tree: dict = {}
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "@":
            tree = _merge(tree, load_toml(argv[i + 1]))
            i += 2
        elif a.startswith("@"):
            tree = _merge(tree, load_toml(a[1:]))
            i += 1
        elif a.startswith("--"):
            key = a[2:]
            if "=" in key:
                key, val = key.split("=", 1)
                _set_dotted(tree, key, _parse_scalar(val))
                i += 1
            elif i + 1 < len(argv) and not argv[i + 1].startswith("--") and argv[i + 1] != "@":
                _set_dotted(tree, key, _parse_scalar(argv[i + 1]))
                i += 2
            else:
                _set_dotted(tree, key, True)
                i += 1
        else:
            raise SystemExit(f"unexpected argument {a!r} (expected `@ file.toml` or `--dotted.key value`)")
    return cls.model_validate(tree)
