"""CPU suite: the reference's trainer config surface (packages/prime-rl-configs/src/prime_rl/configs/trainer.py:597-661) drives
`prime_rl_amd` unchanged — its TOML tables, its `@ file.toml --dotted.key value` launch grammar (the exact argv of
benchmarks/scripts/run_single_benchmark.py:111-191), its defaults, its scheduler arithmetic and its ParallelDims rules."""

import math

import pytest

from prime_rl_amd.configs.trainer import TrainerConfig, parse_argv
from prime_rl_amd.trainer.parallel_dims import ParallelDims, get_parallel_dims


def test_defaults_are_the_references():
    c = TrainerConfig()
    assert (c.model.name, c.model.seq_len, c.model.attn, c.model.impl) == ("Qwen/Qwen3-0.6B", 2048, "auto", "auto")
    assert (c.optim.type, c.optim.lr, c.optim.weight_decay, c.optim.max_norm, c.optim.betas1, c.optim.betas2) == ("adamw", 1e-6, 0.01, 1.0, 0.9, 0.999)
    assert (c.loss.type, c.loss.dppo_mask_low, c.loss.dppo_mask_high, c.loss.adv_tau, c.loss.kl_tau) == ("default", 0.2, 0.2, 1.0, 1e-3)
    assert c.model.fused_lm_head_token_chunk_size == 8192 and c.model.optim_cpu_offload and c.model.reshard_after_forward
    assert c.weight_broadcast.type == "filesystem" and c.rollout_transport.type == "zmq" and c.rollout_transport.port == 5555
    assert c.scheduler.type == "constant" and c.max_steps is None and c.dist_timeout_seconds == 3600 and c.matmul_precision == "high"
    assert c.data.fake is None and c.model.optimization_dtype == "float32" and c.model.reduce_dtype == "float32"


def test_reference_benchmark_argv_parses():
    """The argv `run_single_benchmark.py` builds for an RL run (its flags verbatim, dashes included)."""
    argv = ["--model.name", "Qwen/Qwen3-8B", "--model.seq-len", "2048", "--model.attn", "flash_attention_2", "--max-steps", "4", "--output-dir", "/tmp/bench",
            "--monitors.file.path", "metrics.jsonl", "--model.compile", "--dist-timeout-seconds", "3600", "--model.ac", "--model.fused-lm-head-token-chunk-size",
            "4096", "--data.fake.batch-size", "16"]
    c = parse_argv(argv)
    assert c.model.name == "Qwen/Qwen3-8B" and c.model.seq_len == 2048 and c.max_steps == 4 and c.data.fake.batch_size == 16
    assert c.model.compile is not None and c.model.ac is not None and c.model.fused_lm_head_token_chunk_size == 4096
    assert c.monitors == {"file": {"path": "metrics.jsonl"}}
    c.validate_supported()


def test_toml_tables_and_overrides(tmp_path):
    f = tmp_path / "rl.toml"
    f.write_text('max_steps = 5\n\n[data.fake]\nbatch_size = 2\n\n[model]\nseq_len = 1024\ncompile = "None"\n\n[loss]\ntype = "ipo"\nipo_threshold = 0.05\n\n'
                 '[weight_broadcast]\ntype = "nccl"\nport = 29777\ninference_world_size = 4\n\n[scheduler]\ntype = "cosine"\nwarmup_steps = 2\n')
    c = parse_argv(["@", str(f), "--model.seq-len", "4096", "--optim.lr=3e-6"])
    assert c.max_steps == 5 and c.model.seq_len == 4096 and c.model.compile is None and c.loss.type == "ipo" and c.loss.ipo_threshold == 0.05
    assert c.weight_broadcast.type == "nccl" and c.weight_broadcast.port == 29777 and c.weight_broadcast.inference_world_size == 4
    assert c.optim.lr == 3e-6 and c.scheduler.type == "cosine"
    with pytest.raises(Exception):
        parse_argv(["--model.not-a-key", "1"])  # unknown keys are errors, as in the reference
    with pytest.raises(Exception):
        parse_argv(["--scheduler.type", "cosine"])  # cosine needs max_steps (validate_scheduler, configs/trainer.py:408-412)


def test_out_of_scope_keys_are_refused_loudly():
    for argv in (["--model.lora.rank", "8"], ["--model.moe-fused-kernel"], ["--optim.type", "muon"], ["--enable-token-export"]):
        with pytest.raises(NotImplementedError):
            parse_argv(argv).validate_supported()


@pytest.mark.parametrize("kind", ["constant", "linear", "cosine"])
def test_scheduler_equals_torch_schedulers_the_reference_builds(kind):
    """trainer/scheduler.py builds ConstantLR / LinearLR / CosineAnnealingLR / SequentialLR objects; `setup_scheduler` must hand the
    fused AdamW kernel the same lr at every step.  Spot values: warm-up starts at lr*1e-8-ish floor, peaks at lr, ends at min_lr."""
    from prime_rl_amd.trainer.rl.train import setup_scheduler

    lr, steps = 2e-5, 12
    sched = {"constant": {"type": "constant"}, "linear": {"type": "linear", "warmup_steps": 3, "decay_steps": 4, "min_lr": 2e-6},
             "cosine": {"type": "cosine", "warmup_steps": 3, "min_lr": 2e-6}}[kind]
    c = TrainerConfig.model_validate(dict(max_steps=steps, optim=dict(lr=lr), scheduler=sched))
    lr_at = setup_scheduler(c)
    got = [lr_at(s) for s in range(1, steps + 1)]
    if kind == "constant":
        assert got == [lr] * steps
    else:
        assert got[0] == pytest.approx(2e-6) and got[3] == pytest.approx(lr) and max(got) == pytest.approx(lr)
        assert got[-1] < got[3] and all(g >= 2e-6 - 1e-12 for g in got)
        if kind == "linear":
            assert got[3:8] == pytest.approx([lr] * 5)  # constant phase between warm-up and decay
        else:
            mid = 3 + (steps - 3) // 2
            assert got[mid] == pytest.approx(2e-6 + (lr - 2e-6) * (1 + math.cos(math.pi * (mid - 3) / (steps - 3))) / 2, rel=1e-6)


def test_parallel_dims_rules():
    d = ParallelDims(dp_replicate=1, dp_shard=-1, cp=2, pp=1, ep=1, world_size=8)
    assert (d.dp_shard, d.dp_size, d.fsdp_gradient_divide_factor, d.non_data_parallel_size, d.seq_len_divisor) == (4, 4, 8, 2, 4)
    assert d.cp_enabled and d.dp_enabled and not d.ep_enabled and not d.pp_enabled
    assert [d.dp_rank_of(r) for r in range(8)] == [0, 0, 1, 1, 2, 2, 3, 3]  # dp_rank = rank // non_dp_size (data.py:174-175)
    with pytest.raises(ValueError):
        ParallelDims(dp_replicate=3, dp_shard=-1, cp=1, pp=1, ep=1, world_size=8)
    c = TrainerConfig.model_validate(dict(model=dict(cp=2, seq_len=2050)))
    with pytest.raises(ValueError, match="seq_len_divisor"):
        get_parallel_dims(c.model, c.model.seq_len, world_size=8)
    assert get_parallel_dims(TrainerConfig().model, 2048, world_size=1).dp_size == 1


def test_model_name_resolution(tmp_path):
    import json

    from prime_rl_amd.trainer.model import resolve_model_dims

    c = TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen3-8B", debug=dict(num_layers=2))))
    d = resolve_model_dims(c.model)
    assert (d.hidden_size, d.num_hidden_layers, d.vocab_size, d.num_attention_heads, d.num_key_value_heads) == (4096, 2, 151936, 32, 8)
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="qwen3", vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                                                          num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-6, rope_theta=1e6,
                                                          tie_word_embeddings=True)))
    d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)
    assert (d.vocab_size, d.num_hidden_layers, d.tie_word_embeddings) == (1000, 3, True)
    with pytest.raises(ValueError):
        resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="meta-llama/Nope"))).model)
    # the three dense families = two attention switches (AttentionConfig.use_qk_norm / .attention_bias, models/layers/attn.py:36-37)
    d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen2.5-1.5B-Instruct"))).model)  # BASELINE.json configs[1]
    assert (d.hidden_size, d.num_attention_heads, d.num_key_value_heads, d.head_dim, d.qk_norm, d.attention_bias, d.tie_word_embeddings) == (1536, 12, 2, 128, False, True, True)
    assert d.num_params() == 1_543_714_304  # the published parameter count of Qwen2.5-1.5B
    d3 = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen3-8B"))).model)
    assert d3.qk_norm and not d3.attention_bias and d3.num_params() == 8_190_735_360
    base = dict(vocab_size=1000, hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2)
    for family, want in (("llama", (False, False)), ("qwen2", (False, True)), ("qwen3", (True, False))):
        (tmp_path / "config.json").write_text(json.dumps(dict(model_type=family, **base)))
        d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)
        assert (d.qk_norm, d.attention_bias) == want and d.head_dim == 128
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="llama", rope_scaling=dict(rope_type="yarn", factor=8.0), **base)))
    with pytest.raises(NotImplementedError, match="rope_scaling"):
        resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)
    l3 = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="llama", rope_scaling=l3, rope_theta=500000.0, max_position_embeddings=131072, **base)))
    d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)
    assert d.rope_scaling == l3 and d.rope_theta == 500000.0 and not d.qk_norm
    d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="meta-llama/Llama-3.1-8B-Instruct"))).model)
    assert d.rope_scaling["rope_type"] == "llama3" and d.num_params() == 8_030_261_248  # the published size of Llama-3.1-8B
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="llama", mlp_bias=True, **base)))
    with pytest.raises(NotImplementedError, match="mlp_bias"):
        resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="mistral", **base)))
    with pytest.raises(NotImplementedError, match="model_type"):
        resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name=str(tmp_path)))).model)


def test_rope_frequency_scaling_equals_transformers():
    """utils/rope.py against transformers' ROPE_INIT_FUNCTIONS (what the reference's RotaryEmbedding calls, rotary_emb.py:41-57): "llama3" with the
    Llama-3.1 and 3.2 parameters and "linear", bit for bit; the default table; unsupported types refused."""
    import torch
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS

    from prime_rl_amd.utils.rope import scaled_inv_freq

    for factor in (8.0, 32.0):
        rs = dict(rope_type="llama3", factor=factor, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
        cfg = LlamaConfig(hidden_size=512, num_attention_heads=4, rope_theta=500000.0, rope_scaling=rs, max_position_embeddings=131072)
        want, scale = ROPE_INIT_FUNCTIONS["llama3"](cfg, "cpu")
        got = scaled_inv_freq(128, 500000.0, rs)
        assert scale == 1.0 and torch.equal(got, want) and not torch.equal(got, scaled_inv_freq(128, 500000.0))
    rs = dict(rope_type="linear", factor=4.0)
    want, scale = ROPE_INIT_FUNCTIONS["linear"](LlamaConfig(hidden_size=512, num_attention_heads=4, rope_theta=10000.0, rope_scaling=rs), "cpu")
    assert scale == 1.0 and torch.equal(scaled_inv_freq(128, 10000.0, rs), want)
    assert torch.equal(scaled_inv_freq(128, 1e6), 1.0 / (1e6 ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128)))
    with pytest.raises(NotImplementedError):
        scaled_inv_freq(128, 1e6, dict(rope_type="yarn", factor=4.0))


def test_64_wide_heads_are_stored_zero_padded_and_exposed_unpadded():
    """Qwen2.5-0.5B (BASELINE.json configs[0]) and Llama-3.2-1B have 64-wide heads; the kernels are built for 128.  ParamStore keeps every head
    zero-padded — rotary half 0 at [0, 32), half 1 at [64, 96) of the padded head — and speaks HF shapes at its boundary: load_state_dict pads,
    state_dict / grad_state_dict de-pad, the parameter and flop counts are the model's own."""
    import torch

    from prime_rl_amd.trainer.engine import ModelDims, ParamStore
    from prime_rl_amd.trainer.model import resolve_model_dims

    d = resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="Qwen/Qwen2.5-0.5B-Instruct"))).model)
    assert (d.head_dim, d.kernel_head_dim, d.head_padded, d.q_dim, d.kv_dim) == (64, 128, True, 14 * 128, 2 * 128)
    assert d.num_params() == 494_032_768  # the published size of Qwen2.5-0.5B: padding is storage, not parameters
    assert resolve_model_dims(TrainerConfig.model_validate(dict(model=dict(name="meta-llama/Llama-3.2-1B"))).model).num_params() == 1_235_814_400
    assert not ModelDims.qwen3_8b().head_padded
    dims = ModelDims(97, 128, 256, 2, 2, 1, 64, tie_word_embeddings=False, qk_norm=False, attention_bias=True, name="pad64")
    st = ParamStore(dims, "cpu")
    st.random_init(seed=3)
    sd = {k: v.clone() for k, v in st.state_dict().items()}
    assert sd["model.layers.1.self_attn.q_proj.weight"].shape == (128, 128) and sd["model.layers.1.self_attn.k_proj.bias"].shape == (64,)
    assert sd["model.layers.0.self_attn.o_proj.weight"].shape == (128, 128) and all(float(v.abs().sum()) > 0 for v in sd.values())
    qkv = st.w["model.layers.0.self_attn.qkv_proj.weight"].view(4, 128, 128)  # (2 q + 1 k + 1 v) heads x padded width x hidden
    assert float(qkv[:, 32:64].abs().sum()) == 0 and float(qkv[:, 96:].abs().sum()) == 0 and float(qkv[:, :32].abs().sum()) > 0
    assert torch.equal(qkv[0, 64:96], sd["model.layers.0.self_attn.q_proj.weight"][32:64])  # second rotary half of q head 0
    o = st.w["model.layers.0.self_attn.o_proj.weight"].view(128, 2, 128)
    assert float(o[:, :, 32:64].abs().sum()) == 0 and float(o[:, :, 96:].abs().sum()) == 0
    st2 = ParamStore(dims, "cpu")
    st2.load_state_dict(sd)
    assert torch.equal(st2.params, st.params) and torch.equal(st2.master, st.master)
    g = st.grad_state_dict()
    assert set(g) == set(sd) and all(g[k].shape == sd[k].shape and g[k].dtype == torch.float32 for k in sd)


def test_ep_auto_resolves_like_the_reference():
    """model.ep = "auto" (the default): MoE models -> min(world // dp_replicate, 8), dense models -> 1 (parallel_dims.py:281-317)."""
    from prime_rl_amd.configs.trainer import ModelConfig
    from prime_rl_amd.trainer.parallel_dims import get_parallel_dims, resolve_ep

    dense = ModelConfig(name="Qwen/Qwen3-8B")
    moe = ModelConfig(name="Qwen/Qwen3-30B-A3B")
    assert dense.ep == "auto" and moe.ep == "auto"
    assert resolve_ep(dense, 8) == 1
    assert resolve_ep(moe, 8) == 8 and resolve_ep(moe, 4) == 4 and resolve_ep(moe, 1) == 1 and resolve_ep(moe, 16) == 8
    assert resolve_ep(ModelConfig(name="Qwen/Qwen3-30B-A3B", dp_replicate=2), 8) == 4
    assert resolve_ep(ModelConfig(name="Qwen/Qwen3-30B-A3B", ep=2), 8) == 2
    assert get_parallel_dims(moe, world_size=8).ep == 8 and get_parallel_dims(dense, world_size=8).ep == 1
    # the two remaining branches of the reference rule (parallel_dims.py:291-295, 311-316)
    assert resolve_ep(ModelConfig(name="Qwen/Qwen3-30B-A3B", impl="hf"), 8) == 1  # expert parallelism needs the custom implementation
    import pytest

    with pytest.raises(ValueError, match="requires ep > 1"):
        resolve_ep(ModelConfig(name="Qwen/Qwen3-30B-A3B", ep_comm_backend="deepep"), 1)
    assert resolve_ep(ModelConfig(name="Qwen/Qwen3-30B-A3B", ep_comm_backend="deepep"), 8) == 8  # (rejected later by validate_supported, not by the resolution)


def test_model_ac_maps_onto_the_engine_recompute_arguments(monkeypatch):
    """`model.ac` (ActivationCheckpointConfig; applied by the reference in trainer/model.py:1141-1185): what the user WROTE is honoured — mode full -> block-level
    recomputation, selective -> everything but the attention core, `freq` -> every freq-th layer only —, the schema default (checkpoint everything: sized for
    80 GB parts) is not applied on a 288 GB part; PM_RECOMPUTE / PM_RECOMPUTE_FREQ override either."""
    from prime_rl_amd.trainer.model import recompute_from_config

    monkeypatch.delenv("PM_RECOMPUTE", raising=False)
    monkeypatch.delenv("PM_RECOMPUTE_FREQ", raising=False)
    assert recompute_from_config(TrainerConfig().model) == {"recompute": "none", "recompute_freq": 1}
    assert recompute_from_config(parse_argv(["--model.ac"]).model) == {"recompute": "layer", "recompute_freq": 1}  # the reference benchmark's bare flag
    assert recompute_from_config(parse_argv(["--model.ac.mode", "selective", "--model.ac.freq", "2"]).model) == {"recompute": "selective", "recompute_freq": 2}
    monkeypatch.setenv("PM_RECOMPUTE", "none")
    assert recompute_from_config(parse_argv(["--model.ac"]).model)["recompute"] == "none"
