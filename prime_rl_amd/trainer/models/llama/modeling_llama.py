"""Llama and Qwen2 on the HIP layer seams — mirrors `prime_rl.trainer.models.llama.modeling_llama`
(src/prime_rl/trainer/models/llama/modeling_llama.py:52-100,131-260).

The reference's custom Llama class is its Qwen3 class with `use_qk_norm=False` in the AttentionConfig (:59-60) and
`head_dim = hidden_size // num_attention_heads` (:56); parameter names, packed-batch forward contract and the injected LM head are the
same.  Qwen2 / Qwen2.5 (BASELINE.json configs[0-1]) has no custom class in the reference — `Qwen2Config` is not registered, so it
loads HF's `Qwen2ForCausalLM` (models/__init__.py:41-59, model.py:723-764) — whose arithmetic is the Llama block plus a bias on the q/k/v
projections (oracle pinned to HF Qwen2 and HF Llama: tests/test_oracle_golden.py).  Both are therefore the dense decoder of
`models/qwen3/modeling_qwen3.py` with two switches.  head_dim = 128 models run on the MFMA attention kernels as they are (Llama-2-7B+, Llama-3.1-8B /
70B, Llama-3.2-3B, Qwen2.5-1.5B / 3B / 7B / 14B / 32B / 72B); the 64-wide heads of Qwen2.5-0.5B (BASELINE configs[0]) and Llama-3.2-1B run through the
flat-buffer engine zero-padded to 128 (exact: trainer/engine.py `ModelDims.kernel_head_dim`, `ParamStore._pad_heads`; 2x attention / projection work).
Llama-3.x checkpoints scale their RoPE frequencies (`rope_scaling.rope_type = "llama3"`): `config.rope_scaling` -> utils/rope.py."""

from __future__ import annotations

from dataclasses import dataclass

from ..qwen3.modeling_qwen3 import Qwen3Config, Qwen3ForCausalLM


@dataclass
class LlamaConfig(Qwen3Config):
    rope_theta: float = 500_000.0
    rms_norm_eps: float = 1e-5
    use_qk_norm: bool = False
    attention_bias: bool = False


@dataclass
class Qwen2Config(Qwen3Config):
    use_qk_norm: bool = False
    attention_bias: bool = True
    tie_word_embeddings: bool = True


class LlamaForCausalLM(Qwen3ForCausalLM):
    def __init__(self, config: LlamaConfig):
        if config.use_qk_norm:
            raise ValueError("Llama has no per-head q/k norm (models/llama/modeling_llama.py:60)")
        super().__init__(config)


class Qwen2ForCausalLM(Qwen3ForCausalLM):
    def __init__(self, config: Qwen2Config):
        if config.use_qk_norm or not config.attention_bias:
            raise ValueError("Qwen2 = q/k/v bias, no per-head q/k norm")
        super().__init__(config)
