from .modeling_qwen3_moe import Qwen3MoeConfig, Qwen3MoeDecoderLayer, Qwen3MoeForCausalLM, Qwen3MoeModel

__all__ = ["Qwen3MoeConfig", "Qwen3MoeDecoderLayer", "Qwen3MoeForCausalLM", "Qwen3MoeModel"]
