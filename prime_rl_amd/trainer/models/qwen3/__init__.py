from .modeling_qwen3 import Qwen3Config, Qwen3DecoderLayer, Qwen3ForCausalLM, Qwen3Model  # noqa: F401
