from .modeling_llama import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM

__all__ = ["LlamaConfig", "LlamaForCausalLM", "Qwen2Config", "Qwen2ForCausalLM"]
