"""Dense Qwen3 on the HIP layer seams — mirrors `prime_rl.trainer.models.qwen3.modeling_qwen3`
(src/prime_rl/trainer/models/qwen3/modeling_qwen3.py:38-269): same module tree and parameter names
(`model.embed_tokens`, `model.layers.{i}.{input_layernorm, self_attn.{q,k,v,o}_proj, self_attn.{q,k}_norm,
post_attention_layernorm, mlp.{gate,up,down}_proj}`, `model.norm`, `lm_head`), same packed-batch forward contract
(`input_ids [1,N]`, `position_ids [1,N]`, keyword `seq_lens`, optional `labels` / `temperature` once
`inject_prime_lm_head` replaced the head).  This is the torch.autograd flavour used behind the reference's own train loop;
`prime_rl_amd.trainer.engine.GRPOEngine` is the flat-buffer flavour of the same arithmetic used by the benchmark."""

from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
from torch import nn

from .... import ops
from ..layers.attn import ATTN_IMPL2CLASS, AttentionConfig
from ..layers.linear import add_rms_norm, final_norm
from ..layers.lm_head import PrimeLmOutput
from ..layers.mlp import MLP, MLPConfig
from ..layers.norms import RMSNorm, RMSNormConfig
from ..layers.rotary_emb import RotaryEmbedding


@dataclass
class Qwen3Config:
    vocab_size: int = 151936
    hidden_size: int = 4096
    intermediate_size: int = 12288
    num_hidden_layers: int = 36
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    hidden_act: str = "silu"
    attention_bias: bool = False
    tie_word_embeddings: bool = False
    _attn_implementation: str = "hip_varlen"
    rope_scaling: dict | None = None  # "linear" / "llama3" scaling of the RoPE frequencies (utils/rope.py); None = default
    use_qk_norm: bool = True  # False: the Llama decoder block (models/llama/modeling_llama.py:52-62) / HF Qwen2 with attention_bias=True


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids):
        ctx.save_for_backward(ids)
        ctx.shape = table.shape
        return ops.embedding_fwd(table, ids)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        gt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        ops.embedding_bwd(g.contiguous(), ids, gt)
        return gt.to(g.dtype), None


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, config: Qwen3Config, layer_idx: int):
        super().__init__()
        self.hidden_size = config.hidden_size
        attn_config = AttentionConfig(hidden_size=config.hidden_size, head_dim=config.head_dim, num_attention_heads=config.num_attention_heads,
                                      num_key_value_heads=config.num_key_value_heads, is_causal=True, attention_bias=config.attention_bias,
                                      output_bias=False, use_qk_norm=config.use_qk_norm, rms_norm_eps=config.rms_norm_eps)
        self.self_attn = ATTN_IMPL2CLASS[config._attn_implementation](attn_config)
        self.mlp = MLP(MLPConfig(hidden_size=config.hidden_size, intermediate_size=config.intermediate_size, gate_act=config.hidden_act, bias=False))
        self.input_layernorm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))
        self.post_attention_layernorm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))

    def forward(self, hidden_states, position_embeddings, cu_seqlens=None, max_seqlen=None, pending=None, return_pending: bool = False):
        """The reference's block (modeling_qwen3.py:65-85).  Both residual adds ride in the RMSNorm that follows them (layers/linear.py add_rms_norm):
        `pending` = the previous block's MLP output, not yet added to `hidden_states`; with `return_pending` this block returns (residual stream after
        attention, its own MLP output) and leaves the second add to the next block's / the model's final norm.  Called the reference's way
        (`pending=None`, `return_pending=False`) it returns the summed hidden states."""
        if pending is None:
            h1 = self.input_layernorm(hidden_states)
        else:
            h1, hidden_states = add_rms_norm(pending.reshape(-1, pending.shape[-1]), hidden_states.reshape(-1, hidden_states.shape[-1]),
                                             self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
            h1, hidden_states = h1.view(pending.shape), hidden_states.view(pending.shape)
        attn_out, _ = self.self_attn(h1, position_embeddings=position_embeddings, cu_seqlens=cu_seqlens, max_seqlen=max_seqlen)
        shape = hidden_states.shape
        h2, hidden_states = add_rms_norm(attn_out.reshape(-1, shape[-1]), hidden_states.reshape(-1, shape[-1]), self.post_attention_layernorm.weight,
                                         self.post_attention_layernorm.variance_epsilon)
        mlp_out = self.mlp(h2.view(shape))
        hidden_states = hidden_states.view(shape)
        return (hidden_states, mlp_out) if return_pending else hidden_states + mlp_out


class Qwen3Model(nn.Module):
    def __init__(self, config: Qwen3Config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))
        self.rotary_emb = RotaryEmbedding(config.head_dim, config.rope_theta, rope_scaling=getattr(config, "rope_scaling", None))

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, *, seq_lens, seq_lens_are_pre_shard: bool = False):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = _EmbeddingFn.apply(self.embed_tokens.weight, input_ids.reshape(-1).contiguous()).unsqueeze(0)
        n = inputs_embeds.shape[1]
        if position_ids is None:
            position_ids = torch.arange(n, device=inputs_embeds.device).unsqueeze(0)
        if seq_lens.ndim != 1 or seq_lens.numel() == 0:
            raise ValueError(f"seq_lens must be a non-empty 1D tensor, got shape={tuple(seq_lens.shape)}")  # utils/sequence.py:38-43
        # cu_seqlens without the reference's three host syncs (utils/sequence.py:44-52): stays on the device
        cu_seqlens = torch.zeros(seq_lens.numel() + 1, dtype=torch.int32, device=inputs_embeds.device)
        cu_seqlens[1:] = seq_lens.to(inputs_embeds.device).cumsum(0)
        position_embeddings = self.rotary_emb(inputs_embeds, position_ids)
        hidden_states, pending = inputs_embeds, None
        for layer in self.layers:
            hidden_states, pending = layer(hidden_states, position_embeddings=position_embeddings, cu_seqlens=cu_seqlens, max_seqlen=None, pending=pending,
                                           return_pending=True)
        return SimpleNamespace(last_hidden_state=final_norm(self.norm, hidden_states, pending))


class Qwen3ForCausalLM(nn.Module):
    def __init__(self, config: Qwen3Config):
        super().__init__()
        self.config = config
        self.model = Qwen3Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, labels=None, logits_to_keep: int = 0, temperature=None, *, seq_lens,
                seq_lens_are_pre_shard: bool = False, **kwargs) -> PrimeLmOutput:
        hs = self.model(input_ids=input_ids, position_ids=position_ids, inputs_embeds=inputs_embeds, seq_lens=seq_lens).last_hidden_state
        sl = slice(-logits_to_keep, None) if isinstance(logits_to_keep, int) and logits_to_keep > 0 else slice(None)
        out = self.lm_head(hs[:, sl, :])
        return out if isinstance(out, dict) else PrimeLmOutput(logits=out)
