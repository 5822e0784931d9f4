"""Qwen3-MoE (config 4 of BASELINE.json: Qwen3-30B-A3B) on the HIP layer seams — mirrors
`prime_rl.trainer.models.qwen3_moe.modeling_qwen3_moe` (src/prime_rl/trainer/models/qwen3_moe/modeling_qwen3_moe.py:43-323) and the fields
of `configuration_qwen3_moe.py:150-205` this path reads: the dense Qwen3 decoder layer with the MLP replaced by `MoE` on every layer that
is not in `mlp_only_layers` and whose `(layer_idx + 1) % decoder_sparse_step == 0` (:79-86), MoE arguments as the reference builds them
(:61-72: softmax scores, `route_norm = norm_topk_prob`, no shared expert, scores applied after the experts), the packed-batch forward
contract of the dense model plus `routed_experts [1, N, num_hidden_layers, top_k]` (router replay, :70-99).  Parameter names are the
reference's (`model.layers.{i}.mlp.{router.gate.weight, experts.{w1,w2,w3}}`), i.e. what `converting_qwen3_moe.py` maps HF checkpoints to.

`ep_group`: the experts of every MoE layer are sharded over that process group (`E / ep` experts per rank) and tokens are dispatched
with all-to-all (`prime_rl_amd.trainer.distributed.expert_parallel`); attention, norms, router and LM head are replicated."""

from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace

import torch
import torch.distributed as dist
from torch import nn

from ..layers.attn import ATTN_IMPL2CLASS, AttentionConfig
from ..layers.linear import add_rms_norm, final_norm
from ..layers.lm_head import PrimeLmOutput
from ..layers.mlp import MLP, MLPConfig
from ..layers.moe import MoE, MoEArgs
from ..layers.norms import RMSNorm, RMSNormConfig
from ..layers.rotary_emb import RotaryEmbedding
from ..qwen3.modeling_qwen3 import _EmbeddingFn


@dataclass
class Qwen3MoeConfig:
    vocab_size: int = 151936
    hidden_size: int = 2048
    intermediate_size: int = 6144
    num_hidden_layers: int = 48
    num_attention_heads: int = 32
    num_key_value_heads: int = 4
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    hidden_act: str = "silu"
    attention_bias: bool = False
    tie_word_embeddings: bool = False
    decoder_sparse_step: int = 1
    moe_intermediate_size: int = 768
    num_experts_per_tok: int = 8
    num_experts: int = 128
    norm_topk_prob: bool = True  # Qwen3-30B-A3B's config.json (the class default in configuration_qwen3_moe.py:166 is False)
    mlp_only_layers: list[int] = field(default_factory=list)
    load_balance_coeff: float | None = None
    use_grouped_mm: bool = True
    _attn_implementation: str = "hip_varlen"


class Qwen3MoeDecoderLayer(nn.Module):
    def __init__(self, config: Qwen3MoeConfig, layer_idx: int, ep_group: dist.ProcessGroup | None = None):
        super().__init__()
        self.hidden_size = config.hidden_size
        attn_config = AttentionConfig(hidden_size=config.hidden_size, head_dim=config.head_dim, num_attention_heads=config.num_attention_heads,
                                      num_key_value_heads=config.num_key_value_heads, is_causal=True, attention_bias=config.attention_bias,
                                      output_bias=config.attention_bias, use_qk_norm=True, rms_norm_eps=config.rms_norm_eps)
        self.self_attn = ATTN_IMPL2CLASS[config._attn_implementation](attn_config)
        moe_args = MoEArgs(num_experts=config.num_experts, num_shared_experts=0, score_func="softmax", route_norm=config.norm_topk_prob, route_scale=1.0,
                           score_before_experts=False, top_k=config.num_experts_per_tok, use_grouped_mm=config.use_grouped_mm,
                           load_balance_coeff=config.load_balance_coeff)
        if layer_idx not in config.mlp_only_layers and config.num_experts > 0 and (layer_idx + 1) % config.decoder_sparse_step == 0:
            self.mlp = MoE(moe_args, dim=config.hidden_size, hidden_dim=config.moe_intermediate_size, ep_group=ep_group)
        else:
            self.mlp = MLP(MLPConfig(hidden_size=config.hidden_size, intermediate_size=config.intermediate_size, gate_act=config.hidden_act, bias=False))
        self.input_layernorm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))
        self.post_attention_layernorm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))

    def forward(self, hidden_states, position_embeddings, cu_seqlens=None, max_seqlen=None, routed_experts=None, pending=None, return_pending: bool = False):
        """As Qwen3DecoderLayer.forward (models/qwen3/modeling_qwen3.py): both residual adds ride in the RMSNorm that follows them; `pending` / `return_pending`
        hand the not-yet-added MLP output from block to block."""
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
        mlp_out = self.mlp(h2.view(shape), routed_experts=routed_experts)
        hidden_states = hidden_states.view(shape)
        return (hidden_states, mlp_out) if return_pending else hidden_states + mlp_out


class Qwen3MoeModel(nn.Module):
    def __init__(self, config: Qwen3MoeConfig, ep_group: dist.ProcessGroup | None = None):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3MoeDecoderLayer(config, i, ep_group) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(RMSNormConfig(hidden_size=config.hidden_size, eps=config.rms_norm_eps))
        self.rotary_emb = RotaryEmbedding(config.head_dim, config.rope_theta)

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, routed_experts=None, *, seq_lens, seq_lens_are_pre_shard: bool = False):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = _EmbeddingFn.apply(self.embed_tokens.weight, input_ids.reshape(-1).contiguous()).unsqueeze(0)
        n = inputs_embeds.shape[1]
        if position_ids is None:
            position_ids = torch.arange(n, device=inputs_embeds.device).unsqueeze(0)
        if seq_lens.ndim != 1 or seq_lens.numel() == 0:
            raise ValueError(f"seq_lens must be a non-empty 1D tensor, got shape={tuple(seq_lens.shape)}")
        cu_seqlens = torch.zeros(seq_lens.numel() + 1, dtype=torch.int32, device=inputs_embeds.device)
        cu_seqlens[1:] = seq_lens.to(inputs_embeds.device).cumsum(0)
        position_embeddings = self.rotary_emb(inputs_embeds, position_ids)
        hidden_states, pending = inputs_embeds, None
        for i, layer in enumerate(self.layers):
            layer_routing = routed_experts[:, :, i, :] if routed_experts is not None else None  # modeling_qwen3_moe.py:93 of the reference file
            hidden_states, pending = layer(hidden_states, position_embeddings=position_embeddings, cu_seqlens=cu_seqlens, max_seqlen=None,
                                           routed_experts=layer_routing, pending=pending, return_pending=True)
        return SimpleNamespace(last_hidden_state=final_norm(self.norm, hidden_states, pending))


class Qwen3MoeForCausalLM(nn.Module):
    def __init__(self, config: Qwen3MoeConfig, ep_group: dist.ProcessGroup | None = None):
        super().__init__()
        self.config = config
        self.model = Qwen3MoeModel(config, ep_group)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def forward(self, input_ids=None, position_ids=None, inputs_embeds=None, labels=None, logits_to_keep: int = 0, temperature=None,
                routed_experts=None, *, seq_lens, seq_lens_are_pre_shard: bool = False, **kwargs) -> PrimeLmOutput:
        hs = self.model(input_ids=input_ids, position_ids=position_ids, inputs_embeds=inputs_embeds, routed_experts=routed_experts,
                        seq_lens=seq_lens).last_hidden_state
        sl = slice(-logits_to_keep, None) if isinstance(logits_to_keep, int) and logits_to_keep > 0 else slice(None)
        out = self.lm_head(hs[:, sl, :])
        return out if isinstance(out, dict) else PrimeLmOutput(logits=out)

    def moe_layers(self):
        return [layer.mlp for layer in self.model.layers if isinstance(layer.mlp, MoE)]
