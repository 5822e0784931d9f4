"""HF <-> trainer state-dict names for Qwen3-MoE — what `prime_rl.trainer.models.qwen3_moe.converting_qwen3_moe.conversion_chain`
(src/prime_rl/trainer/models/qwen3_moe/converting_qwen3_moe.py:8-14, built from conversion_ops.routed_experts_op :324-364) does:

  HF checkpoint / vLLM                                   trainer module tree
  model.layers.{i}.mlp.gate.weight            [E, H]  <-> model.layers.{i}.mlp.router.gate.weight
  model.layers.{i}.mlp.experts.{e}.gate_proj.weight   <-> model.layers.{i}.mlp.experts.w1[e]   [E, I, H]
  model.layers.{i}.mlp.experts.{e}.down_proj.weight   <-> model.layers.{i}.mlp.experts.w2[e]   [E, H, I]
  model.layers.{i}.mlp.experts.{e}.up_proj.weight     <-> model.layers.{i}.mlp.experts.w3[e]   [E, I, H]
  (transformers v5 fused form: mlp.experts.gate_up_proj [E, 2I, H] = w1 | w3 along dim 1, mlp.experts.down_proj = w2; accepted on input)

`experts` selects a contiguous slice of experts (an expert-parallel rank's share) when reading a full HF checkpoint."""

from __future__ import annotations

import re

import torch

_PROJ = (("w1", "gate_proj"), ("w2", "down_proj"), ("w3", "up_proj"))
_EXPERT_KEY = re.compile(r"^(model\.layers\.\d+)\.mlp\.experts\.(\d+)\.(gate_proj|down_proj|up_proj)\.weight$")


def hf_to_prime(sd: dict[str, torch.Tensor], num_experts: int, experts: range | None = None) -> dict[str, torch.Tensor]:
    experts = experts if experts is not None else range(num_experts)
    out: dict[str, torch.Tensor] = {}
    per_layer: dict[str, dict[str, dict[int, torch.Tensor]]] = {}
    for key, t in sd.items():
        m = _EXPERT_KEY.match(key)
        if m:
            per_layer.setdefault(m.group(1), {}).setdefault(m.group(3), {})[int(m.group(2))] = t
        elif key.endswith(".mlp.gate.weight"):
            out[key[: -len("gate.weight")] + "router.gate.weight"] = t
        elif key.endswith(".mlp.experts.gate_up_proj"):
            base = key[: -len("gate_up_proj")]
            half = t.shape[1] // 2
            out[base + "w1"], out[base + "w3"] = t[experts.start:experts.stop, :half].contiguous(), t[experts.start:experts.stop, half:].contiguous()
        elif key.endswith(".mlp.experts.down_proj"):
            out[key[: -len("down_proj")] + "w2"] = t[experts.start:experts.stop].contiguous()
        else:
            out[key] = t
    for prefix, projs in per_layer.items():
        for wn, hf_proj in _PROJ:
            have = projs.get(hf_proj, {})
            missing = [e for e in experts if e not in have]
            if missing:
                raise KeyError(f"{prefix}.mlp.experts.{missing[0]}.{hf_proj}.weight is missing from the checkpoint")
            out[f"{prefix}.mlp.experts.{wn}"] = torch.stack([have[e] for e in experts], 0)
    return out


def prime_to_hf(sd: dict[str, torch.Tensor], first_expert: int = 0) -> dict[str, torch.Tensor]:
    """Per-expert HF names (views, no copies).  `first_expert`: global index of row 0 of the stacked tensors (a local expert shard)."""
    out: dict[str, torch.Tensor] = {}
    for key, t in sd.items():
        if key.endswith(".mlp.router.gate.weight"):
            out[key[: -len("router.gate.weight")] + "gate.weight"] = t
            continue
        hit = next(((wn, hf_proj) for wn, hf_proj in _PROJ if key.endswith(f".mlp.experts.{wn}")), None)
        if hit is None:
            out[key] = t
            continue
        base = key[: -len(hit[0])]
        for e in range(t.shape[0]):
            out[f"{base}{first_expert + e}.{hit[1]}.weight"] = t[e]
    return out
