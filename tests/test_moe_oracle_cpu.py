"""The MoE oracle against the reference's own MoE module (tests/golden/moe.npz, produced by tests/golden/make_moe_golden.py)."""

from __future__ import annotations

import ast
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import moe_oracle as M

GOLD = np.load(Path(__file__).parent / "golden" / "moe.npz", allow_pickle=False)
CASES = ["qwen3", "default", "forced", "sigmoid_norm"]


def load_case(name, device="cpu", dtype=torch.bfloat16):
    kw = ast.literal_eval(str(GOLD[f"{name}.args"]))
    args = M.MoEArgs(num_experts=kw["num_experts"], num_shared_experts=kw["num_shared_experts"], score_func=kw["score_func"],
                     route_norm=kw["route_norm"], route_scale=kw["route_scale"], score_before_experts=kw["score_before_experts"],
                     top_k=kw["top_k"], load_balance_coeff=kw["load_balance_coeff"])
    t = lambda k: torch.from_numpy(GOLD[f"{name}.{k}"]).to(device)
    params = {k[len(name) + 3:]: t(k[len(name) + 1:]).to(dtype) for k in GOLD.files if k.startswith(f"{name}.w.")}
    grads = {k[len(name) + 3:]: t(k[len(name) + 1:]) for k in GOLD.files if k.startswith(f"{name}.g.")}
    x = t("x").to(dtype).reshape(-1, kw["dim"])
    gy = t("gy").to(dtype).reshape(-1, kw["dim"])
    bias = t("expert_bias") if f"{name}.expert_bias" in GOLD.files else None
    routed = t("routed").reshape(-1, kw["top_k"]).long() if f"{name}.routed" in GOLD.files else None
    return args, params, grads, x, gy, bias, routed, kw


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_moe_module(name):
    args, params, grads, x, gy, bias, routed, kw = load_case(name)
    x = x.clone().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    y, aux = M.moe_forward(x, params, args, expert_bias=bias, routed_experts=routed)
    y.backward(gy)
    # routing decisions and gating values: exact (same ops on the same CPU)
    assert torch.equal(aux["top_idx"].int(), torch.from_numpy(GOLD[f"{name}.top_idx"]))
    assert torch.equal(aux["counts"].float(), torch.from_numpy(GOLD[f"{name}.counts"]))
    assert torch.equal(aux["top_scores"].detach().float(), torch.from_numpy(GOLD[f"{name}.top_scores"]))
    assert float(aux["confidence"]) == pytest.approx(float(GOLD[f"{name}.confidence"]), rel=1e-6)
    # layer output and gradients: the oracle runs the same bf16 ops, so it must be bit-identical to the reference module
    assert torch.equal(y.detach().float(), torch.from_numpy(GOLD[f"{name}.y"]).reshape(-1, kw["dim"]))
    assert torch.equal(x.grad.float(), torch.from_numpy(GOLD[f"{name}.gx"]).reshape(-1, kw["dim"]))
    for k, g in grads.items():
        assert torch.equal(params[k].grad.float(), g), k


def test_ep_dispatch_layout_bookkeeping():
    counts = [torch.tensor([2, 0, 1, 3]), torch.tensor([1, 1, 0, 2])]
    plan, totals = M.ep_dispatch_layout(counts, ep=2)
    assert totals == [[3, 1], [1, 5]]
    assert plan[0] == [(0, 0, 2), (1, 0, 1), (0, 2, 0), (1, 1, 1)]
    assert plan[1] == [(0, 2, 1), (1, 2, 0), (0, 3, 3), (1, 2, 2)]
