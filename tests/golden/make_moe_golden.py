"""Generate tests/golden/moe.npz from the REFERENCE's own MoE layer (build container only; /root/reference is absent on the GPU box):

    python tests/golden/make_moe_golden.py

File-loads (never copies) /root/reference/src/prime_rl/trainer/models/layers/moe.py with stub modules for the imports that are not
installed here (torchtitan's `expert_parallel` decorator — identity on one rank —, prime_rl.configs.trainer.EPCommBackend), builds
`MoE` modules on the CPU with seeded bf16 weights, runs forward + backward on seeded inputs and stores inputs, weights, router outputs,
layer output and gradients.  Cases: the Qwen3-MoE arguments (softmax, route_norm, no shared expert, scores after the experts), the module
defaults (sigmoid, shared expert, scores before the experts, load-balance bias), forced `routed_experts`.  The for-loop expert path is used
(`use_grouped_mm=False`): on one rank it is the same arithmetic as the grouped path, and it runs on the CPU.
"""

from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/src/prime_rl/trainer/models/layers/moe.py")
OUT = Path(__file__).resolve().parent / "moe.npz"


def load_reference_moe():
    tt = types.ModuleType("torchtitan")
    ttd = types.ModuleType("torchtitan.distributed")
    tte = types.ModuleType("torchtitan.distributed.expert_parallel")
    tte.expert_parallel = lambda f: f
    sys.modules.update({"torchtitan": tt, "torchtitan.distributed": ttd, "torchtitan.distributed.expert_parallel": tte})
    for name in ("prime_rl", "prime_rl.configs"):
        sys.modules.setdefault(name, types.ModuleType(name))
    cfg = types.ModuleType("prime_rl.configs.trainer")
    cfg.EPCommBackend = str
    sys.modules["prime_rl.configs.trainer"] = cfg
    spec = importlib.util.spec_from_file_location("ref_moe", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CASES = {
    # name: (MoEArgs kwargs, dim, hidden, n_tokens, forced routing?, expert bias?)
    "qwen3": (dict(num_experts=8, num_shared_experts=0, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False,
                   top_k=2, use_grouped_mm=False, load_balance_coeff=None), 64, 32, 96, False, False),
    "default": (dict(num_experts=4, num_shared_experts=1, score_func="sigmoid", route_norm=False, route_scale=2.5, score_before_experts=True,
                     top_k=1, use_grouped_mm=False, load_balance_coeff=1e-3), 32, 48, 50, False, True),
    "forced": (dict(num_experts=8, num_shared_experts=0, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False,
                    top_k=4, use_grouped_mm=False, load_balance_coeff=None), 64, 32, 40, True, False),
    "sigmoid_norm": (dict(num_experts=16, num_shared_experts=0, score_func="sigmoid", route_norm=True, route_scale=1.0,
                          score_before_experts=False, top_k=8, use_grouped_mm=False, load_balance_coeff=None), 32, 16, 64, False, False),
}


def main() -> None:
    ref = load_reference_moe()
    # torch.histc has no integer kernel on the CPU (the reference runs it on the GPU): route integer input through float and back (the GPU kernel returns the input dtype)
    _histc = torch.histc
    torch.histc = lambda t, **kw: _histc(t.float(), **kw).to(t.dtype) if not t.is_floating_point() else _histc(t, **kw)
    out: dict[str, np.ndarray] = {}
    for ci, (name, (kw, dim, hidden, n, forced, bias)) in enumerate(CASES.items()):
        g = torch.Generator().manual_seed(1000 + ci)
        moe = ref.MoE(ref.MoEArgs(**kw), dim=dim, hidden_dim=hidden)
        with torch.no_grad():
            for p in moe.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.dim() == 2 and p.shape[0] == kw["num_experts"] else 0.15))
            if bias:
                moe.expert_bias.copy_(torch.randn(kw["num_experts"], generator=g) * 0.3)
        moe = moe.to(torch.bfloat16)
        if bias:
            moe.expert_bias = moe.expert_bias.float()  # the buffer stays f32 in the reference (model.py keeps buffers out of the bf16 cast)
        x = (torch.randn(1, n, dim, generator=g)).to(torch.bfloat16).requires_grad_(True)
        routed = None
        if forced:
            routed = torch.stack([torch.randperm(kw["num_experts"], generator=g)[: kw["top_k"]] for _ in range(n)]).reshape(1, n, kw["top_k"])
        top_scores, idx, counts, conf = moe.router(x.detach().view(-1, dim), moe.expert_bias, routed_experts=None if routed is None else routed.reshape(-1, kw["top_k"]))
        y = moe(x, routed_experts=routed)
        gy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
        y.backward(gy)
        pre = f"{name}."
        out[pre + "x"] = x.detach().float().numpy()
        out[pre + "gy"] = gy.float().numpy()
        out[pre + "y"] = y.detach().float().numpy()
        out[pre + "gx"] = x.grad.float().numpy()
        out[pre + "top_scores"] = top_scores.detach().float().numpy()
        out[pre + "top_idx"] = idx.numpy().astype(np.int32)
        out[pre + "counts"] = counts.detach().float().numpy()
        out[pre + "confidence"] = np.asarray(float(conf))
        out[pre + "tokens_per_expert"] = moe.tokens_per_expert.float().numpy()
        if routed is not None:
            out[pre + "routed"] = routed.numpy().astype(np.int32)
        if bias:
            out[pre + "expert_bias"] = moe.expert_bias.float().numpy()
        for pn, p in moe.named_parameters():
            out[pre + "w." + pn] = p.detach().float().numpy()
            out[pre + "g." + pn] = p.grad.float().numpy()
        out[pre + "args"] = np.asarray(repr(dict(kw, dim=dim, hidden=hidden, n=n)))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
