"""How balanced is the routing of the synthetic Qwen3-30B-A3B-width model of `bench.py`'s moe_step / `tools/microbench.py moestep` (random weights, random
tokens)?  Prints, per MoE layer, max / mean / min tokens per expert of one 8 x 2048-token pass."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from prime_rl_amd.trainer.models.layers.moe import MoE  # noqa: E402

import microbench  # noqa: E402

seen = []
orig = MoE.forward


def spy(self, x, routed_experts=None):
    out = orig(self, x, routed_experts)
    offs = self.router.last_sort[3]
    c = (offs[1:] - offs[:-1]).float()
    seen.append((float(c.max()), float(c.mean()), float(c.min()), int((c == 0).sum())))
    return out


MoE.forward = spy
res = []
microbench.bench_moe_step(res, layers=int(sys.argv[1]) if len(sys.argv) > 1 else 4, n_mb=8, steps=1)
n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i, (mx, mean, mn, empty) in enumerate(seen[-n_layers:]):
    print(json.dumps(dict(layer=i, max_tokens=mx, mean_tokens=mean, min_tokens=mn, empty_experts=empty, max_over_mean=round(mx / mean, 2))))
print(res)
