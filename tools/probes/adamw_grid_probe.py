"""AdamW launch time against the grid cap (PM_UPDATE_GRID is read once per process: run one process per value).
    for g in 128 256 512 1024 4096; do PM_UPDATE_GRID=$g python tools/probes/adamw_grid_probe.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from prime_rl_amd import ops  # noqa: E402

dev = "cuda:0"
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000_000
master = torch.empty(n, dtype=torch.float32, device=dev).normal_(0, 0.02)
m1 = torch.empty(n, dtype=torch.float32, device=dev).normal_(0, 1e-3)
m2 = torch.empty(n, dtype=torch.float32, device=dev).uniform_(1e-8, 1e-6)
g = torch.empty(n, dtype=torch.float32, device=dev).normal_(0, 1e-3)
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, dtype=torch.float32, device=dev)
ts = []
for i in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.adamw_step(master, m1, m2, g, pb, 1e-6, 0.9, 0.999, 1e-8, 0.01, i + 1, 1.0, coef, zero_grad=False)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
best = min(ts[1:])
print(f"PM_UPDATE_GRID={os.environ.get('PM_UPDATE_GRID', 'default 4096')}: {n / 1e9:.1f} G parameters, {best:.2f} ms, {30 * n / best / 1e9:.2f} TB/s (30 B / parameter)", flush=True)
