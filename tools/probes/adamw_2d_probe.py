"""pm_adamw_step_2d against pm_adamw_step + pm_transpose_2d on one matrix: time per launch and effective bytes / s (30 B / parameter flat, 32 B with the
transposed copy, + 4 B for a separate transpose)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from prime_rl_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=6):
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts[1:])


for rows, cols in ((24576, 4096), (4096, 12288), (98304, 8192)):
    n = rows * cols
    p = torch.empty(n, device=dev).normal_(0, 0.02)
    m = torch.empty(n, device=dev).normal_(0, 1e-3)
    v = torch.empty(n, device=dev).uniform_(1e-8, 1e-6)
    g = torch.empty(n, device=dev).normal_(0, 1e-3)
    pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    pt = torch.empty(cols, rows, dtype=torch.bfloat16, device=dev)
    coef = torch.ones(1, device=dev)
    t_flat = timed(lambda: ops.adamw_step(p, m, v, g, pb, 1e-6, 0.9, 0.999, 1e-8, 0.01, 3, 1.0, coef, zero_grad=False))
    t_tr = timed(lambda: ops.transpose_2d(pb.view(rows, cols), out=pt))
    t_2d = timed(lambda: ops.adamw_step_2d(p, m, v, g, pb, pt, 1e-6, 0.9, 0.999, 1e-8, 0.01, 3, 1.0, coef, zero_grad=False))
    print(f"[{rows} x {cols}] flat {t_flat:.3f} ms ({30 * n / t_flat / 1e9:.2f} TB/s) + transpose {t_tr:.3f} ms ({4 * n / t_tr / 1e9:.2f} TB/s) = {t_flat + t_tr:.3f};  "
          f"2-D {t_2d:.3f} ms ({32 * n / t_2d / 1e9:.2f} TB/s)", flush=True)
