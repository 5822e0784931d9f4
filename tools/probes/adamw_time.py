"""pm_adamw_step on 2 Gi parameters (the flat engine launches it once over 8.2 G): ms and TB/s at 30 B per parameter."""
import torch

from prime_rl_amd import ops

n = 2 << 30
dev = "cuda:0"
master, m, v, g = (torch.zeros(n, device=dev) for _ in range(4))
g.normal_()
pb = torch.zeros(n, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, device=dev)
for sizes in (n, n // 64, 1 << 20):
    ts = []
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.adamw_step(master[:sizes], m[:sizes], v[:sizes], g[:sizes], pb[:sizes], 1e-3, 0.9, 0.999, 1e-8, 0.01, it + 1, grad_scale=1.0, clip=coef, zero_grad=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    best = min(ts)
    print(f"{sizes:>12d} parameters: {best:8.3f} ms  {30.0 * sizes / (best * 1e-3) / 1e12:5.2f} TB/s")
