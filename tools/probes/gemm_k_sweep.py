"""Per-tile fixed cost of the GEMM core: time of one forward product [16384, N] = [16384, K] x [N, K]^T over a sweep of K at fixed M, N — the intercept of the
line is what a tile costs besides its K loop (epilogue, tile switch, pipeline refill), the slope the K-loop rate.  PM_PROBE_ZERO=1: zero-filled operands."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from prime_rl_amd import ops  # noqa: E402

DEV = "cuda:0"
variants = [int(v) for v in os.environ.get("PM_GEMM_VARIANTS", "7").split(",")]
m, n = 16384, int(os.environ.get("PM_PROBE_N", "24576"))
zero = os.environ.get("PM_PROBE_ZERO") == "1"


def timeit(fn, iters=8, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for k in (2048, 4096, 8192, 16384):
    x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
    w = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
    if zero:
        x.zero_(); w.zero_()
    y = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    rec = {"k": k, "zero": zero, "lib_ms": timeit(lambda: torch.mm(x, w.t(), out=y))}
    for v in variants:
        rec[f"v{v}_ms"] = timeit(lambda: ops.gemm(x, w, out=y, variant=v))
    tiles_per_wg = (m // 256) * (n // 256) / 256
    rec["tiles_per_workgroup"] = tiles_per_wg
    print(json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in rec.items()}), flush=True)
    del x, w, y
