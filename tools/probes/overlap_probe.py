"""Does an HBM-bound element-wise kernel (the fused AdamW) overlap with MFMA-bound GEMMs when both are queued on different HIP streams of one MI355X?
Times: the GEMM sequence alone, AdamW alone, both at once (AdamW on a side stream).  Library GEMM (one wave per SIMD, 256 VGPRs) and the own core (two waves per
SIMD, all 512 registers of a SIMD lane) are tried separately: only the former leaves register-file room for a second kernel's waves."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from prime_rl_amd import ops  # noqa: E402

dev = "cuda:0"
T, H, I = 16384, 4096, 12288
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = (torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)
dy = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
gw = torch.empty(2 * I, H, dtype=torch.float32, device=dev)
n = 2_000_000_000
master = torch.zeros(n, dtype=torch.float32, device=dev)
m1, m2 = torch.zeros_like(master), torch.zeros_like(master)
g = torch.full((n,), 1e-3, dtype=torch.float32, device=dev)
pb = torch.zeros(n, dtype=torch.bfloat16, device=dev)
coef = torch.ones(1, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()

def gemms(kind, reps=12):
    for _ in range(reps):
        if kind == "lib":
            torch.mm(x, w.t())
        else:
            ops.gemm(dy.t(), x.t(), out=gw)

def adamw():
    ops.adamw_step(master, m1, m2, g, pb, 1e-6, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, coef, zero_grad=False)

def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)

def both(kind):
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        adamw()
        done = torch.cuda.Event()
        done.record(side)
    gemms(kind)
    torch.cuda.current_stream().wait_event(done)

for kind in ("lib", "own"):
    for _ in range(2):
        a = timed(lambda: gemms(kind)); b = timed(adamw); c = timed(lambda: both(kind))
    print(f"{kind}: GEMMs alone {a:.2f} ms, AdamW (2 G parameters) alone {b:.2f} ms, together {c:.2f} ms (serial would be {a + b:.2f})", flush=True)
