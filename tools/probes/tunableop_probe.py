"""Does PyTorch's TunableOp (a timed search over the hipBLASLt / rocBLAS solutions of one GEMM shape) find anything faster than the default
heuristic pick for the library products of the step (forward / dgrad, NT layout, 16 384 tokens)?  Same process: default pick timed first, then the
tuned pick.  Writes the chosen solutions to gpurun_out/tunableop/results.csv."""
import os
import sys
import time

import torch
import torch.cuda.tunable as tun

dev = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
shapes = [("qkv fwd", 6144, 4096), ("o fwd", 4096, 4096), ("gate_up fwd", 24576, 4096), ("down fwd", 4096, 12288),
          ("qkv dgrad", 4096, 6144), ("gate_up dgrad", 4096, 24576), ("down dgrad", 12288, 4096)]
if os.environ.get("PM_TUNE_HEAD"):
    shapes += [("head fwd", 151936, 4096), ("head dgrad", 4096, 151936)]


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


os.makedirs("gpurun_out/tunableop", exist_ok=True)
tun.set_filename("gpurun_out/tunableop/results.csv")
tun.set_max_tuning_duration(int(os.environ.get("PM_TUNE_MS", "8")))
tun.set_max_tuning_iterations(int(os.environ.get("PM_TUNE_ITERS", "4")))
t_start = time.time()
for name, n, k in shapes:
    x = torch.randn(T, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(T, n, dtype=torch.bfloat16, device=dev)
    fn = lambda: torch.mm(x, w.t(), out=out)
    tun.enable(False)
    base = timeit(fn)
    ref = out.clone()
    tun.enable(True)
    tun.tuning_enable(True)
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    t_tune = time.time() - t0
    tun.tuning_enable(False)
    tuned = timeit(fn)
    same = bool(torch.equal(ref, out))
    tun.enable(False)
    base2 = timeit(fn)
    fl = 2.0 * T * n * k
    print(f"{name:14s} default {base:.4f} / {base2:.4f} ms ({fl / base / 1e9:.0f} TF)  tuned {tuned:.4f} ms ({fl / tuned / 1e9:.0f} TF)  x{base / tuned:.3f}  "
          f"search {t_tune:.1f} s  bit-equal-to-default {same}", flush=True)
    if time.time() - t_start > float(os.environ.get("PM_TUNE_BUDGET_S", "150")):
        print("time budget reached", flush=True)
        break
tun.enable(True)
tun.write_file() if hasattr(tun, "write_file") else None
print(open("gpurun_out/tunableop/results.csv").read()[:3000] if os.path.exists("gpurun_out/tunableop/results.csv") else "no results file")
