"""dgrad through the library WITHOUT the [in, out] weight copy: dx = dy @ W with W row-major [out, in] (NN).  The default pick is 10-25 % slower than the
NT product on the transposed copy (which is why the engine keeps the copies); does TunableOp's search find an NN solution that closes the gap?"""
import os
import sys
import time

import torch
import torch.cuda.tunable as tun

dev = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
shapes = [("gate_up dgrad", 24576, 4096), ("down dgrad", 4096, 12288), ("qkv dgrad", 6144, 4096), ("o dgrad", 4096, 4096)]


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
tun.set_filename("gpurun_out/tunableop/results_nn.csv")
tun.set_max_tuning_duration(int(os.environ.get("PM_TUNE_MS", "6")))
tun.set_max_tuning_iterations(int(os.environ.get("PM_TUNE_ITERS", "3")))
t_start = time.time()
for name, n_out, k_in in shapes:
    dy = torch.randn(T, n_out, device=dev).to(torch.bfloat16)
    w = (torch.randn(n_out, k_in, device=dev) * 0.02).to(torch.bfloat16)
    wt = w.t().contiguous()  # [in, out]: the copy the engine keeps
    out = torch.empty(T, k_in, dtype=torch.bfloat16, device=dev)
    nt = lambda: torch.mm(dy, wt.t(), out=out)
    nn = lambda: torch.mm(dy, w, out=out)
    tun.enable(False)
    t_nt = timeit(nt)
    ref = out.clone()
    t_nn = timeit(nn)
    same_default = bool(torch.equal(ref, out))
    tun.enable(True)
    tun.tuning_enable(True)
    t0 = time.time()
    nn()
    torch.cuda.synchronize()
    t_search = time.time() - t0
    tun.tuning_enable(False)
    t_tuned = timeit(nn)
    same = bool(torch.equal(ref, out))
    tun.enable(False)
    fl = 2.0 * T * n_out * k_in
    print(f"{name:14s} NT on the copy {t_nt:.4f} ms ({fl / t_nt / 1e9:.0f} TF)  NN default {t_nn:.4f} ({fl / t_nn / 1e9:.0f} TF)  NN tuned {t_tuned:.4f} "
          f"({fl / t_tuned / 1e9:.0f} TF)  search {t_search:.1f} s  equal to NT: default {same_default}, tuned {same}", flush=True)
    if time.time() - t_start > float(os.environ.get("PM_TUNE_BUDGET_S", "150")):
        print("time budget reached", flush=True)
        break
print(open("gpurun_out/tunableop/results_nn.csv").read()[-1500:] if os.path.exists("gpurun_out/tunableop/results_nn.csv") else "no results file yet (written at exit)")
