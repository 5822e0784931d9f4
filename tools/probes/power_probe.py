"""Board power and clocks (rocm-smi) while one kernel class loops: library GEMM, own GEMM core, AdamW, attention forward.  Evidence for the
'power-managed matrix kernels' reading of profiles/r04_step_clocks.txt."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from prime_rl_amd import ops  # noqa: E402

dev = "cuda:0"
T, H, I = 16384, 4096, 12288
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w = (torch.randn(2 * I, H, device=dev) * 0.02).to(torch.bfloat16)
dy = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
gw = torch.empty(2 * I, H, dtype=torch.float32, device=dev)
n = 1_000_000_000
master = torch.empty(n, device=dev).normal_(0, 0.02); m1 = torch.zeros_like(master); m2 = torch.full_like(master, 1e-6); g = torch.empty(n, device=dev).normal_(0, 1e-3)
pb = torch.empty(n, dtype=torch.bfloat16, device=dev); coef = torch.ones(1, device=dev)
q = torch.randn(T, 32, 128, device=dev).to(torch.bfloat16); k = torch.randn(T, 8, 128, device=dev).to(torch.bfloat16); v = torch.randn(T, 8, 128, device=dev).to(torch.bfloat16)
gu = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16); dact = torch.randn(T, I, device=dev).to(torch.bfloat16); wn = torch.ones(H, dtype=torch.bfloat16, device=dev)
ss = ops.seq_start_from_cu(torch.tensor([0] + [2048 * (i + 1) for i in range(8)], dtype=torch.int32, device=dev), T)
work = {
    "library GEMM (gate|up forward)": lambda: torch.mm(x, w.t()),
    "own GEMM core (gate|up wgrad)": lambda: ops.gemm(dy.t(), x.t(), out=gw),
    "AdamW (1 G parameters)": lambda: ops.adamw_step(master, m1, m2, g, pb, 1e-6, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, coef, zero_grad=False),
    "attention forward": lambda: ops.attn_varlen_fwd(q, k, v, ss, 128**-0.5),
    "SwiGLU backward": lambda: ops.swiglu_bwd(dact, gu),
    "SwiGLU forward": lambda: ops.swiglu_fwd(gu),
    "RMSNorm forward (+ residual)": lambda: ops.rmsnorm_fwd(x, wn, 1e-6, residual=x),
}
if os.environ.get("PM_POWER_ONLY"):
    work = {k_: v_ for k_, v_ in work.items() if any(t in k_ for t in os.environ["PM_POWER_ONLY"].split(","))}


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    mx = re.findall(r"Max Graphics Package Power \(W\): ([\d.]+)", out)
    sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (pw[0] if pw else "?"), (mx[0] if mx else "?"), (sclk[0] if sclk else "?"), out


print("idle:", smi()[:3], flush=True)
for name, fn in work.items():
    stop = False

    def loop():
        while not stop:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    th = threading.Thread(target=loop); th.start()
    time.sleep(2.0)
    samples = [smi()[:3] for _ in range(3)]
    stop = True; th.join()
    print(f"{name:34s} power W / cap W / sclk MHz: {samples}", flush=True)
print(smi()[3][:1500])
