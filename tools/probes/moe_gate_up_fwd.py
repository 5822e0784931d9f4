"""The experts' gate|up forward of a Qwen3-30B-A3B layer: grouped GEMM + pm_swiglu_fwd against the fused epilogue (pm_grouped_gemm_nt_swiglu), interleaved."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from prime_rl_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main(n=16384, e=128, k=8, h=2048, hidden=768):
    g = torch.Generator(device="cpu").manual_seed(0)
    logits = (torch.randn(n, e, generator=g) * 2).to(torch.bfloat16).to(DEV)
    scores, top, idx, counts, conf = ops.moe_router_fwd(logits, k, "softmax", True, 1.0)
    dest, slot, offsets = ops.moe_sort(idx.reshape(-1), e)
    y = torch.randn(n * k, h, generator=g).to(torch.bfloat16).to(DEV)
    both = (torch.randn(2, e, hidden, h, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    w1, w3 = both[0], both[1]
    cases = {
        "grouped GEMM (gate|up) alone": lambda: ops.grouped_gemm_nt(y, w1, offsets, w2=w3),
        "grouped GEMM + pm_swiglu_fwd": lambda: ops.swiglu_fwd(ops.grouped_gemm_nt(y, w1, offsets, w2=w3)),
        "pm_grouped_gemm_nt_swiglu (fused epilogue)": lambda: ops.grouped_gemm_nt_swiglu(y, w1, w3, offsets),
    }
    res = {nm: [] for nm in cases}
    for _ in range(4):
        for nm, fn in cases.items():
            res[nm].append(timeit(fn))
    for nm in cases:
        print(json.dumps(dict(kernel=nm, ms=round(sorted(res[nm])[1], 4))), flush=True)


main()
