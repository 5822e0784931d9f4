"""The grouped weight-gradient product of a Qwen3-30B-A3B expert layer: bf16 results vs f32 written / accumulated into the main-gradient buffer
(ops.grouped_gemm_tn vs ops.grouped_gemm_tn_f32), gate|up shape (two outputs) and down shape."""
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
    m = n * k
    y = torch.randn(m, h, generator=g).to(torch.bfloat16).to(DEV)
    dgu = torch.randn(m, 2 * hidden, generator=g).to(torch.bfloat16).to(DEV)
    hmid = torch.randn(m, hidden, generator=g).to(torch.bfloat16).to(DEV)
    gy = torch.randn(m, h, generator=g).to(torch.bfloat16).to(DEV)
    mg1, mg3 = (torch.zeros(e, hidden, h, dtype=torch.float32, device=DEV) for _ in range(2))
    mg2 = torch.zeros(e, h, hidden, dtype=torch.float32, device=DEV)
    cases = {
        "gate|up bf16 (two outputs)": (lambda: ops.grouped_gemm_tn(dgu, y, offsets, n_first=hidden), 2.0 * m * h * 2 * hidden),
        "gate|up f32 write": (lambda: ops.grouped_gemm_tn_f32(dgu, y, offsets, mg1, mg3, accumulate=False), 2.0 * m * h * 2 * hidden),
        "gate|up f32 accumulate": (lambda: ops.grouped_gemm_tn_f32(dgu, y, offsets, mg1, mg3, accumulate=True), 2.0 * m * h * 2 * hidden),
        "down bf16": (lambda: ops.grouped_gemm_tn(gy, hmid, offsets), 2.0 * m * h * hidden),
        "down f32 write": (lambda: ops.grouped_gemm_tn_f32(gy, hmid, offsets, mg2, None, accumulate=False), 2.0 * m * h * hidden),
        "down f32 accumulate": (lambda: ops.grouped_gemm_tn_f32(gy, hmid, offsets, mg2, None, accumulate=True), 2.0 * m * h * hidden),
    }
    res = {nm: [] for nm in cases}
    for _ in range(3):
        for nm, (fn, fl) in cases.items():
            res[nm].append(timeit(fn))
    for nm, (fn, fl) in cases.items():
        t = sorted(res[nm])[1]
        print(json.dumps(dict(kernel=f"grouped weight gradient {nm}", ms=round(t, 4), TF=round(fl / t / 1e9, 1))), flush=True)


main()
