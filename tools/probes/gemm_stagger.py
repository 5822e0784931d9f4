"""EXPERIMENT (measured, not adopted; the kernel hook it drove was removed again — see profiles/r05_gemm.md section 7 for the patch and the result): the dense core with its
workgroups started in four phases a quarter of a tile apart (schedule word bit 9) against the lock-step start, on the
products of one Qwen3-8B layer at 16 384 tokens; interleaved in one process."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from prime_rl_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def timeit(fn, iters=8, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main(t=16384, h=4096, inter=12288):
    r = lambda *s, sc=1.0: (torch.randn(*s, device=DEV) * sc).to(torch.bfloat16)  # noqa: E731
    x, dy, wgu, wd, gu = r(t, h), r(t, h), r(2 * inter, h, sc=0.02), r(h, inter, sc=0.02), r(t, 2 * inter)
    dgu, act = r(t, 2 * inter), r(t, inter)
    g32 = torch.zeros(2 * inter, h, dtype=torch.float32, device=DEV)
    o1, o2, o3 = torch.empty(t, 2 * inter, dtype=torch.bfloat16, device=DEV), torch.empty(t, inter, dtype=torch.bfloat16, device=DEV), torch.empty(t, h, dtype=torch.bfloat16, device=DEV)
    cases = {
        "gate|up forward + SwiGLU epilogue": lambda v: ops.gemm_swiglu(x, wgu, gate_up=o1, act=o2, variant=v),
        "down row gradient + SwiGLU-backward epilogue": lambda v: ops.gemm_swiglu_bwd(dy, wd, gu, out=o1, variant=v),
        "down forward": lambda v: ops.gemm(act, wd, out=o3, variant=v),
        "gate|up row gradient": lambda v: ops.gemm(dgu, wgu.t(), out=o3, variant=v),
        "gate|up weight gradient": lambda v: ops.gemm(dgu.t(), x.t(), out=g32, variant=v),
    }
    res = {(nm, v): [] for nm in cases for v in (7, 7 + 512)}
    for _ in range(4):
        for nm, fn in cases.items():
            for v in (7, 7 + 512):
                res[(nm, v)].append(timeit(lambda: fn(v)))
    for nm in cases:
        a, b = sorted(res[(nm, 7)])[1], sorted(res[(nm, 7 + 512)])[1]
        print(json.dumps(dict(kernel=nm, lockstep_ms=round(a, 4), staggered_ms=round(b, 4), ratio=round(b / a, 4))), flush=True)


main()
