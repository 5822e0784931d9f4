"""Does a power-of-two row stride cost the GEMM kernels L2 hits?  Same product with contiguous operands (row stride K = 4096 elements = 8 KiB) and with operands
whose rows are PAD elements further apart (views into wider buffers).  Own core (variant 7) and the library, interleaved, random operands."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from prime_rl_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    t, o, i = 16384, 24576, 4096
    fl = 2.0 * t * o * i
    for pad in (0, 64, 128, 192):
        xb = torch.randn(t, i + pad, device=DEV).to(torch.bfloat16)
        wb = (torch.randn(o, i + pad, device=DEV) * 0.02).to(torch.bfloat16)
        x, w = xb[:, :i], wb[:, :i]
        y = torch.empty(t, o, dtype=torch.bfloat16, device=DEV)
        ops.gemm_set_variant(7)
        res = {}
        for rnd in range(3):
            res.setdefault("own", []).append(timeit(lambda: ops.gemm(x, w, out=y)))
            res.setdefault("lib", []).append(timeit(lambda: torch.mm(x, w.t(), out=y)))
        rec = {"pad_elements": pad, "row_stride_bytes": 2 * (i + pad)}
        for k, v in res.items():
            v.sort()
            rec[k + "_TF"] = round(fl / v[1] / 1e12, 1)
        print(json.dumps(rec), flush=True)
        del xb, wb, y


if __name__ == "__main__":
    main()
