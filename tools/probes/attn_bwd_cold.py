"""dK/dV builds with the caches flushed between launches (the in-step condition: Q / K / V / dO come from HBM, not from a warm L2 / Infinity Cache).
python tools/probes/attn_bwd_cold.py 7,0 [docs]   (VARIANTS=1 library)"""
import sys
import json
import torch
sys.path.insert(0, "tools")
import microbench as M
from prime_rl_amd import _lib, ops

lib = _lib.load()
variants = [int(x) for x in sys.argv[1].split(",")]
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, hq, hkv, d = 16384, 32, 8, 128
q = torch.randn(n, hq, d, device="cuda").to(torch.bfloat16)
k = torch.randn(n, hkv, d, device="cuda").to(torch.bfloat16)
v = torch.randn(n, hkv, d, device="cuda").to(torch.bfloat16)
cu = torch.tensor([0] + [n // docs * (i + 1) for i in range(docs)], dtype=torch.int32, device="cuda")
ss = ops.seq_start_from_cu(cu, n)
o, lse = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
go = torch.randn_like(o)
res = {var: [] for var in variants}
for _ in range(5):
    for var in variants:
        lib.pm_attn_debug_set_variant(1, var)
        med, best = M.timeit(lambda: ops.attn_varlen_bwd(go, q, k, v, o, lse, ss, d**-0.5), iters=10, warmup=2, flush_mb=1024)
        res[var].append(med)
lib.pm_attn_debug_set_variant(1, 0)
for var in variants:
    ts = sorted(res[var])
    print(json.dumps(dict(variant=var, docs=docs, cold_median_ms=round(ts[len(ts) // 2] * 1e3, 4), cold_min_ms=round(ts[0] * 1e3, 4))))
