"""Where does a forward variant differ from the shipped kernel?  python tools/probes/attn_p3_diff.py VARIANT 200 [hq hkv]  (VARIANTS=1 library)"""
import sys
import numpy as np
import torch
from prime_rl_amd import _lib, ops

lib = _lib.load()
var = int(sys.argv[1])
seqs = [int(x) for x in sys.argv[2].split(",")]
hq, hkv = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4, 2)
n, d = sum(seqs), 128
torch.manual_seed(0)
q = torch.randn(n, hq, d, device="cuda").to(torch.bfloat16)
k = torch.randn(n, hkv, d, device="cuda").to(torch.bfloat16)
v = torch.randn(n, hkv, d, device="cuda").to(torch.bfloat16)
ss = ops.seq_start_from_cu(torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32, device="cuda"), n)
o0, l0 = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
assert lib.pm_attn_debug_set_variant(0, var) == 0
o1, l1 = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
lib.pm_attn_debug_set_variant(0, -1)
do = (o0.float() - o1.float()).abs()
dl = (l0 - l1).abs()
print("out max diff", float(do.max()), "lse max diff", float(dl.max()), "nan", bool(torch.isnan(o1.float()).any()))
rows = torch.nonzero(do.amax(dim=(1, 2)) > 0)[:, 0].tolist()
print("rows with out diff:", rows[:40], "count", len(rows))
heads = torch.nonzero(do.amax(dim=(0, 2)) > 0)[:, 0].tolist()
print("heads:", heads)
if rows:
    r = rows[0]
    print("row", r, "old", o0[r, heads[0], :8].float().tolist(), "new", o1[r, heads[0], :8].float().tolist())
lr = torch.nonzero(dl.amax(dim=0) > 0)[:, 0].tolist()
print("rows with lse diff:", lr[:40], "count", len(lr))
