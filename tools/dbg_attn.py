import sys, numpy as np, torch
sys.path.insert(0, '.')
from prime_rl_amd import ops
from oracle import grpo_oracle as O
sys.path.insert(0, 'tests')
import test_kernels_gpu as T
DEV='cuda:0'
for seq_lens,hq,hkv in (([64, 1, 100, 37, 98], 8, 2), ([300,212],8,8)):
    n=sum(seq_lens); d=128
    q,k,v = T._attn_case(n, seq_lens, hq, hkv, n)
    scale=d**-0.5
    cu = torch.tensor([0] + list(np.cumsum(seq_lens)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    out, lse = ops.attn_varlen_fwd(q.to(DEV), k.to(DEV), v.to(DEV), ss, scale)
    # reference lse
    bounds=np.cumsum([0]+seq_lens)
    ref=torch.empty(hq,n)
    for h in range(hq):
        for s,e in zip(bounds[:-1],bounds[1:]):
            sc=(q[s:e,h].float()@k[s:e,h//(hq//hkv)].float().T)*scale
            sc=sc.masked_fill(torch.ones(e-s,e-s,dtype=torch.bool).triu(1), float('-inf'))
            ref[h,s:e]=torch.logsumexp(sc,-1)
    l=lse.cpu()
    bad=~torch.isfinite(l)
    print(seq_lens, 'nonfinite lse:', int(bad.sum()), 'max err on finite', float((l-ref)[~bad].abs().max()))
    if bad.any():
        idx=bad.nonzero()[:10].tolist(); print('first bad (head,row):', idx, [float(l[i,j]) for i,j in idx])
    print('out finite:', bool(torch.isfinite(out.float()).all()))
