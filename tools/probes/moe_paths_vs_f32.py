"""Probe (GPU): the MoE layer on the MFMA grouped-GEMM / fused-SwiGLU path and on the eager per-expert loop, each against an f32 evaluation of
the same layer with the same routing (oracle/moe_oracle.py).  Result on the MI355X: identical outputs; row gradient 0.46 % (fused) vs 0.50 %
(loop) mean error; w1 gradient 0.40 % vs 0.43 %."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs
DEV="cuda:0"
torch.manual_seed(0)
dim, hidden, e, k, n = 256, 128, 8, 2, 777
mk = lambda grouped: MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k,
                                 use_grouped_mm=grouped, load_balance_coeff=None), dim, hidden).to(DEV).to(torch.bfloat16)
a, b = mk(True), mk(False)
a.init_weights(0.05); a.experts.grouped_impl = "hip"
b.load_state_dict(a.state_dict())
c = mk(False).float(); c.load_state_dict({k_: v.float() for k_, v in a.state_dict().items()})
x = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
g = torch.randn(1, n, dim, device=DEV).to(torch.bfloat16)
outs = {}
for name, m, xx in (("hip", a, x), ("loop", b, x)):
    xi = xx.clone().requires_grad_(True)
    y = m(xi); y.backward(g)
    outs[name] = (y.float(), xi.grad.float(), m.experts.w1.grad.float(), m.experts.w2.grad.float(), m.router.gate.weight.grad.float())
# f32 truth with the same routing: use oracle-ish eager f32 of the loop path with forced routing from a
from oracle import moe_oracle as M
args = M.MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, route_scale=1.0, score_before_experts=False, top_k=k, load_balance_coeff=None)
params = {k_: v.float().clone().requires_grad_(True) for k_, v in a.state_dict().items() if v.dtype == torch.bfloat16}
xi = x.float().reshape(-1, dim).clone().requires_grad_(True)
idx = a.router.last_sort[0]
y, _ = M.moe_forward(xi, params, args, routed_experts=idx.long())
y.backward(g.float().reshape(-1, dim))
ref = (y.detach(), xi.grad, params["experts.w1"].grad, params["experts.w2"].grad, params["router.gate.weight"].grad)
for i, nm in enumerate(["y", "gx", "gw1", "gw2", "ggate"]):
    r = ref[i].reshape(outs["hip"][i].shape)
    eh = (outs["hip"][i] - r).abs(); el = (outs["loop"][i] - r).abs()
    print(nm, "ref mean", float(r.abs().mean()), "hip err mean/max", float(eh.mean()), float(eh.max()), "loop err mean/max", float(el.mean()), float(el.max()))
