"""tools/parity_bisect.py — where does the full-depth loss difference between the engine and the reference-dtype oracle come from?

VERDICT r3 "next" 1a/1b.  Qwen3-8B at its real depth, `--rollouts` x 2048-token parity-variant rollouts (full-vocab ids, 256 masked
prompt tokens, lp_inf = lp + N(0, 0.02^2), group-relative advantages).  Runs ON THE GPU BOX (test / measurement infrastructure: it imports `oracle/`).

What it prints (and writes as JSON to --out):
  1. the oracles against each other: A (f32 end to end), B (the reference's bf16 rounding points, exact GEMMs), C (B with torch bf16 GEMMs),
     B_fa (B + flash-attention's bf16 P) — scalar loss, PAIRED per-token standard error of every loss difference, per-token log-prob statistics;
  2. the engine (gemm=f32: hand-written kernels only; bf16: as shipped) against B the same way;
  3. the BISECTION: the gemm=f32 engine with ONE kernel class at a time replaced by the oracle-B torch op (rmsnorm + residual, q/k-norm + RoPE,
     attention, SwiGLU, K1), and the complementary runs (everything torch EXCEPT one class) — loss(E) - loss(B) with its standard error per swap;
  4. LOCAL mismatch of every kernel class against the torch op ON IDENTICAL INPUTS, accumulated over all 36 layers of a real forward: fraction of
     bf16 outputs that differ, and the largest difference in bf16 ulps.

The standard error: loss = (1 / N) * sum_i l_i over the N trained tokens, so loss(X) - loss(B) = mean_i (l_i(X) - l_i(B)); its 1-sigma sampling
error is std_i(l_i(X) - l_i(B)) / sqrt(N).  A difference inside ~2 SE is not resolvable on that many tokens, whatever its sign."""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cpu_trainer as C  # noqa: E402
from oracle import grpo_oracle as O  # noqa: E402

DEV = "cuda:0"


def parity_variant_mbs(vocab, seq_len, n_rollouts, prompt_len, seed=0):
    """tests/test_engine_gpu.py::_parity_variant_mbs (BASELINE.md section 3 'parity variant')."""
    g = torch.Generator().manual_seed(seed)
    rewards = (torch.rand(8, 16, generator=g) < 0.5).float()
    adv = (rewards - rewards.mean(1, keepdim=True)).reshape(-1)
    pick = [(i * 128) // n_rollouts + (i % 16) for i in range(n_rollouts)]
    mbs = []
    for r in pick:
        mask = torch.ones(1, seq_len, dtype=torch.bool)
        mask[0, :prompt_len] = False
        a = float(adv[r % 128]) if float(adv[r % 128]) != 0 else 0.5
        mbs.append(dict(input_ids=torch.randint(0, vocab, (1, seq_len), generator=g), position_ids=torch.arange(seq_len)[None],
                        advantages=a * mask.float(), inference_logprobs=torch.zeros(1, seq_len), loss_mask=mask,
                        temperatures=torch.full((1, seq_len), 0.7 if r % 2 else 1.0), seq_lens=torch.tensor([seq_len]),
                        sequence_lengths=[seq_len], env_names=["parity"] * seq_len))
    return mbs


def per_token_loss(lp, inf, adv, mask, cfg: O.LossConfig):
    """default_loss_fn (trainer/rl/loss.py:109-161) per token, float64 on the host: l_i such that loss = sum_i l_i / N_trained."""
    lp, inf, adv = lp.double(), inf.double(), adv.double()
    lr = lp - inf
    pd = torch.exp(lp) - torch.exp(inf)
    invalid = torch.where(adv > 0, pd > cfg.dppo_mask_high, pd < -cfg.dppo_mask_low)
    keep = mask & ~invalid
    return (-(keep * (cfg.adv_tau * adv) * torch.exp(lr)) + cfg.kl_tau * (mask * lr**2)), invalid & mask


def bf16_ulps(a: torch.Tensor, b: torch.Tensor):
    """(fraction of elements that differ, max difference in bf16 ulps) of two bf16 tensors of equal shape."""
    def ordered(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    d = (ordered(a) - ordered(b)).abs()
    return float((d != 0).float().mean()), int(d.max())


class Swapper:
    """Monkeypatches prime_rl_amd.ops forward kernels with the oracle-B torch ops (same rounding points as the reference's eager bf16 modules)."""

    CLASSES = ("rmsnorm", "qknorm_rope", "attention", "swiglu", "k1")

    def __init__(self, ops, seq_lens, eps):
        self.ops, self.seq_lens, self.eps = ops, seq_lens, eps
        self.orig = {k: getattr(ops, k) for k in ("rmsnorm_fwd", "qknorm_rope_fwd", "attn_varlen_fwd", "swiglu_fwd", "logprob_entropy_fwd")}
        self.torch_for: set[str] = set()
        self.compare = False
        self.attn_p_bf16 = False
        self.stats: dict[str, list] = {}

    # -- the torch side (oracle-B rounding points, on the device) ----------------------------------------------------
    def t_rmsnorm(self, x, weight, eps, residual=None):
        s = None
        if residual is not None:
            s = residual + x  # bf16 add, rounded once (modeling_qwen3.py:80,85: hidden_states = residual + hidden_states)
        y = O.rmsnorm(s if s is not None else x, weight, eps)
        return y, None, s

    def t_qknorm_rope(self, qkv, qw, kw, cos, sin, hq, hkv, d, eps):
        n = qkv.shape[0]
        q = qkv[:, : hq * d].reshape(n, hq, d)
        k = qkv[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
        if qw is not None:
            q, k = O.rmsnorm(q, qw, eps), O.rmsnorm(k, kw, eps)
        cf, sf = torch.cat((cos, cos), -1), torch.cat((sin, sin), -1)
        return O.apply_rope(q, cf, sf).contiguous(), O.apply_rope(k, cf, sf).contiguous(), None

    def t_attn(self, q, k, v, seq_start, scale):
        n = q.shape[0]
        lens = self.seq_lens[: n // self.seq_lens[0]] if sum(self.seq_lens) != n else self.seq_lens
        o = O.varlen_causal_attention(q, k, v, lens, scale, p_bf16=self.attn_p_bf16).to(q.dtype)
        return o, None

    def t_swiglu(self, gu, out=None):
        i = gu.shape[1] // 2
        return O.swiglu(gu[:, :i], gu[:, i:])

    def t_k1(self, logits, labels, inv_temp):
        # the reference divides by the temperature (train.py:441); inv_temp = 1 / T with T in {0.7, 1.0}
        t = torch.round(10.0 / inv_temp) / 10.0
        n = logits.shape[0]
        lp = torch.empty(n, dtype=torch.float32, device=logits.device)
        en, lz = torch.empty_like(lp), torch.empty_like(lp)
        for s in range(0, n, 2048):
            x = logits[s:s + 2048].float() / t[s:s + 2048, None]
            z = torch.logsumexp(x, -1)
            lp[s:s + 2048] = torch.log_softmax(x, -1).gather(-1, labels[s:s + 2048, None])[:, 0]
            en[s:s + 2048] = z - (torch.softmax(x, -1) * x).sum(-1)
            lz[s:s + 2048] = z
        return lp, en, lz

    # -- dispatch ------------------------------------------------------------------------------------------------------
    def _note(self, cls, a, b):
        frac, ulp = bf16_ulps(a, b)
        self.stats.setdefault(cls, []).append((frac, ulp))

    def install(self):
        ops, S = self.ops, self

        def rmsnorm_fwd(x, weight, eps, residual=None):
            if "rmsnorm" in S.torch_for:
                return S.t_rmsnorm(x, weight, eps, residual)
            out = S.orig["rmsnorm_fwd"](x, weight, eps, residual)
            if S.compare:
                ty, _, ts = S.t_rmsnorm(x, weight, eps, residual)
                S._note("rmsnorm", out[0], ty)
                if ts is not None:
                    S._note("residual_add", out[2], ts)
            return out

        def qknorm_rope_fwd(qkv, qw, kw, cos, sin, hq, hkv, d, eps):
            if "qknorm_rope" in S.torch_for:
                return S.t_qknorm_rope(qkv, qw, kw, cos, sin, hq, hkv, d, eps)
            out = S.orig["qknorm_rope_fwd"](qkv, qw, kw, cos, sin, hq, hkv, d, eps)
            if S.compare:
                tq, tk, _ = S.t_qknorm_rope(qkv, qw, kw, cos, sin, hq, hkv, d, eps)
                S._note("qknorm_rope(q)", out[0], tq)
                S._note("qknorm_rope(k)", out[1], tk)
            return out

        def attn_varlen_fwd(q, k, v, seq_start, scale):
            if "attention" in S.torch_for:
                return S.t_attn(q, k, v, seq_start, scale)
            out = S.orig["attn_varlen_fwd"](q, k, v, seq_start, scale)
            if S.compare:
                for pb, nm in ((False, "attention vs f32-P"), (True, "attention vs bf16-P")):
                    S.attn_p_bf16 = pb
                    S._note(nm, out[0], S.t_attn(q, k, v, seq_start, scale)[0])
                S.attn_p_bf16 = False
            return out

        def swiglu_fwd(gu, out=None):
            if "swiglu" in S.torch_for:
                return S.t_swiglu(gu)
            o = S.orig["swiglu_fwd"](gu, out)
            if S.compare:
                S._note("swiglu", o, S.t_swiglu(gu))
            return o

        def logprob_entropy_fwd(logits, labels, inv_temp):
            if "k1" in S.torch_for:
                return S.t_k1(logits, labels, inv_temp)
            out = S.orig["logprob_entropy_fwd"](logits, labels, inv_temp)
            if S.compare:
                tl = S.t_k1(logits, labels, inv_temp)
                S.stats.setdefault("k1 logprob max|d| (f32)", []).append((float((out[0] - tl[0]).abs().max()), 0))
            return out

        for k, f in dict(rmsnorm_fwd=rmsnorm_fwd, qknorm_rope_fwd=qknorm_rope_fwd, attn_varlen_fwd=attn_varlen_fwd, swiglu_fwd=swiglu_fwd,
                         logprob_entropy_fwd=logprob_entropy_fwd).items():
            setattr(ops, k, f)

    def restore(self):
        for k, f in self.orig.items():
            setattr(self.ops, k, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=32)
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--prompt-len", type=int, default=256)
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--per-pass", type=int, default=8, help="rollouts per engine pass")
    ap.add_argument("--bisect-rollouts", type=int, default=None, help="rollouts used by the swap runs (default: all)")
    ap.add_argument("--skip-bisect", action="store_true")
    ap.add_argument("--skip-a", action="store_true", help="no f32 oracle: the sampling policy is oracle B + N(0, 0.02^2) (the large-N run)")
    ap.add_argument("--modes", default="bf16,f32", help="engine GEMM modes to run")
    ap.add_argument("--ref", default="b", choices=["b", "bfa"], help="the oracle every row is paired against: B (f32 softmax in attention) or B_fa (B + flash-attention's bf16 P)")
    ap.add_argument("--out", default="gpurun_out/parity_bisect.json")
    args = ap.parse_args()

    import dataclasses

    from prime_rl_amd import ops
    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches

    dims = dataclasses.replace(ModelDims.qwen3_8b(), num_hidden_layers=args.layers)
    cfg = O.TinyQwen3Config(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=dims.num_hidden_layers,
                            num_attention_heads=dims.num_attention_heads, num_key_value_heads=dims.num_key_value_heads, head_dim=dims.head_dim,
                            vocab_size=dims.vocab_size, rms_norm_eps=dims.rms_norm_eps, rope_theta=dims.rope_theta, qk_norm=dims.qk_norm,
                            attention_bias=dims.attention_bias, rope_scaling=dims.rope_scaling)
    lcfg = O.LossConfig()
    n, R = args.seq_len, args.rollouts
    t00 = time.time()
    seed_eng = GRPOEngine(dims, device=DEV, seed=0, optim=OptimConfig(type="sign_sgd"))
    sd = {k: v.detach().clone() for k, v in seed_eng.store.state_dict().items()}
    del seed_eng
    torch.cuda.empty_cache()
    mbs = parity_variant_mbs(dims.vocab_size, n, R, args.prompt_len)
    dev = torch.device(DEV)
    mbs_o = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in mb.items()} for mb in mbs]
    mask = torch.cat([mb["loss_mask"][0] for mb in mbs])
    adv = torch.cat([mb["advantages"][0] for mb in mbs])
    n_tr = int(mask.sum())
    pad = math.log(1.0 / dims.vocab_size)
    report: dict = {"model": dims.name, "layers": dims.num_hidden_layers, "rollouts": R, "seq_len": n, "trained_tokens": n_tr, "rows": []}

    # ---- oracles ---------------------------------------------------------------------------------------------------
    def oracle(params, **kw):
        t0 = time.time()
        with torch.no_grad():
            _, lps, _ = C.step_logprobs_and_loss(params, cfg, mbs_o, lcfg, 1, **kw)
        torch.cuda.synchronize()
        return torch.cat(lps).cpu(), time.time() - t0

    lp_a, ta = None, 0.0
    if not args.skip_a:
        p32 = {k: v.float().to(dev) for k, v in sd.items()}
        lp_a, ta = oracle(p32, f32_logits=True)
        del p32
        torch.cuda.empty_cache()
    p16 = {k: v.to(torch.bfloat16).to(dev) for k, v in sd.items()}
    lp_b, tb = oracle(p16, mm=O.exact_mm, f32_logits=True)
    g = torch.Generator().manual_seed(1234)
    inf = (lp_a if lp_a is not None else lp_b) + 0.02 * torch.randn(lp_b.shape, generator=g)
    for i, mb in enumerate(mbs):
        mb["inference_logprobs"] = inf[i * n:(i + 1) * n][None].clone()
    lp_c, tc = oracle(p16)
    lp_f, tf = oracle(p16, mm=O.exact_mm, f32_logits=True, attn_p_bf16=True)
    del p16
    torch.cuda.empty_cache()
    print(f"[oracles] A {ta:.1f}s  B {tb:.1f}s  C {tc:.1f}s  B_fa {tf:.1f}s   ({time.time() - t00:.0f}s since start)", flush=True)

    ref_name = "B"
    if args.ref == "bfa":  # pair everything against the restatement that shares flash-attention's rounding point (what the reference actually runs)
        lp_b, lp_f, ref_name = lp_f, lp_b, "B_fa"
    report["paired_against"] = ref_name
    l_b, inv_b = per_token_loss(lp_b, inf, adv, mask, lcfg)
    loss_b = float(l_b.sum() / n_tr)

    def row(name, lp, note=""):
        """`lp` may cover only the first k rollouts (the swap runs): every statistic is then taken over those tokens, B included."""
        m = lp.numel()
        mk = mask[:m]
        nt = int(mk.sum())
        l_x, inv_x = per_token_loss(lp, inf[:m], adv[:m], mk, lcfg)
        d = (l_x - l_b[:m])[mk]
        delta = float(d.sum() / nt)
        se = float(d.std() / math.sqrt(nt))
        dl = (lp - lp_b[:m])[mk].double()
        inv_x = torch.cat([inv_x, inv_b[m:]])
        n_tr = nt
        r = dict(name=name, tokens=nt, loss=float(l_x.sum() / n_tr), minus_B=delta, se=se, z=delta / se if se > 0 else 0.0,
                 lp_mean_shift=float(dl.mean()), lp_mean_shift_se=float(dl.std() / math.sqrt(n_tr)), lp_abs_mean=float(dl.abs().mean()),
                 lp_rms=float(dl.pow(2).mean().sqrt()), lp_max=float(dl.abs().max()), flips_vs_B=int((inv_x != inv_b).sum()), note=note)
        report["rows"].append(r)
        print(f"  {name:<46s} loss {r['loss']:+.7f}  -B {delta:+.2e} +- {se:.1e} (z {r['z']:+.1f})  lp shift {r['lp_mean_shift']:+.2e} +- {r['lp_mean_shift_se']:.1e}  "
              f"|dlp| mean {r['lp_abs_mean']:.2e} rms {r['lp_rms']:.2e} max {r['lp_max']:.2e}  flips {r['flips_vs_B']}  {note}", flush=True)
        return r

    print(f"[{dims.name} x{dims.num_hidden_layers}, {R} x {n} tokens, {n_tr} trained] loss({ref_name}) = {loss_b:+.7f}   (every '-B' column below is paired against {ref_name})")
    if lp_a is not None:
        row("oracle A (f32 end to end)", lp_a)
    row("oracle C (bf16 points, torch bf16 GEMMs)", lp_c)
    row("oracle B_fa (B + bf16 P into P.V)" if args.ref == "b" else "oracle B (f32 softmax into P.V)", lp_f)

    # ---- engine runs -----------------------------------------------------------------------------------------------
    sw = Swapper(ops, [n] * args.per_pass, dims.rms_norm_eps)
    sw.install()

    def engine_lp(eng, n_roll):
        out = []
        for s in range(0, n_roll, args.per_pass):
            chunk = mbs[s:s + args.per_pass]
            sw.seq_lens = [n] * len(chunk)
            pb = pack_micro_batches(chunk, DEV)
            ctx = eng.forward(pb)
            lp = ctx["logprobs"].cpu()
            out.append(torch.cat([torch.cat([lp.new_full((1,), pad), lp[i * n:(i + 1) * n - 1]]) for i in range(len(chunk))]))
            del ctx, pb
        return torch.cat(out)

    def sub(lp, n_roll):
        return lp

    try:
        for mode in [m for m in ("bf16", "f32") if m in args.modes.split(",")]:
            eng = GRPOEngine(dims, device=DEV, optim=OptimConfig(type="sign_sgd"), loss=LossConfig(), seed=0, gemm_dtype=mode, recompute="layer")
            eng.store.load_state_dict({k: v.to(DEV) for k, v in sd.items()})
            t0 = time.time()
            sw.torch_for, sw.compare = set(), False
            lp_e = engine_lp(eng, R)
            torch.cuda.synchronize()
            row(f"engine gemm={mode} (all kernels ours)", lp_e, note=f"{time.time() - t0:.1f}s")
            if mode == "bf16":
                # K2 against the host formula on the first pass
                pb = pack_micro_batches(mbs[: args.per_pass], DEV)
                ctx = eng.forward(pb)
                eng.loss(ctx, (pb.local_counts[0], 1, 1))
                k2 = ctx["loss"].item()
                sl = slice(0, args.per_pass * n)
                l_h, _ = per_token_loss(lp_e[sl], inf[sl], adv[sl], mask[sl], lcfg)
                host = float(l_h.sum() / int(mask[sl].sum()))
                print(f"  K2 (pm_grpo_loss_fwd_bwd) on pass 0: {k2:+.8f} vs the float64 host formula on the same log-probs {host:+.8f} (d {k2 - host:+.1e})")
                report["k2_minus_host"] = k2 - host
                del ctx, pb
                del eng
                torch.cuda.empty_cache()
                continue
            if not args.skip_bisect:
                nb = args.bisect_rollouts or R
                scale_note = f"first {nb} rollouts" if nb != R else ""
                # local mismatch on identical inputs, all layers of one pass
                sw.compare = True
                sw.stats = {}
                engine_lp(eng, min(args.per_pass, R))
                sw.compare = False
                print("[local mismatch of each kernel class vs the oracle-B torch op on IDENTICAL inputs (one pass, all layers)]")
                report["local"] = {}
                for cls, vals in sw.stats.items():
                    fr = [v[0] for v in vals]
                    ul = [v[1] for v in vals]
                    report["local"][cls] = dict(calls=len(vals), frac_diff_mean=sum(fr) / len(fr), frac_diff_max=max(fr), max_ulp=max(ul))
                    print(f"  {cls:<28s} calls {len(vals):3d}  differing elements: mean {sum(fr) / len(fr):.3e} max {max(fr):.3e}   max bf16 ulps {max(ul)}", flush=True)
                print("[bisection A: gemm=f32 engine with ONE class replaced by the torch op]")
                for cls in Swapper.CLASSES:
                    sw.torch_for = {cls}
                    t0 = time.time()
                    row(f"E32, torch {cls}", sub(engine_lp(eng, nb), nb), note=f"{scale_note} {time.time() - t0:.1f}s")
                sw.torch_for = {"attention"}
                sw.attn_p_bf16 = True
                row("E32, torch attention (bf16 P)", sub(engine_lp(eng, nb), nb), note=scale_note)
                sw.attn_p_bf16 = False
                print("[bisection B: everything torch EXCEPT one class (all torch = oracle B itself)]")
                sw.torch_for = set(Swapper.CLASSES)
                row("E32, all five torch", sub(engine_lp(eng, nb), nb), note=scale_note)
                for cls in Swapper.CLASSES:
                    sw.torch_for = set(Swapper.CLASSES) - {cls}
                    row(f"E32, only {cls} ours", sub(engine_lp(eng, nb), nb), note=scale_note)
            del eng
            torch.cuda.empty_cache()
    finally:
        sw.restore()
    report["seconds"] = time.time() - t00
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(f"wrote {args.out} ({report['seconds']:.0f}s)")


if __name__ == "__main__":
    main()
