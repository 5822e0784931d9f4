#!/usr/bin/env python3
"""Round 6 — bisection of the driver-run realistic-parity delta of round 5 (two-layer leg: engine - B = -4.1e-4 fresh / -1.05e-3 stale on 14 336 tokens).

Same weights, same rollouts (bench.realistic_parity's generator), 64 x 2048 tokens = 114 688 trained tokens, the engine's FORWARD configurations swapped
one at a time:  own GEMMs for everything ("all") | library forward GEMMs ("wgrad") | exact f32 GEMMs (gemm_dtype="f32": what is left is the hand-written
non-GEMM kernels alone) | SwiGLU epilogue off; temperature 0.25 vs 1.0.  Printed per configuration: loss delta to B and B_fa with the paired SE, the mean
log-prob shift; and — the statistic with the small SE — the PAIRED difference of the per-token loss terms between configurations.

    python tools/r6_parity_bisect.py [rollouts=64] > gpurun_out/r06_parity_bisect.json
"""
import dataclasses
import json
import math
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import grpo_oracle as O  # noqa: E402
from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches  # noqa: E402

DEV = torch.device("cuda:0")
N_ROLL = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SEQ, PROMPT, GROUP = 2048, 256, 8
dims = dataclasses.replace(ModelDims.qwen3_8b(), num_hidden_layers=2)
lcfg = O.LossConfig()
pad = math.log(1.0 / dims.vocab_size)


def make_rollouts(count, seed, temperature):
    g = torch.Generator().manual_seed(seed)
    mbs = []
    for g0 in range(0, count, GROUP):
        gs = min(GROUP, count - g0)
        rewards = torch.zeros(gs)
        rewards[torch.randperm(gs, generator=g)[: gs // 2]] = 1.0
        adv_g = rewards - rewards.mean()
        for r in range(gs):
            mask = torch.ones(1, SEQ, dtype=torch.bool)
            mask[0, :PROMPT] = False
            a = float(adv_g[r]) or 0.5
            mbs.append(dict(input_ids=torch.randint(0, dims.vocab_size, (1, SEQ), generator=g), position_ids=torch.arange(SEQ)[None], advantages=a * mask.float(),
                            inference_logprobs=torch.zeros(1, SEQ), loss_mask=mask, temperatures=torch.full((1, SEQ), temperature), seq_lens=torch.tensor([SEQ]),
                            sequence_lengths=[SEQ], env_names=["parity"] * SEQ))
    return mbs


def oracle_lp(sd, cfg, mbs, labels, mm, temperature, p_bf16=False):
    lps, labs = [], []
    with torch.no_grad():
        for i, mb in enumerate(mbs):
            h = O.qwen3_forward(sd, cfg, mb["input_ids"][0].to(DEV), mb["position_ids"][0].to(DEV), [SEQ], mm=mm, attn_p_bf16=p_bf16)
            hw = sd["lm_head.weight"]
            logits = (mm(h, hw) if mm is not O.exact_mm else h.float() @ hw.float().T).float() / temperature
            lab = logits.argmax(-1) if labels is None else labels[i]
            lp = torch.log_softmax(logits, -1).gather(-1, lab[:, None])[:, 0]
            lps.append(torch.cat([lp.new_full((1,), pad), lp[:-1]]).cpu())
            labs.append(lab)
    return torch.cat(lps), labs


def per_token(lp, inf, adv, mask):
    lp, inf, a = lp.double(), inf.double(), adv.double()
    lr = lp - inf
    pd = torch.exp(lp) - torch.exp(inf)
    invalid = torch.where(a > 0, pd > lcfg.dppo_mask_high, pd < -lcfg.dppo_mask_low) & mask
    keep = mask & ~invalid
    return -(keep * (lcfg.adv_tau * a) * torch.exp(lr)) + lcfg.kl_tau * (mask * lr**2), invalid


def engine_lp(eng, mbs, labels, chunk=8):
    out = []
    for c0 in range(0, len(mbs), chunk):
        part = mbs[c0:c0 + chunk]
        pb = pack_micro_batches(part, DEV)
        pb.labels = torch.cat(labels[c0:c0 + chunk]).to(torch.int64)
        ctx = eng.forward(pb)
        raw = ctx["logprobs"].cpu()
        out.append(torch.cat([torch.cat([raw.new_full((1,), pad), raw[i * SEQ:(i + 1) * SEQ - 1]]) for i in range(len(part))]))
        del ctx, pb
    return torch.cat(out)


def main():
    torch_mm = lambda a, w, b=None: torch.nn.functional.linear(a, w, b)  # noqa: E731
    cfg = O.TinyQwen3Config(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size, num_hidden_layers=2, num_attention_heads=dims.num_attention_heads,
                            num_key_value_heads=dims.num_key_value_heads, head_dim=dims.head_dim, vocab_size=dims.vocab_size, rms_norm_eps=dims.rms_norm_eps,
                            rope_theta=dims.rope_theta, qk_norm=dims.qk_norm, attention_bias=dims.attention_bias, rope_scaling=dims.rope_scaling)
    base = GRPOEngine(dims, device=DEV, optim=OptimConfig(type="sign_sgd"), loss=LossConfig(), seed=0)
    sd = {k: v.clone() for k, v in base.store.state_dict().items()}
    result = {"rollouts": N_ROLL, "weights": "GRPOEngine(seed=0), 2 layers at Qwen3-8B width", "temperatures": {}}
    for temperature in (0.25, 1.0):
        mbs = make_rollouts(N_ROLL, 20260924, temperature)
        mask = torch.cat([mb["loss_mask"][0] for mb in mbs])
        adv = torch.cat([mb["advantages"][0] for mb in mbs])
        n_tr = int(mask.sum())
        lp_b, labels = [], []
        lp_f, lp_c, lp_d = [], [], []
        for c0 in range(0, N_ROLL, 8):
            part = mbs[c0:c0 + 8]
            b, labs = oracle_lp(sd, cfg, part, None, O.exact_mm, temperature)
            lp_b.append(b)
            labels += labs
            lp_f.append(oracle_lp(sd, cfg, part, labs, O.exact_mm, temperature, p_bf16=True)[0])
            lp_c.append(oracle_lp(sd, cfg, part, labs, torch_mm, temperature)[0])
            lp_d.append(oracle_lp(sd, cfg, part, labs, torch_mm, temperature, p_bf16=True)[0])
        lp_b, lp_f, lp_c, lp_d = (torch.cat(x) for x in (lp_b, lp_f, lp_c, lp_d))
        lps = {"oracle_C(torch bf16 GEMMs)": lp_c, "oracle_Bfa(bf16 P)": lp_f, "oracle_D(torch bf16 GEMMs + bf16 P)": lp_d}
        configs = [("own_all", dict(own="all")), ("own_all_no_swiglu_epilogue", dict(own="all", swiglu=False)), ("library_fwd(own=wgrad)", dict(own="wgrad")),
                   ("library_all(own=none)", dict(own="none")), ("exact_f32_gemms", dict(f32=True))]
        for name, c in configs:
            os.environ["PM_GEMM_OWN"] = c.get("own", "all")
            eng = GRPOEngine(dims, device=DEV, optim=OptimConfig(type="sign_sgd"), loss=LossConfig(), seed=0, gemm_dtype="f32" if c.get("f32") else "bf16",
                             fused_swiglu_fwd=c.get("swiglu", True))
            eng.store.load_state_dict(sd)
            lps[name] = engine_lp(eng, mbs, labels)
            del eng
            torch.cuda.empty_cache()
        os.environ["PM_GEMM_OWN"] = "all"
        tres = {"trained_tokens": n_tr, "policies": {}}
        gi = torch.Generator().manual_seed(77)
        for pol, sigma in (("fresh", 0.02), ("stale", 0.3)):
            inf = lp_b + sigma * torch.randn(lp_b.shape, generator=gi)
            l_b, inv_b = per_token(lp_b, inf, adv, mask)
            l_f, _ = per_token(lp_f, inf, adv, mask)
            terms = {}
            rows = {}
            for name, lp in lps.items():
                l_e, inv_e = per_token(lp, inf, adv, mask)
                terms[name] = l_e
                d, df = (l_e - l_b)[mask], (l_e - l_f)[mask]
                rows[name] = {"loss_delta_vs_B": float(d.sum() / n_tr), "se": float(d.std() / math.sqrt(n_tr)), "loss_delta_vs_Bfa": float(df.sum() / n_tr),
                              "se_vs_Bfa": float(df.std() / math.sqrt(n_tr)), "flips_vs_B": int((inv_e != inv_b).sum()),
                              "lp_mean_shift_vs_B": float((lp - lp_b)[mask].mean()), "lp_mean_shift_se": float((lp - lp_b)[mask].std() / math.sqrt(n_tr)),
                              "lp_abs_mean_vs_B": float((lp - lp_b)[mask].abs().mean())}
            paired = {}
            ref = "own_all"
            for name in lps:
                if name == ref:
                    continue
                d = (terms[ref] - terms[name])[mask]
                paired[f"{ref} - {name}"] = {"loss_delta": float(d.sum() / n_tr), "se": float(d.std() / math.sqrt(n_tr)),
                                             "lp_mean_shift": float((lps[ref] - lps[name])[mask].mean()),
                                             "lp_mean_shift_se": float((lps[ref] - lps[name])[mask].std() / math.sqrt(n_tr))}
            tres["policies"][pol] = {"vs_oracles": rows, "paired_between_configurations": paired}
        result["temperatures"][str(temperature)] = tres
        print(f"[T={temperature}] done", file=sys.stderr, flush=True)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
