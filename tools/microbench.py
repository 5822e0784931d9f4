"""Per-kernel microbenchmarks on one MI355X (HIP events on the launch stream).  Not the graded bench (that is bench.py);
used while iterating on kernels:  python tools/microbench.py [k1|attn|norm|all]"""

from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from prime_rl_amd import ops  # noqa: E402

DEV = "cuda:0"
HBM_PEAK = 8.0e12


def timeit(fn, iters=20, warmup=3, flush_mb=0):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    flush = torch.empty(flush_mb * 1024 * 1024, dtype=torch.uint8, device=DEV) if flush_mb else None
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.zero_()  # evict L2 / Infinity Cache between iterations
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def bench_k1(out):
    n, v = 2048, 151936
    for dt in (torch.float32, torch.bfloat16):
        logits = (torch.randn(n, v, device=DEV) * 4).to(dt)
        labels = torch.randint(0, v, (n,), device=DEV)
        inv_t = torch.ones(n, device=DEV)
        med, best = timeit(lambda: ops.logprob_entropy_fwd(logits, labels, inv_t), flush_mb=512)
        nbytes = n * (logits.element_size() * v + 20)
        out.append(dict(kernel=f"logprob_entropy_fwd[{dt}]", ms=med * 1e3, best_ms=best * 1e3, GBps=nbytes / med / 1e9, frac_hbm=nbytes / med / HBM_PEAK))
        _, _, logz = ops.logprob_entropy_fwd(logits, labels, inv_t)
        g = torch.randn(n, device=DEV)
        buf = logits.clone()
        med, best = timeit(lambda: ops.logprob_bwd(buf, labels, inv_t, logz, g, out=buf), flush_mb=512)
        nbytes = n * 2 * logits.element_size() * v
        out.append(dict(kernel=f"logprob_bwd[{dt} in place]", ms=med * 1e3, best_ms=best * 1e3, GBps=nbytes / med / 1e9, frac_hbm=nbytes / med / HBM_PEAK))
        del logits, buf


def bench_k1_step_shape(out):
    """The launch the trainer step makes: 16384 rows of bf16 logits (used for the PMC HBM-traffic passes)."""
    n, v = 16384, 151936
    logits = torch.empty(n, v, device=DEV, dtype=torch.bfloat16).normal_(0, 4)
    labels = torch.randint(0, v, (n,), device=DEV)
    inv_t = torch.ones(n, device=DEV)
    med, best = timeit(lambda: ops.logprob_entropy_fwd(logits, labels, inv_t), iters=5, warmup=1)
    nbytes = n * (2 * v + 20)
    out.append(dict(kernel="logprob_entropy_fwd[bf16 16384 rows]", ms=med * 1e3, best_ms=best * 1e3, GBps=nbytes / med / 1e9, frac_hbm=nbytes / med / HBM_PEAK))


def bench_k1_after_gemm(out, rounds=5):
    """K1 right after the LM-head GEMM that produced its logits, as in the step: does WHO wrote the 5 GB of logits (hipBLASLt or the own core, and in
    which tile order / grid / store policy) change how fast K1 reads them?  HIP events around the K1 launch only; producers interleaved."""
    n, h, v = 16384, 4096, 151936
    hid = torch.randn(n, h, device=DEV).to(torch.bfloat16)
    w = (torch.randn(v, h, device=DEV) * 0.02).to(torch.bfloat16)
    logits = torch.empty(n, v, device=DEV, dtype=torch.bfloat16)
    labels = torch.randint(0, v, (n,), device=DEV)
    inv_t = torch.ones(n, device=DEV)
    producers = {"hipBLASLt": lambda: torch.mm(hid, w.t(), out=logits)}
    for var, label in ((7, "own core, persistent grid"), (3, "own core, one workgroup per tile"), (7 + 512, "own core, non-temporal stores"),
                       (7 + 1024, "own core, round-3 patch sweep order")):
        producers[label] = (lambda vv: (lambda: ops.gemm(hid, w, out=logits, variant=vv)))(var)
    scratch = torch.zeros(512 << 20, dtype=torch.uint8, device=DEV)

    def own_then_flush():
        ops.gemm(hid, w, out=logits, variant=7)
        scratch.add_(1)  # 512 MiB read + written: pushes the GEMM's last (dirty) output lines out of the 256 MiB Infinity Cache before K1 starts

    producers["own core, then a 512 MiB read-modify-write of unrelated memory"] = own_then_flush
    producers["none (logits already resident, back-to-back K1)"] = lambda: None
    # the SAME bytes read tail first: the own GEMM writes the logits column panel by column panel, so its last ~7k columns of every row are what sits (dirty) in the
    # Infinity Cache when K1 starts; a first launch over those columns reads them there, a second streams the rest from HBM, the online state carries over
    tail = int(os.environ.get("PM_K1_TAIL", "7168"))
    def k1_tail_first():
        st = ops.ChunkedLogprobState(n, torch.device(DEV))
        st.update(logits[:, v - tail:], v - tail, labels, inv_t)
        st.update(logits[:, :v - tail], 0, labels, inv_t)
        return st.finalize()

    producers["own core, persistent grid [tail first]"] = producers["own core, persistent grid"]
    producers["none [tail first]"] = lambda: None
    res = {k: [] for k in producers}
    gem = {k: [] for k in producers}
    for _ in range(rounds):
        for name, fn in producers.items():
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            fn()
            e1.record()
            if name.endswith("[tail first]"):
                k1_tail_first()
            else:
                ops.logprob_entropy_fwd(logits, labels, inv_t)
            e2.record()
            torch.cuda.synchronize()
            gem[name].append(e0.elapsed_time(e1))
            res[name].append(e1.elapsed_time(e2))
    # Where should the step's 2.5 GB zero fill of the embedding gradient go: after K1 (A: the GEMM's dirty lines are written back under K1's reads) or between
    # the GEMM and K1 (B: under the fill's writes)?  Time from the end of the GEMM to the end of both, orders interleaved.
    emb = torch.empty(151936 * 4096, dtype=torch.float32, device=DEV)
    order = {"A: GEMM, K1, fill": [], "B: GEMM, fill, K1": []}
    k1_in = {k: [] for k in order}
    for _ in range(rounds + 2):
        for name in order:
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            ops.gemm(hid, w, out=logits, variant=7)
            e0.record()
            if name.startswith("B"):
                emb.zero_()
            e1.record()
            ops.logprob_entropy_fwd(logits, labels, inv_t)
            e2.record()
            if name.startswith("A"):
                emb.zero_()
            e3.record()
            torch.cuda.synchronize()
            order[name].append(e0.elapsed_time(e3))
            k1_in[name].append(e1.elapsed_time(e2))
    nbytes = n * (2 * v + 20)
    for name in order:
        ts, ks = sorted(order[name]), sorted(k1_in[name])
        out.append(dict(kernel=f"fill placement {name}", k1_plus_fill_ms=ts[len(ts) // 2], k1_ms=ks[len(ks) // 2], k1_frac_hbm=nbytes / (ks[len(ks) // 2] * 1e-3) / HBM_PEAK))
    for name in producers:
        ts, gs = sorted(res[name]), sorted(gem[name])
        med = ts[len(ts) // 2] * 1e-3
        out.append(dict(kernel=f"K1 after: {name}", k1_ms=med * 1e3, k1_min_ms=ts[0], frac_hbm=nbytes / med / HBM_PEAK, producer_ms=gs[len(gs) // 2]))


def bench_attn(out, cases=((2048, [2048]), (16384, [2048] * 8))):
    hq, hkv, d = 32, 8, 128
    if os.environ.get("PM_ATTN_VARIANTS"):  # "kernel:variant,..." (0 forward, 1 dK/dV, 2 dQ): which build the PMC passes of tools/pmc_attn.sh look at
        from prime_rl_amd import _lib

        for kv in os.environ["PM_ATTN_VARIANTS"].split(","):
            kern, var = (int(x) for x in kv.split(":"))
            assert _lib.load().pm_attn_debug_set_variant(kern, var) == 0
    for n, seqs in cases:
        q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
        k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
        v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
        cu = torch.tensor([0] + list(torch.tensor(seqs).cumsum(0)), dtype=torch.int32, device=DEV)
        ss = ops.seq_start_from_cu(cu, n)
        med, best = timeit(lambda: ops.attn_varlen_fwd(q, k, v, ss, d**-0.5), iters=10)
        flops = sum(2 * s * s * hq * d for s in seqs)  # causal: 2 matmuls x 2 flop x s^2/2
        out.append(dict(kernel=f"attn_fwd[n={n}]", ms=med * 1e3, TFLOPs=flops / med / 1e12))
        o, lse = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
        go = torch.randn_like(o)
        med, best = timeit(lambda: ops.attn_varlen_bwd(go, q, k, v, o, lse, ss, d**-0.5), iters=10)
        out.append(dict(kernel=f"attn_bwd[n={n}]", ms=med * 1e3, TFLOPs=2.5 * flops / med / 1e12))


def bench_norm(out):
    n, h = 16384, 4096
    x = torch.randn(n, h, device=DEV).to(torch.bfloat16)
    w = torch.ones(h, device=DEV, dtype=torch.bfloat16)
    med, _ = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6), flush_mb=512)
    out.append(dict(kernel="rmsnorm_fwd[16384x4096]", ms=med * 1e3, GBps=n * h * 4 / med / 1e9))
    y, rstd, _ = ops.rmsnorm_fwd(x, w, 1e-6)
    gw = torch.zeros(h, device=DEV)
    med, _ = timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, gw), flush_mb=512)
    out.append(dict(kernel="rmsnorm_bwd[16384x4096]", ms=med * 1e3, GBps=n * h * 6 / med / 1e9))
    dres = torch.randn(n, h, device=DEV).to(torch.bfloat16)
    med, _ = timeit(lambda: ops.rmsnorm_bwd(y, x, w, rstd, gw, dresidual=dres), flush_mb=512)
    out.append(dict(kernel="rmsnorm_bwd+dres[16384x4096]", ms=med * 1e3, GBps=n * h * 8 / med / 1e9))
    for r, c in ((16384, 4096), (16384, 12288), (24576, 4096), (151936, 4096)):
        s = torch.randn(r, c, device=DEV).to(torch.bfloat16)
        d = torch.empty(c, r, dtype=torch.bfloat16, device=DEV)
        med, _ = timeit(lambda: ops.transpose_2d(s, out=d), flush_mb=512)
        out.append(dict(kernel=f"transpose_2d[{r}x{c}]", ms=med * 1e3, GBps=r * c * 4 / med / 1e9))
        del s, d
    gu = torch.randn(n, 2 * 12288, device=DEV).to(torch.bfloat16)
    med, _ = timeit(lambda: ops.swiglu_fwd(gu), flush_mb=512)
    out.append(dict(kernel="swiglu_fwd[16384x12288]", ms=med * 1e3, GBps=n * 12288 * 6 / med / 1e9))
    dact = torch.randn(n, 12288, device=DEV).to(torch.bfloat16)
    med, _ = timeit(lambda: ops.swiglu_bwd(dact, gu), flush_mb=512)
    out.append(dict(kernel="swiglu_bwd[16384x12288]", ms=med * 1e3, GBps=n * 12288 * 10 / med / 1e9))
    del dact
    a = torch.randn(n, 4096, device=DEV).to(torch.bfloat16)
    wt = torch.randn(24576, 4096, device=DEV).to(torch.bfloat16)
    med, _ = timeit(lambda: a @ wt.T, iters=10)
    out.append(dict(kernel="hipBLASLt gemm 16384x4096x24576 bf16", ms=med * 1e3, TFLOPs=2 * n * 4096 * 24576 / med / 1e12))
    a2 = a[:2048]
    med, _ = timeit(lambda: a2 @ wt.T, iters=10)
    out.append(dict(kernel="hipBLASLt gemm 2048x4096x24576 bf16", ms=med * 1e3, TFLOPs=2 * 2048 * 4096 * 24576 / med / 1e12))


def bench_gemm(out):
    """Operand-orientation study for the library GEMMs of one Qwen3-8B layer at T = 16384 tokens (what the step runs): which layout
    of W / dY / X does hipBLASLt run fastest for forward, dgrad and the f32-accumulating wgrad."""
    import os

    if os.environ.get("PM_BLAS"):  # "cublas" = rocBLAS, "cublaslt" = hipBLASLt (torch's names)
        torch.backends.cuda.preferred_blas_library(os.environ["PM_BLAS"])
    t = 16384
    shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (24576, 4096), "down": (4096, 12288)}
    for name, (o, i) in shapes.items():
        x = torch.randn(t, i, device=DEV).to(torch.bfloat16)
        dy = torch.randn(t, o, device=DEV).to(torch.bfloat16)
        w = (torch.randn(o, i, device=DEV) * 0.02).to(torch.bfloat16)
        wt = w.t().contiguous()
        g = torch.zeros(o, i, dtype=torch.float32, device=DEV)
        gt = torch.zeros(i, o, dtype=torch.float32, device=DEV)
        gb = torch.zeros(o, i, dtype=torch.bfloat16, device=DEV)
        dyt, xt = dy.t().contiguous(), x.t().contiguous()
        y, dx = torch.empty(t, o, dtype=torch.bfloat16, device=DEV), torch.empty(t, i, dtype=torch.bfloat16, device=DEV)
        fl = 2 * t * o * i
        variants = {
            "fwd  x @ W.T        (W [out,in])": lambda: torch.mm(x, w.t(), out=y),
            "fwd  x @ Wt         (Wt [in,out])": lambda: torch.mm(x, wt, out=y),
            "dgrad dy @ W        (W [out,in])": lambda: torch.mm(dy, w, out=dx),
            "dgrad dy @ Wt.T     (Wt [in,out])": lambda: torch.mm(dy, wt.t(), out=dx),
            "wgrad dy.T @ x -> f32 [out,in]   (current)": lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=g),
            "wgrad addmm dy.T @ x -> f32      (accumulate)": lambda: torch.addmm(g, dy.t(), x, out_dtype=torch.float32, out=g),
            "wgrad dyt @ x -> f32             (dy pre-transposed)": lambda: torch.mm(dyt, x, out_dtype=torch.float32, out=g),
            "wgrad dyt @ xt.T -> f32          (both K-contiguous)": lambda: torch.mm(dyt, xt.t(), out_dtype=torch.float32, out=g),
            "wgrad dy.T @ xt.T -> f32         (x pre-transposed)": lambda: torch.mm(dy.t(), xt.t(), out_dtype=torch.float32, out=g),
            "wgrad x.T @ dy -> f32 [in,out]   (transposed grad)": lambda: torch.mm(x.t(), dy, out_dtype=torch.float32, out=gt),
            "wgrad dy.T @ x -> bf16": lambda: torch.mm(dy.t(), x, out=gb),
            "transpose dy -> dyt": lambda: dyt.copy_(dy.t()),
            "transpose x -> xt": lambda: xt.copy_(x.t()),
        }
        for label, fn in variants.items():
            try:
                med, best = timeit(fn, iters=8, warmup=2)
            except Exception as ex:  # a layout the library refuses
                out.append(dict(kernel=f"gemm {name} {label}", error=str(ex)[:80]))
                continue
            rec = dict(kernel=f"gemm {name} {label}", ms=med * 1e3)
            if not label.startswith("transpose"):
                rec["TFLOPs"] = fl / med / 1e12
            out.append(rec)
        del x, dy, w, wt, g, gt, gb, dyt, xt, y, dx
        torch.cuda.empty_cache()


def bench_gemm_core(out, t=16384, variants=(7, 39, 3), rounds=int(os.environ.get("PM_GEMM_ROUNDS", "3"))):  # 39 = 7 without the split-K of a half-empty last round; 68 = the one-wave-per-SIMD kernel
    if os.environ.get("PM_GEMM_VARIANTS"):
        variants = tuple(int(v) for v in os.environ["PM_GEMM_VARIANTS"].split(","))
    """The hand-written MFMA GEMM core (csrc/gemm.hip) against the library call it replaces, per Qwen3-8B projection and layout, interleaved in one
    process (boxes differ by a few %).  Random operands (guide §5.4 rule 25: zero-filled operands clock ~15 % higher)."""
    shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (24576, 4096), "down": (4096, 12288)}
    if os.environ.get("PM_GEMM_HEAD", "1") == "1":
        shapes["head"] = (151936, 4096)
    for name, (o, i) in shapes.items():
        x = torch.randn(t, i, device=DEV).to(torch.bfloat16)
        dy = torch.randn(t, o, device=DEV).to(torch.bfloat16)
        w = (torch.randn(o, i, device=DEV) * 0.02).to(torch.bfloat16)
        wt = w.t().contiguous()
        y, dx = torch.empty(t, o, dtype=torch.bfloat16, device=DEV), torch.empty(t, i, dtype=torch.bfloat16, device=DEV)
        g = torch.zeros(o, i, dtype=torch.float32, device=DEV)
        fl = 2.0 * t * o * i
        cases = {
            "fwd": (lambda: torch.mm(x, w.t(), out=y), lambda: ops.gemm(x, w, out=y)),
            "dgrad": (lambda: torch.mm(dy, wt.t(), out=dx), lambda: ops.gemm(dy, w.t(), out=dx)),
            "wgrad": (lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=g), lambda: ops.gemm(dy.t(), x.t(), out=g)),
        }
        for cname, (lib_fn, own_fn) in cases.items():
            # correctness on the spot (sampled rows: the f32 reference of the head shape would be 10 GB)
            own_fn()
            idx = torch.randint(0, t if cname != "wgrad" else o, (64,), device=DEV)
            if cname == "fwd":
                ref, got = x[idx].float() @ w.float().t(), y[idx].float()
            elif cname == "dgrad":
                ref, got = dy[idx].float() @ w.float(), dx[idx].float()
            else:
                ref, got = dy[:, idx].float().t() @ x.float(), g[idx]
            rel = ((got - ref).abs().max() / (ref.abs().max() + 1e-9)).item()
            res = {"lib": []}
            base = (y if cname == "fwd" else dx if cname == "dgrad" else g).clone()
            same = {}
            for v in variants:
                res[f"v{v}"] = []
                ops.gemm_set_variant(v)
                own_fn()
                same[f"v{v}"] = bool(torch.equal(y if cname == "fwd" else dx if cname == "dgrad" else g, base))
            del base
            for _ in range(rounds):
                med, _ = timeit(lib_fn, iters=5, warmup=1)
                res["lib"].append(med)
                for v in variants:
                    ops.gemm_set_variant(v)
                    med, _ = timeit(own_fn, iters=5, warmup=1)
                    res[f"v{v}"].append(med)
            ops.gemm_set_variant(7)
            rec = dict(kernel=f"gemm_core {name} {cname} t={t}", max_rel_err=rel, bit_identical_to_default=same)
            for key, ts in res.items():
                ts.sort()
                rec[key + "_ms"] = ts[len(ts) // 2] * 1e3
                rec[key + "_TF"] = fl / ts[len(ts) // 2] / 1e12
            print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
        del x, dy, w, wt, y, dx, g
        torch.cuda.empty_cache()


def bench_lmhead(out, n=16384, h=4096, v=151936):
    """J1 (fused MFMA LM head) against the library path it can replace: hipBLASLt GEMM -> bf16 logits in HBM -> K1."""
    hidden = torch.randn(n, h, device=DEV).to(torch.bfloat16)
    weight = (torch.randn(v, h, device=DEV) * 0.02).to(torch.bfloat16)
    labels = torch.randint(0, v, (n,), device=DEV)
    inv_t = torch.ones(n, device=DEV)
    flop = 2.0 * n * v * h
    med, best = timeit(lambda: ops.lmhead_logprob_fwd(hidden, weight, labels, inv_t), iters=8, warmup=2)
    out.append(dict(kernel=f"J1 lmhead_logprob_fwd n={n}", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    store = torch.empty(n, v, dtype=torch.bfloat16, device=DEV)
    med, best = timeit(lambda: ops.lmhead_logprob_fwd(hidden, weight, labels, inv_t, logits_out=store), iters=8, warmup=2)
    out.append(dict(kernel="J1 fwd + logits store", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    _, _, logz = ops.lmhead_logprob_fwd(hidden, weight, labels, inv_t)
    g = torch.randn(n, device=DEV)
    med, best = timeit(lambda: ops.lmhead_logprob_bwd(hidden, weight, labels, inv_t, logz, g, out=store), iters=8, warmup=2)
    out.append(dict(kernel="J1 lmhead_logprob_bwd (recompute + dlogits)", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))

    def lib():
        torch.mm(hidden, weight.t(), out=store)
        return ops.logprob_entropy_fwd(store, labels, inv_t)

    med, best = timeit(lib, iters=8, warmup=2)
    out.append(dict(kernel="library: hipBLASLt logits + K1", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    def own():
        ops.gemm(hidden, weight, out=store)
        return ops.logprob_entropy_fwd(store, labels, inv_t)

    med, best = timeit(own, iters=8, warmup=2)
    out.append(dict(kernel="stored logits: own GEMM core + K1 (the step's default)", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    med, best = timeit(lambda: ops.gemm(hidden, weight, out=store), iters=8, warmup=2)
    out.append(dict(kernel="own GEMM core, logits only", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    med, best = timeit(lambda: torch.mm(hidden, weight.t(), out=store), iters=8, warmup=2)
    out.append(dict(kernel="library: hipBLASLt logits only", ms=med * 1e3, best_ms=best * 1e3, TFLOPs=flop / med / 1e12, frac_mfma=flop / med / 2.5e15))
    med, best = timeit(lambda: ops.logprob_bwd(store, labels, inv_t, logz, g, out=store), iters=8, warmup=2)
    out.append(dict(kernel="library path: K1 bwd in place", ms=med * 1e3, best_ms=best * 1e3))


def bench_attn_variants(out, kernel, variants, n=16384, seqs=(2048,) * 8, rounds=5):
    """In-process interleaved A/B of attention kernel variants (boxes differ by ~5 %, so variants must share a process)."""
    from prime_rl_amd import _lib

    lib = _lib.load()
    hq, hkv, d = 32, 8, 128
    q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
    k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(torch.tensor(seqs).cumsum(0)), dtype=torch.int32, device=DEV)
    ss = ops.seq_start_from_cu(cu, n)
    o, lse = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
    go = torch.randn_like(o)
    fn = (lambda: ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)) if kernel == 0 else (lambda: ops.attn_varlen_bwd(go, q, k, v, o, lse, ss, d**-0.5))
    ref = None
    diffs = {}
    times = {var: [] for var in variants}
    for _ in range(rounds):
        for var in variants:
            lib.pm_attn_debug_set_variant(kernel, var)
            res = fn()
            if ref is None:
                ref = [t.clone() for t in res]
            elif _ == 0:
                diffs[var] = max(float((a.float() - b.float()).abs().max()) for a, b in zip(res, ref))
                for a, b in zip(res, ref):
                    assert os.environ.get("PM_PROBE") or torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2), f"variant {var} differs by {diffs[var]}"
            med, best = timeit(fn, iters=10, warmup=2)
            times[var].append(med)
    lib.pm_attn_debug_set_variant(kernel, -1)
    flops = sum(2 * s * s * hq * d for s in seqs) * (1.0 if kernel == 0 else 2.5)
    for var in variants:
        ts = sorted(times[var])
        out.append(dict(kernel=f"attn_{('fwd', 'bwd[dkdv]', 'bwd[dq]')[kernel]} variant {var}", median_ms=ts[len(ts) // 2] * 1e3, min_ms=ts[0] * 1e3,
                        TFLOPs=flops / ts[len(ts) // 2] / 1e12, max_abs_diff_vs_first=diffs.get(var, 0.0)))


def bench_moe(out, n=16384, e=128, k=8, h=2048, hidden=768):
    """Qwen3-30B-A3B layer shapes: routing / permutation kernels, and the expert GEMMs as torch._grouped_mm vs the per-expert loop."""
    g = torch.Generator(device="cpu").manual_seed(0)
    logits = (torch.randn(n, e, generator=g) * 2).to(torch.bfloat16).to(DEV)
    x = torch.randn(n, h, generator=g).to(torch.bfloat16).to(DEV)
    med, _ = timeit(lambda: ops.moe_router_fwd(logits, k, "softmax", True, 1.0, want_counts=False), iters=20)
    out.append(dict(kernel=f"moe_router_fwd[n={n},E={e},k={k}]", us=med * 1e6, GBs=(n * e * (2 + 4) + n * k * 8) / med / 1e9))
    scores, top, idx, counts, conf = ops.moe_router_fwd(logits, k, "softmax", True, 1.0)
    med, _ = timeit(lambda: ops.moe_sort(idx.reshape(-1), e), iters=20)
    out.append(dict(kernel=f"moe_sort[{n * k} slots] (4 launches)", us=med * 1e6))
    dest, slot, offsets = ops.moe_sort(idx.reshape(-1), e)
    med, _ = timeit(lambda: ops.moe_gather_rows(x, slot, k), iters=20)
    out.append(dict(kernel=f"moe_gather_rows[{n * k} x {h}]", us=med * 1e6, GBs=2 * n * k * h * 2 / med / 1e9))
    y = ops.moe_gather_rows(x, slot, k)
    med, _ = timeit(lambda: ops.moe_combine(y, dest, k, top.reshape(-1)), iters=20)
    out.append(dict(kernel=f"moe_combine[{n} x {k} x {h}]", us=med * 1e6, GBs=(n * k * h * 2 + n * h * 2) / med / 1e9))
    w1 = (torch.randn(e, hidden, h, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    w3 = (torch.randn(e, hidden, h, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    w2 = (torch.randn(e, h, hidden, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    from prime_rl_amd.trainer.models.layers.moe import _experts_for_loop, _experts_grouped, _experts_grouped_hip

    flops = 3 * 2 * n * k * h * hidden
    med, _ = timeit(lambda: _experts_for_loop(w1, w2, w3, y, offsets), iters=5, warmup=1)
    out.append(dict(kernel="experts per-expert loop (3 x 128 library GEMMs + host sync)", ms=med * 1e3, TFLOPs=flops / med / 1e12))
    try:
        ref = _experts_for_loop(w1, w2, w3, y, offsets)
        got = _experts_grouped(w1, w2, w3, y, offsets)
        err = float((got.float() - ref.float()).abs().max())
        med, _ = timeit(lambda: _experts_grouped(w1, w2, w3, y, offsets), iters=5, warmup=1)
        out.append(dict(kernel="experts torch._grouped_mm (3 grouped GEMMs)", ms=med * 1e3, TFLOPs=flops / med / 1e12, max_abs_diff_vs_loop=err))
    except Exception as ex:  # noqa: BLE001
        out.append(dict(kernel="experts torch._grouped_mm", error=repr(ex)[:200]))
    with torch.no_grad():
        got = _experts_grouped_hip(w1, w2, w3, y, offsets)
        err = float((got.float() - ref.float()).abs().max())
        med, _ = timeit(lambda: _experts_grouped_hip(w1, w2, w3, y, offsets), iters=10, warmup=2)
    out.append(dict(kernel="experts pm_grouped_gemm_nt (gate|up in one launch) + pm_swiglu_fwd + pm_grouped_gemm_nt", ms=med * 1e3, TFLOPs=flops / med / 1e12,
                    max_abs_diff_vs_loop=err))
    from prime_rl_amd import _lib as _l

    for variant, label in ((1, "128 x 128 tiles"), (2, "256 x 256 tiles, round-2 kernel"), (0, "256 x 256 tiles, ping-pong core")):
        gsched = ((variant << 16) | 7) if variant else None
        ts = []
        for _ in range(3):
            med, _ = timeit(lambda: ops.grouped_gemm_nt(y, w1, offsets, variant=gsched), iters=10, warmup=2)
            ts.append(med)
        med = sorted(ts)[1]
        out.append(dict(kernel=f"pm_grouped_gemm_nt [{n * k} x {h}] x [{e} x {hidden} x {h}] ({label})", ms=med * 1e3, TFLOPs=2 * n * k * h * hidden / med / 1e12))
    hmid = ops.grouped_gemm_nt(y, w1, offsets)
    for variant, label in ((1, "128 x 128 tiles"), (2, "256 x 256 tiles, round-2 kernel"), (0, "256 x 256 tiles, ping-pong core")):
        gsched = ((variant << 16) | 7) if variant else None
        med, _ = timeit(lambda: ops.grouped_gemm_nn(hmid, w1, offsets, variant=gsched), iters=10, warmup=2)
        out.append(dict(kernel=f"pm_grouped_gemm_nn (row gradient of the same; {label})", ms=med * 1e3, TFLOPs=2 * n * k * h * hidden / med / 1e12))
        med, _ = timeit(lambda: ops.grouped_gemm_tn(hmid, y, offsets, variant=gsched), iters=10, warmup=2)
        out.append(dict(kernel=f"pm_grouped_gemm_tn (weight gradient of the same; {label})", ms=med * 1e3, TFLOPs=2 * n * k * h * hidden / med / 1e12))


def bench_moe_gemm_variants(out, variants=(7, 32773), n=16384, e=128, k=8, h=2048, hidden=768, rounds=5):
    """The three grouped products of a Qwen3-30B-A3B expert layer on the gemm.hip core (units of its tile stream) under the dense core's K-loop schedules,
    interleaved in one process; results must be bit-identical between schedules."""
    if os.environ.get("PM_GEMM_VARIANTS"):
        variants = tuple(int(v) for v in os.environ["PM_GEMM_VARIANTS"].split(","))
    g = torch.Generator(device="cpu").manual_seed(0)
    logits = (torch.randn(n, e, generator=g) * 2).to(torch.bfloat16).to(DEV)
    x = torch.randn(n, h, generator=g).to(torch.bfloat16).to(DEV)
    scores, top, idx, counts, conf = ops.moe_router_fwd(logits, k, "softmax", True, 1.0)
    dest, slot, offsets = ops.moe_sort(idx.reshape(-1), e)
    y = ops.moe_gather_rows(x, slot, k)
    w1 = (torch.randn(e, hidden, h, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    w3 = (torch.randn(e, hidden, h, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    hmid = ops.grouped_gemm_nt(y, w1, offsets)
    dgu = torch.randn(n * k, 2 * hidden, generator=g).to(torch.bfloat16).to(DEV)
    cases = {
        "nt (forward, gate|up)": (lambda: ops.grouped_gemm_nt(y, w1, offsets, w2=w3), 2.0 * n * k * h * 2 * hidden),
        "nn (row gradient, gate|up chained)": (lambda: ops.grouped_gemm_nn(dgu, w1, offsets, w2=w3), 2.0 * n * k * h * 2 * hidden),
        "tn (weight gradient)": (lambda: ops.grouped_gemm_tn(hmid, y, offsets), 2.0 * n * k * h * hidden),
    }
    for name, (fn, fl) in cases.items():
        res, same, base = {v: [] for v in variants}, {}, None
        try:
            for v in variants:
                ops.gemm_set_variant(v)
                r = fn()
                r = r[0] if isinstance(r, (tuple, list)) else r
                base = r.clone() if base is None else base
                same[v] = bool(torch.equal(r, base))
            for _ in range(rounds):
                for v in variants:
                    ops.gemm_set_variant(v)
                    med, _ = timeit(fn, iters=5, warmup=1)
                    res[v].append(med)
        finally:
            ops.gemm_set_variant(7)
        rec = dict(kernel=f"grouped {name}", bit_identical=same)
        for v in variants:
            ts = sorted(res[v])
            rec[f"v{v}_ms"] = ts[len(ts) // 2] * 1e3
            rec[f"v{v}_TF"] = fl / ts[len(ts) // 2] / 1e12
        out.append(rec)


def bench_moe_layer(out, n=16384, e=128, k=8, h=2048, hidden=768, iters=int(os.environ.get("PM_MOE_ITERS", "5"))):
    """One Qwen3-30B-A3B MoE layer, forward + backward (for rocprofv3 --kernel-trace)."""
    from prime_rl_amd.trainer.models.layers.moe import MoE, MoEArgs

    layer = MoE(MoEArgs(num_experts=e, num_shared_experts=0, score_func="softmax", route_norm=True, score_before_experts=False, top_k=k, use_grouped_mm=True,
                        load_balance_coeff=None), h, hidden).to(DEV).to(torch.bfloat16)
    layer.init_weights(0.02)
    x = torch.randn(1, n, h, device=DEV).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(1, n, h, device=DEV).to(torch.bfloat16)

    def fb():
        x.grad = None
        for p_ in layer.parameters():  # no `.grad +=` passes: a trainer's expert gradients go straight into the f32 main-grad buffer
            p_.grad = None
        layer(x).backward(gy)

    med, _ = timeit(fb, iters=iters, warmup=2)
    ex = layer.experts
    out.append(dict(kernel=f"MoE layer fwd+bwd [{n} tokens, {e} experts top-{k}, {h}/{hidden}]", ms=med * 1e3,
                    fused_gate_up_forward=bool(ops.grouped_gemm_nt_swiglu_supported(x.detach().reshape(-1, h), ex.w1, ex.w3)),
                    w3_minus_w1_bytes=ex.w3.data_ptr() - ex.w1.data_ptr()))


def trace_mark():
    """One launch of the 16 x 16 x 32 MFMA probe kernel: the marker `tools/rocprof_summary.py --window` cuts a kernel trace at."""
    a = torch.zeros(16, 32, dtype=torch.bfloat16, device=DEV)
    ops.mfma_probe(a, a)


def bench_moe_step(out, layers=8, n_mb=8, seq=2048, steps=3):
    """The module-engine trainer step (trainer/module_engine.py) on a Qwen3-30B-A3B-width model cut to `layers` layers (the full 48 need the
    expert-parallel node: 30.5 B parameters x 18 B of f32 state): forward + loss + backward + clip + fused AdamW on one GPU."""
    from prime_rl_amd.trainer.engine import OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.models.layers.lm_head import inject_prime_lm_head
    from prime_rl_amd.trainer.models.qwen3_moe import Qwen3MoeConfig, Qwen3MoeForCausalLM
    from prime_rl_amd.trainer.module_engine import ModuleEngine
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    cfg = Qwen3MoeConfig(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=4,
                         num_experts=128, num_experts_per_tok=8, moe_intermediate_size=768, norm_topk_prob=True)
    with torch.device(DEV):
        model = Qwen3MoeForCausalLM(cfg)
    model = model.to(torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.02).to(p.dtype))
    inject_prime_lm_head(model, chunk_size=8192)
    eng = ModuleEngine(model, DEV, optim=OptimConfig())
    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=n_mb), seq_len=seq)
    batches = [[pack_micro_batches(dl.get_batch(), DEV)] for _ in range(steps + 1)]
    eng.step(batches[0])
    torch.cuda.synchronize()
    trace_mark()
    t0 = time.perf_counter()
    for b in batches[1:]:
        eng.step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    trace_mark()
    tok = n_mb * seq
    flops = eng.dims.flops_per_token(seq) * tok
    out.append(dict(kernel=f"module-engine step, Qwen3-30B-A3B width, {layers} of 48 layers, {n_mb} x {seq} tokens", ms=dt * 1e3, tokens_per_s=tok / dt,
                    samples_per_s=n_mb / dt, mfu_pct=100 * flops / dt / 2.5e15, params_B=sum(p.numel() for p in model.parameters()) / 1e9,
                    peak_mem_GiB=torch.cuda.max_memory_allocated() / 2**30))


def bench_step_32k(out, seq=32768, rollouts=1, steps=2, recompute="layer"):
    """BASELINE.json configs[4] on one GPU: the Qwen3-8B policy-update step on `rollouts` x 32 768-token documents with block-level recomputation (the
    engine configuration of bench.py's extra.long_context), bracketed by trace markers for `rocprof_summary.py --window` (profiles/r06_bench_32k.md)."""
    from prime_rl_amd.trainer.engine import GRPOEngine, LossConfig, ModelDims, OptimConfig, pack_micro_batches
    from prime_rl_amd.trainer.rl.data import FakeDataLoader, FakeDataLoaderConfig

    dims = ModelDims.qwen3_8b()
    eng = GRPOEngine(dims, device=torch.device(DEV), optim=OptimConfig(), loss=LossConfig(), seed=0, recompute=recompute)
    dl = FakeDataLoader(FakeDataLoaderConfig(batch_size=rollouts), seq_len=seq)
    bs = [[pack_micro_batches([mb], torch.device(DEV)) for mb in dl.get_batch()] for _ in range(steps + 1)]
    eng.step(bs[0])
    torch.cuda.synchronize()
    trace_mark()
    t0 = time.perf_counter()
    for b in bs[1:]:
        eng.step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    trace_mark()
    out.append(dict(kernel=f"GRPO step, Qwen3-8B, {rollouts} x {seq} tokens, recompute={recompute}", ms=dt * 1e3, tokens_per_s=rollouts * seq / dt,
                    mfu_ref_formula=dims.flops_per_token(seq) * rollouts * seq / dt / 2.5e15, peak_mem_GiB=torch.cuda.max_memory_allocated() / 2**30))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    out = []
    if which == "step32k":
        bench_step_32k(out, rollouts=int(sys.argv[2]) if len(sys.argv) > 2 else 1, recompute=sys.argv[3] if len(sys.argv) > 3 else "layer")
    if which == "moestep":
        bench_moe_step(out, layers=int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    if which == "attnfwd":
        docs = int(sys.argv[3]) if len(sys.argv) > 3 else 8  # 16 384 tokens as `docs` equal documents
        bench_attn_variants(out, 0, [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 0, 1, 2], seqs=(16384 // docs,) * docs)
    if which == "attnprobe":  # phase timers of the ping-pong forward kernel (variants 64 + 32 + ablation bits): cycles per kv tile of block 0's waves
        from prime_rl_amd import _lib

        lib = _lib.load()
        n, hq, hkv, d = 16384, 32, 8, 128
        docs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
        q = torch.randn(n, hq, d, device=DEV).to(torch.bfloat16)
        k = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
        v = torch.randn(n, hkv, d, device=DEV).to(torch.bfloat16)
        cu = torch.tensor([0] + [n // docs * (i + 1) for i in range(docs)], dtype=torch.int32, device=DEV)
        ss = ops.seq_start_from_cu(cu, n)
        for var in [int(x) for x in sys.argv[2].split(",")]:
            lib.pm_attn_debug_set_variant(0, var)
            for _ in range(3):
                o, lse = ops.attn_varlen_fwd(q, k, v, ss, d**-0.5)
            torch.cuda.synchronize()
            for g in range(2):
                for wl in range(4):
                    r = lse[g, n - 128 + wl * 32:n - 128 + wl * 32 + 7].tolist()
                    it = max(r[6], 1)
                    out.append(dict(variant=var, wave=g * 4 + wl, tiles=int(it), L=round(r[0] / it), C1=round(r[1] / it), V=round(r[2] / it), C2=round(r[3] / it),
                                    per_tile=round(r[4] / it), total_cyc=int(r[4]), realtime_ticks=int(r[5])))
        lib.pm_attn_debug_set_variant(0, -1)
    if which == "attnbwd":
        bench_attn_variants(out, 1, [int(x) for x in sys.argv[2].split(",")])
    if which == "attndq":
        bench_attn_variants(out, 2, [int(x) for x in sys.argv[2].split(",")])
    if which == "moe":
        bench_moe(out)
    if which == "gemmswiglu":  # the gate|up forward of a Qwen3-8B layer: library GEMM / own GEMM + pm_swiglu_fwd against the fused epilogue
        t, inter, k = int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 12288, 4096
        x = torch.randn(t, k, device=DEV).to(torch.bfloat16)
        w = (torch.randn(2 * inter, k, device=DEV) * 0.02).to(torch.bfloat16)
        gu = torch.empty(t, 2 * inter, dtype=torch.bfloat16, device=DEV)
        act = torch.empty(t, inter, dtype=torch.bfloat16, device=DEV)
        cases = {"library GEMM + pm_swiglu_fwd": lambda: ops.swiglu_fwd(torch.mm(x, w.t(), out=gu), out=act),
                 "own GEMM + pm_swiglu_fwd": lambda: ops.swiglu_fwd(ops.gemm(x, w, out=gu), out=act),
                 "own GEMM alone": lambda: ops.gemm(x, w, out=gu), "library GEMM alone": lambda: torch.mm(x, w.t(), out=gu),
                 "pm_swiglu_fwd alone": lambda: ops.swiglu_fwd(gu, out=act),
                 "pm_gemm_swiglu_bf16 (fused epilogue)": lambda: ops.gemm_swiglu(x, w, gate_up=gu, act=act)}
        ref_act = ops.swiglu_fwd(ops.gemm(x, w)).clone()
        ops.gemm_swiglu(x, w, gate_up=gu, act=act)
        same = bool(torch.equal(act, ref_act))
        res = {k2: [] for k2 in cases}
        for _ in range(5):
            for name, fn in cases.items():
                med, _ = timeit(fn, iters=5, warmup=1)
                res[name].append(med)
        for name, ts in res.items():
            ts.sort()
            out.append(dict(kernel=f"gate|up forward t={t}: {name}", ms=ts[len(ts) // 2] * 1e3, bit_identical_act=same))
    if which == "gemmswiglubwd":  # the down projection's row gradient of a Qwen3-8B layer + the SwiGLU backward: two launches against the fused epilogue
        t, inter, k = int(sys.argv[2]) if len(sys.argv) > 2 else 16384, 12288, 4096
        dy = torch.randn(t, k, device=DEV).to(torch.bfloat16)
        wd = (torch.randn(k, inter, device=DEV) * 0.02).to(torch.bfloat16)
        gu = torch.randn(t, 2 * inter, device=DEV).to(torch.bfloat16)
        dact = torch.empty(t, inter, dtype=torch.bfloat16, device=DEV)
        dgu = torch.empty(t, 2 * inter, dtype=torch.bfloat16, device=DEV)
        wdt = wd.t().contiguous()
        cases = {"library GEMM (K-contiguous weight copy) + pm_swiglu_bwd": lambda: ops.swiglu_bwd(torch.mm(dy, wdt.t(), out=dact), gu, out=dgu),
                 "own GEMM + pm_swiglu_bwd": lambda: ops.swiglu_bwd(ops.gemm(dy, wd.t(), out=dact), gu, out=dgu),
                 "own GEMM alone": lambda: ops.gemm(dy, wd.t(), out=dact), "pm_swiglu_bwd alone": lambda: ops.swiglu_bwd(dact, gu, out=dgu),
                 "pm_gemm_swiglu_bwd_bf16 (fused epilogue)": lambda: ops.gemm_swiglu_bwd(dy, wd, gu, out=dgu)}
        ref = ops.swiglu_bwd(ops.gemm(dy, wd.t()), gu).clone()
        same = bool(torch.equal(ops.gemm_swiglu_bwd(dy, wd, gu, out=dgu), ref))
        res = {k2: [] for k2 in cases}
        for _ in range(5):
            for name, fn in cases.items():
                med, _ = timeit(fn, iters=5, warmup=1)
                res[name].append(med)
        for name, ts in res.items():
            ts.sort()
            out.append(dict(kernel=f"down-projection row gradient + SwiGLU backward t={t}: {name}", ms=ts[len(ts) // 2] * 1e3, bit_identical=same))
    if which == "moegemm":
        bench_moe_gemm_variants(out)
    if which == "moelayer":
        bench_moe_layer(out)
    if which == "lmhead":
        bench_lmhead(out)
    if which == "lmhead4k":
        bench_lmhead(out, n=4096)
    if which == "gemm":
        bench_gemm(out)
    if which == "gemmprobe":  # a few launches of each layout of the MFMA GEMM core at the gate|up shape (for tools/pmc.sh)
        t, o, i = 16384, 24576, 4096
        x = torch.randn(t, i, device=DEV).to(torch.bfloat16)
        dy = torch.randn(t, o, device=DEV).to(torch.bfloat16)
        w = (torch.randn(o, i, device=DEV) * 0.02).to(torch.bfloat16)
        g = torch.zeros(o, i, dtype=torch.float32, device=DEV)
        ops.gemm_set_variant(int(os.environ.get("PM_GEMM_VARIANT", "7")))
        wt = w.t().contiguous()
        if os.environ.get("PM_PROBE_ZERO") == "1":  # DVFS check (guide: zero-filled operands clock ~15-19 % higher)
            x.zero_(); dy.zero_(); w.zero_(); wt.zero_()
        if os.environ.get("PM_PROBE_TIME") == "1":
            def own_v(v):
                def f():
                    ops.gemm_set_variant(v)
                    ops.gemm(x, w)
                    ops.gemm_set_variant(7)
                return f
            cases = [("own fwd", lambda: ops.gemm(x, w)), ("lib fwd", lambda: torch.mm(x, w.t())), ("own wgrad", lambda: ops.gemm(dy.t(), x.t(), out=g)),
                     ("lib wgrad", lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=g))]
            cases += [(f"own fwd variant {v} ({d})", own_v(v)) for v, d in ((1, "one WG per tile"), (4, "no setprio"), (7, "balanced reads"), (13, "ABLATION no DMA"), (21, "ABLATION no reads"), (29, "ABLATION no DMA, no reads"),
                                                                             (68, "4-wave kernel, persistent"), (64, "4-wave kernel, one WG per tile"), (76, "4-wave ABLATION no DMA"),
                                                                             (84, "4-wave ABLATION no reads"), (92, "4-wave ABLATION no DMA, no reads"),
                                                                             (132, "4-wave kernel, register staging"), (260, "4-wave kernel, fine schedule"), (268, "4-wave fine ABLATION no DMA"),
                                                                             (276, "4-wave fine ABLATION no reads"), (7 + 512, "8-wave core, nt result stores"), (260 + 512, "4-wave fine, nt result stores"),
                                                                             (7 + 2048, "8-wave, DMA piece 0 before the reads"), (7 + 4096, "8-wave, both DMA pieces before the reads"),
                                                                             (7 + 32768, "8-wave, two-phase K-tile, W0 pre-read"), (5 + 32768, "8-wave, two-phase K-tile, plain"),
                                                                             (5 + 32768 + 65536, "8-wave, two-phase, sc0 DMA"), (5 + 32768 + 131072, "8-wave, two-phase, split B segment"),
                                                                             (4 + 32768, "8-wave, two-phase, no setprio"), (5 + 8 + 32768, "8-wave, two-phase ABLATION no DMA"),
                                                                             (5 + 16 + 32768, "8-wave, two-phase ABLATION no reads"),
                                                                             (5 + 32768 + 262144, "8-wave, two-phase ABLATION DMA never waited for"),
                                                                             (5 + 32768 + 524288, "8-wave, two-phase ABLATION DMA always hits L2"),
                                                                             (5 + 32768 + 1048576 + (2 << 24), "8-wave, two-phase, L2 prefetch 4 K-tiles ahead"),
                                                                             (5 + 32768 + 1048576 + (3 << 24), "8-wave, two-phase, L2 prefetch 6 K-tiles ahead"),
                                                                             (5 + 32768 + 1048576 + (4 << 24), "8-wave, two-phase, L2 prefetch 8 K-tiles ahead"),
                                                                             (5 + 32768 + 1048576 + (6 << 24), "8-wave, two-phase, L2 prefetch 12 K-tiles ahead"))]
            if os.environ.get("PM_PROBE_VARIANTS"):  # only these variant numbers (+ the library)
                keepv = {int(v) for v in os.environ["PM_PROBE_VARIANTS"].split(",")}
                cases = [c for c in cases if c[0] == "lib fwd" or any(f"variant {v} " in c[0] for v in keepv)]
            if os.environ.get("PM_PROBE_ONLY"):
                keep = os.environ["PM_PROBE_ONLY"].split(",")
                cases = [c for c in cases if any(k in c[0] for k in keep)]
            for label, fn in cases:
                try:
                    med, best = timeit(fn, iters=10, warmup=2)
                except Exception as ex:  # a variant word of a `make VARIANTS=1` library on the product build
                    print(json.dumps(dict(kernel=f"gemmprobe {label}", skipped=repr(ex)[:120])), flush=True)
                    ops.gemm_set_variant(7)
                    continue
                print(json.dumps(dict(kernel=f"gemmprobe {label} zero={os.environ.get('PM_PROBE_ZERO', '0')}", ms=round(med * 1e3, 4), TF=round(2.0 * t * o * i / med / 1e12, 1))), flush=True)
            return
        for _ in range(4):
            ops.gemm(x, w)
            ops.gemm(dy, w.t())
            ops.gemm(dy.t(), x.t(), out=g)
            if os.environ.get("PM_PROBE_LIB", "1") == "1":  # the library kernels next to ours, for the same counters
                torch.mm(x, w.t())
                torch.mm(dy, wt.t())
                torch.mm(dy.t(), x, out_dtype=torch.float32, out=g)
        torch.cuda.synchronize()
    if which == "gemmcore":
        bench_gemm_core(out, t=int(sys.argv[2]) if len(sys.argv) > 2 else 16384)
    if which in ("k1", "all"):
        bench_k1(out)
    if which == "k1aftergemm":
        bench_k1_after_gemm(out)
    if which == "k1step":
        bench_k1_step_shape(out)
    if which in ("attn", "all"):
        bench_attn(out)
    if which == "attn16k":
        bench_attn(out, cases=((16384, [2048] * 8),))
    if which in ("norm", "all"):
        bench_norm(out)
    for r in out:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
