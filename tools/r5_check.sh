#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5chk}; mkdir -p $O
true
true
time (timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err); tail -3 $O/bench_default.err
python - <<PY
import json
r=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "gemm_own", r["config"]["gemm_own"], "roofline", r["roofline"]["frac"])
ex=r["config"]["extra"]
for k in ("mb2","library_mix","all_own","forced_collectives","gemm_roofline","j1_roofline","moe_layer"):
    v=ex.get(k); print(k, {a:b for a,b in (v or {}).items() if a not in ("note","kernel","shape","pattern")})
p=r["parity"]; print("parity gates", p["gates"]); print("realistic gates", p["realistic"].get("gates") if "realistic" in p else None)
if "realistic" in p and "two_layer" in p["realistic"]:
    for d in ("two_layer","full_depth"):
        x=p["realistic"][d]; print(d, "noise_ratio", x["noise_ratio"], {k:(x[k]["loss_abs_delta"], x[k]["loss_abs_delta_vs_bfa"], x[k]["yardstick_loss_abs_delta"], x[k]["yardstick_loss_delta_se"], x[k]["flips"], x[k]["flips_vs_bfa"], x[k]["yardstick_flips"], x[k]["yardstick_bfa_flips"]) for k in ("fresh","stale")})
print("cpu", r.get("cpu_baseline",{}).get("value"))
PY
