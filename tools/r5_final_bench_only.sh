#!/bin/bash
# Round 5: the default `python bench.py` of the committed build (no trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5fin4}; mkdir -p $O
time (timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err); tail -2 $O/bench_default.err
python - <<PY
import json
r=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "roofline", r["roofline"]["frac"])
ex=r["config"]["extra"]
for k in ("mb2","library_mix","all_own","forced_collectives","moe_step","gemm_roofline","moe_layer"):
    v=ex.get(k); print(k, {a:b for a,b in (v or {}).items() if a not in ("note","kernel","shape","pattern","model")})
p=r["parity"]; print("parity gates", all(p["gates"].values()), "realistic gates", all(p["realistic"]["gates"].values()))
PY
