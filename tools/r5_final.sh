#!/bin/bash
# Round 5, end-of-round runs of the committed build: default bench.py (driver-like), kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5fin}; mkdir -p $O
time (timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err); tail -2 $O/bench_default.err
python - <<PY
import json
r=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "gemm_own", r["config"]["gemm_own"], "roofline", r["roofline"]["frac"], r["roofline"]["launch_ms"])
ex=r["config"]["extra"]
for k in ("mb2","library_mix","all_own","forced_collectives","moe_step","gemm_roofline","j1_roofline","moe_layer"):
    v=ex.get(k); print(k, {a:b for a,b in (v or {}).items() if a not in ("note","kernel","shape","pattern","model")})
p=r["parity"]; print("parity gates", all(p["gates"].values()), "realistic gates", all(p["realistic"]["gates"].values()) if "realistic" in p else None)
print("cpu", r.get("cpu_baseline",{}).get("value"))
PY
CMD="python bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mb8 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_trace.err)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r05_bench_mb8_v2 "$CMD"
rm -rf $O/prof
tail -1 $O/bench_trace.json | cut -c1-200
head -24 $O/r05_bench_mb8_v2.md | cut -c1-200
