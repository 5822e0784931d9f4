#!/bin/bash
# in-step A/B: which GEMM classes run on the own core (PM_GEMM_OWN), alternating in one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5own}; mkdir -p $O
for i in 1 2; do
for own in wgrad all; do
  PM_GEMM_OWN=$own timeout 600 python bench.py --steps 6 --warmup 2 --no-extra --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${own}_$i.json
  python - <<PY
import json; r=json.load(open("$O/bench_${own}_$i.json")); print("$own", $i, r["ms_per_step"], r["value"], r["config"]["gemm_own"])
PY
done; done
