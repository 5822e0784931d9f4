#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5k1b}; mkdir -p $O
for i in 1 2; do
for own in all wgrad; do
  PM_GEMM_OWN=$own timeout 600 python bench.py --no-extra --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${own}_$i.json
  python - <<PY
import json; r=json.load(open("$O/bench_${own}_$i.json")); print("$own", r["ms_per_step"], r["value"], "K1 frac", r["roofline"]["frac"], r["roofline"]["launch_ms"])
PY
done; done
