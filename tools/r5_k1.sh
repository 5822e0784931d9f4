#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5k1}; mkdir -p $O
for v in 7 519 7 519; do
  PM_GEMM_VARIANT=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-extra --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$v.json
  python - <<PY
import json; r=json.load(open("$O/bench_$v.json")); print("variant $v", r["ms_per_step"], r["value"], "K1 frac", r["roofline"]["frac"], r["roofline"]["mean_launch_ms"])
PY
done
PM_GEMM_OWN=wgrad timeout 600 python bench.py --steps 6 --warmup 2 --no-extra --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_lib.json
python - <<PY
import json; r=json.load(open("$O/bench_lib.json")); print("library mix", r["ms_per_step"], r["value"], "K1 frac", r["roofline"]["frac"], r["roofline"]["mean_launch_ms"])
PY
