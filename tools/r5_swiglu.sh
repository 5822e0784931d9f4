#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5sw}; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/test_gemm.log; cat $O/test_gemm.log
timeout 600 python tools/microbench.py gemmswiglu > $O/gemmswiglu.log 2>&1; grep "^{" $O/gemmswiglu.log
for own in wgrad all; do
  PM_GEMM_OWN=$own timeout 600 python bench.py --steps 6 --warmup 2 --no-extra --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_${own}.json
  python - <<PY
import json; r=json.load(open("$O/bench_${own}.json")); print("$own", r["ms_per_step"], r["value"], r["config"]["gemm_own"], r["config"]["loss_last"])
PY
done
