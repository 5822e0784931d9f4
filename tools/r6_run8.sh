#!/bin/bash
# round 6, run 8: the step with the fused LM head (PM_LM_HEAD=fused) next to the default, headline only
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r8; mkdir -p $O
for head in library fused; do
  PM_LM_HEAD=$head timeout 900 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-parity > $O/bench_$head.json 2> $O/bench_$head.err
  python - $O/bench_$head.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"].get("lm_head"), d["ms_per_step"], d["value"], d["roofline"]["kernel"][:60], d["roofline"]["frac"], "peak_mem", d["config"].get("peak_hbm_gb"))
PY
done
