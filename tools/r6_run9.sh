#!/bin/bash
# round 6, run 9: recompute_freq — the bit-identity test, the train CLI rehearsals, and the bench line's long_context block
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r9; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -k "recompute" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_multirank_one_gpu.py -q -x -k "train_cli" 2>&1 | tail -3
timeout 1500 python bench.py --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r9/bench.json").read().strip().splitlines()[-1])
c=d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in c.items() if k.startswith("long_context") or k in ("j1_frac","mb2_samples_per_s","moe_step_ms")})
print(json.load(open("gpurun_out/bench_detail.json")).get("extra",{}).get("long_context"))
PY
