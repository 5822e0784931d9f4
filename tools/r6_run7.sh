#!/bin/bash
# round 6, run 7: engine GPU tests + the default bench line after moving the embedding-gradient zero fill in front of K1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r7; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_multirank_one_gpu.py -q -x 2>&1 | tail -4
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json | head -c 2500; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r7/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["mean_launch_ms"], {k:v for k,v in d["config"].items() if k in ("mb2_samples_per_s","long_context_ms","long_context_selective_ms","moe_step_ms","parity_gates_all_true","all_own_ms","library_mix_ms")})
PY
