#!/bin/bash
# round 6, end of round (third pass): smoke, the whole GPU suite and the default bench line of the committed build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final4; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final4/bench.json").read().strip().splitlines()[-1])
c=d["config"]
print(d["value"], d["ms_per_step"], "K1", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], {k:v for k,v in c.items() if k.startswith("long_context") and k.endswith("_ms") or k in ("j1_frac","mb2_samples_per_s","moe_step_ms","parity_gates_all_true","parity_realistic_gates_all_true","parity_kernel_gates_all_true")})
print(len(open("gpurun_out/final4/bench.json").read().strip().splitlines()[-1]), "bytes")
PY
