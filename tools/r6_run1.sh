#!/bin/bash
# round 6, first GPU call: smoke the new bench code on the tiny model, the default-protocol bench (driver's: --steps 20 --warmup 5), the parity bisection
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 python bench.py --model tiny --seq-len 256 --steps 2 --warmup 1 --micro-batches 4 --tokens-per-pass 512 --parity-rollouts 8 > $O/r06_bench_tiny.json 2> $O/r06_bench_tiny.err
echo "tiny rc=$?"; tail -c 600 $O/r06_bench_tiny.err; head -c 1500 $O/r06_bench_tiny.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_v1.json 2> $O/r06_bench_v1.err
echo "bench rc=$?"; tail -c 800 $O/r06_bench_v1.err
cp $O/bench_detail.json $O/r06_bench_v1_detail.json 2>/dev/null; cp $O/bench_parity_realistic.json $O/r06_bench_v1_parity_realistic.json 2>/dev/null
python - <<'PY'
import json
r=json.loads([l for l in open("gpurun_out/r06_bench_v1.json") if l.startswith("{")][-1])
print("value", r["value"], "ms", r["ms_per_step"], "roofline", r["roofline"]["frac"], "line bytes", len(json.dumps(r)))
c=r["config"]; print({k:v for k,v in c.items() if k!="extra" and k!="workload"})
print("gates", r["parity"]["gates"]); print("realistic", json.dumps(r["parity"].get("realistic"), indent=0)[:3000])
PY
timeout 900 python tools/r6_parity_bisect.py 64 > $O/r06_parity_bisect.json 2> $O/r06_parity_bisect.err
echo "bisect rc=$?"; tail -c 500 $O/r06_parity_bisect.err
