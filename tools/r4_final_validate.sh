mkdir -p gpurun_out/r4z
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r4z/pytest_gpu.log 2>&1; tail -3 gpurun_out/r4z/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4z/smoke.log 2>&1; tail -1 gpurun_out/r4z/smoke.log
timeout 900 python bench.py > gpurun_out/r4z/bench_default.json 2> gpurun_out/r4z/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4z/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
print(json.dumps(d['config']['extra'].get('mb2')))
print(json.dumps(d['parity']['realistic']['gates']))
PY
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r4z/prof -o mb8 -- $CMD > gpurun_out/r4z/bench_trace.json 2> gpurun_out/r4z/bench_trace.err
DB=$(find gpurun_out/r4z/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r4z/r04_bench_mb8_v2 "$CMD"
rm -rf gpurun_out/r4z/prof
