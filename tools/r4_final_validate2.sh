mkdir -p gpurun_out/r4zz
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r4zz/pytest_gpu.log 2>&1; grep -n "passed\|failed\|error" gpurun_out/r4zz/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4zz/smoke.log 2>&1; tail -1 gpurun_out/r4zz/smoke.log
timeout 900 python bench.py > gpurun_out/r4zz/bench_default.json 2> gpurun_out/r4zz/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4zz/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
print(json.dumps(d['config']['extra'].get('mb2')))
print(all(d['parity']['realistic']['gates'].values()), json.dumps(d['parity'].get('gates', {}))[:300])
PY
