mkdir -p gpurun_out/r4q
timeout 900 python bench.py > gpurun_out/r4q/bench_default.json 2> gpurun_out/r4q/bench_default.err
tail -c 3000 gpurun_out/r4q/bench_default.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r4q/prof -o mb8 -- python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-parity > gpurun_out/r4q/bench_trace.json 2> gpurun_out/r4q/bench_trace.err
DB=$(find gpurun_out/r4q/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r4q/r04_bench_mb8_v1 "python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
rm -rf gpurun_out/r4q/prof
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4q/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
print(json.dumps(d['config']['extra'].get('mb2')), json.dumps(d['config']['extra'].get('gemm_roofline'))[:300])
print(json.dumps(d['parity']['realistic']['gates']))
for k in ('two_layer','full_depth'):
    for pol in ('fresh','stale'):
        r=d['parity']['realistic'][k][pol]; print(k,pol,{x:r[x] for x in ('loss_abs_delta','loss_delta_se','yardstick_loss_abs_delta','flips','yardstick_flips','tokens_within_0.02_of_threshold','masked_tokens_engine','masked_tokens_oracle')})
PY
