mkdir -p gpurun_out/r4ab
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -k "update_under or recompute_gives or optimizer" > gpurun_out/r4ab/test.log 2>&1; tail -3 gpurun_out/r4ab/test.log
B="--steps 4 --warmup 2 --no-cpu-baseline --no-parity"
for f in "" "--no-overlap-update" "" "--no-overlap-update"; do timeout 400 python bench.py $B $f > gpurun_out/r4ab/bench.json 2>gpurun_out/r4ab/bench.err || tail -5 gpurun_out/r4ab/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r4ab/bench.json').read().strip().splitlines()[-1]); print('flags [$f]', d['ms_per_step'], d['value'], 'mb2', d['config']['extra'].get('mb2'))"; done
