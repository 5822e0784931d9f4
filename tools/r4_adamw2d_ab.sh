mkdir -p gpurun_out/r4aj
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "adamw" > gpurun_out/r4aj/test.log 2>&1; tail -3 gpurun_out/r4aj/test.log
B="--steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-parity"
for v in 0 1 0 1; do PM_ADAMW_2D=$v timeout 300 python bench.py $B > gpurun_out/r4aj/bench_$v.json 2>gpurun_out/r4aj/bench_$v.err || tail -3 gpurun_out/r4aj/bench_$v.err; python -c "
import json; d=json.loads(open('gpurun_out/r4aj/bench_$v.json').read().strip().splitlines()[-1]); print('PM_ADAMW_2D=$v', d['ms_per_step'], d['value'])"; done
