mkdir -p gpurun_out/r4aa
timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q > gpurun_out/r4aa/test_gemm.log 2>&1; tail -2 gpurun_out/r4aa/test_gemm.log
PM_GEMM_HEAD=1 PM_GEMM_VARIANTS=7,1031 timeout 400 python tools/microbench.py gemmcore 16384 > gpurun_out/r4aa/gemmcore.log 2>&1; cut -c1-300 gpurun_out/r4aa/gemmcore.log
B="--steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-parity"
for v in 1031 7 1031 7; do PM_GEMM_VARIANT=$v timeout 300 python bench.py $B > gpurun_out/r4aa/bench_v$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4aa/bench_v$v.json').read().strip().splitlines()[-1]); print('variant $v', d['ms_per_step'], d['value'])"; done
