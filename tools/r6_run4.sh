#!/bin/bash
# round 6: product library with the hand-placed pipelined attention forward: whole GPU suite, then the default bench line
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/r06_pytest_gpu_v3.log
tail -5 $O/r06_pytest_gpu_v3.log
timeout 1500 python bench.py > $O/r06_bench_v2.json 2> $O/r06_bench_v2.err; tail -c 1500 $O/r06_bench_v2.json; tail -3 $O/r06_bench_v2.err
