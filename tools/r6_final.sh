#!/bin/bash
# round 6, end of round: smoke, the whole GPU suite, the default bench line of the committed build
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/r06_smoke_final.log; tail -2 $O/r06_smoke_final.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/r06_pytest_gpu_final.log
grep -E "passed|failed" $O/r06_pytest_gpu_final.log | tail -2
timeout 1500 python bench.py > $O/r06_bench_final.json 2> $O/r06_bench_final.err; tail -c 600 $O/r06_bench_final.json; tail -2 $O/r06_bench_final.err
