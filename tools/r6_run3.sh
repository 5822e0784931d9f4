#!/bin/bash
# round-6 run 3: the new module-engine test + the K1 read-order experiment
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=$PWD; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_module_engine_gpu.py -q -x -k "direct_parameters" > gpurun_out/r3/pytest.log 2>&1; tail -5 gpurun_out/r3/pytest.log
for t in 7168 4096 14336; do PM_K1_TAIL=$t timeout 600 python tools/microbench.py k1aftergemm > gpurun_out/r3/k1_$t.log 2>&1; tail -12 gpurun_out/r3/k1_$t.log; done
