#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5swb; mkdir -p $O
timeout 900 python -m pytest tests/test_moe_gpu.py tests/test_module_engine_gpu.py tests/test_multirank_one_gpu.py -q -x -k "moe or expert or module" > $O/pytest_moe.txt 2>&1; grep -E "passed|failed|Error|assert" $O/pytest_moe.txt | head -20
PM_MOE_ITERS=30 timeout 300 python tools/microbench.py moelayer 2>&1 | grep -v "^\[W" | tee -a $O/moelayer6.txt
PM_MOE_ITERS=30 timeout 300 python tools/microbench.py moelayer 2>&1 | grep -v "^\[W" | tee -a $O/moelayer6.txt
