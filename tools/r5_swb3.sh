#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5swb; mkdir -p $O
timeout 1200 python -m pytest tests/test_moe_gpu.py tests/test_module_engine_gpu.py tests/test_multirank_one_gpu.py -q -x > $O/pytest_moe.txt 2>&1; grep -E "passed|failed|Error|assert" $O/pytest_moe.txt | head -20
timeout 600 python tools/microbench.py moestep 2>&1 | grep -v "^\[W" | tee -a $O/moestep8.txt
