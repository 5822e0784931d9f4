#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5t}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 ${R5_PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; tail -n 40 $O/pytest_gpu.log
