#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5t2}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=12 ${R5_PYTEST_ARGS} > $O/pytest_gpu.log 2>&1; tail -n 60 $O/pytest_gpu.log
