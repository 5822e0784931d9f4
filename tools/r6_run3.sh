#!/bin/bash
# round 6, third GPU call: the whole GPU suite (no -x) on the current build
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/r06_pytest_gpu_v2.log
tail -15 $O/r06_pytest_gpu_v2.log
