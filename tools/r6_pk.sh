#!/bin/bash
# round 6: packed multiplies in the dK/dV stream (VARIANTS library in the tree): bit-identity tests, then the A/B (0 = packed, 12 + 32*7 = scalar, 7 = round-5 kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -3
timeout 600 python tools/microbench.py attnbwd 0,236,7 2>&1 | grep -v amdgpu | tail -8 | cut -c1-250
