#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5epi}; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu 2>&1 | tail -3
PM_GEMM_ROUNDS=5 PM_GEMM_HEAD=1 PM_GEMM_VARIANTS=7 timeout 900 python tools/microbench.py gemmcore 16384 > $O/gemmcore.log 2>&1
python tools/r5_summ.py $O
timeout 600 python tools/microbench.py gemmswiglu 2>&1 | grep "^{" | tee $O/gemmswiglu.log
