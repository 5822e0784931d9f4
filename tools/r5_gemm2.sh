#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5f}; mkdir -p $O
PM_GEMM_ROUNDS=7 PM_GEMM_HEAD=1 PM_GEMM_VARIANTS=7,32773,163845 timeout 900 python tools/microbench.py gemmcore 16384 > $O/gemmcore.log 2>&1
PM_GEMM_VARIANTS=7,32773,163845 timeout 600 python tools/microbench.py moegemm > $O/moegemm.log 2>&1
python tools/r5_summ.py $O; cat $O/moegemm.log | tail -5
