#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5g}; mkdir -p $O
PM_GEMM_ROUNDS=7 PM_GEMM_HEAD=1 PM_GEMM_VARIANTS=${R5_VARIANTS:-163845,164357} timeout 900 python tools/microbench.py gemmcore 16384 > $O/gemmcore.log 2>&1
python tools/r5_summ.py $O
