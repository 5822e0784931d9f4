#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5seam}; mkdir -p $O
python tools/probes/_dbg_epi.py 2>&1 | grep "zero=" | grep "wave0" > $O/dbg.log; cat $O/dbg.log
PM_GEMM_ROUNDS=5 PM_GEMM_HEAD=1 PM_GEMM_VARIANTS=7,8199 timeout 900 python tools/microbench.py gemmcore 16384 > $O/gemmcore.log 2>&1
PM_GEMM_VARIANTS=7,8199 timeout 600 python tools/microbench.py moegemm > $O/moegemm.log 2>&1
python tools/r5_summ.py $O; grep "^{" $O/moegemm.log
