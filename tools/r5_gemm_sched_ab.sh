#!/bin/bash
# Round 5: A/B of the 8-wave GEMM core's K-loop schedules (pm_gemm_debug_set_variant) against the default and the library.
#   R5_OUT=<dir under gpurun_out>  R5_VARIANTS=<gemmcore variants>  R5_PROBE=<gemmprobe variants>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5}; mkdir -p $O
PM_GEMM_HEAD=${R5_HEAD:-0} PM_GEMM_VARIANTS=${R5_VARIANTS:-7,32773} timeout 900 python tools/microbench.py gemmcore 16384 > $O/gemmcore.log 2>&1
if [ -n "$R5_PROBE" ]; then
PM_PROBE_TIME=1 PM_PROBE_ZERO=1 PM_PROBE_VARIANTS=$R5_PROBE timeout 300 python tools/microbench.py gemmprobe > $O/probe_zero.log 2>&1
PM_PROBE_TIME=1 PM_PROBE_ZERO=0 PM_PROBE_VARIANTS=$R5_PROBE timeout 300 python tools/microbench.py gemmprobe > $O/probe_rand.log 2>&1
fi
python tools/r5_summ.py $O
