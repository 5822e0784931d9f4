#!/bin/bash
# Round 5: kernel trace of the default step (all GEMMs on the own core) + PMC passes over the GEMM core's three layouts next to the library's kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5prof}; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mb8 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_trace.err)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r05_bench_mb8_v1 "$CMD"
rm -rf $O/prof
tail -1 $O/bench_trace.json | cut -c1-300
head -40 $O/r05_bench_mb8_v1.md
PM_GEMM_VARIANT=7 bash tools/pmc.sh r5gemm "gemm_kernel|Cijk" python tools/microbench.py gemmprobe > $O/pmc_gemm.log 2>&1
cp gpurun_out/pmc_r5gemm/summary.txt $O/pmc_gemm_summary.txt; rm -rf gpurun_out/pmc_r5gemm/p*/
