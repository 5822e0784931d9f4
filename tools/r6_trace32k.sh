#!/bin/bash
# round 6: kernel trace of the 32 768-token step (BASELINE.json configs[4] on one GPU, recompute=layer) -> profiles/r06_bench_32k.md
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O
CMD="python tools/microbench.py step32k 1"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_32k -o t32k -- bash -c "cd $R && $CMD" > $R/$O/r06_step32k_trace.log 2>&1)
tail -2 $O/r06_step32k_trace.log | cut -c1-400
DB=$(find $O/prof_32k -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_bench_32k "$CMD" --window
rm -rf $O/prof_32k
head -40 $O/r06_bench_32k.md | cut -c1-200
