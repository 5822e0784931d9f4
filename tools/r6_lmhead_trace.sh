#!/bin/bash
# round 6: kernel trace of the fused LM head on the tile stream next to its alternatives -> profiles/r06_lmhead_trace.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r6lm; R=$GRAFT_REPO_ROOT; mkdir -p $O
CMD="python tools/microbench.py lmhead"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lm -- bash -c "cd $R && $CMD" > $R/$O/lmhead.log 2>&1)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_lmhead_trace "$CMD"
rm -rf $O/prof
grep "kernel" $O/lmhead.log | cut -c1-200
head -24 $O/r06_lmhead_trace.md | cut -c1-220
