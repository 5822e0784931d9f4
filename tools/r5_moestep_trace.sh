#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5moestep; mkdir -p $O
CMD="python tools/microbench.py moestep"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o moe -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/moestep.log 2>&1)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r05_moe_step "$CMD"
rm -rf $O/prof
tail -3 $O/moestep.log | cut -c1-400
head -50 $O/r05_moe_step.md | cut -c1-200
