#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${R5_OUT:-r5moe}; mkdir -p $O
CMD="python tools/microbench.py moelayer"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o moe -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/moelayer.log 2>&1)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r05_moe_layer "$CMD"
rm -rf $O/prof
tail -3 $O/moelayer.log | cut -c1-300
head -45 $O/r05_moe_layer.md | cut -c1-230
