#!/bin/bash
# round 6: kernel trace of the module-engine MoE step (configs[3] slice) with the final build -> profiles/r06_moe_step_v2.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r6moestep; R=$GRAFT_REPO_ROOT; mkdir -p $O
CMD="python tools/microbench.py moestep"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o moe -- bash -c "cd $R && $CMD" > $R/$O/moestep.log 2>&1)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_moe_step_v2 "$CMD" --window
rm -rf $O/prof
tail -3 $O/moestep.log | cut -c1-400
head -64 $O/r06_moe_step_v2.md | cut -c1-200
