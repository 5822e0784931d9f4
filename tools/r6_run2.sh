#!/bin/bash
# round 6, second GPU call: the GPU suite on the stateless-ABI / own-GEMM nn.Module build, then the MoE module-engine step (timing + windowed kernel trace)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/r06_pytest_gpu_v1.log
cat $O/r06_pytest_gpu_v1.log | tail -12
timeout 300 python tools/microbench.py moestep > $O/r06_moestep_v1.log 2>&1; tail -2 $O/r06_moestep_v1.log | cut -c1-500
CMD="python tools/microbench.py moestep"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_moe -o moe -- bash -c "cd $R && $CMD" > $R/$O/r06_moestep_trace.log 2>&1)
DB=$(find $O/prof_moe -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_moe_step_v1 "$CMD" --window
rm -rf $O/prof_moe
head -45 $O/r06_moe_step_v1.md | cut -c1-220
