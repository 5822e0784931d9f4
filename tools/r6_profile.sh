#!/bin/bash
# Round 6: kernel trace of the default step (the command of the bench line, extras / CPU baseline / parity legs off), then PMC counters of the attention kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r6prof; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mb8 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/bench_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_trace.err)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_bench_mb8_v1 "$CMD"
rm -rf $O/prof
tail -1 $O/bench_trace.json | cut -c1-400
head -36 $O/r06_bench_mb8_v1.md | cut -c1-200
bash tools/pmc_attn.sh > $O/r06_attn_pmc_final.txt 2>&1
grep -A25 "attn_bwd_dkdv\|attn_fwd" $O/r06_attn_pmc_final.txt | head -60
