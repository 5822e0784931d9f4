#!/bin/bash
# Round 6, final build: kernel trace of the default step (the command of the bench line, extras / CPU baseline / parity legs off) -> profiles/r06_bench_mb8_v2.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/r6prof2; R=$GRAFT_REPO_ROOT; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o mb8 -- bash -c "cd $R && $CMD" > $R/$O/bench_trace.json 2> $R/$O/bench_trace.err)
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" $O/r06_bench_mb8_v2 "$CMD"
rm -rf $O/prof
tail -1 $O/bench_trace.json | cut -c1-300
head -30 $O/r06_bench_mb8_v2.md | cut -c1-200
