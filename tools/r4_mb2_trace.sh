mkdir -p gpurun_out/r4ae
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --micro-batches 2 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --no-parity"
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r4ae/prof -o mb2 -- $CMD > gpurun_out/r4ae/bench_trace.json 2> gpurun_out/r4ae/bench_trace.err
DB=$(find gpurun_out/r4ae/prof -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r4ae/r04_bench_mb2_v1 "$CMD"
rm -rf gpurun_out/r4ae/prof
tail -1 gpurun_out/r4ae/bench_trace.json | cut -c1-300
