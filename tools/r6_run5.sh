#!/bin/bash
# round 6, run 5: the extended direct-parameter test, the 32 768-token step's kernel trace with the final build (layer and selective recompute), and a
# world-2 rehearsal of the bench launch on one GPU (gloo, correctness of the N > 1 path only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out; R=$GRAFT_REPO_ROOT; mkdir -p $O/r5
timeout 600 python -m pytest tests/test_module_engine_gpu.py -q -x -k "direct_parameters" 2>&1 | tail -4
for mode in layer selective; do
  CMD="python tools/microbench.py step32k 1 $mode"
  (cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_32k -o t32k -- bash -c "cd $R && $CMD" > $R/$O/r5/step32k_$mode.log 2>&1)
  tail -2 $O/r5/step32k_$mode.log | cut -c1-300
  DB=$(find $O/prof_32k -name "*results.db" | head -1)
  python tools/rocprof_summary.py "$DB" $O/r5/r06_bench_32k_$mode "$CMD" --window
  rm -rf $O/prof_32k
  head -24 $O/r5/r06_bench_32k_$mode.md | cut -c1-180
done
PM_BENCH_DRYRUN_SHARED_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-parity > $O/r5/bench_world2.json 2> $O/r5/bench_world2.err
tail -c 1500 $O/r5/bench_world2.json; tail -3 $O/r5/bench_world2.err
