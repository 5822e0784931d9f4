#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out; mkdir -p $O/r6
timeout 600 python -m pytest tests/test_module_engine_gpu.py -q -x -k "direct_parameters" 2>&1 | grep -v "^$" | tail -40 | cut -c1-250
PM_BENCH_DRYRUN_SHARED_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --micro-batches 2 --recompute layer --no-extra --no-cpu-baseline --no-parity > $O/r6/bench_world2.json 2> $O/r6/bench_world2.err
tail -c 1200 $O/r6/bench_world2.json; grep -v "^\s*$" $O/r6/bench_world2.err | tail -5 | cut -c1-300
