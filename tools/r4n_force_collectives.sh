mkdir -p gpurun_out/r4n
B="--steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-parity"
for v in 3 7; do PM_GEMM_VARIANT=$v timeout 300 python bench.py --force-collectives $B > gpurun_out/r4n/fc_v$v.json 2> gpurun_out/r4n/fc_v$v.err; done
timeout 300 python bench.py $B > gpurun_out/r4n/plain.json 2> gpurun_out/r4n/plain.err
PM_GEMM_OWN=all timeout 300 python bench.py $B > gpurun_out/r4n/all_own.json 2> gpurun_out/r4n/all_own.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4n/prof_fc -o fc -- python bench.py --force-collectives --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-parity > gpurun_out/r4n/fc_trace.json 2> gpurun_out/r4n/fc_trace.err
DB=$(find gpurun_out/r4n/prof_fc -name "*results.db" | head -1)
python tools/rocprof_summary.py "$DB" gpurun_out/r4n/r04_force_collectives "python bench.py --force-collectives --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-parity" 
rm -rf gpurun_out/r4n/prof_fc
for f in fc_v3 fc_v7 plain all_own fc_trace; do echo $f; python -c "
import json,sys
d=json.loads(open('gpurun_out/r4n/$f.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['comm'], d['config'].get('gemm_own'))"; done
