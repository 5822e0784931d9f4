#!/bin/bash
# round 6: VARIANTS=1 library with the hand-placed pipelined forward as the default: attention tests, A/B timing against the round-5 kernel, PMC counters of both
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -8 > $O/r06_attn_tests_variants_build.log; tail -4 $O/r06_attn_tests_variants_build.log
for d in 8 1; do timeout 300 python tools/microbench.py attnfwd 15,128,129,130 $d > $O/r06_attn_fwd_ab_docs$d.log 2>&1; tail -4 $O/r06_attn_fwd_ab_docs$d.log | cut -c1-200; done
bash tools/pmc_attn.sh PM_ATTN_VARIANTS=0:15 > $O/r06_attn_pmc_r5kernel.txt 2>&1
bash tools/pmc_attn.sh > $O/r06_attn_pmc_p3a.txt 2>&1
grep -A25 "attn_fwd" $O/r06_attn_pmc_r5kernel.txt | head -27
grep -A25 "attn_fwd" $O/r06_attn_pmc_p3a.txt | head -27
