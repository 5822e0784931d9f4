#!/bin/bash
# round 6: timing ablations of the hand-placed forward stream + PMC counters of the shipped forward (15) and the hand-placed one (129)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out; mkdir -p $O
PM_PROBE=1 timeout 300 python tools/microbench.py attnfwd 15,129,130,131,132,133 1 > $O/r06_attn_abl_1x16384.log 2>&1; tail -6 $O/r06_attn_abl_1x16384.log | cut -c1-200
PM_PROBE=1 timeout 300 python tools/microbench.py attnfwd 15,129,130,131,132,133 8 > $O/r06_attn_abl_8x2048.log 2>&1; tail -6 $O/r06_attn_abl_8x2048.log | cut -c1-200
if [ -z "$NOPMC" ]; then
bash tools/pmc_attn.sh > $O/r06_attn_pmc_default.txt 2>&1
bash tools/pmc_attn.sh PM_ATTN_VARIANTS=0:129 > $O/r06_attn_pmc_p3a.txt 2>&1
grep -A30 "attn_fwd" $O/r06_attn_pmc_default.txt | head -40
grep -A30 "attn_fwd" $O/r06_attn_pmc_p3a.txt | head -40
fi
