#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5swb; mkdir -p $O
timeout 300 python tools/microbench.py gemmswiglubwd 2>&1 | grep -v "^\[W" | tee $O/gemmswiglubwd.txt
for i in 1 2; do timeout 300 python tools/microbench.py moelayer 2>&1 | grep -v "^\[W" | tee -a $O/moelayer.txt; done
(timeout 900 python tools/r5_swb_ab.py 8; timeout 600 python tools/r5_swb_ab.py 2) 2>&1 | grep -v "^\[W" | tee $O/step_ab.txt
