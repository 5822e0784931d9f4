#!/bin/bash
# round 6, end of round (second pass): smoke + the whole GPU suite on the committed build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/smoke.log; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
