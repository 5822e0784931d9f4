#!/bin/bash
# round 6: the three-stage pipelined forward (VARIANTS=1 library): bit-identity against the shipped kernel, then interleaved timing at 8 x 2048 and 1 x 16384
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "fwd_variants_are_bit_identical" 2>&1 | tail -15 > $O/r06_attn_p3_identity.log
tail -6 $O/r06_attn_p3_identity.log
timeout 300 python tools/microbench.py attnfwd ${VARS:-15,128,129} 8 > $O/r06_attn_p3_8x2048.log 2>&1; tail -5 $O/r06_attn_p3_8x2048.log
timeout 300 python tools/microbench.py attnfwd ${VARS:-15,128,129} 1 > $O/r06_attn_p3_1x16384.log 2>&1; tail -5 $O/r06_attn_p3_1x16384.log
