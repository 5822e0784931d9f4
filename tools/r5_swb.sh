#!/bin/bash
# Round 5: the SwiGLU backward in the down-projection row-gradient epilogue — parity tests, microbench, MoE layer, short step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5swb; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_moe_gpu.py -q -x -k "swiglu or one_autograd_node or moe_layer" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "swiglu_backward_in or whichever or routing_can" 2>&1 | tail -5
timeout 300 python tools/microbench.py gemmswiglubwd 2>&1 | grep -v "^\[W" | tee $O/gemmswiglubwd.txt
timeout 300 python tools/microbench.py moelayer 2>&1 | grep -v "^\[W" | tee $O/moelayer.txt
