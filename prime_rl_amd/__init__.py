"""prime_rl_amd — MI355X-native GRPO policy-update hot path behind the prime-rl trainer API.

Layout: `csrc/` (HIP kernels + C ABI, built into `lib/libprime_mi355x.so`), `_lib.py` (ctypes binding), `ops.py`
(tensor-level wrappers), and host-side mirrors of the reference's seams (`trainer/`, `orchestrator/`, `transports/`).
"""

import os as _os

# Multi-process GPU work (RCCL, one process per GPU) needs dmabuf IPC on this driver stack; the variable is read when HIP
# initialises, i.e. at the first device call, so setting it at package import is early enough.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
