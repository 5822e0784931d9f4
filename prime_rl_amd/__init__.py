"""prime_rl_amd — MI355X-native GRPO policy-update hot path behind the prime-rl trainer API.

Layout: `csrc/` (HIP kernels + C ABI, built into `lib/libprime_mi355x.so`), `_lib.py` (ctypes binding), `ops.py`
(tensor-level wrappers), and host-side mirrors of the reference's seams (`trainer/`, `orchestrator/`, `transports/`).
"""

__version__ = "0.1.0"
