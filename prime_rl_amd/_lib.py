"""ctypes binding of libprime_mi355x.so (the C ABI declared in include/prime_mi355x.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  The product
path never routes through `oracle/` or through eager PyTorch re-implementations of the kernels.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "lib" / "libprime_mi355x.so"
CSRC_DIR = _PKG_DIR / "csrc"

PM_DTYPE_F32, PM_DTYPE_BF16, PM_DTYPE_I64, PM_DTYPE_U8 = 0, 1, 2, 3
PM_LOSS_DEFAULT, PM_LOSS_IPO = 0, 1
PM_ADV_GRPO, PM_ADV_MAXRL = 0, 1
PM_LOSS_NSTATS = 20
PM_LOSS_NMETRICS = 12

# indices into sample_stats (enum PM_STAT_* in the header)
STAT = {
    name: i
    for i, name in enumerate(
        [
            "RL_LOSS", "RL_N_MASK", "RL_N_DROP", "RL_N_KEEP", "RL_MKL_DROP", "RL_MKL_KEEP", "RL_N_LOW", "RL_N_HIGH",
            "RL_N_DROP_POS", "RL_N_DROP_NEG", "CE_LOSS", "CE_N", "CE_NLL", "REFKL_LOSS", "REFKL_N", "REFKL_N_DROP",
            "REFKL_N_KEEP", "REFKL_MKL_DROP", "REFKL_MKL_KEEP", "REFKL_SUM",
        ]
    )
}
# indices into sample_metrics (enum PM_MET_*), with the reference's metric keys (trainer/rl/loss.py:151-159,234-241,255-258)
METRIC_KEYS = [
    "masked_mismatch_kl", "unmasked_mismatch_kl", "is_masked", "is_masked_low", "is_masked_high",
    "masked_advantage_positive", "masked_advantage_negative", "nll",
    "ref_kl/masked_mismatch_kl", "ref_kl/unmasked_mismatch_kl", "ref_kl/is_masked", "ref_kl",
]


class LossParams(Structure):
    """pm_loss_params (include/prime_mi355x.h)."""

    _fields_ = [
        ("loss_type", c_int32),
        ("dppo_mask_low", c_float),
        ("dppo_mask_high", c_float),
        ("adv_tau", c_float),
        ("kl_tau", c_float),
        ("ipo_threshold", c_float),
        ("rl_scale", c_float),
        ("ce_scale", c_float),
        ("ref_kl_scale", c_float),
        ("shift", c_int32),
        ("pad_logprob", c_float),
    ]


P, I64, I32, F, I = c_void_p, c_int64, c_int32, c_float, c_int

# name -> (restype, argtypes); every symbol include/prime_mi355x.h declares
SIGNATURES: dict[str, tuple] = {
    "pm_abi_version": (c_int, []),
    "pm_last_error": (c_char_p, []),
    "pm_device_arch": (c_int, [c_char_p, c_int]),
    "pm_logprob_workspace_bytes": (I64, [I64, I64]),
    "pm_logprob_entropy_fwd": (I, [P, I, I64, I64, I64, P, P, P, P, P, P, I64, P]),
    "pm_logprob_bwd": (I, [P, I, I64, I64, I64, P, P, P, P, P, I, I64, P]),
    "pm_logprob_chunk_init": (I, [P, P, P, P, I64, P]),
    "pm_logprob_chunk_update": (I, [P, I, I64, I64, I64, I64, P, P, P, P, P, P, P]),
    "pm_logprob_chunk_finalize": (I, [P, P, P, P, I64, P, P, P, P]),
    "pm_logprob_chunk_bwd": (I, [P, I, I64, I64, I64, I64, P, P, P, P, P, I, I64, P]),
    "pm_lmhead_workspace_bytes": (I64, [I64, I64]),
    "pm_lmhead_logprob_fwd": (I, [P, P, I64, I64, I64, P, P, I32, P, P, P, P, I64, P, I64, P]),
    "pm_lmhead_logprob_bwd": (I, [P, P, I64, I64, I64, P, P, P, P, I32, P, I64, P]),
    "pm_grpo_advantage": (I, [P, P, I32, I, P, P, P, F, F, F, P, P]),
    "pm_advantage_broadcast": (I, [P, P, I32, P, P, I64, P]),
    "pm_grpo_loss_fwd_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I32, I64, POINTER(LossParams), P, P, P, P, P]),
    "pm_embedding_fwd": (I, [P, P, I64, I64, I64, P, P]),
    "pm_embedding_bwd": (I, [P, P, P, I64, I64, I64, P, P]),
    "pm_embedding_bwd_window": (I, [P, P, P, I64, I64, I64, I64, I64, P, P]),
    "pm_rmsnorm_fwd": (I, [P, P, P, I64, I64, F, P, P, P, P]),
    "pm_rmsnorm_bwd_workspace_bytes": (I64, [I64, I64]),
    "pm_rmsnorm_bwd": (I, [P, P, P, P, P, I64, I64, P, P, P, I64, P]),
    "pm_rope_table": (I, [P, I64, P, I32, P, P, P]),
    "pm_qknorm_rope_fwd": (I, [P, P, P, P, P, I64, I32, I32, I32, F, P, P, P, P]),
    "pm_qknorm_rope_bwd_workspace_bytes": (I64, [I64, I32, I32]),
    "pm_qknorm_rope_bwd": (I, [P, P, P, P, P, P, P, P, I64, I32, I32, I32, P, P, P, P, I64, P]),
    "pm_column_sum_workspace_bytes": (I64, [I64]),
    "pm_column_sum_bf16": (I, [P, I64, I64, I64, P, P, I64, P]),
    "pm_seq_start_from_cu": (I, [P, I32, I64, P, P]),
    "pm_attn_fwd_workspace_bytes": (I64, [I64, I32, I32]),
    "pm_attn_varlen_fwd": (I, [P, P, P, I64, P, I64, I32, I32, I32, F, P, P, P, I64, P]),
    "pm_attn_bwd_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "pm_attn_varlen_bwd": (I, [P, P, P, P, I64, P, P, P, I64, I32, I32, I32, F, P, P, P, I64, P, I64, P]),
    "pm_attn_varlen_fwd_rows": (I, [P, P, P, I64, P, I64, I64, I32, I32, I32, F, P, P, P]),
    "pm_attn_varlen_bwd_rows": (I, [P, P, P, P, I64, P, P, P, I64, I64, I32, I32, I32, F, P, P, P, I64, P, I64, P]),
    "pm_transpose_2d": (I, [P, I64, I64, I64, P, I64, P]),
    "pm_swiglu_fwd": (I, [P, I64, I64, P, P]),
    "pm_swiglu_bwd": (I, [P, P, I64, I64, P, P]),
    "pm_grad_accumulate": (I, [P, P, I64, P]),
    "pm_sumsq_workspace_bytes": (I64, [I64]),
    "pm_sumsq": (I, [P, I64, P, P, I64, P]),
    "pm_sumsq_segments": (I, [P, P, I32, P, P]),
    "pm_sum_partials": (I, [P, I64, P, P]),
    "pm_clip_coef": (I, [P, F, F, P, P, P]),
    "pm_adamw_step": (I, [P, P, P, P, P, I64, F, F, F, F, F, I64, F, P, I32, P]),
    "pm_sgd_step": (I, [P, P, P, P, I64, F, F, F, I32, I64, F, P, I32, P]),
    "pm_sign_sgd_step": (I, [P, P, P, I64, F, F, F, P, I32, P]),
    "pm_comm_available": (c_int, []),
    "pm_comm_rccl_version": (c_int, []),
    "pm_comm_unique_id": (I, [P]),
    "pm_comm_init": (I, [POINTER(c_void_p), P, I32, I32]),
    "pm_comm_destroy": (I, [P]),
    "pm_comm_rank_size": (I, [P, POINTER(c_int32), POINTER(c_int32)]),
    "pm_comm_all_reduce": (I, [P, P, P, I64, I, P]),
    "pm_comm_broadcast": (I, [P, P, P, I64, I, I32, P]),
    "pm_comm_reduce_scatter": (I, [P, P, P, I64, I, P]),
    "pm_comm_all_gather": (I, [P, P, P, I64, I, P]),
    "pm_comm_all_to_all": (I, [P, P, P, I64, I, P]),
    "pm_comm_send": (I, [P, P, I64, I, I32, P]),
    "pm_comm_recv": (I, [P, P, I64, I, I32, P]),
    "pm_comm_group_start": (I, []),
    "pm_comm_group_end": (I, []),
    "pm_mfma_probe_16x16x32": (I, [P, P, P, P]),
    "pm_attn_debug_set_variant": (I, [I32, I32]),
    "pm_moe_router_fwd": (I, [P, I32, P, P, I64, I32, I32, I32, I32, F, P, P, P, P, P, P]),
    "pm_moe_router_bwd": (I, [P, P, P, I64, I32, I32, I32, I32, F, P, P]),
    "pm_moe_sort_workspace_bytes": (I64, [I64, I32]),
    "pm_moe_sort": (I, [P, I64, I32, P, P, P, P, I64, P]),
    "pm_moe_gather_rows": (I, [P, P, I32, P, I64, I64, P, P]),
    "pm_moe_combine": (I, [P, P, P, P, I64, I32, I64, P, P]),
    "pm_moe_slot_grads": (I, [P, P, P, I32, P, I64, I64, P, P, P]),
    "pm_moe_ep_row_map": (I, [P, I32, I32, P, P, P, P]),
    "pm_grouped_gemm_supported": (I, [I64, I64, I64]),
    "pm_grouped_gemm_debug_set_variant": (I, [I32]),
    "pm_grouped_gemm_nt": (I, [P, P, P, P, I32, I64, I64, I64, I64, P, I32, P]),
    "pm_grouped_gemm_nn": (I, [P, P, P, P, I32, I64, I64, I64, I64, P, I32, P]),
    "pm_grouped_gemm_nt_swiglu_supported": (I, [I64, I64, I64]),
    "pm_grouped_gemm_nt_swiglu": (I, [P, P, P, P, I32, I64, I64, I64, P, P, I32, P]),
    "pm_grouped_gemm_nn_swiglu_bwd_supported": (I, [I64, I64, I64]),
    "pm_grouped_gemm_nn_swiglu_bwd": (I, [P, P, P, I32, I64, I64, I64, P, P, I32, P]),
    "pm_grouped_gemm_tn": (I, [P, P, P, I32, I64, I64, I64, I64, P, P, I32, P]),
    "pm_grouped_gemm_tn_f32": (I, [P, P, P, I32, I64, I64, I64, I64, P, P, I32, I32, P]),
    "pm_grouped_gemm_tn_sumsq_slots": (I64, [I32, I64, I64]),
    "pm_grouped_gemm_tn_f32_sumsq": (I, [P, P, P, I32, I64, I64, I64, I64, P, P, I32, P, I64, I32, P]),
    "pm_ds_read_tr16_probe": (I, [P, P, P]),
    "pm_gemm_supported": (I, [I64, I64, I64, I32, I32]),
    "pm_gemm_bf16": (I, [P, I64, I32, P, I64, I32, P, I64, I32, P, I32, I64, I64, I64, I32, P]),
    "pm_gemm_sumsq_slots": (I64, [I64, I64]),
    "pm_gemm_workspace_bytes": (I64, [I64, I64, I64]),
    "pm_gemm_bf16_ws": (I, [P, I64, I32, P, I64, I32, P, I64, I32, P, I32, I64, I64, I64, P, I64, P, I64, I32, P]),
    "pm_gemm_bf16_sumsq": (I, [P, I64, I32, P, I64, I32, P, I64, I32, I64, I64, I64, P, I64, I32, P]),
    "pm_gemm_swiglu_supported": (I, [I64, I64, I64]),
    "pm_gemm_swiglu_bf16": (I, [P, I64, P, I64, P, I64, P, I64, I64, I64, I64, I32, P]),
    "pm_gemm_swiglu_bwd_supported": (I, [I64, I64, I64]),
    "pm_gemm_swiglu_bwd_bf16": (I, [P, I64, P, I64, P, I64, P, I64, I64, I64, I64, I32, P]),
    "pm_gemm_debug_set_variant": (I, [I32]),
}


class PrimeMI355XError(RuntimeError):
    pass


def build(verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into prime_rl_amd/lib/libprime_mi355x.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(CSRC_DIR), "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise PrimeMI355XError(f"building libprime_mi355x.so failed (exit {res.returncode})")
    return LIB_PATH


_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """Load the shared library and bind every declared symbol.  Raises loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PrimeMI355XError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {CSRC_DIR}`). "
            "prime_rl_amd has no CPU or eager-PyTorch fallback."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - a stale .so
            raise PrimeMI355XError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.pm_abi_version()
    if got != 2:
        raise PrimeMI355XError(f"ABI version mismatch: library {got}, binding 2")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pm_last_error()
        raise PrimeMI355XError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
