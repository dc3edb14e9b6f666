"""ctypes binding of libmmvae_hip.so (the C-ABI declared in include/mmvae_hip.h).

The library is the product: if it is missing or fails to load, every HIP-path call raises -- there is no
fallback to torch ops or to the oracle ("no CPU fallback").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmmvae_hip.so")
if os.environ.get("MMVAE_LIB"):  # A/B runs of two builds of the same source tree (diagnostics)
    LIB_PATH = os.environ["MMVAE_LIB"]

OK, ERR_ARG, ERR_LAUNCH, ERR_WORKSPACE = 0, 1, 2, 3
_ERR_NAMES = {1: "MMVAE_ERR_ARG", 2: "MMVAE_ERR_LAUNCH", 3: "MMVAE_ERR_WORKSPACE"}

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
GEMM_RELU, GEMM_ACCUMULATE, GEMM_RAW_SLABS, GEMM_OPERAND_SLACK = 1, 2, 4, 8
GEMM_PRECISION_F32, GEMM_PRECISION_BF16X3 = 0, 1
ADAM_STATE_FLOATS = 8
PREPARE_NORM, PREPARE_ADVANCE = 1, 2

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_u = C.c_uint
_z = C.c_size_t
_u64 = C.c_uint64


class BnParams(C.Structure):
    """mirror of struct mmvae_bn_params"""

    _fields_ = [
        ("gamma", _p),
        ("beta", _p),
        ("running_mean", _p),
        ("running_var", _p),
        ("num_batches_tracked", _p),
        ("momentum", _f),
        ("eps", _f),
    ]


# name -> (restype, argtypes).  Kept in the order of include/mmvae_hip.h; tests/test_host_logic.py
# (test_abi_header_symbols_are_exported_and_bound) checks that every
# symbol the header declares is listed here and exported by the .so.
PROTOTYPES = {
    "mmvae_abi_version": (_i, []),
    "mmvae_build_arch": (C.c_char_p, []),
    "mmvae_gemm_set_precision": (_i, [_i]),
    "mmvae_gemm_set_workgroup_cap": (_i, [_i]),
    "mmvae_adam_set_workgroups": (_i, [_i]),
    "mmvae_recon_set_h_kpad": (_i, [_i]),
    "mmvae_adam_get_workgroups": (_i, []),
    "mmvae_gemm_set_x3w": (_i, [_i]),
    "mmvae_gemm_get_x3w": (_i, []),
    "mmvae_gemm_get_precision": (_i, []),
    "mmvae_gemm_plan": (_i, [_i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "mmvae_gemm_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "mmvae_gemm_f32": (_i, [_i, _i, _i, _i, _f, _p, _l, _p, _l, _p, _l, _p, _u, _i, _p, _z, _p]),
    "mmvae_recon_tiles": (_i, [_i]),
    "mmvae_decoder_recon_f32": (_i, [_i, _i, _i, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p]),
    "mmvae_decoder_recon_rows_f32": (_i, [_i, _i, _i, _i, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p]),
    "mmvae_recon_row_tiles": (_i, [_i]),
    "mmvae_decoder_recon_rows_colsum_f32": (_i, [_i, _i, _i, _i, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p, _p]),
    "mmvae_fc_workspace_bytes": (_z, [_i, _i]),
    "mmvae_fc_epilogue_fwd": (
        _i,
        [_i, _i, _p, _l, _i, _p, C.POINTER(BnParams), _i, _i, _p, _f, _p, _p, _p, _l, _p, _p, _p, _z, _p],
    ),
    "mmvae_fc_epilogue_bwd": (
        _i,
        [_i, _i, _p, _l, _i, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i, _p, _l, _p, _p, _p, _p, _z, _p],
    ),
    "mmvae_layernorm_fwd": (_i, [_i, _i, _p, _l, _f, _p, _l, _p, _p, _p]),
    "mmvae_layernorm_bwd": (_i, [_i, _i, _p, _l, _p, _l, _p, _p, _l, _p]),
    "mmvae_reparam_kl_fwd": (_i, [_i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p, _p]),
    "mmvae_reparam_kl_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p]),
    "mmvae_mse_sum_fwd_bwd": (_i, [_i, _i, _p, _l, _p, _l, _p, _p, _l, _p, _f, _p]),
    "mmvae_elbo_finalize": (_i, [_i, _i, _i, _p, _p, _p, _i, _p, _f, _p, _p, _p, _p]),
    "mmvae_iwae_logratio": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "mmvae_elbo_finalize_iwae": (_i, [_i, _i, _i, _p, _p, _p, _i, _p, _f, _p, _p, _p, _p]),
    "mmvae_iwae_bwd_terms": (_i, [_i, _i, _i, _p, _f, _p, _p, _p, _p, _p, _p]),
    "mmvae_cross_entropy_sum": (_i, [_i, _i, _p, _l, _p, _p, _p, _l, _p, _f, _p]),
    "mmvae_cross_entropy_heads": (_i, [_i, _i, _i, _p, _p, _p, _l, _p, _p, _p, _l, _f, _p]),
    "mmvae_sum_f32": (_i, [_l, _p, _p, _i, _p]),
    "mmvae_debug_occupy": (_i, [_i, _i, _i, _p, _p]),
    "mmvae_sum_rows_f32": (_i, [_i, _l, _p, _l, _p, _p, _p]),
    "mmvae_sqnorm_partials": (_l, [_l]),
    "mmvae_grad_sqnorm": (_i, [_l, _p, _p, _p]),
    "mmvae_adam_prepare": (_i, [_l, _p, _f, _f, _f, _f, _p, _u, _p]),
    "mmvae_grad_sqnorm_ranges_prepare": (_i, [_i, _p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _p, _u, _p]),
    "mmvae_adam_step": (_i, [_l, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _f, _p]),
    "mmvae_adam_step_copy": (_i, [_l, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _f, _i, _p, _p, _p]),
    "mmvae_adam_step_jobs": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _f, _f, _f, _f, _f, _p]),
    "mmvae_grad_sqnorm_jobs": (_i, [_i, _p, _p, _p, _p]),
    "mmvae_grad_zero_flagged_jobs": (_i, [_i, _p, _p, _p]),
    "mmvae_jobs_pack": (_i, [_i, _p, _p, _p, _p]),
    "mmvae_jobs_unpack": (_i, [_i, _p, _p, _p, _p]),
    "mmvae_philox_keep_mask": (_i, [_l, _f, _p, _p, _u64, _i, _p]),
    "mmvae_philox_normal": (_i, [_l, _p, _p, _u64, _i, _p]),
    "mmvae_philox_advance": (_i, [_p, _u64, _p]),
    "mmvae_philox_fill_jobs": (_i, [_i, _p, _l, _p, _p]),
    "mmvae_philox_fill_jobs_advance": (_i, [_i, _p, _l, _p, C.c_uint64, _p, _p]),
    "mmvae_axpby": (_i, [_l, _f, _p, _f, _p, _p]),
    "mmvae_upload_words": (_i, [_l, _p, _p, _p]),
    "mmvae_scale_rows": (_i, [_i, _i, _p, _l, _p, _p, _l, _p]),
    "mmvae_debug_stamp": (_i, [_p, _i, _p]),
    "mmvae_weighted_colsum_chunks": (_i, [_i]),
    "mmvae_weighted_colsum_f32": (_i, [_i, _i, _p, _l, _p, _p, _p]),
    "mmvae_sum_parts_batch": (_i, [_i, _p, _l, _p]),
    "mmvae_gemm_sq_partials": (_i, [_i, _i, _i, _i, _i]),
    "mmvae_gemm_f32_sq": (_i, [_i, _i, _i, _i, _f, _p, _l, _p, _l, _p, _l, _p, _u, _p, _l, _p]),
    "mmvae_split_planes_f32": (_i, [_i, _i, _p, _l, _p, _l, _l, _p]),
    "mmvae_gemm_planes_f32": (
        _i,
        [_i, _i, _i, _i, _f, _p, _l, _p, _l, _l, _p, _l, _p, _l, _l, _p, _l, _p, _u, _i, _p, _z, _p, _l, _p],
    ),
    "mmvae_gemm_planes_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "mmvae_decoder_recon_planes_f32": (
        _i,
        [_i, _i, _i, _i, _p, _l, _p, _l, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _l, _l, _p, _p, _p],
    ),
    "mmvae_decoder_recon_wplanes_f32": (
        _i,
        [_i, _i, _i, _i, _p, _l, _p, _l, _l, _p, _l, _p, _l, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p, _p],
    ),
    "mmvae_fc_epilogue_fwd_planes": (
        _i,
        [_i, _i, _p, _l, _i, _p, C.POINTER(BnParams), _i, _i, _p, _f, _p, _p, _p, _l, _p, _p, _p, _z, _p, _l, _l, _p],
    ),
    "mmvae_fc_epilogue_fwd_split": (
        _i,
        [_i, _i, _p, _l, _i, _p, C.POINTER(BnParams), _i, _i, _p, _f, _p, _p, _p, _l, _p, _p, _p, _z,
         _i, _i, _p, _l, _p, _l, _l, _p],
    ),
    "mmvae_fc_epilogue_bwd_planes": (
        _i,
        [_i, _i, _p, _l, _i, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i, _p, _l, _p, _p, _p, _p, _z, _p, _l, _l, _p],
    ),
    "mmvae_csr_to_dense_f32": (_i, [_i, _i, _l, _p, _p, _p, _p, _l, _p]),
    "mmvae_csr_to_dense_i32_f32": (_i, [_i, _i, _l, _p, _p, _p, _p, _l, _p]),
    "mmvae_csr_spmm_wt_i32_f32": (_i, [_i, _i, _i, _l, _p, _p, _p, _p, _l, _p, _p, _l, _p]),
    "mmvae_cond_linear_fwd": (_i, [_i, _i, _i, _p, _l, _p, _p, _p, _p, _p, _p, _l, _p]),
    "mmvae_cond_linear_bwd_dx": (_i, [_i, _i, _i, _p, _l, _p, _p, _p, _p, _p, _l, _i, _p]),
    "mmvae_cond_linear_bwd_dw": (_i, [_i, _p, _p, _p, _p, _i, _i, _p, _l, _p, _l, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "mmvae_cond_linear_fwd_multi": (_i, [_i, _i, _i, _i, _p, _l, _l, _p, _p, _p, _p, _p, _l, _p, _l, _l, _p]),
    "mmvae_cond_linear_bwd_dx_multi": (_i, [_i, _i, _i, _i, _p, _l, _l, _p, _p, _p, _l, _p, _l, _i, _p]),
    "mmvae_cond_linear_bwd_dw_multi": (_i, [_i, _i, _p, _p, _p, _p, _l, _i, _i, _p, _l, _l, _p, _l, _l, _p, _p, _p, _i, _p,
                                            _p, _p, _p, _l, _p]),
    "mmvae_gemm_batch_job_ok": (_i, [_p]),
    "mmvae_gemm_batch_prepare": (_i, [_i, _p, C.POINTER(_i)]),
    "mmvae_gemm_batch_f32": (_i, [_i, _p, _i, _p]),
    "mmvae_adv_pass_plan": (_i, [_p, _i, C.POINTER(_i), C.POINTER(_z), C.POINTER(_l), C.POINTER(_i)]),
    "mmvae_adv_pass_f32": (_i, [_i, _p, _i, _i, _i, _i, _z, _p]),
    "mmvae_adv_dw_prepare": (_i, [_i, _p, C.POINTER(_i), C.POINTER(_i)]),
    "mmvae_adv_dw_f32": (_i, [_i, _p, _i, _i, _p, _p, _p, _i, _p, _i, _p]),
    "mmvae_adam_step_multi": (_i, [_i, _p, _l, _p]),
    "mmvae_ell_from_dense_f32": (_i, [_i, _i, _p, _l, _i, _p, _p, _p, _p]),
    "mmvae_dw_sparse_ell_f32": (_i, [_i, _i, _i, _p, _l, _p, _p, _p, _i, _p, _l, _p]),
}


class SumJob(C.Structure):
    """mmvae_sum_job (include/mmvae_hip.h): one fixed-order reduction of mmvae_sum_parts_batch."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("part_stride", C.c_int64), ("ld_src", C.c_int64),
                ("ld_dst", C.c_int64), ("n_parts", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("alpha", C.c_float), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class PhiloxJob(C.Structure):
    """mmvae_philox_job of include/mmvae_hip.h."""
    _fields_ = [("out", C.c_void_p), ("n", C.c_int64), ("stream_id", C.c_uint64), ("p_drop", C.c_float),
                ("kind", C.c_int32)]


class GemmJob(C.Structure):
    """mmvae_gemm_job (include/mmvae_hip.h): one GEMM of a grouped mmvae_gemm_batch_f32 launch."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("lda", C.c_int64),
                ("ldb", C.c_int64), ("ldc", C.c_int64), ("layout", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("alpha", C.c_float), ("flags", C.c_uint32), ("first_block", C.c_int32),
                ("n_blocks", C.c_int32)]


ADV_MAX_LAYERS, ADV_MAX_HEADS = 4, 8


class AdvJob(C.Structure):
    """mmvae_adv_job (include/mmvae_hip.h): one adversary in one phase of mmvae_adv_pass_f32."""
    _fields_ = [("x", _p), ("ldx", _l), ("W", _p * ADV_MAX_LAYERS), ("b", _p * ADV_MAX_LAYERS),
                ("mask", _p * ADV_MAX_LAYERS), ("act", _p * ADV_MAX_LAYERS), ("dz", _p * ADV_MAX_LAYERS), ("Wh", _p),
                ("bh", _p), ("labels", _p), ("logits", _p), ("lse", _p), ("loss_rows", _p), ("gx", _p), ("partials", _p),
                ("loss_each", _p), ("loss_total", _p), ("total_loss", _p), ("total_scale", _f),
                ("gscale", _f), ("p_drop", _f * ADV_MAX_LAYERS), ("relu", C.c_int32 * ADV_MAX_LAYERS),
                ("width", C.c_int32 * (ADV_MAX_LAYERS + 1)), ("n_layers", C.c_int32), ("H", C.c_int32),
                ("Ct", C.c_int32), ("B", C.c_int32), ("col", C.c_int32 * ADV_MAX_HEADS),
                ("classes", C.c_int32 * ADV_MAX_HEADS), ("seg_lo", (C.c_int16 * ADV_MAX_HEADS) * 8),
                ("seg_hi", (C.c_int16 * ADV_MAX_HEADS) * 8)]


class AdvDwJob(C.Structure):
    """mmvae_adv_dw_job: one Linear's weight / bias gradient of mmvae_adv_dw_f32."""
    _fields_ = [("dz", _p), ("ld_dz", _l), ("inp", _p), ("ld_inp", _l), ("gW", _p), ("gb", _p), ("lse", _p),
                ("labels", _p), ("gscale", _f), ("M", C.c_int32), ("N", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
                ("opt", C.c_int32), ("col", C.c_int32 * ADV_MAX_HEADS), ("classes", C.c_int32 * ADV_MAX_HEADS),
                ("first_block", C.c_int32), ("n_blocks", C.c_int32)]


class AdvOpt(C.Structure):
    """mmvae_adv_opt: one optimiser of a mmvae_adv_dw_f32 launch (norm / clip / step bookkeeping)."""
    _fields_ = [("state", _p), ("norm_out", _p), ("max_norm", _f), ("grad_scale", _f), ("beta1", _f), ("beta2", _f),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class AdamArena(C.Structure):
    """mmvae_adam_arena: one optimiser's arenas in a mmvae_adam_step_multi launch."""
    _fields_ = [("p", _p), ("g", _p), ("m", _p), ("v", _p), ("state", _p), ("n", _l), ("lr", _f), ("beta1", _f),
                ("beta2", _f), ("eps", _f), ("weight_decay", _f), ("grad_scale", _f)]


class HipLibraryError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmmvae_hip.so (built by __graft_entry__.build() / `make -C mmvae_amd/csrc`).  Raises loudly."""
    global _lib
    if _lib is not None:
        return _lib
    # The library works on torch's streams and device memory, so it must share torch's HIP runtime: torch ships its own
    # libamdhip64, and whichever copy is loaded first serves the whole process.  Loading this library before torch binds
    # its code objects to the system runtime, and every launch on a torch stream then fails (MMVAE_ERR_LAUNCH).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C mmvae_amd/csrc` (or __graft_entry__.build()). "
            "mmvae_amd has no non-HIP execution path for device tensors."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise HipLibraryError(f"{what} failed: {_ERR_NAMES.get(rc, rc)}")
