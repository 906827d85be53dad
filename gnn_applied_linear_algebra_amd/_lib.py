"""ctypes binding of libgla_hip.so (the C ABI declared in include/gla.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
torch is imported first so that the process-wide HIP runtime (libamdhip64.so.7 shipped inside the
torch wheel) is the one libgla_hip.so binds to; device pointers and streams are then shared.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede loading libgla_hip.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLA_LIB_PATH") or os.path.join(_HERE, "libgla_hip.so")      # (GLA_LIB_PATH: A/B builds, tools/)

GLA_F32, GLA_F64 = 0, 1
_DTYPES = {torch.float32: GLA_F32, torch.float64: GLA_F64}


class GlaError(RuntimeError):
    code = None                      # the library's status code (include/gla.h), when the error came from a call


class CsrInfo(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_int64), ("n_cols", ctypes.c_int64), ("nnz", ctypes.c_int64),
                ("sorted_input", ctypes.c_int32), ("max_row_len", ctypes.c_int32),
                ("n_tiles", ctypes.c_int32), ("tile_nnz", ctypes.c_int32), ("lds_elems", ctypes.c_int32),
                ("long_rows", ctypes.c_int32), ("val_dtype", ctypes.c_int32), ("packed", ctypes.c_int32),
                ("col16", ctypes.c_int32), ("col16_wide_tiles", ctypes.c_int32), ("dia", ctypes.c_int32),
                ("dia_stride", ctypes.c_int32), ("dia_row_lo", ctypes.c_int64), ("dia_row_hi", ctypes.c_int64),
                ("dia_A", ctypes.c_int32), ("dia_B", ctypes.c_int32), ("dia_canon5", ctypes.c_int32),
                ("dia_sym", ctypes.c_int32), ("dia_full", ctypes.c_int32)]


_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I = ctypes.c_int
_D = ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/gla.h declares (tests check this)
SIGNATURES = {
    "gla_version": (_I, []),
    "gla_error_string": (ctypes.c_char_p, [_I]),
    "gla_set_option": (_I, [ctypes.c_char_p, _I64]),
    "gla_csr_create_from_coo": (_I, [_I64, _I64, _I64, _P, _P, _P, ctypes.POINTER(_P)]),
    "gla_csr_destroy": (_I, [_P]),
    "gla_csr_set_values": (_I, [_P, _P, _I64, _I, _P]),
    "gla_csr_info": (_I, [_P, ctypes.POINTER(CsrInfo)]),
    "gla_csr_export": (_I, [_P, _P, _P, _P, _P]),
    "gla_csr_classify_tiles": (_I, [_P, _I64, _P, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "gla_spmm": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _I, _P]),
    "gla_spmm_add": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P, _I64, _I, _P]),
    "gla_dense_apply": (_I, [_I, _I64, _I64, _P, _I64, _I, _P, _I64, _P, _I64, _P]),
    "gla_residual": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P, _I64, _I, _P]),
    "gla_jacobi_sweep": (_I, [_P, _I, _I, _P, _P, _I64, _P, _I64, _P, _I64, _D, _I, _P]),
    "gla_jacobi_sweeps": (_I, [_P, _I, _I, _P, _P, _I64, _P, _I64, _P, _P, _I64, _D, _I, _P]),
    "gla_jacobi_sweeps_aos": (_I, [_P, _I, _P, _I64, _P, _P, _P, _P, _D, _I, _P]),
    "gla_jacobi_sweeps_schedule": (_I, [_P, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gla_jacobi_fused": (_I, [_P, _I, _P, _P, _P, _P, _D, _I, _I, _P]),
    "gla_jacobi_fused_parts": (_I, [_P, _I, _P, _P, _P, _P, _D, _I, _I, _I, _P]),
    "gla_power_rayleigh": (_I, [_P, _I, _I, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "gla_cheby_fused_next": (_I, [_P, _I, _I, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _D, _P, _P, _P]),
    "gla_cheby_steps": (_I, [_P, _I, _I, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "gla_copy_batch": (_I, [_I, _I, _P, _P, _P, _P, _P]),
    "gla_jacobi_fused_residual": (_I, [_P, _I, _P, _P, _P, _P, _P, _D, _I, _P]),
    "gla_jacobi_fused_parts_aos": (_I, [_P, _I, _P, _I64, _I64, _I64, _P, _P, _P, _D, _I, _I, _I, _P]),
    "gla_cheby_first": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _D, _I, _P]),
    "gla_cheby_next": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _D, _D, _D, _I, _P]),
    "gla_cheby_fused": (_I, [_P, _I, _I, _I, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64,
                             ctypes.POINTER(_D), ctypes.POINTER(_D), _P]),
    "gla_dot_work_bytes": (_I64, [_I64, _I, _I]),
    "gla_col_dot": (_I, [_I, _I64, _I, _P, _I64, _P, _I64, _P, _P, _P]),
    "gla_col_scale_rsqrt": (_I, [_I, _I64, _I, _P, _I64, _P, _P, _I64, _P]),
    "gla_power_work_bytes": (_I64, [_P, _I, _I]),
    "gla_power_fused_max_iters": (_I, [_P, _I, _I]),
    "gla_power_fused_work_bytes": (_I64, [_P, _I, _I, _I]),
    "gla_power_fused": (_I, [_P, _I, _I, _P, _I64, _P, _P, _I64, _P, _P, _I, _P]),
    "gla_power_step": (_I, [_P, _I, _I, _P, _I64, _P, _P, _P, _I64, _P, _P, _I, _P]),
    "gla_segment_reduce": (_I, [_P, _I, _I, _P, _I64, _I, _P, _I64, _P]),
    "gla_segment_stats": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P]),
    "gla_edge_dot": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P, _P]),
    "gla_mlp_work_bytes": (_I64, [_I]),
    "gla_mlp_relu3": (_I, [_I, _I64, _I, _I, _I, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gla_split_columns": (_I, [_I, _I64, _I, _P, _I64, ctypes.POINTER(_P), _P]),
    "gla_edge_products": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P]),
    "gla_edge_attr": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _P]),
    "gla_soc_classic": (_I, [_P, _I, _D, _P, _P, _P]),
    "gla_direct_interp": (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "gla_soc_sa": (_I, [_P, _I, _P, _P, _P]),
    "gla_laplacian2d_nnz": (_I64, [_I64, _I64, _I64, _I64]),
    "gla_laplacian2d_coo": (_I, [_I64, _I64, _I64, _I64, _P, _P, _P, _I, _P]),
    "gla_compact_work_bytes": (_I64, [_I64]),
    "gla_remove_diag_count": (_I, [_I64, _P, _P, _P, ctypes.POINTER(_I64), _P]),
    "gla_remove_diag_apply": (_I, [_I64, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P]),
    "gla_coo_coalesce": (_I, [_I64, _I64, _I64, _P, _P, _P, _I64, _I, _P, _P, _P, ctypes.POINTER(_I64), _P]),
    "gla_release_scratch": (_I, []),
    "gla_spgemm_count": (_I, [_P, _P, ctypes.POINTER(_I64), _P]),
    "gla_spgemm": (_I, [_P, _P, _I, _I64, _P, _P, _P, ctypes.POINTER(_I64), _P]),
    "gla_pmis_split": (_I, [_P, _P, _P, _I, ctypes.POINTER(_I), _P]),
    "gla_comm_unique_id": (_I, [_P]),
    "gla_comm_probe": (_I, []),
    "gla_comm_create": (_I, [_P, _I, _I, ctypes.POINTER(_P)]),
    "gla_comm_destroy": (_I, [_P]),
    "gla_comm_ranks": (_I, [_P, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gla_halo_begin": (_I, [_P, _I, _I, _P, _I64, _P, _I64, _I64, _P, _P]),
    "gla_halo_end": (_I, [_P, _P]),
    "gla_comm_all_gather_begin": (_I, [_P, _I, _P, _P, _I64, _P]),
    "gla_comm_all_gather": (_I, [_P, _I, _P, _P, _I64, _P]),
    "gla_comm_all_reduce_sum": (_I, [_P, _I, _P, _I64, _P]),
    "gla_dist_jacobi_sweeps": (_I, [_P, _P, _I, _I, _P, _P, _I64, _P, _P, _I64, _P, _I64, _I64, _P, _D, _I, _I, _P]),
    "gla_dist_jacobi_temporal": (_I, [_P, _P, _I, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _D,
                                      ctypes.POINTER(_I), _I, _P, _P]),
    "gla_dist_jacobi_temporal_graph": (_I, [_P, _P, _I, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _D,
                                            ctypes.POINTER(_I), _I, ctypes.POINTER(_P)]),
    "gla_graph_launch": (_I, [_P, _P]),
    "gla_graph_destroy": (_I, [_P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlaError(
                f"{LIB_PATH} not found: the HIP library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or csrc/build.sh). "
                "There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().gla_error_string(rc)
        err = GlaError(f"libgla_hip error {rc}: {msg.decode() if msg else '?'}")
        err.code = rc
        raise err


def dtype_code(dt):
    try:
        return _DTYPES[dt]
    except KeyError:
        raise TypeError(f"libgla_hip supports float32/float64, got {dt}") from None


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise GlaError("gnn_applied_linear_algebra_amd needs an AMD GPU (HIP device); none is visible and there is "
                       "no CPU fallback")
