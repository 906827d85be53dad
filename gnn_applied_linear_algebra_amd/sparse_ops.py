"""Device-side sparse algebra for the V-cycle setup: coalesce and sparse-sparse product (csrc/gla_sparse.hip).

Replaces the torch CPU sparse calls of the reference's setup path (`A.coalesce()` pytorch/UtilsGNN.py:74-78,
`P.t() @ (A @ P)` pytorch/VCycle.py:209) with libgla_hip kernels; results come back as coalesced COO pairs
(edge_index int64 [2, nnz], values [nnz]) that feed `plan_for` through the sorted-input fast path.
"""
import ctypes

import torch

from ._lib import GlaError, check, dtype_code, lib, ptr, stream_ptr
from .plan import CsrPlan, _DeviceGuard


def coalesce(edge_index, values, n_rows, n_cols):
    """Sort by (row, col) and add duplicates in their original order -> (edge_index, values)."""
    if not edge_index.is_cuda or not values.is_cuda:
        raise ValueError("coalesce expects GPU tensors")
    ei = edge_index.contiguous()
    vals = values.reshape(-1).contiguous()
    nnz = ei.shape[1]
    if vals.numel() != nnz:
        raise ValueError("values and edge_index disagree on the number of entries")
    out_i = torch.empty((2, nnz), dtype=torch.long, device=ei.device)
    out_v = torch.empty(nnz, dtype=vals.dtype, device=ei.device)
    n_out = ctypes.c_int64()
    with _DeviceGuard(ei.device):
        try:                                            # the library validates the indices in its key pass
            check(lib().gla_coo_coalesce(int(n_rows), int(n_cols), nnz, ptr(ei[0]), ptr(ei[1]), ptr(vals), 1,
                                         dtype_code(vals.dtype), ptr(out_i[0]), ptr(out_i[1]), ptr(out_v),
                                         ctypes.byref(n_out), stream_ptr(ei.device)))
        except GlaError as exc:
            if exc.code == -2:
                raise ValueError("edge index out of range") from None
            raise
    return out_i[:, :n_out.value].contiguous(), out_v[:n_out.value].contiguous()


def spgemm(A: CsrPlan, B: CsrPlan):
    """C = A @ B of two plans with values -> coalesced (edge_index [2, nnz], values [nnz]), shape A.n_rows x B.n_cols."""
    if A.n_cols != B.n_rows:
        raise ValueError(f"shape mismatch: {A.n_rows}x{A.n_cols} @ {B.n_rows}x{B.n_cols}")
    if A.val_dtype is None or A.val_dtype != B.val_dtype:
        raise ValueError("both plans need values of the same dtype")
    dev, dt = A.device, A.val_dtype
    n_prod = ctypes.c_int64()
    with _DeviceGuard(dev):
        check(lib().gla_spgemm_count(A._h, B._h, ctypes.byref(n_prod), stream_ptr(dev)))
        out_i = torch.empty((2, n_prod.value), dtype=torch.long, device=dev)
        out_v = torch.empty(n_prod.value, dtype=dt, device=dev)
        n_out = ctypes.c_int64()
        check(lib().gla_spgemm(A._h, B._h, dtype_code(dt), n_prod.value, ptr(out_i[0]), ptr(out_i[1]), ptr(out_v),
                               ctypes.byref(n_out), stream_ptr(dev)))
    return out_i[:, :n_out.value].contiguous(), out_v[:n_out.value].contiguous()


def plan_of(edge_index, values, n_rows, n_cols):
    p = CsrPlan(edge_index, n_rows, n_cols)
    p.set_values(values)
    return p


def galerkin(A_ei, A_val, n, P_ei, P_val, n_c):
    """Ac = P^T (A P) (VCycle.py:209) -> coalesced (edge_index, values) of the n_c x n_c coarse operator, plus the
    plans of P and P^T for restriction / prolongation."""
    A = plan_of(A_ei, A_val, n, n)
    P = plan_of(P_ei, P_val, n, n_c)
    Pt_ei, Pt_val = coalesce(P_ei.flip(0), P_val, n_c, n)
    Pt = plan_of(Pt_ei, Pt_val, n_c, n)
    AP_ei, AP_val = spgemm(A, P)
    AP = plan_of(AP_ei, AP_val, n, n_c)
    Ac_ei, Ac_val = spgemm(Pt, AP)
    P.coo, Pt.coo = (P_ei, P_val), (Pt_ei, Pt_val)        # kept for row-block slicing (dist_amg.py)
    return Ac_ei, Ac_val, P, Pt
