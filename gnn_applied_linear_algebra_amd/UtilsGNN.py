"""Graph / matrix builders -- device-side mirror of /root/reference/pytorch/UtilsGNN.py:53-78.

Same names, arguments and return contracts as the reference (edge_index int64 [2, nnz], edge_val [nnz, 1]);
the tensors are produced on the GPU by libgla_hip kernels instead of scipy.sparse.kron on the host.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, stream_ptr
from ._common import compute_device


def laplacianfun_torch(N, Ny=None, row_range=None, dtype=torch.float64, device=None):
    """UtilsGNN.py:53-67.  Negative 5-point Laplacian on an N x N grid (diag -4, neighbours +1), COO sorted
    row-major with ascending columns, fp64 like the reference.  Extensions: `Ny` (rectangular Ny x N grid)
    and `row_range=(lo, hi)` (only those rows, global numbering) for row-block partitions."""
    _lib.require_gpu()
    Nx = int(N)
    Ny = Nx if Ny is None else int(Ny)
    lo, hi = (0, Nx * Ny) if row_range is None else (int(row_range[0]), int(row_range[1]))
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    nnz = lib().gla_laplacian2d_nnz(Nx, Ny, lo, hi)
    if nnz < 0:
        raise ValueError(f"bad grid / row range: N={Nx}, Ny={Ny}, rows=({lo},{hi})")
    edge_index = torch.empty((2, nnz), dtype=torch.long, device=dev)
    edge_val = torch.empty((nnz, 1), dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        check(lib().gla_laplacian2d_coo(Nx, Ny, lo, hi, ptr(edge_index[0]), ptr(edge_index[1]), ptr(edge_val),
                                        dtype_code(dtype), stream_ptr(dev)))
    return edge_index, edge_val


def remove_diag_entries(edge_index, edge_val):
    """UtilsGNN.py:69-72 (PyG remove_self_loops): order-preserving removal of the row == col edges."""
    dev = compute_device(edge_index, edge_val)
    ei = edge_index.to(dev).contiguous()
    ev = edge_val.to(dev)
    if ev.shape[0] != ei.shape[1]:
        raise ValueError("edge_val and edge_index disagree on the number of edges")
    width = 1 if ev.dim() == 1 else ev.shape[1]
    if ev.dim() > 2 or width != 1:
        # wider attribute rows: compact the indices once, then gather rows with the kept mask (rare path)
        mask = ei[0] != ei[1]
        return _ret(edge_index, ei[:, mask]), _ret(edge_val, ev[mask])
    ev = ev.contiguous()
    nnz = ei.shape[1]
    work = torch.empty(lib().gla_compact_work_bytes(nnz), dtype=torch.uint8, device=dev)
    kept = ctypes.c_int64()
    with torch.cuda.device(dev):
        check(lib().gla_remove_diag_count(nnz, ptr(ei[0]), ptr(ei[1]), ptr(work), ctypes.byref(kept),
                                          stream_ptr(dev)))
        out_i = torch.empty((2, kept.value), dtype=torch.long, device=dev)
        out_v = torch.empty((kept.value,) + tuple(ev.shape[1:]), dtype=ev.dtype, device=dev)
        if kept.value > 0:
            check(lib().gla_remove_diag_apply(nnz, ptr(ei[0]), ptr(ei[1]), ptr(ev), 1, dtype_code(ev.dtype),
                                              ptr(work), ptr(out_i[0]), ptr(out_i[1]), ptr(out_v), stream_ptr(dev)))
    return _ret(edge_index, out_i), _ret(edge_val, out_v)


def _ret(like, t):
    return t if like.is_cuda else t.to(like.device)


def coo_to_gnn_input(A):
    """UtilsGNN.py:74-78: coalesced indices [2, nnz] and values [nnz, 1] of a torch sparse COO tensor.
    The pair is cached on the sparse tensor object so that repeated adapter calls (VCycle.py calls this on every
    smoother / residual invocation, :63,77,96,141,157) reuse the same index tensor and hence its CSR plan."""
    cached = getattr(A, "_gla_gnn_input", None)
    if cached is not None:
        return cached
    Ac = A.coalesce()
    out = (Ac.indices(), Ac.values().reshape(-1, 1))
    try:
        A._gla_gnn_input = out
    except Exception:
        pass
    return out
