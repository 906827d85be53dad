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
    if A.is_cuda and not A.is_coalesced():
        from .sparse_ops import coalesce           # libgla_hip sort + in-order duplicate sums
        ei, ev = coalesce(A._indices(), A._values(), A.shape[0], A.shape[1])
        out = (ei, ev.reshape(-1, 1))
    else:
        Ac = A.coalesce()
        out = (Ac.indices(), Ac.values().reshape(-1, 1))
    try:
        A._gla_gnn_input = out
    except Exception:
        pass
    return out


def heateqnfem2dfun(num_cells, h_all, bcs=(2, 2), dtype=torch.float64, device=None):
    """matlab/heateqnfem2dfun.m:52-172 (the only definition of the FEM mesh of BASELINE config 5): Q1 stiffness matrix
    on a cx x cy cell grid with stretch alpha = hy/hx.  Every interior node sees all four of its elements, so the
    assembled interior rows are the constant 9-point stencil of :105-110,
        (1/6/alpha) * [4(2a^2+2) self, 2(-2a^2+1) x-neighbours, 2(a^2-2) y-neighbours, (-1-a^2) corners],
    truncated at the (eliminated) boundary.  bcs = [2,2]: interior unknowns only (:164-168); bcs = [1,1]: all nodes,
    identity rows/columns on the boundary (:143-161).  Returns (edge_index int64 [2,nnz], edge_val [nnz,1]) in
    coalesced row-major order, generated on the device."""
    _lib.require_gpu()
    cx, cy = int(num_cells[0]), int(num_cells[1])
    if tuple(bcs) not in ((1, 1), (2, 2)):
        raise ValueError("bcs must be [1,1] or [2,2] (mixed modes are rejected by the reference)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    alpha = float(h_all[1]) / float(h_all[0])
    a2 = alpha ** 2
    s = 1 / 6 / alpha
    e_self, e_x, e_y, e_c = s * (2 * a2 + 2), s * (-2 * a2 + 1), s * (a2 - 2), s * (-1 - a2)
    # assembled entries: the element contributions of one entry are equal, so their sum is an exact multiple
    stencil = [(-1, -1, e_c), (0, -1, e_y + e_y), (1, -1, e_c),
               (-1, 0, e_x + e_x), (0, 0, (e_self + e_self) + (e_self + e_self)), (1, 0, e_x + e_x),
               (-1, 1, e_c), (0, 1, e_y + e_y), (1, 1, e_c)]
    mx, my = cx - 1, cy - 1
    k = torch.arange(mx * my, device=dev)
    i, j = k % mx, k // mx
    cand_c = torch.stack([(j + dy) * mx + (i + dx) for dx, dy, _ in stencil], 1)
    valid = torch.stack([(i + dx >= 0) & (i + dx < mx) & (j + dy >= 0) & (j + dy < my) for dx, dy, _ in stencil], 1)
    vals = torch.tensor([c for _, _, c in stencil], dtype=torch.float64, device=dev).expand(k.numel(), 9)
    rows = k[:, None].expand(-1, 9)
    r, c, v = rows[valid], cand_c[valid], vals[valid]
    if tuple(bcs) == (1, 1):                     # embed the interior block, identity on the boundary nodes
        nx = cx + 1
        to_full = lambda q: (q // mx + 1) * nx + (q % mx + 1)
        kk = torch.arange(nx * (cy + 1), device=dev)
        bnd = kk[(kk % nx == 0) | (kk % nx == cx) | (kk // nx == 0) | (kk // nx == cy)]
        r = torch.cat([to_full(r), bnd])
        c = torch.cat([to_full(c), bnd])
        v = torch.cat([v, torch.ones(bnd.numel(), dtype=torch.float64, device=dev)])
        order = torch.argsort(r * (nx * (cy + 1)) + c)
        r, c, v = r[order], c[order], v[order]
    return torch.stack([r, c]), v.to(dtype).reshape(-1, 1)
