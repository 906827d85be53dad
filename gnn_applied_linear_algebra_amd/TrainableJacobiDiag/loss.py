"""Damping-factor loss of the learned Jacobi diagonal -- mirror of
/root/reference/pytorch/TrainableJacobiDiag/loss.py.

df(A, omega, d) ~ rho(I - omega D^-1 A) is estimated with Gelfand's formula on m test vectors:
max_j ||T^K y_j||^(1/K), T = I - omega D^-1 A (:92-128).  The K products `T @ Y` are gla_spmm launches on T's CSR plan
(Y is [n, m] row-major: every gather reads m contiguous values) and stay differentiable with respect to the learned
diagonal through gnn_applied_linear_algebra_amd.autograd (gla_edge_dot + the transposed-plan gla_spmm).  The `exact`
variants (:59-65, :197-233) are dense torch eigen-solves exactly as in the reference -- validation only, tiny sizes.
"""
import math

import torch

from ..autograd import SparseMatrix
from ..plan import plan_for
from ..sparse_ops import coalesce


def _as_sparse(A, device=None):
    if isinstance(A, SparseMatrix):
        return A
    if isinstance(A, torch.Tensor) and A.layout == torch.sparse_coo:
        cached = getattr(A, "_gla_sparse", None)        # device copy (and its cached plans) kept with the tensor
        if cached is None:
            cached = SparseMatrix.from_torch(A, device)
            try:
                A._gla_sparse = cached
            except Exception:
                pass
        return cached
    raise TypeError("expected a SparseMatrix or a torch sparse COO tensor")


class _RowGather(torch.autograd.Function):
    """d[row_e] for every edge e of a matrix pattern.  Backward = the edge-to-vertex sum of the incoming gradient on
    the pattern's CSR plan (gla_segment_reduce: edge order within a row, deterministic) instead of torch's generic
    index_put accumulation, which radix-sorts the 2e7 indices on every call (1.5 ms of a 4.4 ms training step)."""

    @staticmethod
    def forward(ctx, d, edge_index, n_rows, n_cols):
        ctx.edge_index, ctx.shape = edge_index, (n_rows, n_cols)
        return d[edge_index[0]]

    @staticmethod
    def backward(ctx, g):
        plan = plan_for(ctx.edge_index, ctx.shape[0], ctx.shape[1])
        return plan.segment_reduce(g.contiguous(), "sum"), None, None, None


def damping_factor(A, omega, diagonal_value, xy=None, exact=False):
    """:43-90  max |eig| of I - omega * (D \\ A); `exact` converts to dense (no autograd), else Gelfand K=3, m=20."""
    N = A.shape[0]
    if exact:
        Adense = A.to_dense()
        D = torch.diagflat(diagonal_value).to(Adense.device)
        J = torch.eye(N, device=Adense.device, dtype=Adense.dtype) - omega * (torch.linalg.solve(D, Adense))
        return max(abs(torch.linalg.eigvals(J.cpu())))
    K = 3
    m = 20
    T = build_error_matrix(A, diagonal_value, omega)
    return eigval_approx(K, m, T, xy=xy, method='high_freq')


def build_error_matrix(A, diag, omega):
    """:92-102  T = I - omega D^-1 A as a coalesced sparse matrix, differentiable in `diag`.
    The reference stacks A's scaled entries and an identity and lets coalesce() add the duplicates; here the union
    pattern is coalesced once (integer work, gla_coo_coalesce) and the two value lists are added into their slots, the
    scaled entry first, then the 1 -- the same order of addition."""
    A = _as_sparse(A, diag.device if diag.is_cuda else None)
    dev = A.device
    A_ind, A_val = A.indices(), A.values()
    n = A.shape[0]
    diag = diag.to(dev)
    T_val = -omega * (1 / _RowGather.apply(diag.reshape(-1), A_ind, n, A.shape[1])) * A_val
    pattern = getattr(A, "_error_pattern", None)
    if pattern is None:
        # pattern of A union I and the slot of every stacked entry in the coalesced list: integer work done once per
        # matrix (the training loop calls this every step with a new diagonal), which also keeps T's edge-index
        # tensor -- and with it the CSR plans of T and T^T cached on that tensor -- alive across steps
        eye = torch.arange(n, device=dev)
        stacked = torch.hstack((A_ind, torch.stack((eye, eye))))
        T_ind, _ = coalesce(stacked, torch.zeros(stacked.shape[1], dtype=A_val.dtype, device=dev), n, A.shape[1])
        keys = T_ind[0] * A.shape[1] + T_ind[1]
        slot_A = torch.searchsorted(keys, A_ind[0] * A.shape[1] + A_ind[1])
        slot_I = torch.searchsorted(keys, eye * A.shape[1] + eye)
        pattern = (T_ind, slot_A, slot_I)
        A._error_pattern = pattern
    T_ind, slot_A, slot_I = pattern
    out = torch.zeros(T_ind.shape[1], dtype=T_val.dtype, device=dev)
    out = out.index_add(0, slot_A, T_val)                       # A is coalesced: one entry per slot
    out = out.index_add(0, slot_I, torch.ones(n, dtype=T_val.dtype, device=dev))
    return SparseMatrix(T_ind, out, A.shape)


def eigval_approx(K, m, A, xy=None, method='uniform', Y=None):
    """:105-128  max_j ||A^K y_j||^(1/K); Y (normalised columns) may be passed instead of being drawn."""
    if Y is None:
        if method == 'uniform':
            Y = get_random_on_sphere(A.shape[1], m)
        elif method == 'high_freq':
            Y = get_random_high_freq(A.shape[1], m, xy)
        else:
            raise ValueError(f'method {method} not recognized')
    A = _as_sparse(A)
    Y = Y.to(device=A.device, dtype=A.dtype).contiguous()
    for _ in range(K):
        Y = A @ Y
    return torch.max(torch.linalg.vector_norm(Y, dim=0)) ** (1 / K)


def get_random_on_sphere(N, m):
    """:130-137  (drawn with the CPU generator, like the reference, so seeds reproduce its vectors)"""
    Y = torch.randn((N, m), dtype=torch.float)
    return normalize_vectors(Y)


def get_random_high_freq(N, m, xy=None):
    """:139-170  m random high-frequency sine modes on the sqrt(N) x sqrt(N) grid (CPU generator, same draw order)."""
    n = int(math.sqrt(N))
    if xy is None:
        xx = torch.zeros(n * n, 1)
        yy = torch.zeros(n * n, 1)
        # the reference fills index (j-1)*n + i with 0-based i, j (:147-151): negative indices wrap around
        j, i = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
        me = ((j - 1) * n + i).reshape(-1)
        xx[me, 0] = ((i + 1).double() / (n + 1)).reshape(-1).to(xx.dtype)      # Python-float quotient, then fp32
        yy[me, 0] = ((j + 1).double() / (n + 1)).reshape(-1).to(yy.dtype)
    else:
        xy = xy.cpu() if isinstance(xy, torch.Tensor) else torch.as_tensor(xy)
        xx = xy[:, 0]
        yy = xy[:, 1]
    Y = torch.zeros(N, m)
    nHigh = 0
    while nHigh < m:
        thetax, thetay = (n - 1) * torch.rand(2) + 1
        if thetax > n / 2 or thetay > n / 2:
            t = torch.sin(thetax * torch.pi * xx) * torch.sin(thetay * torch.pi * yy)
            Y[:, nHigh] = t.squeeze()
            nHigh += 1
    return normalize_vectors(Y)


def normalize_vectors(Y):
    """:172-176"""
    col_norm = torch.linalg.vector_norm(Y, dim=0)
    col_scale = torch.diagflat(1 / col_norm)
    return Y @ col_scale


def jacobi_dl(dvals, omega, A, x):
    """:179-183  one damped-Jacobi error propagation x - omega D^-1 A x"""
    Ax = _as_sparse(A) @ x
    DinvA = Ax / dvals
    return x - omega * DinvA


def loss_batch(model, batch):
    """:185-204  mean damping factor over the graphs of a (PyG-style) batch: batch.x, .edgeij_pair, .edge_attr,
    .batch, .num_graphs, .matrix[i], .coords[i]."""
    vertex_attr, _, _ = model(batch.x, batch.edgeij_pair, batch.edge_attr, [], batch.batch)
    omega = 2. / 3.
    loss = 0.
    for i in range(batch.num_graphs):
        A = batch.matrix[i]
        xy = batch.coords[i]
        dvals = vertex_attr[batch.batch.to(vertex_attr.device) == i]
        loss = loss + damping_factor(A, omega, dvals, xy=xy, exact=False)
    return loss / batch.num_graphs


def loss_optimal_jacobi(batch):
    """:206-220  (dense)"""
    loss = 0
    for i in range(batch.num_graphs):
        A = batch.matrix[i]
        dvals = torch.diag(A.to_dense())
        omega = optimal_omega(A, dvals)
        loss += damping_factor(A, omega, dvals, exact=True)
    return loss / batch.num_graphs


def optimal_omega(A, dvals):
    """:222-233  2 / (lmax + lmin) of D^-1 A (dense)"""
    D = torch.diagflat(dvals)
    DinvA = torch.linalg.solve(D, A.to_dense().to(D.device))
    EVals = torch.linalg.eigvals(DinvA.cpu())
    return 2 / (max(abs(EVals)) + min(abs(EVals)))
