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


def _dense_jacobi_spectrum(A, omega, diagonal_value):
    """Eigenvalues of I - omega D^-1 A from dense factorisations (validation only: O(n^2) memory, host eigen-solver)."""
    M = A.to_dense()
    scaled = torch.linalg.solve(torch.diagflat(diagonal_value).to(M.device), M)
    J = torch.eye(A.shape[0], device=M.device, dtype=M.dtype) - omega * scaled
    return torch.linalg.eigvals(J.cpu())


def damping_factor(A, omega, diagonal_value, xy=None, exact=False):
    """:43-90  spectral radius of I - omega * (D \\ A).  exact=True: dense eigenvalues (no autograd, small matrices);
    default: Gelfand's formula with K = 3 products on m = 20 high-frequency test vectors (sparse, differentiable)."""
    if exact:
        return max(abs(_dense_jacobi_spectrum(A, omega, diagonal_value)))
    return eigval_approx(3, 20, build_error_matrix(A, diagonal_value, omega), xy=xy, method='high_freq')


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
    """:130-137  m Gaussian directions, normalised (CPU generator, like the reference, so seeds reproduce its vectors)."""
    return normalize_vectors(torch.randn((N, m), dtype=torch.float))


def _grid_coordinates(n):
    """The reference's default coordinates (:143-151): entry (j-1)*n + i holds ((i+1)/(n+1), (j+1)/(n+1)) for 0-based
    i, j -- the negative indices of j = 0 wrap to the end, which is reproduced, not corrected."""
    j, i = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
    slot = ((j - 1) * n + i).reshape(-1)
    xx, yy = torch.zeros(n * n, 1), torch.zeros(n * n, 1)
    xx[slot, 0] = ((i + 1).double() / (n + 1)).reshape(-1).to(xx.dtype)      # Python-float quotient, then fp32
    yy[slot, 0] = ((j + 1).double() / (n + 1)).reshape(-1).to(yy.dtype)
    return xx, yy


def get_random_high_freq(N, m, xy=None):
    """:139-170  m random high-frequency sine modes sin(pi tx x) sin(pi ty y) on the sqrt(N) x sqrt(N) grid, normalised.
    Frequencies are drawn two at a time from the CPU generator and rejected unless one of them exceeds n/2 -- the
    reference's draw order, so a seed gives the reference's vectors."""
    n = int(math.sqrt(N))
    if xy is None:
        xx, yy = _grid_coordinates(n)
    else:
        pts = xy.cpu() if isinstance(xy, torch.Tensor) else torch.as_tensor(xy)
        xx, yy = pts[:, 0], pts[:, 1]
    modes = []
    while len(modes) < m:
        tx, ty = (n - 1) * torch.rand(2) + 1
        if tx > n / 2 or ty > n / 2:
            modes.append((torch.sin(tx * torch.pi * xx) * torch.sin(ty * torch.pi * yy)).reshape(-1))
    return normalize_vectors(torch.stack(modes, 1).to(torch.float).reshape(N, m))


def normalize_vectors(Y):
    """:172-176  unit columns (multiplication by the reciprocal norms, as the reference's Y @ diag(1/norm))."""
    return Y * (1 / torch.linalg.vector_norm(Y, dim=0))


def jacobi_dl(dvals, omega, A, x):
    """:179-183  one damped-Jacobi error propagation x - omega D^-1 A x"""
    return x - omega * ((_as_sparse(A) @ x) / dvals)


def loss_batch(model, batch):
    """:185-204  mean damping factor over the graphs of a (PyG-style) batch: batch.x, .edgeij_pair, .edge_attr,
    .batch, .num_graphs, .matrix[i], .coords[i]; the model's output are the learned diagonals."""
    learned, _, _ = model(batch.x, batch.edgeij_pair, batch.edge_attr, [], batch.batch)
    owner = batch.batch.to(learned.device)
    total = 0.
    for g in range(batch.num_graphs):
        total = total + damping_factor(batch.matrix[g], 2. / 3., learned[owner == g], xy=batch.coords[g])
    return total / batch.num_graphs


def optimal_omega(A, dvals):
    """:222-233  2 / (lmax + lmin) of |eig(D^-1 A)| (dense)"""
    dense = A.to_dense()
    spectrum = abs(torch.linalg.eigvals(torch.linalg.solve(torch.diagflat(dvals).to(dense.device), dense).cpu()))
    return 2 / (max(spectrum) + min(spectrum))


def loss_optimal_jacobi(batch):
    """:206-220  mean exact damping factor with the matrix' own diagonal and its optimal omega (dense)"""
    total = 0
    for A in batch.matrix[:batch.num_graphs]:
        own_diag = torch.diag(A.to_dense())
        total += damping_factor(A, optimal_omega(A, own_diag), own_diag, exact=True)
    return total / batch.num_graphs
