"""Differentiable sparse products on the libgla_hip kernels.

The reference's learned-Jacobi loss differentiates through `A @ Y` on torch sparse tensors
(/root/reference/pytorch/TrainableJacobiDiag/loss.py:92-128, train.py:105 `loss.backward()`).  Here the forward is
gla_spmm on the CSR plan of the edge list; the backward is
    dL/dvalues[e] = sum_k G[row_e, k] * X[col_e, k]      (gla_edge_dot, edge order)
    dL/dX         = A^T G                                (gla_spmm on the plan of the flipped edge list)
`SparseMatrix` is the small operand type that lets reference-style code (`T @ Y`, `.coalesce()`, `.indices()`,
`.values()`, `.shape`) run unchanged.  No CPU fallback: operands must live on the GPU.
"""
import torch

from .plan import plan_for


def _transposed_plan(plan, edgeij_pair):
    """Plan of A^T, cached on A's plan (the flipped edge list is kept alive with it)."""
    cached = getattr(plan, "_transposed", None)
    if cached is None:
        ei_t = edgeij_pair.flip(0).contiguous()
        cached = (ei_t, plan_for(ei_t, plan.n_cols, plan.n_rows))
        plan._transposed = cached
    return cached[1]


class _SparseMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, X, edgeij_pair, n_rows, n_cols):
        plan = plan_for(edgeij_pair, n_rows, n_cols)
        plan.set_values_cached(values)
        ctx.save_for_backward(values, X)
        ctx.edgeij_pair, ctx.shape = edgeij_pair, (n_rows, n_cols)
        return plan.spmm(X if X.dim() == 2 else X.reshape(-1, 1)).reshape((n_rows,) + tuple(X.shape[1:]))

    @staticmethod
    def backward(ctx, G):
        values, X = ctx.saved_tensors
        n_rows, n_cols = ctx.shape
        plan = plan_for(ctx.edgeij_pair, n_rows, n_cols)
        G2 = G.reshape(n_rows, -1).contiguous()
        X2 = X.reshape(n_cols, -1).contiguous()
        d_values = d_X = None
        if ctx.needs_input_grad[0]:
            d_values = plan.edge_dot(G2, X2).reshape(values.shape)
        if ctx.needs_input_grad[1]:
            plan_t = _transposed_plan(plan, ctx.edgeij_pair)
            plan_t.set_values_cached(values)
            d_X = plan_t.spmm(G2).reshape(X.shape)
        return d_values, d_X, None, None, None


def sparse_mm(edgeij_pair, values, X, n_rows, n_cols=None):
    """Y = A X for A given as an edge list (row 0 = matrix row, row 1 = column) with `values` [nnz] or [nnz, 1];
    differentiable with respect to `values` and `X`.  X: [n_cols] or [n_cols, K]."""
    n_cols = X.shape[0] if n_cols is None else n_cols
    if not (values.is_cuda and X.is_cuda and edgeij_pair.is_cuda):
        raise ValueError("sparse_mm expects GPU tensors (there is no CPU path)")
    if values.dtype != X.dtype:
        raise TypeError(f"values are {values.dtype}, X is {X.dtype}")
    return _SparseMM.apply(values, X, edgeij_pair, int(n_rows), int(n_cols))


class SparseMatrix:
    """Coalesced COO operand (indices int64 [2, nnz], values [nnz], shape) with `@` on the GPU kernels."""

    def __init__(self, indices, values, shape):
        if indices.dim() != 2 or indices.shape[0] != 2 or values.shape[0] != indices.shape[1]:
            raise ValueError("indices must be [2, nnz] and values [nnz]")
        self._indices, self._values, self.shape = indices, values.reshape(-1), (int(shape[0]), int(shape[1]))

    @classmethod
    def from_torch(cls, A, device=None):
        """From a torch sparse COO tensor (what the reference's datasets hold, Data.py)."""
        A = A.coalesce()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        return cls(A.indices().to(dev), A.values().to(dev), A.shape)

    def coalesce(self):
        return self

    def indices(self):
        return self._indices

    def values(self):
        return self._values

    @property
    def device(self):
        return self._values.device

    @property
    def dtype(self):
        return self._values.dtype

    def __matmul__(self, X):
        return sparse_mm(self._indices, self._values, X, self.shape[0], self.shape[1])

    def to_dense(self):
        return torch.sparse_coo_tensor(self._indices, self._values, self.shape).to_dense()

    def to_torch(self):
        return torch.sparse_coo_tensor(self._indices, self._values, self.shape).coalesce()


__all__ = ["sparse_mm", "SparseMatrix"]
