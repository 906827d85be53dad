"""Weighted Jacobi x <- x + w (b - A x) / A_ii -- mirror of /root/reference/pytorch/JacobiGNN.py.

`JacobiGNN().forward(n_iters, vertex_attr, edgeij_pair, edge_attr, g)` (JacobiGNN.py:138-148) runs all sweeps
as fused CSR kernels (gla_jacobi_sweeps): per sweep one pass over val/col, one read of b, A_ii, x and one
write of x; the [A_ij, c_ij] edge tensor the reference re-materialises every sweep is only produced by
`iterate()`, whose signature returns it.
"""
import torch

from . import meta
from ._common import aggregate, back, cols, compute_device, float_dtype, plan_for, scalar0, split_cols, stack_cols, to_dev


def edge_to_vertex_aggregation(edgeij_pair, edge_attr, n_vertices):
    """JacobiGNN.py:52-69: cbar_i = sum_j c_ij (c = column 1 of the edge attributes), as a column vector."""
    return aggregate(edgeij_pair, edge_attr[:, 1], n_vertices, "sum").reshape(-1, 1)


class EdgeUpdate(torch.nn.Module):
    """JacobiGNN.py:71-89: [A_ij, c_ij], c_ij = A_ij * x_j with x = column 2 of the vertex attributes."""
    gla_block = "jacobi.edge"

    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        A = edge_attr[:, 0:1]
        return torch.cat([A, A * vattr_j[:, 2:3]], 1)


class VertexUpdate(torch.nn.Module):
    """JacobiGNN.py:91-123: [A_ii, b_i, x_i + w (b_i - cbar_i) / A_ii], w = g[0]."""
    gla_block = "jacobi.vertex"

    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        A_ii, b, x = vertex_attr[:, 0:1], vertex_attr[:, 1:2], vertex_attr[:, 2:3]
        cbar = self.edge_aggregation_function(edgeij_pair, edge_attr, vertex_attr.shape[0])
        return torch.cat([A_ii, b, x + g[0] * (b - cbar) / A_ii], 1)


def _prep(vertex_attr, edgeij_pair, edge_attr, g, views=False):
    dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
    va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
    float_dtype(va)
    if va.shape[1] != 3:
        raise ValueError("vertex_attr must be [n, 3] = [A_ii, b, x]")
    plan = plan_for(ei, va.shape[0], edge_attr=ea)
    w = scalar0(g)                               # w = g[0] in the tensor's own dtype (JacobiGNN.py:108)
    if va.dim() == 2:
        d_col, b, x = split_cols(va)             # one pass over the AoS rows
        return plan, va, ea, d_col.reshape(-1), b, x, w
    if views and va.stride(2) == 1:
        # [n, 3, K]: b and x are row-major [n, K] operands with leading dimension 3 K as they stand -- the library takes
        # leading dimensions, so the K-wide sweeps read them in place (two contiguous copies were 8.6 GB at 4096^2 x 16)
        return plan, va, ea, va[:, 0, 0].contiguous(), va[:, 1, :], va[:, 2, :], w
    return plan, va, ea, va[:, 0, 0].contiguous(), cols(va, 1), cols(va, 2), w


def _iterate(vertex_attr, edgeij_pair, edge_attr, g):
    """One fused sweep; returns (vertex_attr [n,3], edge_attr [nnz,2] = [A_ij, A_ij*x_j(old)], g)."""
    plan, va, ea, diag, b, x, w = _prep(vertex_attr, edgeij_pair, edge_attr, g)
    ea_out = plan.edge_attr(x)                         # [A_ij, A_ij * x_j(old)] in one pass
    x_new = plan.jacobi_sweeps(diag, b, x, w, 1)
    return (back(stack_cols([cols(va, 0), b, x_new], va), vertex_attr), back(ea_out, edge_attr), g)


class JacobiGNN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.gnn = meta.MetaLayer(EdgeUpdate(), VertexUpdate(edge_to_vertex_aggregation))   # wiring of :128

    def iterate(self, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """One sweep (JacobiGNN.py:130-135): self.gnn is recognised by MetaLayer and runs as one fused kernel."""
        return self.gnn(vertex_attr, edgeij_pair, edge_attr, g, batch=batch)

    def forward(self, n_iters, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """n_iters sweeps -> x [n,1] (or [n,1,K] for the multi-RHS extension vertex_attr [n,3,K])."""
        if vertex_attr.dim() == 2 and int(n_iters) >= 2:
            # AoS fast path: the first temporally blocked pass reads [A_ii, b, x] rows itself (no separate column split)
            dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
            va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
            float_dtype(va)
            if va.shape[1] != 3:
                raise ValueError("vertex_attr must be [n, 3] = [A_ii, b, x]")
            plan = plan_for(ei, va.shape[0], edge_attr=ea)
            x = plan.jacobi_sweeps_aos(va, scalar0(g), int(n_iters))
            if x is not None:
                return back(x, vertex_attr)
        plan, va, ea, diag, b, x, w = _prep(vertex_attr, edgeij_pair, edge_attr, g, views=True)
        x = plan.jacobi_sweeps(diag, b, x, w, int(n_iters))
        return back(x if va.dim() == 2 else x.unsqueeze(1), vertex_attr)


def _fused_layer(layer, x, edgeij_pair, edge_attr, u, batch):
    """MetaLayer(EdgeUpdate(), VertexUpdate(edge_to_vertex_aggregation)) == one fused sweep."""
    if layer.node_model.edge_aggregation_function is not edge_to_vertex_aggregation:
        return meta.generic_forward(layer, x, edgeij_pair, edge_attr, u, batch)
    return _iterate(x, edgeij_pair, edge_attr, u)


meta.register_fused("jacobi.edge", "jacobi.vertex", None, _fused_layer)
