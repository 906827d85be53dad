"""Weighted Jacobi x <- x + w (b - A x) / A_ii -- mirror of /root/reference/pytorch/JacobiGNN.py.

`JacobiGNN().forward(n_iters, vertex_attr, edgeij_pair, edge_attr, g)` (JacobiGNN.py:138-148) runs all sweeps
as fused CSR kernels (gla_jacobi_sweeps): per sweep one pass over val/col, one read of b, A_ii, x and one
write of x; the [A_ij, c_ij] edge tensor the reference re-materialises every sweep is only produced by
`iterate()`, whose signature returns it.
"""
import torch

from ._common import back, cols, compute_device, edge_cols, float_dtype, plan_for, stack_cols, to_dev


class JacobiGNN(torch.nn.Module):
    def _prep(self, vertex_attr, edgeij_pair, edge_attr, g):
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        if va.shape[1] != 3:
            raise ValueError("vertex_attr must be [n, 3] = [A_ii, b, x]")
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        diag = (va[:, 0] if va.dim() == 2 else va[:, 0, 0]).contiguous()
        w = float(g.reshape(-1)[0])                  # w = g[0] in the tensor's own dtype (JacobiGNN.py:108)
        return plan, va, ea, diag, cols(va, 1), cols(va, 2), w

    def iterate(self, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """One sweep; returns (vertex_attr [n,3], edge_attr [nnz,2] = [A_ij, A_ij*x_j(old)], g)  (:130-135)."""
        plan, va, ea, diag, b, x, w = self._prep(vertex_attr, edgeij_pair, edge_attr, g)
        c = plan.edge_products(x)
        x_new = plan.jacobi_sweeps(diag, b, x, w, 1)
        d_col = cols(va, 0)
        return (back(stack_cols([d_col, b, x_new], va), vertex_attr),
                back(edge_cols(ea, c), edge_attr), g)

    def forward(self, n_iters, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """n_iters sweeps -> x [n,1] (or [n,1,K] for the multi-RHS extension vertex_attr [n,3,K])."""
        plan, va, ea, diag, b, x, w = self._prep(vertex_attr, edgeij_pair, edge_attr, g)
        x = plan.jacobi_sweeps(diag, b, x, w, int(n_iters))
        return back(x if va.dim() == 2 else x.unsqueeze(1), vertex_attr)
