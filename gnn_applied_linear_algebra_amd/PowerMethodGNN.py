"""Power iteration + Rayleigh quotient -- mirror of /root/reference/pytorch/PowerMethodGNN.py (:296-334).

Per iteration the reference runs three MetaLayers (b <- A b; n = sqrt(sum b^2); b <- b / n); here an iteration is
ONE fused kernel (gla_power_step): the normalisation b = y / n of the previous iteration is applied to the gathered
operand (reciprocal + one correction step: the reference's quotient except for rare last-place cases), y = A b is stored once and the workgroups leave their partial
sums of y^2 for a fixed-order second stage -- 16 n bytes of vector traffic per iteration instead of 40 n.  Plans /
K without a fused kernel (long rows, odd K > 1) use gla_spmm + gla_col_dot + gla_col_scale_rsqrt.  The Rayleigh
block is an SpMV plus two reductions.
"""
import torch

from . import _lib
from ._common import back, cols, compute_device, edge_cols, float_dtype, plan_for, stack_cols, to_dev
from .plan import col_dot, col_scale_rsqrt


class PowerMethodGNN(torch.nn.Module):
    def __init__(self, num_iter, return_edge_attr=True):
        super().__init__()
        self.num_iter = int(num_iter)
        self.return_edge_attr = return_edge_attr

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """vertex_attr [n,2] = [b, y]; edge_attr [nnz,2] = [A_ij, c_ij]; g [3] = [n, n_A, lambda]
        -> (vertex_attr [n,2] = [b, b^2], edge_attr [nnz,2] = [A_ij, A_ij*b_j], g [3]); lambda_max = g[2].
        Multi-RHS extension: vertex_attr [n,2,K], g [3,K] runs K independent power iterations."""
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        if va.shape[1] != 2:
            raise ValueError("vertex_attr must be [n, 2] = [b, y]")
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        b = cols(va, 0)
        K = b.shape[1]
        out_dtype = torch.result_type(va, g)
        gd = to_dev(g, dev).to(out_dtype).reshape(3, -1)
        nrm, n_A = gd[0].expand(K).clone(), gd[1].expand(K).clone()
        if self.num_iter > 0:
            try:
                y, ss, work = plan.power_step(b)                 # iteration 1: b is used as given
                for _ in range(self.num_iter - 1):
                    y, ss, work = plan.power_step(y, nrm=torch.sqrt(ss), work=work)
            except _lib.GlaError as e:
                if "error -5" not in str(e):
                    raise
                y = b
                for it in range(self.num_iter):                  # unfused path, same arithmetic
                    y = plan.spmm(y if it == 0 else col_scale_rsqrt(y, ss))
                    ss = col_dot(y, y)
            b = col_scale_rsqrt(y, ss)                           # Layer 3 of the last iteration: b <- b / n
            nrm = torch.sqrt(ss).to(out_dtype)
        Ab = plan.spmm(b)                             # Rayleigh layer 1: y = b * (A b); n_A = sum y  (:209-266)
        n_A = col_dot(b, Ab).to(out_dtype)
        lam = n_A / col_dot(b, b).to(out_dtype)       # Rayleigh layer 2: lambda = n_A / sum b^2      (:268-294)
        g_out = torch.stack([nrm, n_A, lam], 0)
        g_out = g_out.reshape(3) if g.dim() == 1 else g_out
        ea_out = edge_cols(ea, plan.edge_products(b)) if self.return_edge_attr else None
        return (back(stack_cols([b, b * b], va), vertex_attr), back(ea_out, edge_attr), back(g_out, g))
