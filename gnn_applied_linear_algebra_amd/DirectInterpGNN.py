"""Direct-interpolation weights -- mirror of /root/reference/pytorch/DirectInterpGNN.py (:155-174).

Both MetaLayers (C_j gather + two row sums + alpha_i, then the per-edge weight) are one fused row-tile kernel
(gla_direct_interp).  IEEE behaviour is kept: C rows evaluate 0 * (+-inf | NaN) = NaN exactly like the
reference; those entries sit in F columns that the V-cycle drops.
"""
import torch

from ._common import aggregate, back, compute_device, float_dtype, plan_for, to_dev


class DirectInterpGNN(torch.nn.Module):
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g=None, batch=None):
        """vertex_attr [n,2] = [A_ii, C_i]; edge_attr [nnz_od,2] = [A_ij, S_ij] -> w_ij [nnz_od] (1-D)."""
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        dt = float_dtype(va)
        if va.dim() != 2 or va.shape[1] != 2 or ea.dim() != 2 or ea.shape[1] != 2:
            raise ValueError("expected vertex_attr [n,2] = [A_ii, C_i] and edge_attr [nnz,2] = [A_ij, S_ij]")
        ea = ea.to(dt)
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        w = plan.direct_interp(va[:, 0].contiguous(), va[:, 1].contiguous(), ea[:, 1].contiguous())
        return back(w, edge_attr)


# ---- the reference's per-layer modules (DirectInterpGNN.py:50-152) --------------------------------------------------
class Layer1_EdgeUpdate(torch.nn.Module):
    """:50-69  append C_j (the C/F flag of the COLUMN vertex)  ->  [A_ij, S_ij, C_j]."""
    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        return torch.cat([edge_attr, vattr_j[:, 1:2]], 1)


def EdgeToVertexAggregation(edgeij_pair, edge_attr, n_vertices):
    """:71-97  gammabar_i = sum_k A_ik / sum_k (A_ik S_ik C_k), [n, 1]."""
    num = aggregate(edgeij_pair, edge_attr[:, 0], n_vertices, "sum")
    den = aggregate(edgeij_pair, edge_attr[:, 0] * edge_attr[:, 1] * edge_attr[:, 2], n_vertices, "sum")
    return (num / den).reshape((-1, 1))


class Layer1_VertexUpdate(torch.nn.Module):
    """:99-131  alpha_i = (1 / A_ii) * gammabar_i  ->  [A_ii, C_i, alpha_i]."""
    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        A_ii = vertex_attr[:, 0:1]
        gamma = self.edge_aggregation_function(edgeij_pair, edge_attr, vertex_attr.shape[0])
        return torch.cat([A_ii, vertex_attr[:, 1:2], (1 / A_ii) * gamma], 1)


class Layer2_EdgeUpdate(torch.nn.Module):
    """:133-152  w_ij = (1 - C_i) * (-A_ij * alpha_i) with the ROW vertex's C_i, alpha_i."""
    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        A, S = edge_attr[:, 0:1], edge_attr[:, 1:2]
        return torch.cat([A, S, (1 - vattr_i[:, 1:2]) * (-A * vattr_i[:, 2:3])], 1)
