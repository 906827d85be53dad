"""r = b - A x -- mirror of /root/reference/pytorch/GNNResidual.py (GNNResidual :121-132)."""
import torch

from . import meta
from ._common import aggregate, back, cols, compute_device, float_dtype, plan_for, to_dev


def edge_to_vertex_aggregation(edgeij_pair, c_ij, n_vertices):
    """GNNResidual.py:43-62: cbar_i = sum_j c_ij."""
    return aggregate(edgeij_pair, c_ij, n_vertices, "sum")


class EdgeUpdate(torch.nn.Module):
    """GNNResidual.py:64-86: [A_ij, c_ij], c_ij = A_ij * x_j with x = column 1 of the vertex attributes [b, x]."""
    gla_block = "residual.edge"

    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        return torch.cat([edge_attr, edge_attr * vattr_j[:, 1:2]], 1)


class VertexUpdate(torch.nn.Module):
    """GNNResidual.py:88-118: [b_i, x_i, r_i] with r_i = b_i - aggregation(c_ij)."""
    gla_block = "residual.vertex"

    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        cbar = self.edge_aggregation_function(edgeij_pair, edge_attr[:, 1:2], vertex_attr.shape[0])
        return torch.cat([vertex_attr[:, 0:1], vertex_attr[:, 1:2], vertex_attr[:, 0:1] - cbar], 1)


class GNNResidual(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.gnn = meta.MetaLayer(EdgeUpdate(), VertexUpdate(edge_to_vertex_aggregation))   # wiring of :124

    def forward(self, vertex_attr, edgeij_pair, edge_attr, batch=None):
        """vertex_attr [n,2] = [b, x]; edge_attr [nnz,1] = A_ij  ->  r [n,1]   (GNNResidual.py:126-132).
        Multi-RHS extension: vertex_attr [n,2,K] -> r [n,1,K]."""
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        r = plan.residual(cols(va, 0), cols(va, 1))
        return back(r if va.dim() == 2 else r.unsqueeze(1), vertex_attr)


def _fused_layer(layer, x, edgeij_pair, edge_attr, u, batch):
    """MetaLayer(EdgeUpdate(), VertexUpdate(edge_to_vertex_aggregation)): [b, x, r] and [A_ij, c_ij] from the fused
    residual kernel + one edge-product pass."""
    if layer.node_model.edge_aggregation_function is not edge_to_vertex_aggregation or x.dim() != 2:
        return meta.generic_forward(layer, x, edgeij_pair, edge_attr, u, batch)
    dev = compute_device(x, edgeij_pair, edge_attr)
    va, ei, ea = to_dev(x, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
    float_dtype(va)
    plan = plan_for(ei, va.shape[0], edge_attr=ea)
    b, xx = cols(va, 0), cols(va, 1)
    r = plan.residual(b, xx)
    return back(torch.cat([b, xx, r], 1), x), back(plan.edge_attr(xx), edge_attr), u


meta.register_fused("residual.edge", "residual.vertex", None, _fused_layer)
