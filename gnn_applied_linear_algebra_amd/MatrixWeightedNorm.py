"""sqrt(x^T W x) as one graph-network layer -- mirror of /root/reference/pytorch/MatrixWeightedNorm.py (:49-161).

    gnn = MetaLayer(EdgeUpdate(), VertexUpdate(EdgeToVertexAggregation), GlobalUpdate(VertexToGlobalAggregation))
    node_attr, edge_attr, g = gnn(node_attr, edgeij_pair, edge_attr, g, batch=batch)
"""
import torch

from . import meta
from ._common import back, compute_device, float_dtype, plan_for, to_dev
from .plan import col_dot


def EdgeToVertexAggregation(edgeij_pair, edge_attr, num_nodes):
    from .MatVecGNN import edge_to_vertex_aggregation
    return edge_to_vertex_aggregation(edgeij_pair, edge_attr[:, 1], num_nodes).reshape((-1, 1))


def VertexToGlobalAggregation(node_attr):
    return torch.sum(node_attr[:, 1])


class EdgeUpdate(torch.nn.Module):
    gla_block = "wnorm.edge"

    def forward(self, src, dest, edge_attr, g, batch):
        return torch.cat([edge_attr, edge_attr * dest], 1)


class VertexUpdate(torch.nn.Module):
    gla_block = "wnorm.vertex"

    def __init__(self, EdgeToVertexAggregation):
        super().__init__()
        self.EdgeToVertexAggregation = EdgeToVertexAggregation

    def forward(self, node_attr, edgeij_pair, edge_attr, g, batch):
        return torch.cat([node_attr, node_attr * self.EdgeToVertexAggregation(edgeij_pair, edge_attr,
                                                                              node_attr.shape[0])], 1)


class GlobalUpdate(torch.nn.Module):
    gla_block = "wnorm.global"

    def __init__(self, VertexToGlobalAggregation):
        super().__init__()
        self.VertexToGlobalAggregation = VertexToGlobalAggregation

    def forward(self, node_attr, edgeij_pair, edge_attr, g, batch):
        return torch.sqrt(self.VertexToGlobalAggregation(node_attr))


def _fused(layer, x, edgeij_pair, edge_attr, u, batch):
    dev = compute_device(x, edgeij_pair, edge_attr)
    xd, ei, ea = to_dev(x, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
    float_dtype(xd)
    plan = plan_for(ei, xd.shape[0], edge_attr=ea)
    X = xd.contiguous()
    Wx = plan.spmm(X)                                # (W x)_i
    y = X * Wx
    gn = torch.sqrt(col_dot(X, Wx)[0])               # sqrt(sum_i x_i (W x)_i)
    return back(torch.cat([xd, y], 1), x), back(plan.edge_attr(X), edge_attr), back(gn, x)


meta.register_fused("wnorm.edge", "wnorm.vertex", "wnorm.global", _fused)
