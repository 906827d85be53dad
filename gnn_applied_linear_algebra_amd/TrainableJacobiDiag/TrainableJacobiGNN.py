"""Learned Jacobi diagonal -- mirror of /root/reference/pytorch/TrainableJacobiDiag/TrainableJacobiGNN.py.

One MetaLayer with a vertex model only (:49-51): the vertex update concatenates the vertex attribute (A_ii) with the
min / mean / sum / max of the row's off-diagonal entries (:53-70, four torch_scatter.scatter calls in the reference,
ONE pass of gla_segment_stats here) and feeds the 5 numbers through a 5-50-20-1 ReLU MLP (:86-91).  The MLP is plain
torch (rocBLAS/hipBLASLt GEMMs with autograd); the edge attributes are data, so the aggregation needs no backward.
"""
import torch
from torch.nn import Linear, ReLU, Sequential

from .._common import compute_device, plan_for, to_dev
from ..meta import MetaLayer


def get_model():
    """:49-51"""
    return MetaLayer(None, VertexUpdate(edge_to_vertex_aggregation), None)


def edge_to_vertex_aggregation(edgeij_pair, edge_attr, num_nodes):
    """:53-70  cat(min, mean, sum, max) over the edges leaving each vertex -> [num_nodes, 4 * n_edge_attr];
    vertices without edges get 0 (torch_scatter semantics)."""
    if edge_attr.requires_grad:
        raise ValueError("edge_to_vertex_aggregation: edge attributes are data on this path (no backward kernel)")
    dev = compute_device(edge_attr, edgeij_pair)
    plan = plan_for(to_dev(edgeij_pair, dev), num_nodes)
    out = plan.segment_stats(to_dev(edge_attr, dev))
    return out if edge_attr.is_cuda else out.to(edge_attr.device)


class VertexUpdate(torch.nn.Module):
    """:72-111"""

    def __init__(self, edge_agg):
        super().__init__()
        self.edge_agg = edge_agg
        self.vertex_update = Sequential(Linear(5, 50), ReLU(), Linear(50, 20), ReLU(), Linear(20, 1))
        self.vertex_update.apply(init_weights)

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        agg_edges = self.edge_agg(edgeij_pair, edge_attr, vertex_attr.size(0))
        update_input = torch.cat([vertex_attr, agg_edges.to(vertex_attr.device)], dim=1)
        return self.vertex_update(update_input)


def init_weights(m):
    """:113-117  weights ~ U[0, 1), biases 0.01"""
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.uniform_(m.weight, a=0.0, b=1.0)
        m.bias.data.fill_(0.01)
