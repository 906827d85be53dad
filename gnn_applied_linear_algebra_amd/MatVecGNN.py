"""y = A x as one graph-network layer -- mirror of /root/reference/pytorch/MatVecGNN.py.

    gnn = MetaLayer(EdgeUpdate(), VertexUpdate(edge_to_vertex_aggregation))
    vertex_attr, edge_attr, _ = gnn(x, edgeij_pair, A_ij, None, batch=batch)      # MatVecGNN.py:146-150

returns cat[x, A x] [n, 2k] and cat[A_ij, c_ij] [nnz, 1+k] like the reference, computed by the fused CSR
SpMV/SpMM kernel (gla_spmm) plus one edge-product pass (gla_edge_products) for the returned c_ij.
"""
import torch

from . import meta
from ._common import aggregate, back, compute_device, float_dtype, plan_for, to_dev
from .meta import MetaLayer  # noqa: F401  (re-exported: the reference imports it from torch_geometric.nn)


def edge_to_vertex_aggregation(edgeij_pair, c_ij, n_vertices):
    """MatVecGNN.py:43-62: cbar_i = sum_j c_ij (edge-order row sums) for every attribute column of c_ij."""
    return aggregate(edgeij_pair, c_ij, n_vertices, "sum")


class EdgeUpdate(torch.nn.Module):
    """MatVecGNN.py:64-84: [A_ij, c_ij] with c_ij = A_ij * x_j (broadcast over the k columns of x)."""
    gla_block = "matvec.edge"

    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        return torch.cat([edge_attr, edge_attr * vattr_j], 1)


class VertexUpdate(torch.nn.Module):
    """MatVecGNN.py:86-114: [x_i, y_i] with y_i = aggregation of c_ij."""
    gla_block = "matvec.vertex"

    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        c_ij = edge_attr[:, 1:1 + vertex_attr.shape[1]]
        return torch.cat([vertex_attr, self.edge_aggregation_function(edgeij_pair, c_ij, vertex_attr.shape[0])], 1)


def _fused_matvec(layer, x, edgeij_pair, edge_attr, u, batch):
    if layer.node_model.edge_aggregation_function is not edge_to_vertex_aggregation:
        return None
    dev = compute_device(x, edgeij_pair, edge_attr)
    xd, ei, ea = to_dev(x, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
    float_dtype(xd)
    if xd.dim() != 2 or ea.dim() != 2 or ea.shape[1] != 1:
        raise ValueError("MatVec expects x [n, k] and edge_attr [nnz, 1]")
    plan = plan_for(ei, xd.shape[0], edge_attr=ea)
    X = xd.contiguous()
    y = plan.spmm(X)
    return back(torch.cat([xd, y], 1), x), back(plan.edge_attr(X), edge_attr), u


def _dispatch(layer, x, ei, ea, u, batch):
    out = _fused_matvec(layer, x, ei, ea, u, batch)
    if out is None:                                   # user-supplied aggregation: generic PyG sequence
        return meta.generic_forward(layer, x, ei, ea, u, batch)
    return out


meta.register_fused("matvec.edge", "matvec.vertex", None, _dispatch)
