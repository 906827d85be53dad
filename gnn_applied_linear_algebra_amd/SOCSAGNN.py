"""Smoothed-aggregation strength of connection -- mirror of /root/reference/pytorch/SOCSAGNN.py (:49-71).

    node_attr, edge_attr, g = MetaLayer(EdgeUpdate())(vertex_attr, edgeij_pair, edge_attr, batch=batch)
"""
import torch

from . import meta
from ._common import back, compute_device, float_dtype, plan_for, to_dev


class EdgeUpdate(torch.nn.Module):
    """[A_ij, S_ij] with S_ij = A_ij^2 / (A_ii A_jj); vertex attribute = the diagonal."""
    gla_block = "socsa.edge"

    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        return torch.cat([edge_attr, (edge_attr * edge_attr) / (vattr_i * vattr_j)], 1)


def _fused(layer, x, edgeij_pair, edge_attr, u, batch):
    dev = compute_device(x, edgeij_pair, edge_attr)
    xd, ei, ea = to_dev(x, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
    float_dtype(xd)
    plan = plan_for(ei, xd.shape[0], edge_attr=ea)
    S = plan.soc_sa(xd.reshape(-1).contiguous())
    return x, back(torch.cat([ea, S.reshape(-1, 1)], 1), edge_attr), u


meta.register_fused("socsa.edge", None, None, _fused)
