"""Classical AMG strength of connection -- mirror of /root/reference/pytorch/SOCClassicGNN.py (:131-147).

v_i = max_{j != i}(-A_ij) (segment max, 0 for edge-less vertices) and S_ij = relu(-A_ij / v_i - theta) are one
fused row-tile kernel (gla_soc_classic): A_od is read once, S written once, in the caller's edge order.
"""
import torch

from ._common import aggregate, back, compute_device, float_dtype, plan_for, to_dev


class SOCClassicGNN(torch.nn.Module):
    def __init__(self, theta):
        super().__init__()
        self.theta = theta

    def forward(self, vertex_attr, edgeij_pair, edge_attr, batch=None, return_v=False):
        """vertex_attr [n,1] (placeholder for v_i); edgeij_pair/edge_attr [nnz_od,1]: OFF-diagonal edges
        -> S_ij [nnz_od] (1-D, like edge_attr[:,1] in the reference)."""
        dev = compute_device(edge_attr, edgeij_pair, vertex_attr)
        ei, ea = to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(ea)
        plan = plan_for(ei, vertex_attr.shape[0], edge_attr=ea)
        v, S = plan.soc_classic(self.theta, ea.dtype)
        return (back(S, edge_attr), back(v, edge_attr)) if return_v else back(S, edge_attr)


# ---- the reference's per-layer modules (SOCClassicGNN.py:50-129) ----------------------------------------------------
def edge_to_vertex_aggregation(edgeij_pair, edge_attr, n_vertices):
    """:50-72  cbar_i = max_{j != i} -A_ij (0 for vertices without edges), [n, 1]."""
    return aggregate(edgeij_pair, -1 * edge_attr[:, 0], n_vertices, "max").reshape((-1, 1))


class VertexUpdate(torch.nn.Module):
    """:74-102  v_i = cbar_i (replaces the placeholder vertex attribute)."""
    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return self.edge_aggregation_function(edgeij_pair, edge_attr, vertex_attr.shape[0])


class EdgeUpdate(torch.nn.Module):
    """:104-129  [A_ij, relu(-A_ij / v_i - theta)], v taken at the ROW vertex."""
    def __init__(self, theta):
        super().__init__()
        self.theta = theta

    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        A = edge_attr[:, 0:1]
        return torch.cat([A, torch.relu(-1 * A / vattr_i[:, 0:1] - self.theta)], 1)
