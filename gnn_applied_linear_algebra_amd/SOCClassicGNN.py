"""Classical AMG strength of connection -- mirror of /root/reference/pytorch/SOCClassicGNN.py (:131-147).

v_i = max_{j != i}(-A_ij) (segment max, 0 for edge-less vertices) and S_ij = relu(-A_ij / v_i - theta) are one
fused row-tile kernel (gla_soc_classic): A_od is read once, S written once, in the caller's edge order.
"""
import torch

from ._common import back, compute_device, float_dtype, plan_for, to_dev


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
