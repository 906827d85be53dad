"""r = b - A x -- mirror of /root/reference/pytorch/GNNResidual.py (GNNResidual :121-132)."""
import torch

from ._common import back, cols, compute_device, float_dtype, plan_for, to_dev


class GNNResidual(torch.nn.Module):
    def forward(self, vertex_attr, edgeij_pair, edge_attr, batch=None):
        """vertex_attr [n,2] = [b, x]; edge_attr [nnz,1] = A_ij  ->  r [n,1]   (GNNResidual.py:126-132).
        Multi-RHS extension: vertex_attr [n,2,K] -> r [n,1,K]."""
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        r = plan.residual(cols(va, 0), cols(va, 1))
        return back(r if va.dim() == 2 else r.unsqueeze(1), vertex_attr)
