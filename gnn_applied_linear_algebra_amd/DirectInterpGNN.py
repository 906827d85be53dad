"""Direct-interpolation weights -- mirror of /root/reference/pytorch/DirectInterpGNN.py (:155-174).

Both MetaLayers (C_j gather + two row sums + alpha_i, then the per-edge weight) are one fused row-tile kernel
(gla_direct_interp).  IEEE behaviour is kept: C rows evaluate 0 * (+-inf | NaN) = NaN exactly like the
reference; those entries sit in F columns that the V-cycle drops.
"""
import torch

from ._common import back, compute_device, float_dtype, plan_for, to_dev


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
