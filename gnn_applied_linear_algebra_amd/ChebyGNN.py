"""Chebyshev relaxation of degree `deg` -- mirror of /root/reference/pytorch/ChebyGNN.py (ChebyRelaxGNN :287-353).

The reference chains 2*deg MetaLayers (an SpMV layer and a vector-update layer per degree, :291-321).  Here each
degree is ONE fused kernel: gla_cheby_first (r = b - A x; p = r; x += alpha p) and gla_cheby_next
(r -= alpha_old A p; p = r + beta p; x += alpha_new p).  The scalar recurrences of the global updates
(:123-139, :246-285) are evaluated on the host with 0-dim tensors of g's dtype in the reference's exact
operation order, so alpha/beta are bit-identical.
"""
import torch

from ._common import back, cols, compute_device, edge_cols, float_dtype, plan_for, stack_cols, to_dev


def chebyshev_relax(plan, b, x0, g, deg):
    """Core of ChebyRelaxGNN on SoA operands: b, x0 [n, K] contiguous, g = [c, d] (any device).  Runs `deg` fused
    steps on `plan` and returns (x, r, p, gathered, g_out) where `gathered` is the vector the last SpMV read
    (x0 for deg 1, the previous p otherwise -- the z_ij = A_ij * gathered_j of the returned edge attributes)."""
    gh = g.detach().to("cpu")
    c, d = gh[0], gh[1]
    alpha = 1 / d                                                    # ChebyGNN.py:137
    beta = None
    r, p, x = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
    plan.cheby_first(b, x0, r, p, x, float(alpha))
    gathered = x0
    p_next = torch.empty_like(p) if deg > 1 else None
    for it in range(2, deg + 1):
        beta = 0.5 * (c * alpha) ** 2 if it == 2 else ((c * alpha) / 2) ** 2      # :262 / :282
        alpha_new = 1 / (d - beta / alpha)                                          # :263 / :283
        plan.cheby_next(p, p_next, r, x, float(alpha), float(alpha_new), float(beta))
        gathered, p, p_next = p, p_next, p
        alpha = alpha_new
    g_out = torch.hstack([c, d, alpha] if beta is None else [c, d, alpha, beta]).to(g.device)
    return x, r, p, gathered, g_out


class ChebyRelaxGNN(torch.nn.Module):
    def __init__(self, deg=3, return_edge_attr=True):
        super().__init__()
        self.deg = int(deg)
        self.return_edge_attr = return_edge_attr      # False: skip materialising z_ij (returns edge_attr=None)

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """vertex_attr [n,2] = [b, x]; edge_attr [nnz,1]; g [2] = [c, d]
        -> (vertex_attr [n,4] = [b, x, r, p], edge_attr [nnz,2] = [A_ij, z_ij], g [4] = [c, d, alpha, beta])."""
        if self.deg <= 0:
            return vertex_attr, edge_attr, g
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        if va.shape[1] != 2:
            raise ValueError("vertex_attr must be [n, 2] = [b, x]")
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        b, x0 = cols(va, 0), cols(va, 1)
        x, r, p, gathered, g_out = chebyshev_relax(plan, b, x0, g, self.deg)
        ea_out = edge_cols(ea, plan.edge_products(gathered)) if self.return_edge_attr else None
        return back(stack_cols([b, x, r, p], va), vertex_attr), back(ea_out, edge_attr), g_out
