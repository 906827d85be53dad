"""Power iteration + Rayleigh quotient -- mirror of /root/reference/pytorch/PowerMethodGNN.py (:296-334).

Per iteration the reference runs three MetaLayers (b <- A b; n = sqrt(sum b^2); b <- b / n); here an iteration is
ONE fused kernel (gla_power_step): the normalisation b = y / n of the previous iteration is applied to the gathered
operand (reciprocal + one correction step: the reference's quotient except for rare last-place cases), y = A b is stored once and the workgroups leave their partial
sums of y^2 for a fixed-order second stage -- 16 n bytes of vector traffic per iteration instead of 40 n.  Plans /
K without a fused kernel (long rows, odd K > 1) use gla_spmm + gla_col_dot + gla_col_scale_rsqrt.  The Rayleigh
block is an SpMV plus two reductions.
"""
import torch

from . import _lib
from ._common import aggregate, back, cols, compute_device, float_dtype, plan_for, stack_cols, to_dev
from .plan import col_dot, col_scale_rsqrt


class PowerMethodGNN(torch.nn.Module):
    def __init__(self, num_iter, return_edge_attr=True):
        super().__init__()
        self.num_iter = int(num_iter)
        self.return_edge_attr = return_edge_attr

    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch=None):
        """vertex_attr [n,2] = [b, y]; edge_attr [nnz,2] = [A_ij, c_ij]; g [3] = [n, n_A, lambda]
        -> (vertex_attr [n,2] = [b, b^2], edge_attr [nnz,2] = [A_ij, A_ij*b_j], g [3]); lambda_max = g[2].
        Multi-RHS extension: vertex_attr [n,2,K], g [3,K] runs K independent power iterations."""
        dev = compute_device(vertex_attr, edgeij_pair, edge_attr)
        va, ei, ea = to_dev(vertex_attr, dev), to_dev(edgeij_pair, dev), to_dev(edge_attr, dev)
        float_dtype(va)
        if va.shape[1] != 2:
            raise ValueError("vertex_attr must be [n, 2] = [b, y]")
        plan = plan_for(ei, va.shape[0], edge_attr=ea)
        # multi-RHS: b is read in place from the AoS input and [b, b^2] written straight into the AoS result
        in_place = va.dim() == 3 and va.shape[2] > 1 and va.is_contiguous()
        b = va[:, 0, :] if in_place else cols(va, 0)
        res = torch.empty_like(va) if in_place else None
        K = b.shape[1]
        out_dtype = torch.result_type(va, g)
        gd = to_dev(g, dev).to(out_dtype).reshape(3, -1)
        nrm, n_A = gd[0].expand(K).clone(), gd[1].expand(K).clone()
        tail_done, fused_tail_ok = False, False
        nrm_reported = None
        if self.num_iter > 0:
            tmax = plan.power_fused_max_iters(K, b.dtype) if self.num_iter >= 2 else 0
            blocked = False
            if tmax >= 2:
                # temporally blocked: ceil(num_iter / tmax) passes of equal size +- 1, each m iterations in ONE pass over the
                # matrix (gla_power_fused); the norm the reference reports after an iteration is the ratio of two consecutive
                # levels' norms, the next pass normalises by the absolute norm of the pass' result
                kp = -(-self.num_iter // tmax)
                base, rem = divmod(self.num_iter, kp)
                try:
                    y, nrm_abs = b, None
                    for i in range(kp):
                        m = base + (1 if i < rem else 0)
                        y, ssl = plan.power_fused(y, m, nrm=nrm_abs)
                        nrm_abs = torch.sqrt(ssl[1])
                        nrm_reported = torch.sqrt(ssl[1] / ssl[0]) if m >= 2 else nrm_abs
                    ss = ssl[1]
                    blocked = fused_tail_ok = True
                except _lib.GlaError as e:                           # (operands the K-wide march does not take: step by step)
                    if "error -5" not in str(e):
                        raise
                    nrm_reported = None
            if not blocked:
                try:
                    y, ss, work = plan.power_step(b)                 # iteration 1: b is used as given
                    fused_tail_ok = True
                    for _ in range(self.num_iter - 1):
                        y, ss, work = plan.power_step(y, nrm=torch.sqrt(ss), work=work)
                except _lib.GlaError as e:
                    if "error -5" not in str(e):
                        raise
                    y = b
                    for it in range(self.num_iter):                  # unfused path, same arithmetic
                        y = plan.spmm(y if it == 0 else col_scale_rsqrt(y, ss))
                        ss = col_dot(y, y)
            nrm_dev = torch.sqrt(ss)                             # (the absolute norm of y: what the last layer 3 divides by)
            nrm = (nrm_dev if nrm_reported is None else nrm_reported).to(out_dtype)
            # Layer 3 of the last iteration (b <- b / n), the Rayleigh layers' A b, n_A = sum b (A b), sum b^2 and the
            # module's [b, b^2] result in ONE pass where the plan has the blocked stencil kernel (gla_power_rayleigh)
            if in_place and fused_tail_ok:
                sums = plan.power_rayleigh(y, nrm_dev, res[:, 0, :], res[:, 1, :])
                if sums is not None:
                    tail_done = True
                    b = res[:, 0, :]
                    n_A = sums[0].to(out_dtype)
                    lam = n_A / sums[1].to(out_dtype)
            if not tail_done:
                b = col_scale_rsqrt(y, ss, out=res[:, 0, :] if in_place else None)
        elif in_place:
            res[:, 0, :] = b
            b = res[:, 0, :]
        if not tail_done:
            Ab = plan.spmm(b)                             # Rayleigh layer 1: y = b * (A b); n_A = sum y  (:209-266)
            n_A = col_dot(b, Ab).to(out_dtype)
            lam = n_A / col_dot(b, b).to(out_dtype)       # Rayleigh layer 2: lambda = n_A / sum b^2      (:268-294)
        g_out = torch.stack([nrm, n_A, lam], 0)
        g_out = g_out.reshape(3) if g.dim() == 1 else g_out
        ea_out = plan.edge_attr(b) if self.return_edge_attr else None           # [A_ij, A_ij * b_j] in one pass
        if tail_done:
            pass                                          # [b, b^2] already written by the fused tail
        elif in_place:
            torch.mul(b, b, out=res[:, 1, :])
        else:
            res = stack_cols([b, b * b], va)
        return (back(res, vertex_attr), back(ea_out, edge_attr), back(g_out, g))


# ---- the reference's per-layer update modules (PowerMethodGNN.py:50-294) -------------------------------------------
# vertex attributes [b, y], edge attributes [A_ij, c_ij], globals [n, n_A, lambda].
def vertex_to_global_aggregation(vertex_attr):
    """:50-61  ybar = sum_i y_i (deterministic two-stage column sum on the device)."""
    y = vertex_attr[:, 1:2]
    if not y.is_cuda:
        return torch.sum(vertex_attr[:, 1])
    y = y.contiguous()
    return col_dot(y, torch.ones_like(y))[0]


def edge_to_vertex_aggregation(edgeij_pair, edge_attr, n_vertices):
    """:64-83  cbar_i = sum_j c_ij, [n, 1]."""
    return aggregate(edgeij_pair, edge_attr[:, 1], n_vertices, "sum").reshape((-1, 1))


class Layer_1_EdgeUpdate(torch.nn.Module):
    """:86-106  c_ij = A_ij * b_j."""
    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        A = edge_attr[:, 0:1]
        return torch.cat([A, A * vattr_j[:, 0:1]], 1)


class Layer_2_VertexUpdate(torch.nn.Module):
    """:109-126  y_i = b_i^2."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b = vertex_attr[:, 0:1]
        return torch.cat([b, b * b], 1)


class _EdgeAgg(torch.nn.Module):
    def __init__(self, edge_aggregation_function):
        super().__init__()
        self.edge_aggregation_function = edge_aggregation_function


class _GlobalAgg(torch.nn.Module):
    def __init__(self, vertex_to_global_aggregation_function):
        super().__init__()
        self.vertex_to_global_aggregation_function = vertex_to_global_aggregation_function


class Iterative_Layer_1_VertexUpdate(_EdgeAgg):
    """:129-158  b_i <- cbar_i."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.cat([self.edge_aggregation_function(edgeij_pair, edge_attr, vertex_attr.shape[0]),
                          vertex_attr[:, 1:2]], 1)


class Iterative_Layer_2_GlobalUpdate(_GlobalAgg):
    """:160-185  n = sqrt(ybar)."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.stack([torch.sqrt(self.vertex_to_global_aggregation_function(vertex_attr)).to(g.device),
                            g[1], g[2]])


class Iterative_Layer_3_VertexUpdate(torch.nn.Module):
    """:187-207  b_i <- b_i / n."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.cat([vertex_attr[:, 0:1] / g[0].to(vertex_attr.device), vertex_attr[:, 1:2]], 1)


class Rayleigh_Layer_1_VertexUpdate(_EdgeAgg):
    """:209-237  y_i = b_i * cbar_i."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b = vertex_attr[:, 0:1]
        return torch.cat([b, b * self.edge_aggregation_function(edgeij_pair, edge_attr, vertex_attr.shape[0])], 1)


class Rayleigh_Layer_1_GlobalUpdate(_GlobalAgg):
    """:239-266  n_A = ybar."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.stack([g[0], self.vertex_to_global_aggregation_function(vertex_attr).to(g.device), g[2]])


class Rayleigh_Layer_2_GlobalUpdate(_GlobalAgg):
    """:268-294  lambda = n_A / ybar."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.stack([g[0], g[1], g[1] / self.vertex_to_global_aggregation_function(vertex_attr).to(g.device)])


def unfused_layers(num_iter):
    """The reference's layer list (PowerMethodGNN.py:303-329) from the modules above and this package's MetaLayer."""
    from .meta import MetaLayer
    e2v, v2g = edge_to_vertex_aggregation, vertex_to_global_aggregation
    block = [MetaLayer(Layer_1_EdgeUpdate(), Iterative_Layer_1_VertexUpdate(e2v), None),
             MetaLayer(None, Layer_2_VertexUpdate(), Iterative_Layer_2_GlobalUpdate(v2g)),
             MetaLayer(None, Iterative_Layer_3_VertexUpdate(), None)]
    rayleigh = [MetaLayer(Layer_1_EdgeUpdate(), Rayleigh_Layer_1_VertexUpdate(e2v), Rayleigh_Layer_1_GlobalUpdate(v2g)),
                MetaLayer(None, Layer_2_VertexUpdate(), Rayleigh_Layer_2_GlobalUpdate(v2g))]
    return block * num_iter + rayleigh
