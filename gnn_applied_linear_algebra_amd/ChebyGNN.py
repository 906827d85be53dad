"""Chebyshev relaxation of degree `deg` -- mirror of /root/reference/pytorch/ChebyGNN.py (ChebyRelaxGNN :287-353).

The reference chains 2*deg MetaLayers (an SpMV layer and a vector-update layer per degree, :291-321).  Here each
degree is ONE fused kernel: gla_cheby_first (r = b - A x; p = r; x += alpha p) and gla_cheby_next
(r -= alpha_old A p; p = r + beta p; x += alpha_new p).  The scalar recurrences of the global updates
(:123-139, :246-285) are evaluated on the host with 0-dim tensors of g's dtype in the reference's exact
operation order, so alpha/beta are bit-identical.
"""
import torch

from ._common import aggregate, back, cols, compute_device, float_dtype, plan_for, split_cols, stack_cols, to_dev


FUSED_MAX = 4          # gla_cheby_fused takes the first 2 .. 4 steps in one pass over the data


def _scalars(g, deg):
    """The scalar recurrences of the global updates, all steps up front (they do not depend on the vectors), evaluated in
    g's dtype on the host exactly as the reference's GlobalUpdate modules do."""
    gh = g.detach().to("cpu")
    c, d = gh[0], gh[1]
    alphas, betas = [1 / d], [None]                                  # ChebyGNN.py:137
    for it in range(2, deg + 1):
        a_prev = alphas[-1]
        beta = 0.5 * (c * a_prev) ** 2 if it == 2 else ((c * a_prev) / 2) ** 2      # :262 / :283
        alphas.append(1 / (d - beta / a_prev))                                      # :263 / :283
        betas.append(beta)
    return c, d, alphas, betas


def _g_out(c, d, alphas, betas, g):
    alpha, beta = alphas[-1], betas[-1]
    return torch.hstack([c, d, alpha] if beta is None else [c, d, alpha, beta]).to(g.device)


def chebyshev_relax(plan, b, x0, g, deg, out=None, need_gathered=True, b_copy=None):
    """Core of ChebyRelaxGNN on SoA operands: b, x0 [n, K] (row-major, any leading dimension), g = [c, d] (any
    device).  Runs `deg` steps on `plan` and returns (x, r, p, gathered, g_out) where `gathered` is the vector
    the last SpMV read (x0 for deg 1, the previous p otherwise -- the z_ij = A_ij * gathered_j of the returned edge
    attributes; None when need_gathered is False and the fused path did not materialise it).  out = (x, r, p):
    caller-owned destinations (e.g. column views of the AoS result) the final iterates are written into directly;
    b_copy: where the module's [b, x, r, p] result wants its copy of b (written by the fused pass, else copied here).
    On 2-D stencil plans with K a multiple of the 16-byte vector the first min(deg, 4) steps run as ONE temporally
    blocked pass (gla_cheby_fused, bit-identical); the rest, and every other case, one fused kernel per step."""
    c, d, alphas, betas = _scalars(g, deg)
    if out is None:
        r, p_final, x = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
    else:
        x, r, p_final = out
    gathered = x0
    done = 0
    canon = int(plan.info.dia_canon5) in (1, 9)       # the march kernels: continuation passes exist (gla_cheby_fused_next)
    new = lambda: torch.empty((b.shape[0], b.shape[1]), dtype=b.dtype, device=b.device)   # noqa: E731
    for m in range(min(deg, FUSED_MAX), 1, -1):
        # Fused pass for steps 1 .. m (the most the plan's kernel takes: four, or three for K-wide 9-point passes), then --
        # on canonical plans -- continuation passes of the same kind for the steps that follow and at most one single step
        # at the end (other plans: single steps only).  The buffers are laid out up front so that the last writer of x, r
        # and p is the caller's buffer.
        rest = deg - m
        sizes, single = [], rest                        # continuation pass sizes, single steps after the fused passes
        if canon and rest >= 2:
            n_pass = -(-rest // m)
            base, extra = divmod(rest, n_pass)
            if base >= 2:
                sizes, single = [base + 1] * extra + [base] * (n_pass - extra), 0
            else:
                sizes, single = [2] * (rest // 2), rest % 2
        n_cont = len(sizes)
        # x, r: the passes alternate between the caller's buffers and temporaries, ending in the caller's
        xr = [(x, r), (new(), new())] if n_cont else [(x, r)]
        # p: fused pass j writes p_seq[j], alternating between p_final and a scratch; the last one writes p_final when an even
        # number of single steps follows (they ping-pong), else the scratch
        s1 = new() if (n_cont or single) else None
        other = lambda t: s1 if t is p_final else p_final   # noqa: E731
        p_seq = [p_final if single % 2 == 0 else s1]
        for _ in range(n_cont):
            p_seq.insert(0, other(p_seq[0]))
        # the operand the very last step gathers (z_ij of the returned edge attributes): reported by the last fused pass when
        # no single step follows
        s2 = new() if (need_gathered and single == 0) else None
        cur = xr[n_cont % 2]
        if not plan.cheby_fused(b, x0, cur[0], cur[1], p_seq[0], [float(a) for a in alphas[:m]],
                                [0.0] + [float(v) for v in betas[1:m]], Pprev=s2 if n_cont == 0 else None, Bcopy=b_copy):
            continue
        b_copy = None
        done = m
        for j, sz in enumerate(sizes):
            nxt = xr[(n_cont - 1 - j) % 2]
            ok = plan.cheby_fused_next(cur[1], cur[0], p_seq[j], nxt[0], nxt[1], p_seq[j + 1], float(alphas[done - 1]),
                                       [float(a) for a in alphas[done:done + sz]], [float(v) for v in betas[done:done + sz]],
                                       Pprev=s2 if j == n_cont - 1 else None)
            if not ok:
                # (the march kernels are switched off by option: only the first pass had a kernel.)  Finish step by step from
                # where the iterates are, then move them into the caller's buffers
                n_left = deg - done
                its = range(done + 1, deg + 1)
                pa, pb = p_seq[j], p_seq[j + 1]
                plan.cheby_steps(pa, pb, cur[1], cur[0], [alphas[it - 2] for it in its], [alphas[it - 1] for it in its],
                                 [betas[it - 1] for it in its])
                p_fin, p_prev = (pa, pb) if n_left % 2 == 0 else (pb, pa)
                gathered = p_prev.clone() if need_gathered else None
                if cur[0] is not x:
                    x.copy_(cur[0])
                    r.copy_(cur[1])
                if p_fin is not p_final:
                    p_final.copy_(p_fin)
                return x, r, p_final, gathered, _g_out(c, d, alphas, betas, g)
            done += sz
            cur = nxt
        assert cur[0] is x and cur[1] is r
        p = p_seq[-1]
        p_next = other(p) if single else None
        gathered = s2
        break
    if b_copy is not None:
        b_copy.copy_(b)
    if done == 0:
        p_other = torch.empty((b.shape[0], b.shape[1]), dtype=b.dtype, device=b.device) if deg > 1 else None
        # p ping-pongs deg - 1 times: start so that the last one lands in the caller's buffer
        p, p_next = (p_final, p_other) if (deg - 1) % 2 == 0 else (p_other, p_final)
        plan.cheby_first(b, x0, r, p, x, float(alphas[0]))
        done = 1
    rest_steps = deg - done
    if rest_steps > 0:
        # the remaining steps done + 1 .. deg are enqueued by ONE library call (no host round trip per step)
        its = range(done + 1, deg + 1)
        plan.cheby_steps(p, p_next, r, x, [alphas[it - 2] for it in its], [alphas[it - 1] for it in its],
                         [betas[it - 1] for it in its])
        if rest_steps % 2 == 1:
            p, p_next = p_next, p
        gathered = p_next                                            # the operand the last step gathered
    return x, r, p, gathered, _g_out(c, d, alphas, betas, g)


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
        if va.dim() == 3 and va.shape[2] > 1 and va.is_contiguous():
            # multi-RHS: the kernels take leading dimensions, so b / x0 are read in place from the AoS input and
            # x, r, p are written straight into the AoS result (no split / stack passes over the K-wide vectors)
            res = torch.empty((va.shape[0], 4, va.shape[2]), dtype=va.dtype, device=va.device)
            _, _, _, gathered, g_out = chebyshev_relax(plan, va[:, 0, :], va[:, 1, :], g, self.deg,
                                                       out=(res[:, 1, :], res[:, 2, :], res[:, 3, :]),
                                                       need_gathered=self.return_edge_attr, b_copy=res[:, 0, :])
        else:
            if va.dim() == 2 and 2 <= self.deg <= FUSED_MAX and va.is_contiguous():
                # single right-hand side, up to four steps: the fused pass reads b, x as columns of the AoS input and
                # writes [b, x, r, p] straight into the AoS result (no split / stack passes) where the plan qualifies
                c_, d_, alphas, betas = _scalars(g, self.deg)
                res = torch.empty((va.shape[0], 4), dtype=va.dtype, device=va.device)
                pprev = torch.empty((va.shape[0], 1), dtype=va.dtype, device=va.device) if self.return_edge_attr else None
                if plan.cheby_fused(va[:, 0:1], va[:, 1:2], res[:, 1:2], res[:, 2:3], res[:, 3:4],
                                    [float(a) for a in alphas], [0.0] + [float(v) for v in betas[1:]],
                                    Pprev=pprev, Bcopy=res[:, 0:1]):
                    ea_out = plan.edge_attr(pprev) if self.return_edge_attr else None
                    return back(res, vertex_attr), back(ea_out, edge_attr), _g_out(c_, d_, alphas, betas, g)
            b, x0 = split_cols(va) if va.dim() == 2 else (cols(va, 0), cols(va, 1))
            x, r, p, gathered, g_out = chebyshev_relax(plan, b, x0, g, self.deg, need_gathered=self.return_edge_attr)
            res = stack_cols([b, x, r, p], va)
        ea_out = plan.edge_attr(gathered) if self.return_edge_attr else None    # [A_ij, z_ij] in one pass
        return back(res, vertex_attr), back(ea_out, edge_attr), g_out


# ---- the reference's per-layer update modules (ChebyGNN.py:49-285), usable in a MetaLayer like the originals --------
# Vertex attributes are [b, x, r, p] (columns appear as the layers add them), edge attributes [A_ij, z_ij],
# globals [c, d, alpha, beta].  ChebyRelaxGNN above runs the same sequence as fused kernels.
def _col(t, k):
    return t[:, k:k + 1]


def AllIters_EdgeToVertexAggregation(edgeij_pair, edge_attr, n_vertices):
    """ChebyGNN.py:73-89: zbar_i = sum_j z_ij (1-D)."""
    return aggregate(edgeij_pair, edge_attr[:, 1], n_vertices, "sum")


class Iter1_Layer1_EdgeUpdate(torch.nn.Module):
    """:49-70  z_ij = A_ij * x_j."""
    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        return torch.cat([edge_attr, edge_attr * _col(vattr_j, 1)], 1)


class _AggVertexUpdate(torch.nn.Module):
    def __init__(self, EdgeToVertexAggregation):
        super().__init__()
        self.EdgeToVertexAggregation = EdgeToVertexAggregation

    def _zbar(self, vertex_attr, edgeij_pair, edge_attr):
        return self.EdgeToVertexAggregation(edgeij_pair, edge_attr, vertex_attr.shape[0]).view(-1, 1)


class Iter1_Layer1_VertexUpdate(_AggVertexUpdate):
    """:91-121  r_i = b_i - zbar_i  ->  [b, x, r]."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b, x = _col(vertex_attr, 0), _col(vertex_attr, 1)
        return torch.cat([b, x, b - self._zbar(vertex_attr, edgeij_pair, edge_attr)], 1)


class Iter1_Layer1_GlobalUpdate(torch.nn.Module):
    """:123-139  alpha = 1/d  ->  [c, d, alpha]."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        return torch.hstack([g[0], g[1], 1 / g[1]])


class Iter1_Layer2_VertexUpdate(torch.nn.Module):
    """:141-163  p = r; x += alpha p  ->  [b, x, r, p]."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b, x, r = (_col(vertex_attr, k) for k in range(3))
        return torch.cat([b, x + g[2] * r, r, r], 1)


class IterGreaterThan1_Layer1_EdgeUpdate(torch.nn.Module):
    """:166-183  z_ij = A_ij * p_j."""
    def forward(self, vattr_i, vattr_j, edge_attr, g, batch):
        A = _col(edge_attr, 0)
        return torch.cat([A, A * _col(vattr_j, 3)], 1)


class IterGreaterThan1_Layer1_VertexUpdate(_AggVertexUpdate):
    """:186-216  r -= alpha * zbar (alpha is still the previous one: the global update runs after)."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b, x, r, p = (_col(vertex_attr, k) for k in range(4))
        return torch.cat([b, x, r - g[2] * self._zbar(vertex_attr, edgeij_pair, edge_attr), p], 1)


class IterGreaterThan1_Layer2_VertexUpdate(torch.nn.Module):
    """:219-243  p = r + beta p; x += alpha p."""
    def forward(self, vertex_attr, edgeij_pair, edge_attr, g, batch):
        b, x, r, p = (_col(vertex_attr, k) for k in range(4))
        p = r + g[3] * p
        return torch.cat([b, x + g[2] * p, r, p], 1)


class Iter2_Layer1_GlobalUpdate(torch.nn.Module):
    """:246-265  beta = 0.5 (c alpha)^2; alpha = 1/(d - beta/alpha)."""
    def forward(self, vertex_attr, edge_index, edge_attr, g, batch):
        beta = 0.5 * (g[0] * g[2]) ** 2
        return torch.hstack([g[0], g[1], 1 / (g[1] - beta / g[2]), beta])


class IterGreaterThan2_Layer1_GlobalUpdate(torch.nn.Module):
    """:269-285  beta = ((c alpha)/2)^2; alpha = 1/(d - beta/alpha)."""
    def forward(self, vertex_attr, edge_index, edge_attr, g, batch):
        beta = ((g[0] * g[2]) / 2) ** 2
        return torch.hstack([g[0], g[1], 1 / (g[1] - beta / g[2]), beta])


def unfused_layers(deg):
    """The reference's layer list for degree `deg` (ChebyGNN.py:291-321) built from the modules above and this
    package's MetaLayer -- the unfused formulation, kept for callers that want to interleave their own layers."""
    from .meta import MetaLayer
    agg = AllIters_EdgeToVertexAggregation
    first = [MetaLayer(Iter1_Layer1_EdgeUpdate(), Iter1_Layer1_VertexUpdate(agg), Iter1_Layer1_GlobalUpdate()),
             MetaLayer(None, Iter1_Layer2_VertexUpdate(), None)]
    second = [MetaLayer(IterGreaterThan1_Layer1_EdgeUpdate(), IterGreaterThan1_Layer1_VertexUpdate(agg),
                        Iter2_Layer1_GlobalUpdate()),
              MetaLayer(None, IterGreaterThan1_Layer2_VertexUpdate(), None)]
    later = [MetaLayer(IterGreaterThan1_Layer1_EdgeUpdate(), IterGreaterThan1_Layer1_VertexUpdate(agg),
                       IterGreaterThan2_Layer1_GlobalUpdate()),
             MetaLayer(None, IterGreaterThan1_Layer2_VertexUpdate(), None)]
    return (first if deg > 0 else []) + (second if deg > 1 else []) + later * max(deg - 2, 0)
