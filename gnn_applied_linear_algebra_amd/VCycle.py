"""Two-grid V-cycle glue -- mirror of /root/reference/pytorch/VCycle.py (functions :58-237, script body :239-277),
plus the multi-level generalisation BASELINE config 5 asks for (`AMGHierarchy`).

The reference is a whole-file script whose adapters read module-level globals (`N`, `SOCGNN`, `DIGNN`, `ResidualGNN`,
`ChebyGNN`, `JacGNN`; VCycle.py:239-256).  The same names exist here as module attributes, created by `configure()`;
the adapters keep the reference's names and argument lists.  Differences, all documented in SURVEY.md Appendix A:
  * the C/F splitting is DATA (`configure(splitting=...)`, default every other vertex, matlab/test_vcycle.m:65-67)
    instead of pyamg's randomised CLJP (VCycle.py:114);
  * the true diagonal of A is used where the script hard-codes -4 (VCycle.py:117,165) -- identical on its matrix;
  * P = (I + W)[:, C] is assembled sparse, without the dense n x n detour of VCycle.py:129-137.
Smoothers, residual, strength of connection and interpolation weights run in libgla_hip kernels; restriction and
prolongation are SpMVs on plans of P^T / P; the Galerkin product P^T (A P) (setup, VCycle.py:209) and every
coalesce run in libgla_hip's expand-sort-compress kernels (sparse_ops.py, csrc/gla_sparse.hip).
"""
import torch

from ._common import compute_device, plan_for, to_dev
from . import sparse_ops
from .ChebyGNN import ChebyRelaxGNN, chebyshev_relax
from .DirectInterpGNN import DirectInterpGNN
from .GNNResidual import GNNResidual
from .JacobiGNN import JacobiGNN
from .SOCClassicGNN import SOCClassicGNN
from .UtilsGNN import coo_to_gnn_input, remove_diag_entries

# module-level state, as in the reference script (VCycle.py:239-256)
N = None
theta = 0.25
cheb_deg = 4
SOCGNN = DIGNN = ResidualGNN = ChebyGNN = JacGNN = None
_splitting = None


def default_splitting(n, device):
    s = torch.zeros(n, dtype=torch.int64, device=device)
    s[0::2] = 1
    return s


def pmis_splitting(edgeij_pair_od, strong, n, max_rounds=64):
    """Deterministic PMIS-style C/F splitting in libgla_hip (gla_pmis_split, csrc/gla_sparse.hip); stands in for pyamg's
    randomised CLJP (VCycle.py:114).  `strong[e]` says that vertex row(e) strongly depends on
    col(e).  weight_i = (number of vertices that strongly depend on i, fixed 32-bit hash of i, low bits of i) compared
    lexicographically; a vertex becomes C when its weight beats every undecided strong neighbour (either direction),
    then every undecided vertex that strongly depends on a new C point becomes F.  Vertices without any strong
    connection become C (identity interpolation; the DI formula has no finite weight for them).
    Returns an int64 vector on the caller's device, 1 = C point.  (Checked against a CPU restatement of the same rule kept with the test oracle.)"""
    import ctypes
    from ._lib import check, lib, ptr, stream_ptr
    dev = compute_device(edgeij_pair_od)
    ei = to_dev(edgeij_pair_od, dev)
    flags = to_dev(strong, dev).reshape(-1).to(torch.uint8).contiguous()
    if flags.numel() != ei.shape[1]:
        raise ValueError(f"strong has {flags.numel()} entries, the edge list {ei.shape[1]}")
    plan = plan_for(ei, int(n))
    split = torch.empty(int(n), dtype=torch.int64, device=dev)
    rounds = ctypes.c_int(0)
    with torch.cuda.device(dev):
        check(lib().gla_pmis_split(plan._h, ptr(flags), ptr(split), int(max_rounds), ctypes.byref(rounds),
                                   stream_ptr(dev)))
    return split.to(edgeij_pair_od.device)


class GridSplitting:
    """Standard (full) coarsening of an nx x ny vertex grid numbered k = j*nx + i: C points at even i and even j.
    Callable as splitting(A_level, S, n); it tracks the grid size from level to level."""

    def __init__(self, nx, ny):
        self.nx, self.ny = int(nx), int(ny)

    def __call__(self, A, S, n):
        if n != self.nx * self.ny:
            raise ValueError(f"level has {n} vertices, expected a {self.nx} x {self.ny} grid")
        dev = A.device
        k = torch.arange(n, device=dev)
        s = (((k % self.nx) % 2 == 0) & ((k // self.nx) % 2 == 0)).to(torch.int64)
        self.nx, self.ny = (self.nx + 1) // 2, (self.ny + 1) // 2
        return s


def configure(theta_=0.25, cheb_deg_=4, splitting=None, N_=None):
    """Create the GNN modules (VCycle.py:251-256).  splitting: callable(S_matrix_coo, n) -> int vector, 1 = C point."""
    global theta, cheb_deg, SOCGNN, DIGNN, ResidualGNN, ChebyGNN, JacGNN, _splitting, N
    theta, cheb_deg, N = theta_, cheb_deg_, N_
    SOCGNN = SOCClassicGNN(theta)
    DIGNN = DirectInterpGNN()
    ResidualGNN = GNNResidual()
    ChebyGNN = ChebyRelaxGNN(cheb_deg, return_edge_attr=False)
    JacGNN = JacobiGNN()
    _splitting = splitting


def _ready():
    if SOCGNN is None:
        configure()


def _dev(A):
    dev = compute_device(A)
    return dev, (A if A.device == dev else A.to(dev))


def _diag_of(edgeij_pair, edge_attr, n):
    """True diagonal of A as [n, 1] (zero where the matrix stores none)."""
    d = torch.zeros((n, 1), dtype=edge_attr.dtype, device=edge_attr.device)
    m = edgeij_pair[0] == edgeij_pair[1]
    d[edgeij_pair[0][m], 0] = edge_attr[m, 0]
    return d


def runResidual(A, b, x):
    """VCycle.py:58-70: r = b - A x."""
    _ready()
    dev, A = _dev(A)
    edgeij_pair, edge_attr = coo_to_gnn_input(A)
    return ResidualGNN(torch.cat([to_dev(b, dev), to_dev(x, dev)], 1), edgeij_pair, edge_attr)


def runSOC(A):
    """VCycle.py:72-92: boolean strong-connection flags of the off-diagonal edges, [nnz_od, 1]."""
    _ready()
    dev, A = _dev(A)
    edgeij_pair, edge_attr = remove_diag_entries(*coo_to_gnn_input(A))
    vertex_attr = torch.zeros((A.shape[0], 1), dtype=edge_attr.dtype, device=dev)
    return SOCGNN(vertex_attr, edgeij_pair, edge_attr).reshape(-1, 1) > 0


def runDirectInterp(A, S, N=None, splitting=None):
    """VCycle.py:94-137: prolongation operator P = (I + W)[:, C] as a sparse COO tensor [n, n_c]."""
    _ready()
    dev, A = _dev(A)
    n = A.shape[0]
    ei_all, ea_all = coo_to_gnn_input(A)
    edgeij_pair, edge_attr = remove_diag_entries(ei_all, ea_all)
    dt = edge_attr.dtype
    if splitting is None:
        if _splitting is not None:
            S_matrix = torch.sparse_coo_tensor(edgeij_pair, to_dev(S, dev).flatten().to(dt), (n, n)).coalesce()
            splitting = _splitting(S_matrix, n)
        else:
            splitting = default_splitting(n, dev)
    splitting = torch.as_tensor(splitting).to(dev).reshape(-1)
    vertex_attr = torch.hstack([_diag_of(ei_all, ea_all, n), splitting.reshape(-1, 1).to(dt)])
    w_ij = DIGNN(vertex_attr, edgeij_pair, torch.hstack([edge_attr, to_dev(S, dev).to(dt)]), None)
    # sparse (I + W)[:, C]: keep edges whose column is a C point, plus the identity entries of the C rows
    is_c = splitting > 0
    cmap = torch.cumsum(is_c.to(torch.int64), 0) - 1
    keep = is_c[edgeij_pair[1]]
    c_rows = torch.nonzero(is_c).reshape(-1)
    rows = torch.cat([edgeij_pair[0][keep], c_rows])
    cols = torch.cat([cmap[edgeij_pair[1][keep]], cmap[c_rows]])
    vals = torch.cat([w_ij[keep], torch.ones(c_rows.numel(), dtype=dt, device=dev)])
    n_c = int(is_c.sum())
    P_ei, P_ev = sparse_ops.coalesce(torch.stack([rows, cols]), vals, n, n_c)
    return _as_sparse(P_ei, P_ev, (n, n_c))


def runCheby(A, b, x, c, d):
    """VCycle.py:139-154: Chebyshev relaxation of degree cheb_deg with eigen-bounds (c, d)."""
    _ready()
    dev, A = _dev(A)
    edgeij_pair, edge_attr = coo_to_gnn_input(A)
    vertex_attr = torch.cat([to_dev(b, dev), to_dev(x, dev)], 1)
    vertex_attr, _, _ = ChebyGNN(vertex_attr, edgeij_pair, edge_attr, torch.tensor([c, d]))
    return vertex_attr[:, 1].reshape(-1, 1)


def runJacobi(n_iters, w, A, b, x):
    """VCycle.py:156-173: n_iters weighted-Jacobi sweeps."""
    _ready()
    dev, A = _dev(A)
    edgeij_pair, edge_attr = coo_to_gnn_input(A)
    diag_vals = _diag_of(edgeij_pair, edge_attr, A.shape[0])
    edge_attr = torch.cat([edge_attr, torch.zeros_like(edge_attr)], 1)
    vertex_attr = torch.cat([diag_vals, to_dev(b, dev), to_dev(x, dev)], 1)
    return JacGNN(n_iters, vertex_attr, edgeij_pair, edge_attr, torch.tensor(w).reshape(-1))


def _as_sparse(ei, ev, shape):
    """Coalesced (edge_index, values) -> torch sparse COO carrying its GNN-input cache (no re-coalesce later)."""
    M = torch.sparse_coo_tensor(ei, ev.reshape(-1), shape, is_coalesced=True)
    M._gla_gnn_input = (ei, ev.reshape(-1, 1))
    return M


def _galerkin(A, P):
    """(Ac, plan(P), plan(P^T)) with Ac = P^T (A P) (VCycle.py:209) computed by libgla_hip's SpGEMM."""
    A_ei, A_ev = coo_to_gnn_input(A)
    P_ei, P_ev = coo_to_gnn_input(P)
    Ac_ei, Ac_ev, Pp, Pt = sparse_ops.galerkin(A_ei, A_ev[:, 0].contiguous(), A.shape[0], P_ei,
                                               P_ev[:, 0].contiguous(), P.shape[1])
    return _as_sparse(Ac_ei, Ac_ev, (P.shape[1], P.shape[1])), Pp, Pt


def runVCycle(A, b, x, n_presmooth, n_postsmooth, n_coarsesolve, use_jacobi=True):
    """VCycle.py:175-237: one two-grid cycle (pre-smooth, SOC, DI, Galerkin, residual, restrict, Chebyshev coarse
    'solve', prolong + correct, post-smooth).  n_coarsesolve is accepted and unused, as in the reference."""
    _ready()
    dev, A = _dev(A)
    b, x = to_dev(b, dev), to_dev(x, dev)
    if use_jacobi:
        x = runJacobi(n_presmooth, 0.7, A, b, x)
    else:
        x = runCheby(A, b, x, -3.461, -4.0)
    S = runSOC(A)
    P = runDirectInterp(A, S, N)
    Ac, P_plan, Pt_plan = _galerkin(A, P)                                # VCycle.py:209
    r = runResidual(A, b, x)
    rc = Pt_plan.spmm(r.contiguous())                                    # :215
    xc = runCheby(Ac, rc, torch.zeros_like(rc), -3.4, -4.0)              # :218-223
    x = x + P_plan.spmm(xc.contiguous())                                 # :226
    if use_jacobi:
        x = runJacobi(n_postsmooth, 0.7, A, b, x)
    else:
        x = runCheby(A, b, x, -3.4, -4.0)
    return x


class AMGHierarchy:
    """Multi-level generalisation (BASELINE config 5: 3 levels): the same five operators applied recursively, with
    the setup (SOC -> splitting -> DI -> P -> Galerkin) done ONCE instead of every cycle.

    cycle(): weighted Jacobi pre/post smoothing on every level but the coarsest.  Coarsest 'solve': Chebyshev
    relaxation (the reference's choice, VCycle.py:218-223) of degree coarse_deg over [lambda/kappa, 1.1 lambda] with
    lambda from the power-method GNN, or coarse_solver="dense" (the coarsest operator factored once and its inverse applied
    per cycle, for hierarchies deep enough that it is small).  splitting: None (every other vertex), "pmis" (pmis_splitting) or a callable."""

    # The smoothers' A_ii is the diagonal this class extracts from each level's matrix (`diags`): the march passes are told so
    # and take it from the centre coefficient they load anyway (plan.jacobi_sweeps(..., diag_is_matrix_diagonal=True)).
    # False: stream the extracted vector like any caller's (A/B, tests).
    _diag_from_matrix = True

    def __init__(self, A, levels=3, theta_=0.25, cheb_deg_=4, splitting=None, w=0.7, n_pre=3, n_post=3,
                 coarse_bounds=None, coarse_sweeps=1, min_coarse=0, coarse_deg=None,
                 coarse_kappa=100.0, coarse_solver="cheby"):
        dev, A = _dev(A)
        self.w, self.n_pre, self.n_post, self.coarse_sweeps = w, n_pre, n_post, coarse_sweeps
        self.soc, self.di = SOCClassicGNN(theta_), DirectInterpGNN()
        self.res, self.jac = GNNResidual(), JacobiGNN()
        self.cheby = ChebyRelaxGNN(cheb_deg_ if coarse_deg is None else coarse_deg, return_edge_attr=False)
        self.A, self.P, self.Pt, self.gnn = [A], [], [], []
        for _ in range(levels - 1):
            Al = self.A[-1]
            n = Al.shape[0]
            if n <= min_coarse:
                break
            ei_all, ea_all = coo_to_gnn_input(Al)
            ei, ea = remove_diag_entries(ei_all, ea_all)
            S = self.soc(torch.zeros((n, 1), dtype=ea.dtype, device=dev), ei, ea).reshape(-1, 1) > 0
            if splitting == "pmis":
                split = pmis_splitting(ei, S, n)
            elif splitting is not None:
                split = splitting(Al, S, n)
            else:
                split = default_splitting(n, dev)
            split = split.to(dev).reshape(-1)
            va = torch.hstack([_diag_of(ei_all, ea_all, n), split.reshape(-1, 1).to(ea.dtype)])
            w_ij = self.di(va, ei, torch.hstack([ea, S.to(ea.dtype)]), None)
            is_c = split > 0
            cmap = torch.cumsum(is_c.to(torch.int64), 0) - 1
            # F rows interpolate from their STRONG C neighbours only (classical direct interpolation: the
            # denominator of DirectInterpGNN.py:93 sums exactly those), C rows are identity rows
            keep = is_c[ei[1]] & ~is_c[ei[0]] & S.reshape(-1)
            c_rows = torch.nonzero(is_c).reshape(-1)
            rows = torch.cat([ei[0][keep], c_rows])
            cols = torch.cat([cmap[ei[1][keep]], cmap[c_rows]])
            vals = torch.cat([w_ij[keep], torch.ones(c_rows.numel(), dtype=ea.dtype, device=dev)])
            n_c = int(is_c.sum())
            P_ei, P_ev = sparse_ops.coalesce(torch.stack([rows, cols]), vals, n, n_c)
            Ac_ei, Ac_ev, P_plan, Pt_plan = sparse_ops.galerkin(ei_all, ea_all[:, 0].contiguous(), n, P_ei, P_ev, n_c)
            self.P.append(P_plan)
            self.Pt.append(Pt_plan)
            self.A.append(_as_sparse(Ac_ei, Ac_ev, (n_c, n_c)))
        # per-level GNN inputs, built once
        self.plans, self.diags = [], []
        for Al in self.A:
            ei, ea = coo_to_gnn_input(Al)
            dg = _diag_of(ei, ea, Al.shape[0])
            self.gnn.append((ei, ea, torch.cat([ea, torch.zeros_like(ea)], 1), dg))
            self.plans.append(sparse_ops.plan_of(ei, ea, Al.shape[0], Al.shape[1]))      # values gathered once
            self.diags.append(dg.reshape(-1).contiguous())
        self.coarse_solver = coarse_solver
        self._coarse_lu = self._coarse_inv = None
        if coarse_solver == "dense":
            # classical AMG practice for a coarsest level of a few thousand unknowns: factor it once (torch.linalg, setup
            # time).  Up to DENSE_INVERSE_MAX rows the factors are turned into the explicit inverse, which every cycle then
            # applies as ONE matrix-vector pass (gla_dense_apply): the pair of triangular solves behind lu_solve took
            # 0.66 ms per cycle at 1024 rows -- a quarter of the 8-level cycle on the 4095^2 mesh -- against ~5 us
            Ad = self.A[-1].to_dense()
            lu = torch.linalg.lu_factor(Ad)
            if Ad.shape[0] <= self.DENSE_INVERSE_MAX:
                self._coarse_inv = torch.linalg.lu_solve(lu[0], lu[1], torch.eye(Ad.shape[0], dtype=Ad.dtype,
                                                                                 device=Ad.device)).contiguous()
            else:
                self._coarse_lu = lu
            coarse_bounds = (0.0, 1.0)
        if coarse_bounds is None:
            # dominant eigenvalue of the coarsest operator by the power-method GNN; the Chebyshev 'solve' then
            # targets [lambda/kappa, 1.1 lambda] (c = half-width, d = centre, as in ChebyGNN.py:365-368)
            from .PowerMethodGNN import PowerMethodGNN
            ei, ea, ea2, dg = self.gnn[-1]
            nc = dg.shape[0]
            gen = torch.Generator().manual_seed(24601)
            b0 = torch.rand(nc, 1, generator=gen, dtype=torch.float64).to(dg.dtype).to(dev)
            _, _, gp = PowerMethodGNN(30, return_edge_attr=False)(
                torch.cat([b0, torch.zeros_like(b0)], 1), ei, ea2, torch.zeros(3, dtype=dg.dtype), None)
            lam = float(gp[2])
            hi, lo = 1.1 * lam, lam / coarse_kappa
            coarse_bounds = ((hi - lo) / 2, (hi + lo) / 2)
        self.coarse_bounds = coarse_bounds

    DENSE_INVERSE_MAX = 4096                                 # rows (128 MiB of fp64 inverse at most)

    @staticmethod
    def _apply_dense(M, b):
        """M b by libgla_hip (gla_dense_apply); b [n, K]."""
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        if not b.is_cuda:
            raise RuntimeError("libgla_hip operates on GPU tensors")
        b = b.contiguous()
        if b.dtype != M.dtype:
            raise ValueError(f"coarsest operator is {M.dtype}, right-hand side {b.dtype}")
        y = torch.empty_like(b)
        n, K = b.shape
        check(lib().gla_dense_apply(dtype_code(b.dtype), n, n, ptr(M), M.stride(0), K, ptr(b), b.stride(0), ptr(y),
                                    y.stride(0), stream_ptr(b.device)))
        return y

    def _zero_guess(self, like):
        """The zero initial guess of a coarse level: one read-only buffer per (shape, dtype) instead of a fill launch per level
        and cycle (every consumer -- the smoothers, the coarsest solve -- is out of place)."""
        cache = self.__dict__.setdefault("_zeros", {})
        key = (tuple(like.shape), like.dtype, like.device)
        z = cache.get(key)
        if z is None:
            z = cache[key] = torch.zeros_like(like)
        return z

    def residual(self, b, x, level=0):
        """r = b - A_level x (GNNResidual on the level's plan)."""
        return self.plans[level].residual(b.contiguous(), x.contiguous())

    def _smooth(self, level, n_it, b, x):
        """n_it weighted-Jacobi sweeps (JacobiGNN on the level's plan; w is cast to the operand dtype like g[0])."""
        w = float(torch.tensor(self.w, dtype=b.dtype))
        # (self.diags[level] was extracted from the level's matrix: the march passes are told so and stream no A_ii)
        return self.plans[level].jacobi_sweeps(self.diags[level], b.contiguous(), x.contiguous(), w, n_it,
                                               diag_is_matrix_diagonal=self._diag_from_matrix)

    def capture_cycle(self, n_rhs=1, dtype=torch.float64):
        """Capture one V-cycle into a HIP graph (torch.cuda.CUDAGraph) and return run(b, x) -> x_new.
        Same kernels in the same order: the result is bit-identical to cycle(b, x).  For hosts that cannot keep the
        launch queue ahead of the GPU (the coarse levels of a deep hierarchy issue ~15 launches per level with a few
        microseconds of work each).  Measured on the 8-level 4095^2 FEM hierarchy: 5.11 ms per replay (including the
        operand copies) vs 4.96 ms eager -- there the eager queue already runs ahead, so eager stays the default.
        Operands have the captured shape [n, n_rhs] / dtype."""
        dev = self.plans[0].device
        n = self.A[0].shape[0]
        sb = torch.zeros((n, n_rhs), dtype=dtype, device=dev)
        sx = torch.zeros_like(sb)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # warm-up off the capture: lazy allocations, code loads
            for _ in range(2):
                self.cycle(sb, sx)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.cycle(sb, sx)

        def run(b, x):
            sb.copy_(b)
            sx.copy_(x)
            graph.replay()
            return out.clone()
        run.graph = graph
        return run

    def cycle(self, b, x, level=0):
        if level == len(self.A) - 1:
            if self._coarse_inv is not None:
                return self._apply_dense(self._coarse_inv, b)
            if self._coarse_lu is not None:
                return torch.linalg.lu_solve(self._coarse_lu[0], self._coarse_lu[1], b)
            g = torch.tensor(self.coarse_bounds, dtype=b.dtype)
            for _ in range(self.coarse_sweeps):
                x = chebyshev_relax(self.plans[level], b.contiguous(), x.contiguous(), g, self.cheby.deg,
                                    need_gathered=False)[0]
            return x
        # pre-smoothing and the residual of its result: one pass over the level's matrix where the plan qualifies
        # (gla_jacobi_fused_residual; same arithmetic as the two calls, bit for bit)
        w = float(torch.tensor(self.w, dtype=b.dtype))
        xr = self.plans[level].jacobi_sweeps_residual(self.diags[level], b.contiguous(), x.contiguous(), w, self.n_pre,
                                                      diag_is_matrix_diagonal=self._diag_from_matrix) \
            if (b.dim() == 2 and b.shape[1] == 1) else None
        if xr is not None:
            x, r = xr
        else:
            x = self._smooth(level, self.n_pre, b, x)
            r = self.residual(b, x, level)
        rc = self.Pt[level].spmm(r)
        xc = self.cycle(rc, self._zero_guess(rc), level + 1)
        x = self.P[level].spmm_add(x.contiguous(), xc.contiguous())       # x + P xc in one pass (gla_spmm_add)
        return self._smooth(level, self.n_post, b, x)
