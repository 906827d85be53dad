"""Multi-rank V-cycle (BASELINE config 5: "3-level AMG ... 8 x MI355X"), distributed by contiguous row blocks.

SOLVE phase (`RowBlockAMG.cycle`): each rank keeps its row block of every level operator A_l, of every prolongation
P_l (rows: fine block, input vector: coarse partition) and restriction P_l^T (rows: coarse block, input vector: fine
partition).  A cycle needs, per level, one halo all-gather per Jacobi sweep / residual / restriction / prolongation
(dist.RowBlockPartition, overlapped with interior tiles) and, on the coarsest level, one all-gather of the whole
(small) right-hand side, which every rank solves redundantly.

SETUP phase, two routes:
  * `RowBlockAMG(H)`: every rank builds the whole `AMGHierarchy` H (replicated setup) and keeps its slices;
  * `RowBlockAMG.distributed_setup(...)`: NOTHING of nnz size is replicated.  Each rank starts from its row block of
    A and builds, level by level, its rows of S (strength, row-local), of the interpolation weights (row-local; the
    C/F flags of halo columns come from the index-based splitting rule), of P, of A P (after fetching the P rows of
    its halo columns from their owners), of P^T (after receiving the P entries whose column it owns: the entries a rank
    does not own travel by tensor all-gathers, O(nnz(P)) traffic per level during setup) and of
    A_c = P^T (A P) (after fetching the A P rows of the fine vertices its coarse rows interpolate from).  Every row
    is computed by exactly the kernels, operands and summation order of the single-rank setup, so the hierarchy --
    and with it every cycle -- is bit-identical to the single-rank one (tests/test_gpu_dist.py).  Only O(n) index
    vectors (the C/F flags and the coarse numbering) and the coarsest operator are replicated.

Row arithmetic is rank-independent, so the distributed cycle is bit-identical to the single-rank one.
"""
import torch
import torch.distributed as dist

from .ChebyGNN import chebyshev_relax
from .dist import RowBlockOps, _all_wanted, _gather_var, _publish_rows


def even_blocks(n, world):
    """Contiguous, balanced blocks of [0, n)."""
    base, extra = divmod(int(n), world)
    out, lo = [], 0
    for q in range(world):
        hi = lo + base + (1 if q < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _row_slice(ei, ev, lo, hi):
    """Rows [lo, hi) of a coalesced (row-sorted) COO pair."""
    a = int(torch.searchsorted(ei[0].contiguous(), torch.tensor(lo, device=ei.device)))
    b = int(torch.searchsorted(ei[0].contiguous(), torch.tensor(hi, device=ei.device)))
    return ei[:, a:b].contiguous(), ev.reshape(-1)[a:b].contiguous()


class _CoarseSolve:
    """What the coarsest-level 'solve' needs (taken from an AMGHierarchy whose LAST level is the coarsest operator)."""

    def __init__(self, H):
        self.plan, self.lu, self.bounds = H.plans[-1], H._coarse_lu, H.coarse_bounds
        self.sweeps, self.deg = H.coarse_sweeps, H.cheby.deg


class RowBlockAMG:
    def __init__(self, H, group=None, overlap=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.w, self.n_pre, self.n_post = H.w, H.n_pre, H.n_post
        self.n_levels = len(H.A)
        self.coarse = _CoarseSolve(H)
        self.blocks = [even_blocks(A.shape[0], self.world) for A in H.A]
        self.A_ops, self.P_ops, self.Pt_ops, self.diag = [], [], [], []
        for lv, A in enumerate(H.A):
            lo, hi = self.blocks[lv][self.rank]
            ei, ea, _, dg = H.gnn[lv]
            e_l, v_l = _row_slice(ei, ea, lo, hi)
            self.A_ops.append(RowBlockOps(e_l, v_l, A.shape[0], lo, hi, group, overlap))
            self.diag.append(dg.reshape(-1)[lo:hi].contiguous())
        for lv in range(len(H.A) - 1):
            lo, hi = self.blocks[lv][self.rank]
            clo, chi = self.blocks[lv + 1][self.rank]
            P_ei, P_ev = H.P[lv].coo
            e_l, v_l = _row_slice(P_ei, P_ev, lo, hi)
            self.P_ops.append(RowBlockOps(e_l, v_l, H.A[lv].shape[0], lo, hi, group, overlap,
                                          col_blocks=self.blocks[lv + 1]))
            Pt_ei, Pt_ev = H.Pt[lv].coo
            e_l, v_l = _row_slice(Pt_ei, Pt_ev, clo, chi)
            self.Pt_ops.append(RowBlockOps(e_l, v_l, H.A[lv + 1].shape[0], clo, chi, group, overlap,
                                           col_blocks=self.blocks[lv]))

    # ---- distributed setup ---------------------------------------------------------------------------
    @classmethod
    def distributed_setup(cls, A_ei, A_ev, n, levels=3, splitting=None, theta=0.25, w=0.7, n_pre=3, n_post=3,
                          min_coarse=0, group=None, overlap=True, **coarse_kwargs):
        """Build the hierarchy from THIS RANK'S ROW BLOCK of A: (A_ei, A_ev) = the coalesced (row-sorted, ascending
        columns) COO entries of rows even_blocks(n, world)[rank], global vertex ids.  `splitting`: an index-based
        rule -- None (every other vertex, as VCycle.default_splitting) or a VCycle.GridSplitting; it is evaluated
        on global vertex ids, so no rank needs another rank's strength graph.  coarse_kwargs go to the coarsest
        level's solver (coarse_solver=, coarse_deg=, coarse_kappa=, coarse_bounds=, coarse_sweeps=), as in
        AMGHierarchy.  Returns a RowBlockAMG whose cycle() is bit-identical to the single-rank AMGHierarchy.cycle."""
        from . import sparse_ops
        from . import VCycle as V
        from .plan import CsrPlan
        self = cls.__new__(cls)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.w, self.n_pre, self.n_post = w, n_pre, n_post
        self.blocks, self.A_ops, self.P_ops, self.Pt_ops, self.diag = [], [], [], [], []
        dev = A_ei.device
        ei, ev, n_l = A_ei, A_ev.reshape(-1), int(n)
        grid = splitting                                    # GridSplitting walks down the levels as it is called
        setup_nnz = []
        for lv in range(levels):
            blocks = even_blocks(n_l, self.world)
            lo, hi = blocks[self.rank]
            n_loc = hi - lo
            self.blocks.append(blocks)
            self.A_ops.append(RowBlockOps(ei, ev.reshape(-1, 1), n_l, lo, hi, group, overlap))
            dg = torch.zeros(n_loc, dtype=ev.dtype, device=dev)
            on_diag = ei[0] == ei[1]
            dg[ei[0][on_diag] - lo] = ev[on_diag]
            self.diag.append(dg)
            setup_nnz.append(int(ei.shape[1]))
            if lv == levels - 1 or n_l <= min_coarse:
                break
            # ---- C/F flags and coarse numbering: index-based, evaluated on global ids (O(n) vectors, no exchange)
            split_full = V.default_splitting(n_l, dev) if grid is None else grid(_OnDev(dev), None, n_l)
            split_full = split_full.to(dev).reshape(-1)
            is_c_full = split_full > 0
            cmap_full = torch.cumsum(is_c_full.to(torch.int64), 0) - 1
            n_c = int(is_c_full.sum())
            # ---- strength and interpolation weights of the own rows: plan = [own rows] x [own | halo columns]
            od = ei[0] != ei[1]
            r_od, c_od, v_od = ei[0][od], ei[1][od], ev[od]
            off = (c_od < lo) | (c_od >= hi)
            halo = torch.unique(c_od[off])
            c_loc = torch.where(off, n_loc + torch.searchsorted(halo, c_od), c_od - lo)
            plan_od = CsrPlan(torch.stack([r_od - lo, c_loc]), n_loc, n_loc + int(halo.numel()))
            plan_od.set_values(v_od)
            _, S = plan_od.soc_classic(theta, ev.dtype)
            strong = S > 0
            cf_ext = torch.cat([split_full[lo:hi], split_full[halo]]).to(ev.dtype)
            w_ij = plan_od.direct_interp(dg, cf_ext, strong.to(ev.dtype))
            keep = is_c_full[c_od] & ~is_c_full[r_od] & strong
            c_rows = torch.nonzero(is_c_full[lo:hi]).reshape(-1) + lo
            P_rows = torch.cat([r_od[keep], c_rows])
            P_cols = torch.cat([cmap_full[c_od[keep]], cmap_full[c_rows]])
            P_vals = torch.cat([w_ij[keep], torch.ones(c_rows.numel(), dtype=ev.dtype, device=dev)])
            P_ei, P_ev = sparse_ops.coalesce(torch.stack([P_rows - lo, P_cols]), P_vals, max(n_loc, 1), n_c)
            P_ei = torch.stack([P_ei[0] + lo, P_ei[1]])                       # global fine rows, global coarse columns
            cblocks = even_blocks(n_c, self.world)
            clo, chi = cblocks[self.rank]
            self.P_ops.append(RowBlockOps(P_ei, P_ev.reshape(-1, 1), n_l, lo, hi, group, overlap, col_blocks=cblocks))
            # ---- (A P) rows of the own fine rows: needs the P rows of A's halo columns
            a_off = (ei[1] < lo) | (ei[1] >= hi)
            a_halo = torch.unique(ei[1][a_off])
            if self.world > 1:
                hP_ei, hP_ev = _publish_rows(P_ei, P_ev, lo, hi, _all_wanted(a_halo, group), group)
            else:
                hP_ei, hP_ev = P_ei[:, :0], P_ev[:0]
            a_cols = torch.where(a_off, n_loc + torch.searchsorted(a_halo, ei[1]), ei[1] - lo)
            A_loc = sparse_ops.plan_of(torch.stack([ei[0] - lo, a_cols]), ev, n_loc, n_loc + int(a_halo.numel()))
            Pext_rows = torch.cat([P_ei[0] - lo, n_loc + torch.searchsorted(a_halo, hP_ei[0])])
            P_ext = sparse_ops.plan_of(torch.stack([Pext_rows, torch.cat([P_ei[1], hP_ei[1]])]),
                                       torch.cat([P_ev, hP_ev]), n_loc + int(a_halo.numel()), n_c)
            AP_ei, AP_ev = sparse_ops.spgemm(A_loc, P_ext)
            AP_ei = torch.stack([AP_ei[0] + lo, AP_ei[1]])                    # global fine rows
            # ---- P^T rows of the own coarse rows: P entries whose column this rank owns, from every rank
            mine = (P_ei[1] >= clo) & (P_ei[1] < chi)
            if self.world > 1:
                # (tensor collectives; every rank sees every rank's non-owned P entries and keeps those whose column it owns)
                pool_i = [t_.t() for t_ in _gather_var(P_ei[:, ~mine].t().contiguous(), group)]
                pool_v = _gather_var(P_ev[~mine].contiguous(), group)
                o_i = [pi[:, (pi[1] >= clo) & (pi[1] < chi)] for q, pi in enumerate(pool_i) if q != self.rank]
                o_v = [pv[(pi[1] >= clo) & (pi[1] < chi)] for q, (pi, pv) in enumerate(zip(pool_i, pool_v)) if q != self.rank]
                all_i = torch.cat([P_ei[:, mine]] + o_i, 1)
                all_v = torch.cat([P_ev[mine]] + o_v)
            else:
                all_i, all_v = P_ei[:, mine], P_ev[mine]
            Pt_ei, Pt_ev = sparse_ops.coalesce(torch.stack([all_i[1] - clo, all_i[0]]), all_v, max(chi - clo, 1), n_l)
            Pt_glob = torch.stack([Pt_ei[0] + clo, Pt_ei[1]])                 # global coarse rows, global fine columns
            self.Pt_ops.append(RowBlockOps(Pt_glob, Pt_ev.reshape(-1, 1), n_c, clo, chi, group, overlap, col_blocks=blocks))
            # ---- A_c rows of the own coarse rows = P^T rows x (A P) rows of the fine vertices they touch
            fine = torch.unique(Pt_ei[1])
            f_off = (fine < lo) | (fine >= hi)
            if self.world > 1:
                hAP_ei, hAP_ev = _publish_rows(AP_ei, AP_ev, lo, hi, _all_wanted(fine[f_off], group), group)
            else:
                hAP_ei, hAP_ev = AP_ei[:, :0], AP_ev[:0]
            own_rows = torch.isin(AP_ei[0], fine)
            B_rows = torch.cat([AP_ei[0][own_rows], hAP_ei[0]])
            B_cols = torch.cat([AP_ei[1][own_rows], hAP_ei[1]])
            B_vals = torch.cat([AP_ev[own_rows], hAP_ev])
            Pt_loc = sparse_ops.plan_of(torch.stack([Pt_ei[0], torch.searchsorted(fine, Pt_ei[1])]), Pt_ev,
                                        chi - clo, max(int(fine.numel()), 1))
            AP_ext = sparse_ops.plan_of(torch.stack([torch.searchsorted(fine, B_rows), B_cols]), B_vals,
                                        max(int(fine.numel()), 1), n_c)
            Ac_ei, Ac_ev = sparse_ops.spgemm(Pt_loc, AP_ext)
            ei, ev, n_l = torch.stack([Ac_ei[0] + clo, Ac_ei[1]]), Ac_ev, n_c
        self.n_levels = len(self.A_ops)
        self.setup_nnz_local = setup_nnz
        # ---- coarsest operator: gathered to every rank (small), solved redundantly like the single-rank hierarchy
        if self.world > 1:
            c_ei = torch.cat([t_.t() for t_ in _gather_var(ei.t().contiguous(), group)], 1)
            c_ev = torch.cat(_gather_var(ev.contiguous(), group))
        else:
            c_ei, c_ev = ei, ev
        Hc = V.AMGHierarchy(V._as_sparse(c_ei, c_ev, (n_l, n_l)), levels=1, theta_=theta, w=w, **coarse_kwargs)
        self.coarse = _CoarseSolve(Hc)
        return self

    # ---- local pieces of global vectors -------------------------------------------------------------
    def local(self, v, level=0):
        lo, hi = self.blocks[level][self.rank]
        return v[lo:hi].contiguous()

    def gather(self, v_local, level):
        """All ranks get the whole level vector (blocks padded to a common size for the collective)."""
        if self.world == 1:
            return v_local
        blocks = self.blocks[level]
        m = max(hi - lo for lo, hi in blocks)
        K = v_local.shape[1]
        send = torch.zeros((m, K), dtype=v_local.dtype, device=v_local.device)
        send[:v_local.shape[0]] = v_local
        if v_local.is_cuda and dist.get_backend(self.group) == "gloo":        # test-only route, as in dist.py
            recv = torch.empty((self.world * m, K), dtype=v_local.dtype)
            dist.all_gather_into_tensor(recv, send.cpu(), group=self.group)
            recv = recv.to(v_local.device)
        else:
            recv = torch.empty((self.world * m, K), dtype=v_local.dtype, device=v_local.device)
            dist.all_gather_into_tensor(recv, send, group=self.group)
        return torch.cat([recv[q * m:q * m + (hi - lo)] for q, (lo, hi) in enumerate(blocks)], 0)

    # ---- distributed operators ----------------------------------------------------------------------
    @staticmethod
    def _apply(ops, v_in, launch):
        """Put the local input entries into a halo buffer, exchange, run launch(buffer, which) on all tiles."""
        p = ops.part
        buf = p.new_buffer(v_in.shape[1], v_in.dtype, v_in.device)
        buf[:p.n_in] = v_in
        ops.run(buf, lambda which: launch(buf, which))

    def residual(self, b, x, level=0):
        ops = self.A_ops[level]
        r = torch.empty_like(b)
        self._apply(ops, x, lambda buf, which: ops.plan.residual(b, buf, out=r, which=which))
        return r

    def smooth(self, level, n_it, b, x):
        ops, p = self.A_ops[level], self.A_ops[level].part
        w = float(torch.tensor(self.w, dtype=b.dtype))
        cur, nxt = p.new_buffer(1, b.dtype, b.device), p.new_buffer(1, b.dtype, b.device)
        cur[:p.n_in] = x
        for _ in range(n_it):
            src, dst = cur, nxt[:p.n_local]
            ops.run(src, lambda which: ops.plan.jacobi_sweep(self.diag[level], b, src, dst, w, which))
            cur, nxt = nxt, cur
        return cur[:p.n_local].clone()

    def restrict(self, level, r):
        ops = self.Pt_ops[level]
        rc = torch.empty((ops.part.n_local, r.shape[1]), dtype=r.dtype, device=r.device)
        self._apply(ops, r, lambda buf, which: ops.plan.spmm(buf, out=rc, which=which))
        return rc

    def prolong(self, level, xc):
        ops = self.P_ops[level]
        out = torch.empty((ops.part.n_local, xc.shape[1]), dtype=xc.dtype, device=xc.device)
        self._apply(ops, xc, lambda buf, which: ops.plan.spmm(buf, out=out, which=which))
        return out

    def cycle(self, b, x, level=0):
        """One V-cycle on this rank's rows: b, x are the LOCAL parts ([n_local, 1]) of the level vectors."""
        if level == self.n_levels - 1:
            C = self.coarse
            lo, hi = self.blocks[level][self.rank]
            b_full = self.gather(b, level)
            if C.lu is not None:
                return torch.linalg.lu_solve(C.lu[0], C.lu[1], b_full)[lo:hi].contiguous()
            x_full = self.gather(x, level)
            g = torch.tensor(C.bounds, dtype=b.dtype)
            for _ in range(C.sweeps):
                x_full = chebyshev_relax(C.plan, b_full.contiguous(), x_full.contiguous(), g, C.deg,
                                         need_gathered=False)[0]
            return x_full[lo:hi].contiguous()
        x = self.smooth(level, self.n_pre, b, x)
        r = self.residual(b, x, level)
        rc = self.restrict(level, r)
        xc = self.cycle(rc, torch.zeros_like(rc), level + 1)
        x = x + self.prolong(level, xc)
        return self.smooth(level, self.n_post, b, x)


class _OnDev:
    """Stand-in for the level operator in GridSplitting.__call__(A, S, n), which only reads A.device."""

    def __init__(self, device):
        self.device = device
