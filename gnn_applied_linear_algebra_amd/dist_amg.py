"""Multi-rank V-cycle (BASELINE config 5: "3-level AMG ... 8 x MI355X"), distributed by contiguous row blocks.

SOLVE phase (`RowBlockAMG.cycle`): each rank keeps its row block of every DISTRIBUTED level operator A_l -- whole grid lines
where the level is a 2-D stencil (`line_blocks`) -- of every prolongation P_l (rows: fine block, input vector: coarse
partition) and restriction P_l^T (rows: coarse block, input vector: fine partition).  Per distributed level a cycle is
  * pre-smoothing AND the residual of its result in ONE pass of the temporally blocked kernel (gla_jacobi_fused_residual) on
    the block extended by 8 ghost lines: one ghost-line all-gather of [b, x] (round 2: one halo all-gather per sweep, a
    separate residual launch with its own exchange);
  * restriction: one halo all-gather of r; the coarse-grid correction x + P xc in one pass (gla_spmm_add): one halo all-gather
    of xc; post-smoothing in one pass: one ghost-line all-gather of x
= 4 exchanges per level instead of n_pre + n_post + 3.  Levels the temporally blocked kernels do not take (blocks of fewer
than 8 grid lines, non-stencil operators) keep one exchange per sweep.  Operand buffers are allocated once per operator.
AGGLOMERATION: from the first level with fewer than `agglomerate_below` rows per rank on, the hierarchy is
replicated -- one all-gather of that level's right-hand side, then every rank runs the rest of the cycle on its own GPU with
the single-GPU kernels (AMGHierarchy.cycle) and keeps its rows of the correction.  The coarsest level always is.

SETUP phase, two routes:
  * `RowBlockAMG(H)`: every rank builds the whole `AMGHierarchy` H (replicated setup) and keeps its slices;
  * `RowBlockAMG.distributed_setup(...)`: NOTHING of nnz size is replicated on the distributed levels.  Each rank starts from
    its row block of A and builds, level by level, its rows of S (strength, row-local), of the interpolation weights
    (row-local; the C/F flags of halo columns come from the index-based splitting rule), of P, of A P (after fetching the P
    rows of its halo columns from their owners), of P^T (after receiving the P entries whose column it owns: the entries a
    rank does not own travel by tensor all-gathers, O(nnz(P)) traffic per level during setup) and of A_c = P^T (A P) (after
    fetching the A P rows of the fine vertices its coarse rows interpolate from).  Every row is computed by exactly the
    kernels, operands and summation order of the single-rank setup, so the hierarchy -- and with it every cycle -- is
    bit-identical to the single-rank one (tests/test_gpu_dist.py).  Only O(n) index vectors (the C/F flags and the coarse
    numbering) and the agglomerated levels are replicated.

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


def line_blocks(n, world, stride=0):
    """Contiguous, balanced blocks of [0, n) made of WHOLE grid lines of `stride` vertices when the level is an n / stride-line
    grid with at least one line per rank (what the temporally blocked kernels need); even_blocks otherwise."""
    stride = int(stride or 0)
    if stride > 0 and n % stride == 0 and n // stride >= world:
        return [(lo * stride, hi * stride) for lo, hi in even_blocks(n // stride, world)]
    return even_blocks(n, world)


def _row_slice(ei, ev, lo, hi):
    """Rows [lo, hi) of a coalesced (row-sorted) COO pair."""
    a = int(torch.searchsorted(ei[0].contiguous(), torch.tensor(lo, device=ei.device)))
    b = int(torch.searchsorted(ei[0].contiguous(), torch.tensor(hi, device=ei.device)))
    return ei[:, a:b].contiguous(), ev.reshape(-1)[a:b].contiguous()


class _Sub:
    """The replicated part of the hierarchy: an AMGHierarchy and the level of it at which the distributed cycle hands over."""

    def __init__(self, H, level):
        self.H, self.level = H, int(level)


class RowBlockAMG:
    AGGLOMERATE_BELOW = 1 << 16                              # rows per rank below which a level is replicated (default)
    # One-rank rehearsal (tools/bench_dist_vcycle_one_rank.py): with a process group of ONE rank the levels still take the
    # ghost-line route -- extended-block plans, persistent operand buffers, one library RCCL all-gather per ghost-line exchange
    # (of the rank's own first / last lines; it has no neighbour, so nothing is filled in) -- i.e. every kernel and every host
    # step of a rank's cycle except the neighbours' data.  What it costs next to AMGHierarchy.cycle on the same rows is the
    # per-rank overhead of the distributed cycle before inter-GPU latency.
    REHEARSE_ONE_RANK = False
    # As AMGHierarchy._diag_from_matrix (VCycle.py): the smoothers' A_ii IS each level's matrix diagonal, and the extended blocks hold
    # true global rows, so their extended A_ii is the extended plan's centre diagonal row for row: the march passes are told so and
    # stream no A_ii (8 bytes per row of a level's pass: 23-27 us of each level-0 pass at 16.8 M rows, profiles/r05/dist_vcycle_trace.txt)
    _diag_from_matrix = True

    def __init__(self, H, group=None, overlap=True, agglomerate_below=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.w, self.n_pre, self.n_post = H.w, H.n_pre, H.n_post
        thr = self.AGGLOMERATE_BELOW if agglomerate_below is None else int(agglomerate_below)
        n_all = len(H.A)
        # levels 0 .. n_dist - 1 are distributed; level n_dist (the coarsest at the latest) and everything below it is replicated
        # (level 0 included: a problem that small runs replicated as a whole, n_dist = 0 -- every rank does the single-GPU cycle
        # on gathered vectors and keeps its rows)
        n_dist = n_all - 1
        for lv in range(0, n_all):
            if H.A[lv].shape[0] < thr * self.world:
                n_dist = lv
                break
        self.n_dist, self.n_levels = n_dist, n_all
        self.sub = _Sub(H, n_dist)
        strides = [int(H.plans[lv].info.dia_stride) for lv in range(n_all)]
        self.blocks = [line_blocks(A.shape[0], self.world, strides[lv]) for lv, A in enumerate(H.A)][:n_dist + 1]
        self.A_ops, self.P_ops, self.Pt_ops, self.diag = [], [], [], []
        for lv in range(n_dist):
            A = H.A[lv]
            lo, hi = self.blocks[lv][self.rank]
            ei, ea, _, dg = H.gnn[lv]
            e_l, v_l = _row_slice(ei, ea, lo, hi)
            self.A_ops.append(RowBlockOps(e_l, v_l, A.shape[0], lo, hi, group, overlap))
            self.diag.append(dg.reshape(-1)[lo:hi].contiguous())
            self._ghost_setup(lv, e_l, v_l)
        for lv in range(n_dist):
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

    def _ghost_setup(self, lv, e_l, v_l):
        """Level lv's block extended by ghost lines (RowBlockOps._setup_ghost) + the extended A_ii with its ghost lines filled:
        the temporally blocked smoothing passes of this level need one ghost-line exchange each."""
        ops = self.A_ops[lv]
        ops.ext = None
        if (self.world > 1 or (self.REHEARSE_ONE_RANK and dist.is_initialized())) and e_l.is_cuda:
            ops._setup_ghost(e_l, v_l.reshape(-1, 1))
            if self.world == 1 and ops.ext is not None:      # rehearsal: the ghost-line all-gathers run on a one-rank library communicator
                from .dist import LibComm
                ops.comm = LibComm.for_group(self.group)
        if ops.ext is not None:
            e = ops.ext
            dt, dev = e["dtype"], e["dev"]
            d_ext = torch.ones(e["n_ext"], dtype=dt, device=dev)
            own = d_ext[e["off"]:e["off"] + ops.part.n_local]
            own.copy_(self.diag[lv])
            ops._fill_ghosts([d_ext], ops._ghost_exchange([own]))
            ops._ghost_pending = None
            e["amg"] = dict(d=d_ext, b=torch.zeros(e["n_ext"], dtype=dt, device=dev), x=torch.zeros(e["n_ext"], dtype=dt, device=dev))

    # ---- distributed setup ---------------------------------------------------------------------------
    @classmethod
    def distributed_setup(cls, A_ei, A_ev, n, levels=3, splitting=None, theta=0.25, w=0.7, n_pre=3, n_post=3,
                          min_coarse=0, group=None, overlap=True, agglomerate_below=None, **coarse_kwargs):
        """Build the hierarchy from THIS RANK'S ROW BLOCK of A: (A_ei, A_ev) = the coalesced (row-sorted, ascending
        columns) COO entries of rows line_blocks(n, world, splitting.nx)[rank] (whole grid lines when `splitting` is a
        GridSplitting; even_blocks(n, world)[rank] otherwise), global vertex ids.  `splitting`: an index-based
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
        thr = cls.AGGLOMERATE_BELOW if agglomerate_below is None else int(agglomerate_below)
        dev = A_ei.device
        ei, ev, n_l = A_ei, A_ev.reshape(-1), int(n)
        grid = splitting                                    # GridSplitting walks down the levels as it is called
        stride_of = lambda: (grid.nx if isinstance(grid, V.GridSplitting) else 0)      # noqa: E731  (grid lines of the CURRENT level)
        setup_nnz = []
        blocks = line_blocks(n_l, self.world, stride_of())
        for lv in range(levels):
            lo, hi = blocks[self.rank]
            n_loc = hi - lo
            self.blocks.append(blocks)
            # this level is the first replicated one: the coarsest at the latest, earlier when it is small (agglomeration)
            if lv == levels - 1 or n_l <= min_coarse or (lv >= 1 and n_l < thr * self.world):
                break
            self.A_ops.append(RowBlockOps(ei, ev.reshape(-1, 1), n_l, lo, hi, group, overlap))
            dg = torch.zeros(n_loc, dtype=ev.dtype, device=dev)
            on_diag = ei[0] == ei[1]
            dg[ei[0][on_diag] - lo] = ev[on_diag]
            self.diag.append(dg)
            setup_nnz.append(int(ei.shape[1]))
            self._ghost_setup(lv, ei, ev)
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
            cblocks = line_blocks(n_c, self.world, stride_of())         # (the splitting has moved on to the coarse grid)
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
            ei, ev, n_l, blocks = torch.stack([Ac_ei[0] + clo, Ac_ei[1]]), Ac_ev, n_c, cblocks
        self.n_dist = len(self.A_ops)
        self.setup_nnz_local = setup_nnz + [int(ei.shape[1])]
        # ---- the first replicated level's operator: gathered to every rank, the rest of the hierarchy is built (and cycled)
        #      redundantly by the single-GPU code -- the coarsest level alone, or the agglomerated tail of small levels
        if self.world > 1:
            c_ei = torch.cat([t_.t() for t_ in _gather_var(ei.t().contiguous(), group)], 1)
            c_ev = torch.cat(_gather_var(ev.contiguous(), group))
        else:
            c_ei, c_ev = ei, ev
        Hc = V.AMGHierarchy(V._as_sparse(c_ei, c_ev, (n_l, n_l)), levels=levels - self.n_dist, theta_=theta, w=w,
                            n_pre=n_pre, n_post=n_post, splitting=grid, min_coarse=min_coarse, **coarse_kwargs)
        self.sub = _Sub(Hc, 0)
        self.n_levels = self.n_dist + len(Hc.A)
        return self

    # ---- local pieces of global vectors -------------------------------------------------------------
    def local(self, v, level=0):
        lo, hi = self.blocks[level][self.rank]
        return v[lo:hi].contiguous()

    def gather(self, v_local, level):
        """All ranks get the whole level vector (blocks padded to a common size for the collective)."""
        if self.world == 1:
            return v_local
        from .dist import EXCHANGES
        EXCHANGES[0] += 1
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
        """Put the local input entries into the operator's (persistent) halo buffer, exchange, run launch(buffer, which)."""
        p = ops.part
        buf = ops.io_buffer(v_in.shape[1], v_in.dtype, v_in.device)
        buf[:p.n_in] = v_in
        ops.run(buf, lambda which: launch(buf, which))

    def residual(self, b, x, level=0):
        if level >= self.n_dist:                             # replicated level: the single-GPU residual on gathered vectors
            lo, hi = self.blocks[level][self.rank]
            r_full = self.sub.H.residual(self.gather(b, level).contiguous(), self.gather(x, level).contiguous(), self.sub.level)
            return r_full[lo:hi].contiguous()
        ops = self.A_ops[level]
        r = torch.empty_like(b)
        self._apply(ops, x, lambda buf, which: ops.plan.residual(b, buf, out=r, which=which))
        return r

    def smooth(self, level, n_it, b, x):
        """n_it sweeps with one halo exchange per sweep (levels without a ghost-line plan)."""
        ops, p = self.A_ops[level], self.A_ops[level].part
        w = float(torch.tensor(self.w, dtype=b.dtype))
        bufs = ops._lib_send.get(("smooth", b.dtype))
        if bufs is None:
            bufs = ops._lib_send[("smooth", b.dtype)] = (p.new_buffer(1, b.dtype, b.device), p.new_buffer(1, b.dtype, b.device))
        cur, nxt = bufs
        cur[:p.n_in] = x
        for _ in range(n_it):
            src, dst = cur, nxt[:p.n_local]
            ops.run(src, lambda which: ops.plan.jacobi_sweep(self.diag[level], b, src, dst, w, which))
            cur, nxt = nxt, cur
        return cur[:p.n_local].clone()

    def _ghost_operands(self, level, b, x):
        """The extended operands of a ghost-line level with this rank's b (None: keep the last one) and x and their ghost lines."""
        ops = self.A_ops[level]
        e = ops.ext
        a = e["amg"]
        off, nl = e["off"], ops.part.n_local
        own = lambda v: v[off:off + nl]                      # noqa: E731
        vecs = []
        if b is not None:
            # b is copied into the level's extended block unless it is already there: restrict() writes a coarse level's
            # right-hand side straight into this buffer.  The caller's level-0 b is copied on EVERY cycle unless the caller
            # vouches that it has not changed (cycle(..., b_unchanged=True)): torch's version counter does not see writes
            # made through raw pointers (the library's own out= entry points, HIP-graph replays), so it cannot vouch
            src = a.get("b_src")
            if b.data_ptr() == own(a["b"]).data_ptr() and b.numel() == nl:
                a["b_src"] = None
            elif not (getattr(self, "_b_unchanged", False) and not getattr(self, "_capturing", False)
                      and src is not None and src is b):
                own(a["b"]).copy_(b.reshape(-1))
                a["b_src"] = b
            vecs.append(a["b"])
        x_full = ops._ext_of(x, (1,))                        # x allocated with its ghost rows (correct() does): used in place
        if x_full is not None:
            x_e = x_full.reshape(-1)
        else:
            own(a["x"]).copy_(x.reshape(-1))
            x_e = a["x"]
        vecs.append(x_e)
        ops._fill_ghosts(vecs, ops._ghost_exchange([own(v) for v in vecs]))
        ops._ghost_pending = None

        def own_marked(full):                                # own-row view of a fresh extended result, recognised by _ext_of:
            v = full[off:off + nl]                           # the next smoothing pass (this cycle's or the next one's) reads it in place
            v._gla_ext = full
            return v
        return a["d"], a["b"], x_e, own_marked

    def presmooth_residual(self, level, b, x):
        """n_pre sweeps and r = b - A x of the result.  Ghost-line levels: ONE exchange ([b, x] ghost lines) and ONE pass of the
        temporally blocked kernel (gla_jacobi_fused_residual) on the extended block; other levels: one exchange per sweep + one
        for the residual.  Per row the arithmetic of single sweeps / gla_residual: bit-identical either way."""
        ops = self.A_ops[level]
        w = float(torch.tensor(self.w, dtype=b.dtype))
        if ops.ext is not None and b.shape[1] == 1 and 1 <= self.n_pre < ops.TS_MAX:      # (n_pre sweeps + the residual level <= ghost lines)
            d_e, b_e, x_e, own = self._ghost_operands(level, b, x)
            plan = ops.ext["plan"]
            # Round 5: the residual lands where the restriction reads it.  The restriction's operand is [own rows | gathered halo
            # slots]; the pass writes r at extended-block positions.  One persistent carrier per level holds both views: the pass
            # writes carrier[:n_ext] (own rows at `off`), the restriction reads carrier[off : off + n_local + halo] -- its halo
            # slots overlap the block's lower ghost lines, whose r nobody reads, and are filled by the halo exchange AFTER the pass.
            # Before, restrict() copied r into its own buffer: 134 MB in and out at 16.8 M rows, 55-60 us of a 1.8 ms cycle.
            carrier = self._r_carrier(level, b.dtype)
            xr = plan.jacobi_sweeps_residual(d_e, b_e.reshape(-1, 1), x_e.reshape(-1, 1), w, self.n_pre, r_out=carrier,
                                             diag_is_matrix_diagonal=self._diag_from_matrix)
            if xr is not None:                               # (fresh extended tensors: the caller keeps their own-row views, no copy)
                r_own = own(xr[1])
                if carrier is not None and xr[1].data_ptr() == carrier.data_ptr():
                    r_own = carrier[ops.ext["off"]:ops.ext["off"] + ops.part.n_local].reshape(-1, 1)
                    r_own._gla_io = (carrier, level)
                return own(xr[0]), r_own
            x1 = plan.jacobi_sweeps(d_e, b_e.reshape(-1, 1), x_e.reshape(-1, 1), w, self.n_pre, diag_is_matrix_diagonal=self._diag_from_matrix)       # (rows within n_pre lines of the
            return own(x1), own(plan.residual(b_e.reshape(-1, 1), x1))                               #  extension's ends are not valid: never read)
        x = self.smooth(level, self.n_pre, b, x)
        return x, self.residual(b, x, level)

    def postsmooth(self, level, b, x):
        ops = self.A_ops[level]
        w = float(torch.tensor(self.w, dtype=b.dtype))
        if ops.ext is not None and b.shape[1] == 1 and 1 <= self.n_post <= ops.TS_MAX and 1 <= self.n_pre < ops.TS_MAX:
            d_e, b_e, x_e, own = self._ghost_operands(level, None, x)      # (b and its ghost lines are still there from the pre-smoothing)
            # a coarse level's result is read once, by the prolongation of the level above: it lands in a carrier that is both this
            # pass's output (extended-block positions) and that operator's operand ([own rows | halo slots], filled afterwards)
            carrier = self._x_carrier(level, x.dtype) if level > 0 else None
            got = ops.ext["plan"].jacobi_sweeps(d_e, b_e.reshape(-1, 1), x_e.reshape(-1, 1), w, self.n_post,
                                                diag_is_matrix_diagonal=self._diag_from_matrix, out=carrier)
            res = own(got)
            if carrier is not None and got.data_ptr() == carrier.data_ptr():
                res._gla_px = (carrier, level)
            return res
        return self.smooth(level, self.n_post, b, x)

    def _x_carrier(self, level, dtype):
        """The buffer that is BOTH the post-smoothing pass's output on ghost-line level `level` (>= 1) and the operand of the
        prolongation from it (correct(level - 1)); None where the two layouts do not line up."""
        a_ops, pr = self.A_ops[level], (self.P_ops[level - 1] if 0 < level <= len(self.P_ops) else None)
        e = a_ops.ext
        if e is None or pr is None or pr.part.n_in != a_ops.part.n_local:
            return None
        key = ("x_carrier", dtype)
        buf = e["amg"].get(key)
        if buf is None:
            need = max(e["n_ext"], e["off"] + pr.part.n_in + pr.part.halo_size)
            buf = e["amg"][key] = torch.zeros(need, dtype=dtype, device=e["dev"])
        return buf

    def _r_carrier(self, level, dtype):
        """The buffer that is BOTH the residual output of a ghost-line level's smoothing pass and the restriction's operand
        (presmooth_residual); None where the two layouts do not line up."""
        a_ops, pt = self.A_ops[level], (self.Pt_ops[level] if level < len(self.Pt_ops) else None)
        e = a_ops.ext
        if e is None or pt is None or pt.part.n_in != a_ops.part.n_local:
            return None
        key = ("r_carrier", dtype)
        buf = e["amg"].get(key)
        if buf is None:
            need = max(e["n_ext"], e["off"] + pt.part.n_in + pt.part.halo_size)
            buf = e["amg"][key] = torch.zeros(need, dtype=dtype, device=e["dev"])
        return buf

    def restrict(self, level, r):
        ops = self.Pt_ops[level]
        nxt = self.A_ops[level + 1] if level + 1 < self.n_dist else None
        if nxt is not None and nxt.ext is not None and r.shape[1] == 1 and nxt.ext["dtype"] == r.dtype:
            # the coarse level's right-hand side lands in that level's extended block (its smoothing pass reads it there)
            e = nxt.ext
            rc = e["amg"]["b"][e["off"]:e["off"] + nxt.part.n_local].reshape(-1, 1)
            e["amg"]["b_src"] = None
        else:
            rc = torch.empty((ops.part.n_local, r.shape[1]), dtype=r.dtype, device=r.device)
        io = getattr(r, "_gla_io", None)
        if io is not None and io[1] == level and r.shape[1] == 1:
            # r already sits in the carrier, at the head of the restriction's operand: no copy (presmooth_residual)
            carrier, e = io[0], self.A_ops[level].ext
            buf = carrier[e["off"]:e["off"] + ops.part.n_in + ops.part.halo_size].reshape(-1, 1)
            ops.run(buf, lambda which: ops.plan.spmm(buf, out=rc, which=which))
        else:
            self._apply(ops, r, lambda buf, which: ops.plan.spmm(buf, out=rc, which=which))
        return rc

    def prolong(self, level, xc):
        ops = self.P_ops[level]
        out = torch.empty((ops.part.n_local, xc.shape[1]), dtype=xc.dtype, device=xc.device)
        self._apply(ops, xc, lambda buf, which: ops.plan.spmm(buf, out=out, which=which))
        return out

    def correct(self, level, x, xc):
        """x + P xc in one pass (gla_spmm_add) after one halo exchange of xc."""
        ops = self.P_ops[level]
        a_ops = self.A_ops[level]
        # the result is the post-smoother's operand: allocated with the ghost rows of the level's extended block, so that the
        # smoothing pass reads it in place
        out = a_ops.ext_tensor(1, dtype=x.dtype, zero=False) if (a_ops.ext is not None and x.shape[1] == 1) else torch.empty_like(x)
        px = getattr(xc, "_gla_px", None)
        if px is not None and px[1] == level + 1 and xc.shape[1] == 1:
            # xc already sits at the head of this operator's operand (postsmooth's carrier): no copy
            e = self.A_ops[level + 1].ext
            buf = px[0][e["off"]:e["off"] + ops.part.n_in + ops.part.halo_size].reshape(-1, 1)
            ops.run(buf, lambda which: ops.plan.spmm_add(x, buf, out=out, which=which))
        else:
            self._apply(ops, xc, lambda buf, which: ops.plan.spmm_add(x, buf, out=out, which=which))
        return out

    def capture_cycle(self, dtype=torch.float64):
        """This rank's cycle captured into a HIP graph (torch.cuda.CUDAGraph): returns run(b_local, x_local) -> x_new_local,
        bit-identical to cycle().  A rank's share of a deep hierarchy is ~70 launches of a few microseconds each plus the
        exchanges: driven from Python the host, not the GPU, bounds the cycle once the blocks are small (8 ranks: level 0 is
        2.1 M rows per rank).  Needs the library's RCCL communicator (or a single rank): torch.distributed's host-ordered
        collectives of the gloo rehearsal route cannot be captured.  Every rank must capture and replay alike."""
        dev = self.A_ops[0].plan.device if self.A_ops else self.sub.H.plans[0].device
        lo, hi = self.blocks[0][self.rank]
        sb = torch.zeros((hi - lo, 1), dtype=dtype, device=dev)
        sx = torch.zeros_like(sb)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self._capturing = True                               # (no operand copy may be skipped on the strength of an earlier call)
        try:
            with torch.cuda.stream(side):                    # warm-up off the capture: lazy allocations, code loads
                for _ in range(2):
                    self.cycle(sb, sx)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.cycle(sb, sx)
        finally:
            self._capturing = False
        for ops in self.A_ops:                               # the eager path must not trust what the capture left behind
            if ops.ext is not None and "amg" in ops.ext:
                ops.ext["amg"]["b_src"] = None

        def run(b, x):
            sb.copy_(b)
            sx.copy_(x)
            graph.replay()
            return out.clone()
        run.graph = graph
        return run

    def cycle(self, b, x, level=0, b_unchanged=False):
        """One V-cycle on this rank's rows: b, x are the LOCAL parts ([n_local, 1]) of the level vectors.
        b_unchanged=True: the caller vouches that b is the tensor of the previous cycle() call with the same contents (a solver
        loop on one right-hand side): its copy into the level's extended block is skipped."""
        if level == 0:
            self._b_unchanged = bool(b_unchanged)
        if level == self.n_dist:
            # replicated from here on: one all-gather of the right-hand side, the rest of the cycle on this GPU alone
            lo, hi = self.blocks[level][self.rank]
            b_full = self.gather(b, level).contiguous()
            # a coarse level starts from the zero guess (one cached read-only buffer); at level 0 -- a hierarchy that is
            # replicated as a whole (one level, or level 0 already below the agglomeration threshold) -- x is the CALLER's
            # initial guess and is gathered like b, as AMGHierarchy.cycle(b, x) uses it
            x0 = self.gather(x, level).contiguous() if level == 0 else self.sub.H._zero_guess(b_full)
            x_full = self.sub.H.cycle(b_full, x0, self.sub.level)
            return x_full[lo:hi].contiguous()
        x, r = self.presmooth_residual(level, b, x)
        rc = self.restrict(level, r)
        zeros = self.__dict__.setdefault("_zeros", {})       # the zero initial guess of a coarse level: one read-only buffer per
        key = (tuple(rc.shape), rc.dtype)                    # shape, not a fill launch per level and cycle
        z = zeros.get(key)
        if z is None:
            nxt = self.A_ops[level + 1] if level + 1 < self.n_dist else None
            if nxt is not None and nxt.ext is not None and rc.shape[1] == 1 and nxt.ext["dtype"] == rc.dtype:
                # a ghost-line level reads its x in place when x was allocated WITH its ghost rows (the zero guess' ghost rows
                # are zero on every rank: filling them writes zeros over zeros); a plain zero tensor was copied into the
                # level's block on every cycle (67 MB at 4.2 M rows, 15 us)
                z = zeros[key] = nxt.ext_tensor(1, dtype=rc.dtype)
            else:
                z = zeros[key] = torch.zeros_like(rc)
        xc = self.cycle(rc, z, level + 1)
        x = self.correct(level, x, xc)
        return self.postsmooth(level, b, x)


class _OnDev:
    """Stand-in for the level operator in GridSplitting.__call__(A, S, n), which only reads A.device."""

    def __init__(self, device):
        self.device = device
