"""Multi-rank V-cycle: an `AMGHierarchy` whose SOLVE phase is distributed by contiguous row blocks (BASELINE config 5:
"3-level AMG ... 8 x MI355X").

Setup (SOC -> splitting -> DI -> P -> Galerkin) is replicated: every rank builds the same hierarchy with the device
kernels (0.3 s for 16.8 M unknowns), then keeps only its row block of every level operator A_l, of every
prolongation P_l (rows: fine block, input vector: coarse partition) and restriction P_l^T (rows: coarse block, input
vector: fine partition).  A cycle then needs, per level, one halo all-gather per Jacobi sweep / residual / restriction
/ prolongation (dist.RowBlockPartition, overlapped with interior tiles) and, on the coarsest level, one all-gather of
the whole (small) right-hand side, which every rank solves redundantly.  Row arithmetic is rank-independent, so the
distributed cycle is bit-identical to the single-rank one.
"""
import torch
import torch.distributed as dist

from .ChebyGNN import chebyshev_relax
from .dist import RowBlockOps


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


class RowBlockAMG:
    def __init__(self, H, group=None, overlap=True):
        self.H, self.group = H, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
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
        w = float(torch.tensor(self.H.w, dtype=b.dtype))
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
        H = self.H
        if level == len(H.A) - 1:
            lo, hi = self.blocks[level][self.rank]
            b_full = self.gather(b, level)
            if H._coarse_lu is not None:
                return torch.linalg.lu_solve(H._coarse_lu[0], H._coarse_lu[1], b_full)[lo:hi].contiguous()
            x_full = self.gather(x, level)
            g = torch.tensor(H.coarse_bounds, dtype=b.dtype)
            for _ in range(H.coarse_sweeps):
                x_full = chebyshev_relax(H.plans[level], b_full.contiguous(), x_full.contiguous(), g, H.cheby.deg)[0]
            return x_full[lo:hi].contiguous()
        x = self.smooth(level, H.n_pre, b, x)
        r = self.residual(b, x, level)
        rc = self.restrict(level, r)
        xc = self.cycle(rc, torch.zeros_like(rc), level + 1)
        x = x + self.prolong(level, xc)
        return self.smooth(level, H.n_post, b, x)
