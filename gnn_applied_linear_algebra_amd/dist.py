"""Row-block domain decomposition of the message-passing loop across the GPUs of one node (RCCL over xGMI).

The reference has no distributed code; BASELINE.json defines this layer: the mesh is cut into
contiguous ROW BLOCKS (whole grid lines of the 2-D Laplacian), one block per rank = one process per GPU, and the
only data exchanged per sweep / SpMV are the HALO vertices: the x values of off-block columns.

Layout per rank (rows [row_lo, row_hi) of the global matrix, n_local of them):

    X buffer  [ n_local | P * S ] x K      local vertices, then the ALL-GATHERED send slots of every rank
    columns   local column c        -> c - row_lo
              column owned by rank q -> n_local + q * S + (position of c in q's send list)

S = longest send list (a rank's send list = its rows that ANY other rank reads; for the 5-point grid: its first
and last grid line).  One `all_gather_into_tensor` per sweep writes straight into the tail of the X buffer the
kernels gather from, so there is no unpack pass.  The collective is asynchronous: interior row tiles (no halo
column, classified once by gla_csr_classify_tiles) run while it is in flight, boundary tiles after it landed.

`RowBlockPartition` is pure index logic + the exchange (works on CPU tensors with the gloo backend, which is how
tests/test_dist_cpu.py covers it with world_size 2); `RowBlockJacobi` / `RowBlockOps` bind it to the HIP kernels.
"""
import torch
import torch.distributed as dist

from ._common import scalar0

class _Done:
    def wait(self):
        return True


def _tlog(msg, _t=[None]):
    """Setup-phase progress on stderr when GLA_DIST_LOG is set (rank 0): seconds since the previous mark."""
    import os
    if not os.environ.get("GLA_DIST_LOG") or (dist.is_initialized() and dist.get_rank() != 0):
        return
    import sys
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    now = time.time()
    print(f"  dist setup: +{0.0 if _t[0] is None else now - _t[0]:6.2f} s  {msg}", file=sys.stderr, flush=True)
    _t[0] = now


EXCHANGES = [0]      # collectives on the data path issued by this process (halo / ghost-line all-gathers, vector gathers): for reports

class LibComm:
    """An RCCL communicator owned by libgla_hip (gla_comm_t: ncclCommInitRank + a private communication stream and
    two events, csrc/gla_dist.hip) spanning the ranks of a torch process group.  With it the halo exchange is two
    C calls (gla_halo_begin / gla_halo_end, stream-ordered, no host wait) and JacobiGNN's whole sweep loop one
    (gla_dist_jacobi_sweeps).  Only for GPU ranks of an RCCL ("nccl") group, or a single process."""

    _cache = {}

    def __init__(self, group=None):
        import ctypes
        from ._lib import check, lib
        distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if distributed else 1
        self.rank = dist.get_rank(group) if distributed else 0
        ident = ctypes.create_string_buffer(128)
        rc0 = int(lib().gla_comm_unique_id(ident)) if self.rank == 0 else 0
        if self.world > 1:
            box = [rc0, bytes(ident.raw)]                    # (rank 0's status travels with the id: every rank fails or none)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group,
                                       device=torch.device("cuda", torch.cuda.current_device()))
            rc0, ident = int(box[0]), ctypes.create_string_buffer(box[1], 128)
        check(rc0)
        self._h = ctypes.c_void_p()
        check(lib().gla_comm_create(ident, self.rank, self.world, ctypes.byref(self._h)))

    @classmethod
    def for_group(cls, group=None):
        key = (id(group), torch.cuda.current_device())
        c = cls._cache.get(key)
        if c is None:
            c = cls._cache[key] = cls(group)
        return c

    @staticmethod
    def _agree(flag, group):
        """MIN of a local 0 / 1 over the ranks of `group` (tensor on the device the group's backend moves)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return int(flag)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        ok = torch.tensor([int(flag)], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        return int(ok.item())

    @classmethod
    def try_for_group(cls, group=None):
        """for_group(group), or None ON EVERY RANK when the library's communicator cannot be set up on one of them (librccl not
        loadable through dlopen, ncclGetUniqueId / ncclCommInitRank refusing): the caller then drives torch.distributed's own
        collectives -- slower (host-ordered), never silent: the reason goes to stderr and the driver's `mode` (bench.py's
        `config.exchange`) names the route that ran.

        The ranks agree in two steps.  (1) BEFORE anyone enters the collective ncclCommInitRank: every rank probes locally
        (gla_comm_probe: dlopen + symbols, no GPU call, nothing created) and the group takes the MIN -- a rank without RCCL
        would otherwise return from gla_comm_create at once and leave the others blocked inside the initialisation.
        (2) After the creation (rank 0's ncclGetUniqueId status travels with the id; ncclCommInitRank itself fails or succeeds
        as a group) a second MIN catches whatever is left."""
        import sys
        from ._lib import lib
        dev_id = torch.cuda.current_device() if torch.cuda.is_available() else -1
        fkey = ("unavailable", id(group), dev_id)
        if fkey in cls._cache:
            return None
        c, err = None, None
        ready = int(lib().gla_comm_probe()) == 0
        if not ready:
            err = "librccl could not be loaded on this rank (gla_comm_probe)"
        if cls._agree(ready, group):
            try:
                c = cls.for_group(group)
            except Exception as e:                           # noqa: BLE001 -- any failure takes every rank to the fallback
                err = e
            if not cls._agree(c is not None, group):
                c = None
        if c is None:
            cls._cache[fkey] = True
            print(f"gnn_applied_linear_algebra_amd.dist: libgla_hip's RCCL communicator is unavailable "
                  f"({err if err is not None else 'another rank could not create it'}); halo exchanges go through "
                  f"torch.distributed collectives on this group", file=sys.stderr, flush=True)
        return c

    @classmethod
    def release_all(cls):
        from ._lib import lib
        for c in cls._cache.values():
            if isinstance(c, cls) and c._h:
                lib().gla_comm_destroy(c._h)
                c._h = None
        cls._cache.clear()

    def all_reduce_sum(self, t):
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        assert t.is_cuda and t.is_contiguous()
        check(lib().gla_comm_all_reduce_sum(self._h, dtype_code(t.dtype), ptr(t), t.numel(), stream_ptr(t.device)))
        return t

def _use_lib_comm(group, device, host_loop):
    """The library-side exchange needs GPU operands and either a single process or an RCCL group."""
    if host_loop or device.type != "cuda":
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_backend(group) == "nccl"

def _gather_var(t, group):
    """All-gather of a tensor whose FIRST dimension differs from rank to rank, by tensor collectives (a count exchange, then one
    padded all_gather_into_tensor) -- no pickling, no host round trip on RCCL groups.  Returns the list of every rank's tensor
    (on t's device).  gloo groups with GPU tensors (the one-GPU rehearsals) stage through host memory."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":                   # RCCL moves device memory
        cdev = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    else:
        cdev = torch.device("cpu")
    cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=cdev)
    cnts = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    cnts = [int(c) for c in cnts.tolist()]
    m = max(cnts + [1])
    tail = tuple(t.shape[1:])
    send = torch.zeros((m,) + tail, dtype=t.dtype, device=cdev)
    send[:t.shape[0]] = t.to(cdev)
    recv = torch.empty((world * m,) + tail, dtype=t.dtype, device=cdev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return [recv[q * m:q * m + cnts[q]].to(t.device) for q in range(world)]


def _publish_rows(ei, ev, row_lo, row_hi, wanted_by_rank, group):
    """Sparse-row exchange (setup only): `wanted_by_rank[q]` = sorted global row ids rank q asks for.  Every rank publishes
    those of its rows [row_lo, row_hi) that anyone wants (tensor collectives, _gather_var); returns the (edge_index, values)
    of all published rows of OTHER ranks that this rank asked for, row-sorted, on ei's device."""
    rank = dist.get_rank(group)
    dev = ei.device
    mine = [w[(w >= row_lo) & (w < row_hi)] for q, w in enumerate(wanted_by_rank) if q != rank]
    pub_ids = torch.unique(torch.cat(mine)) if mine else torch.zeros(0, dtype=torch.int64, device=dev)
    if pub_ids.numel():
        # (a lookup table over this rank's rows: torch.isin sorts the whole edge list, seconds at 10^8 edges)
        flag = torch.zeros(row_hi - row_lo, dtype=torch.bool, device=dev)
        flag[(pub_ids - row_lo).to(dev)] = True
        m = flag[ei[0] - row_lo]
        pub_i, pub_v = ei[:, m].t().contiguous(), ev.reshape(-1)[m].contiguous()
    else:
        pub_i, pub_v = torch.zeros((0, 2), dtype=torch.int64, device=dev), torch.zeros(0, dtype=ev.dtype, device=dev)
    _tlog("publish rows: this rank's published rows selected")
    pool_i, pool_v = _gather_var(pub_i, group), _gather_var(pub_v, group)
    _tlog("publish rows: gathered")
    want = wanted_by_rank[rank].to(dev)
    got_i, got_v = [], []
    for q, (pi, pv) in enumerate(zip(pool_i, pool_v)):
        if q == rank or pi.shape[0] == 0:
            continue
        pi = pi.t()
        pos = torch.searchsorted(want, pi[0]).clamp_(max=max(int(want.numel()) - 1, 0))     # `want` is sorted
        m = want[pos] == pi[0] if want.numel() else torch.zeros(pi.shape[1], dtype=torch.bool, device=dev)
        got_i.append(pi[:, m])
        got_v.append(pv[m])
    if not got_i:
        return torch.zeros((2, 0), dtype=torch.int64, device=dev), torch.zeros(0, dtype=ev.dtype, device=dev)
    gi, gv = torch.cat(got_i, 1), torch.cat(got_v)
    order = torch.argsort(gi[0], stable=True)                  # owners are disjoint row ranges: row-sorted overall
    return gi[:, order].contiguous(), gv[order].contiguous()


def _all_wanted(need, group):
    """Every rank's sorted list of wanted global row ids (tensor collective)."""
    return _gather_var(need.to(torch.int64).contiguous(), group)


class RowBlockPartition:
    def __init__(self, edgeij_pair, n_global, row_lo, row_hi, group=None, col_blocks=None):
        """edgeij_pair [2, nnz_local]: the edges whose ROW lies in [row_lo, row_hi), GLOBAL vertex ids.
        Square operators (default): the input vector is partitioned like the rows.  Rectangular operators
        (restriction P^T, prolongation P): `col_blocks` = [(lo, hi)] * world, the partition of the INPUT vector."""
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_global, self.row_lo, self.row_hi = int(n_global), int(row_lo), int(row_hi)
        self.n_local = self.row_hi - self.row_lo
        dev = edgeij_pair.device
        row, col = edgeij_pair[0], edgeij_pair[1]
        if row.numel() and (int(row.min()) < self.row_lo or int(row.max()) >= self.row_hi):
            raise ValueError("edge rows outside this rank's row block")
        if self.distributed:
            blocks = [None] * self.world
            dist.all_gather_object(blocks, (self.row_lo, self.row_hi), group=group)
        else:
            blocks = [(self.row_lo, self.row_hi)]
        los = torch.tensor([b[0] for b in blocks])
        his = torch.tensor([b[1] for b in blocks])
        if int(los[0]) != 0 or int(his[-1]) != self.n_global or bool((los[1:] != his[:-1]).any()):
            raise ValueError(f"row blocks do not tile [0, {self.n_global}): {blocks}")
        self.blocks = blocks
        in_blocks = blocks if col_blocks is None else [(int(lo), int(hi)) for lo, hi in col_blocks]
        if len(in_blocks) != self.world or in_blocks[0][0] != 0 or \
                any(in_blocks[q][1] != in_blocks[q + 1][0] for q in range(self.world - 1)):
            raise ValueError(f"input-vector blocks do not tile their range: {in_blocks}")
        self.in_blocks = in_blocks
        self.in_lo, self.in_hi = in_blocks[self.rank]
        self.n_in = self.in_hi - self.in_lo                       # entries of the input vector this rank owns
        in_his = torch.tensor([b[1] for b in in_blocks])
        if col.numel() and (int(col.min()) < 0 or int(col.max()) >= in_blocks[-1][1]):
            raise ValueError("edge columns outside the input vector")

        _tlog("partition: blocks known")
        # the off-block input entries this rank reads
        off = (col < self.in_lo) | (col >= self.in_hi)
        _tlog("partition: off-block column mask")
        need = torch.unique(col[off]).cpu()                       # sorted global ids
        _tlog("partition: unique off-block columns")
        if self.distributed:
            needs = [t_.cpu() for t_ in _gather_var(need, group)]
        else:
            needs = [need]
        # send list of rank q = sorted union of q's input entries that any rank needs
        all_need = torch.unique(torch.cat(needs)) if self.world > 1 else need
        send_lists = [all_need[(all_need >= lo) & (all_need < hi)] for lo, hi in in_blocks]
        self.S = max([int(s.numel()) for s in send_lists] + [0])
        self.halo_size = self.world * self.S
        self.send_idx = (send_lists[self.rank] - self.in_lo).to(dev)           # local input ids, sorted
        self.n_send = int(self.send_idx.numel())

        # column renumbering: local input entries | gathered slots
        owner = torch.bucketize(need, in_his, right=True)                       # rank owning each halo column
        slot = torch.empty_like(need)
        for q in range(self.world):
            m = owner == q
            if bool(m.any()):
                slot[m] = q * self.S + torch.searchsorted(send_lists[q], need[m])
        _tlog("partition: send lists and slots (host)")
        need_d, slot_d = need.to(dev), slot.to(dev)
        col_local = col - self.in_lo
        if need_d.numel():
            pos = torch.searchsorted(need_d, col[off])
            col_local = col_local.clone()
            col_local[off] = self.n_in + slot_d[pos]
        self.edge_index_local = torch.stack([row - self.row_lo, col_local])
        self.n_cols_local = self.n_in + self.halo_size
        _tlog("partition: halo lists and local column numbering")
        self._send_bufs, self._send_flip = {}, 0

    # ---- buffers / exchange -------------------------------------------------------------------------
    def new_buffer(self, K, dtype, device):
        return torch.zeros((self.n_cols_local, K), dtype=dtype, device=device)

    def exchange(self, X, async_op=False):
        """All-gather the send slots of every rank into X[n_in:].  X: [n_in + P*S, K] (n_in = n_local when square)."""
        if self.world == 1 or self.S == 0:
            return None
        K = X.shape[1]
        key = (K, X.dtype, X.device)
        send = self._send_bufs.get(key)
        if send is None:                       # two buffers per shape: the previous collective may still read one
            self._send_bufs[key] = send = [torch.zeros((self.S, K), dtype=X.dtype, device=X.device) for _ in (0, 1)]
        self._send_flip ^= 1
        send = send[self._send_flip]
        if self.n_send:
            torch.index_select(X[:self.n_in], 0, self.send_idx, out=send[:self.n_send])
        if X.is_cuda and dist.get_backend(self.group) == "gloo":
            # test-only route (2 ranks sharing one GPU, tests/test_gpu_dist.py): gloo moves host memory
            send_h = send.cpu()
            recv_h = torch.empty((self.halo_size, K), dtype=X.dtype)
            dist.all_gather_into_tensor(recv_h, send_h, group=self.group)
            X[self.n_in:] = recv_h.to(X.device)
            return _Done() if async_op else None
        return dist.all_gather_into_tensor(X[self.n_in:], send, group=self.group, async_op=async_op)

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

class RowBlockOps:
    """The partition bound to a libgla_hip CSR plan of the local block (rows local, columns local + halo)."""

    def __init__(self, edgeij_pair, edge_attr, n_global, row_lo, row_hi, group=None, overlap=True, col_blocks=None,
                 host_loop=False):
        from .plan import CsrPlan, ALL_TILES, INTERIOR_TILES, BOUNDARY_TILES
        self.ALL, self.INTERIOR, self.BOUNDARY = ALL_TILES, INTERIOR_TILES, BOUNDARY_TILES
        self.part = RowBlockPartition(edgeij_pair, n_global, row_lo, row_hi, group, col_blocks)
        p = self.part
        self.plan = CsrPlan(p.edge_index_local, p.n_local, p.n_cols_local)
        self.plan.set_values(edge_attr)
        self.overlap = bool(overlap) and p.world > 1 and p.S > 0
        if self.overlap:
            self.plan.classify_tiles(p.n_in)
        _tlog("block plan built")
        # exchange route: libgla_hip's own RCCL communicator (stream-ordered, no host wait) for GPU ranks of an RCCL
        # group; torch.distributed collectives driven from Python otherwise (gloo rehearsals, CPU index tests)
        self.comm = None
        if p.world > 1 and p.S > 0 and _use_lib_comm(group, edgeij_pair.device, host_loop):
            self.comm = LibComm.try_for_group(group)
            if self.comm is not None:
                self.send_idx32 = p.send_idx.to(torch.int32).contiguous()
        self._lib_send = {}
        self.mode = ("no exchange" if p.world == 1 or p.S == 0 else
                     ("library-side: gla_halo_begin/end on libgla_hip's RCCL communicator, event-ordered, "
                      if self.comm is not None else "host-side: torch.distributed all_gather_into_tensor per sweep, ")
                     + ("interior/boundary overlap" if self.overlap else "no overlap"))

    def io_buffer(self, K, dtype, device):
        """The [n_in + P * S, K] operand buffer of this operator (local input entries, then the gathered send slots), allocated
        once per (K, dtype): the exchange overwrites the whole tail, the caller the head."""
        key = ("io", K, dtype)
        buf = self._lib_send.get(key)
        if buf is None:
            buf = self._lib_send[key] = self.part.new_buffer(K, dtype, device)
        return buf

    def _send_buf(self, X):
        key = (X.shape[1], X.dtype)
        buf = self._lib_send.get(key)
        if buf is None:
            buf = self._lib_send[key] = torch.zeros((self.part.S, X.shape[1]), dtype=X.dtype, device=X.device)
        return buf

    def run(self, X, launch):
        """Exchange X's halo and call launch(which_tiles) for all row tiles, overlapping when enabled."""
        p = self.part
        if not (p.S == 0 or (p.world == 1 and self.comm is None)):
            EXCHANGES[0] += 1
        if p.S == 0 or (p.world == 1 and self.comm is None):
            launch(self.ALL)
        elif self.comm is not None:
            from ._lib import check, dtype_code, lib, ptr, stream_ptr
            st = stream_ptr(X.device)
            check(lib().gla_halo_begin(self.comm._h, dtype_code(X.dtype), X.shape[1], ptr(X), p.n_in,
                                       ptr(self.send_idx32), p.n_send, p.S, ptr(self._send_buf(X)), st))
            if self.overlap:
                launch(self.INTERIOR)
            check(lib().gla_halo_end(self.comm._h, st))
            launch(self.BOUNDARY if self.overlap else self.ALL)
        elif self.overlap:
            work = p.exchange(X, async_op=True)
            launch(self.INTERIOR)
            work.wait()
            launch(self.BOUNDARY)
        else:
            p.exchange(X)
            launch(self.ALL)

    # ---- the block extended by ghost grid lines (temporally blocked routes) --------------------------------------------
    TS_MAX = 8                                               # ghost lines either side = the most sweeps / steps a pass carries
    ext = None

    def _setup_ghost(self, edgeij_pair, edge_attr, need_jacobi_passes=True):
        """Build the plan of the block extended by TS_MAX ghost lines either side (true global rows for the ghost lines
        within TS_MAX - 1 of the block, fetched from their owners); leaves self.ext = None when the block is not a 2-D
        stencil, too small, or the extended plan does not qualify for the fused kernel."""
        from .plan import CsrPlan
        p, T = self.part, self.TS_MAX
        info = self.plan.info
        S = int(info.dia_stride)
        ok = int(S > 0 and int(info.dia_A) == 1 and all(hi - lo >= T * S for lo, hi in p.blocks)
                 and all(lo % S == 0 for lo, _ in p.blocks) and p.n_global % S == 0)
        flag = torch.tensor([ok], dtype=torch.int64)
        if p.distributed:                                  # every rank must take the same route (collectives below)
            flags = [None] * p.world
            dist.all_gather_object(flags, ok, group=p.group)
            ok = int(all(flags))
        if not ok:
            return
        lo, hi, n = p.row_lo, p.row_hi, p.n_global
        elo, ehi = max(0, lo - T * S), min(n, hi + T * S)                  # extended x domain
        # Matrix rows of EVERY ghost line are fetched.  The outermost one is needed only for its x values (a pass carries at most T
        # sweeps), but with its true rows -- cut off where they point out of the extended block, exactly like the first / last
        # line of a grid -- the extended block is a proper stencil matrix: in particular a symmetric one where the global matrix
        # is, so that the block's passes read the lower diagonals from their upper partners as the single-GPU plan does (option
        # "sym"; until round 4 the outermost ghost line had no rows and the extended plan could never qualify)
        glo, ghi = elo, ehi
        dev = edgeij_pair.device
        want = torch.cat([torch.arange(glo, lo), torch.arange(hi, ghi)])
        ev = edge_attr[:, 0] if edge_attr.dim() == 2 else edge_attr
        g_ei, g_ev = _publish_rows(edgeij_pair, ev, lo, hi, _all_wanted(want, p.group), p.group)
        _tlog("ghost rows of the matrix fetched")
        all_ei = torch.cat([g_ei[:, g_ei[0] < lo], edgeij_pair, g_ei[:, g_ei[0] >= hi]], 1)
        all_ev = torch.cat([g_ev[g_ei[0] < lo], ev.reshape(-1), g_ev[g_ei[0] >= hi]])
        inside = (all_ei[1] >= elo) & (all_ei[1] < ehi)
        # only the outermost ghost lines may point out of the extended block; anything else is not a stencil this route handles
        outer = (all_ei[0] < elo + S) | (all_ei[0] >= ehi - S)
        good = int(bool((inside | outer).all()))
        if good and not bool(inside.all()):
            all_ei, all_ev = all_ei[:, inside], all_ev[inside]
        plan = None
        _tlog("extended edge list assembled")
        if good:
            plan = CsrPlan(all_ei - elo, ehi - elo, ehi - elo)
            _tlog("extended plan: structure")
            plan.set_values(all_ev)
            sched = plan.jacobi_schedule(4 * T, all_ev.dtype)
            good = int((not need_jacobi_passes or 1 < max(sched) <= T) and int(plan.info.dia_stride) == S)
            self._sched = plan.jacobi_schedule
        _tlog("extended plan built")
        goods = [None] * p.world
        dist.all_gather_object(goods, good, group=p.group)
        if not all(goods):
            return
        up = p.rank > 0 and lo - elo == T * S                # a neighbour above / below supplies T ghost lines
        down = p.rank < p.world - 1 and ehi - hi == T * S
        self.ext = dict(plan=plan, S=S, elo=elo, ehi=ehi, n_ext=ehi - elo, off=lo - elo, up=up, down=down,
                        slab=T * S, dtype=all_ev.dtype, dev=dev)

    def _ghost_exchange(self, owns, begin_only=False, st=None):
        """ONE all-gather of every rank's first and last TS_MAX grid lines of each vector in `owns` ([n_local] or [n_local, K]
        each, rows may be strided); returns per vector the (upper, lower) ghost slabs for this rank, flat (None where there is
        no neighbour).  With the library communicator the gather runs on its side stream: call with begin_only=True, launch
        work that needs no ghosts, then _ghost_finish().  `st`: the HIP stream (ctypes pointer) everything is ordered on;
        default: torch's current stream."""
        e, p = self.ext, self.part
        m = e["slab"]
        EXCHANGES[0] += 1
        widths = [own.numel() // max(own.shape[0], 1) for own in owns]
        chunk = 2 * m * sum(widths)
        if self.comm is not None and all(own.dim() == 1 for own in owns):
            # (the hot case: single-rhs vectors.)  Persistent send / receive buffers per layout, packed by ONE gla_copy_batch launch
            # on the given stream: no torch operator, hence no stream context and no allocator call on this path -- the host
            # cost of a pass is what bounds small blocks (profiles/r03/dist_route_host_cost.txt)
            from ._lib import check, dtype_code, lib, ptr, stream_ptr
            dt, dev = owns[0].dtype, owns[0].device
            key = ("xbuf", tuple(widths), dt)
            bufs = self._lib_send.get(key) if hasattr(self, "_lib_send") else None
            if bufs is None:
                bufs = (torch.empty(chunk, dtype=dt, device=dev), torch.empty(p.world * chunk, dtype=dt, device=dev))
                if not hasattr(self, "_lib_send"):
                    self._lib_send = {}
                self._lib_send[key] = bufs
            send, recv = bufs
            st = stream_ptr(dev) if st is None else st
            pairs, pos = [], 0
            for own in owns:
                pairs += [(own[:m], send[pos:pos + m]), (own[-m:], send[pos + m:pos + 2 * m])]
                pos += 2 * m
            self._copy_batch(pairs, st)
            if not begin_only:
                # nothing to overlap with: the gather runs IN LINE on the caller's stream.  Through the side stream it costs
                # two cross-queue dependencies, 12 us each on this runtime around a 4 us gather (profiles/r05/dist_vcycle_trace.txt)
                check(lib().gla_comm_all_gather(self.comm._h, dtype_code(dt), ptr(send), ptr(recv), chunk, st))
                self._ghost_pending = (send, recv, widths, False)
                return self._ghost_finish(st)
            check(lib().gla_comm_all_gather_begin(self.comm._h, dtype_code(dt), ptr(send), ptr(recv), chunk, st))
            self._ghost_pending = (send, recv, widths, True)
            return None
        send = torch.cat([piece.reshape(-1) for own in owns for piece in (own[:m], own[-m:])])
        recv = torch.empty(p.world * chunk, dtype=send.dtype, device=send.device)
        if self.comm is not None:
            from ._lib import check, dtype_code, lib, ptr, stream_ptr
            check(lib().gla_comm_all_gather_begin(self.comm._h, dtype_code(send.dtype), ptr(send), ptr(recv),
                                                  chunk, stream_ptr(send.device) if st is None else st))
            self._ghost_pending = (send, recv, widths, True)    # keep the buffers alive until the gather has been waited for
        else:
            if send.is_cuda and dist.get_backend(p.group) == "gloo":   # rehearsal route: gloo moves host memory
                recv_h = torch.empty(recv.numel(), dtype=send.dtype)
                dist.all_gather_into_tensor(recv_h, send.cpu(), group=p.group)
                recv = recv_h.to(send.device)
            else:
                dist.all_gather_into_tensor(recv, send, group=p.group)
            self._ghost_pending = (send, recv, widths, False)
        return None if begin_only else self._ghost_finish(st)

    def _ghost_finish(self, st=None):
        e, p = self.ext, self.part
        m = e["slab"]
        send, recv, widths, on_side_stream = self._ghost_pending
        if on_side_stream:
            from ._lib import check, lib, stream_ptr
            check(lib().gla_halo_end(self.comm._h, stream_ptr(recv.device) if st is None else st))
        chunk = 2 * m * sum(widths)                          # a rank's chunk: per vector [first m rows | last m rows]
        out, base = [], 0
        ru, rd = e.get("up_src", p.rank - 1), e.get("down_src", p.rank + 1)      # (overridable: one-rank self-halo tests)
        for w_ in widths:
            up = recv[ru * chunk + base + m * w_:ru * chunk + base + 2 * m * w_] if e["up"] else None
            lo = recv[rd * chunk + base:rd * chunk + base + m * w_] if e["down"] else None
            out.append((up, lo))                             # the neighbour above's LAST lines / below's FIRST lines
            base += 2 * m * w_
        return out

    @staticmethod
    def _copy_batch(pairs, st=None):
        """dst[:] = src for every (src, dst) pair of 1-D tensors (src may be strided) in ONE launch (gla_copy_batch) on the HIP
        stream `st` (default: torch's current stream); plain copies for CPU tensors."""
        pairs = [(s_, d_) for s_, d_ in pairs if d_.numel()]
        if not pairs:
            return
        if not pairs[0][1].is_cuda:
            for s_, d_ in pairs:
                d_.copy_(s_)
            return
        import ctypes
        from ._lib import check, dtype_code, lib, stream_ptr
        for i in range(0, len(pairs), 8):
            chunk = pairs[i:i + 8]
            k = len(chunk)
            for s_, d_ in chunk:
                assert s_.dim() == 1 and d_.dim() == 1 and d_.is_contiguous() and s_.numel() == d_.numel() and s_.dtype == d_.dtype
            src = (ctypes.c_void_p * k)(*[s_.data_ptr() for s_, _ in chunk])
            dst = (ctypes.c_void_p * k)(*[d_.data_ptr() for _, d_ in chunk])
            cnt = (ctypes.c_int64 * k)(*[d_.numel() for _, d_ in chunk])
            strd = (ctypes.c_int64 * k)(*[max(1, s_.stride(0)) for s_, _ in chunk])
            check(lib().gla_copy_batch(dtype_code(chunk[0][1].dtype), k, src, dst, cnt, strd,
                                       stream_ptr(chunk[0][1].device) if st is None else st))

    def _fill_ghosts(self, ext_vecs, slabs, st=None):
        """Ghost lines of the extended vectors ([n_ext] or contiguous [n_ext, K]) from the gathered slabs, one launch."""
        e = self.ext
        pairs = []
        for vec, (upper, lower) in zip(ext_vecs, slabs):
            if upper is not None:
                pairs.append((upper, vec[:e["off"]].reshape(-1)))
            if lower is not None:
                pairs.append((lower, vec[e["off"] + self.part.n_local:].reshape(-1)))
        self._copy_batch(pairs, st)

    def ext_tensor(self, *trailing, dtype=None, zero=True):
        """An operand allocated WITH its ghost rows: the rank's own-row view [n_local, *trailing] of a zero tensor
        [n_ext, *trailing].  The ghost-line routes (RowBlockJacobi on [n_local, 3, K], RowBlockCheby's b and x0) read such an
        operand in place -- ghost rows are filled around it, nothing is copied: a plain tensor costs one copy into the
        extended block per call (2 x 4.2 GB for b and x at 4096^2 x 16, more than half of what the fused Chebyshev pass
        itself moves).  Without a ghost-line route: an ordinary tensor."""
        e, nl = self.ext, self.part.n_local
        if e is None:
            return torch.zeros((nl, *trailing), dtype=dtype or torch.float64,
                               device=self.plan.device if hasattr(self, "plan") else None)
        alloc = torch.zeros if zero else torch.empty         # (zero=False: the caller writes every own row; ghost rows get filled)
        full = alloc((e["n_ext"], *trailing), dtype=dtype or e["dtype"], device=e["dev"])
        view = full[e["off"]:e["off"] + nl]
        view._gla_ext = full
        return view

    def _ext_of(self, t, trailing):
        """The extended tensor behind an own-row view made by ext_tensor() (None for any other tensor)."""
        full, e = getattr(t, "_gla_ext", None), self.ext
        if full is None or e is None or full.shape[0] != e["n_ext"] or tuple(full.shape[1:]) != tuple(trailing) \
                or full.dtype != t.dtype or full.device != t.device or not full.is_contiguous() \
                or t.shape[0] != self.part.n_local or t.data_ptr() != full[e["off"]:].data_ptr():
            return None
        return full

    def all_reduce_sum(self, t):
        """Sum over the ranks (power method partial sums): stream-ordered on the library's communicator when there is
        one, torch.distributed otherwise."""
        if self.comm is not None and t.is_cuda:
            return self.comm.all_reduce_sum(t)
        return self.part.all_reduce_sum(t)

class RowBlockJacobi(RowBlockOps):
    """JacobiGNN.forward on a row block: same arguments as the module, local rows only (JacobiGNN.py:138-148).

    Two routes.  (1) One halo exchange per sweep (north_star's wording): interior / boundary tiles of the block's own
    plan.  (2) `temporal=True` (default) on 2-D stencil blocks with one right-hand side: the temporally blocked kernel
    of the single-GPU path (5, 4 or 2 sweeps per pass over the data) on the block EXTENDED by ghost lines, with ONE
    exchange per pass -- the 5 grid lines either side that a pass can depend on -- instead of one per sweep
    (communication-avoiding; the ghost lines are recomputed redundantly, exactly like the overlap between the line
    blocks of one GPU).  The ghost rows of the matrix are fetched once at construction, those of b and A_ii once per
    forward().  Same per-row arithmetic: bit-identical to route (1) and to the single-GPU module."""

    def __init__(self, edgeij_pair, edge_attr, n_global, row_lo, row_hi, group=None, overlap=True, col_blocks=None,
                 host_loop=False, temporal=True):
        super().__init__(edgeij_pair, edge_attr, n_global, row_lo, row_hi, group, overlap, col_blocks, host_loop)
        self.ext = None
        if temporal and col_blocks is None and self.part.world > 1 and edgeij_pair.is_cuda:
            self._setup_ghost(edgeij_pair, edge_attr)
        if self.ext is not None:
            self.mode = ("temporally blocked: one ghost-line exchange per pass of up to 8 sweeps, "
                         + ("libgla_hip RCCL all-gather on a side stream, interior / boundary line blocks overlapped"
                            if self.comm is not None else "torch.distributed all-gather (host-ordered)"))

    def _forward_temporal(self, n_iters, va, w):
        import ctypes
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        e, p = self.ext, self.part
        n_ext, off, nl = e["n_ext"], e["off"], p.n_local
        dev, dt = va.device, va.dtype
        code, st, h = dtype_code(dt), stream_ptr(dev), e["plan"]._h
        buf = e.get("buffers")
        if buf is None:                                      # persistent extended vectors: A_ii, b, two x buffers
            buf = e["buffers"] = (torch.ones(n_ext, dtype=dt, device=dev), torch.zeros(n_ext, dtype=dt, device=dev),
                                  torch.zeros(n_ext, dtype=dt, device=dev), torch.zeros(n_ext, dtype=dt, device=dev))
        d_ext, b_ext, cur, nxt = buf
        own = lambda v: v[off:off + nl]                      # noqa: E731
        passes = list(self._sched(int(n_iters), dt))         # the single-GPU path's launch plan (fewest passes)
        T, S = self.TS_MAX, e["S"]
        # First pass straight from the AoS rows (as the single-GPU path does): the interior line blocks of a pass read
        # nothing but the rank's own rows, so they can take A_ii, b, x from vertex_attr itself (plan row r = tensor row
        # r - off) and leave the contiguous copies of A_ii and b behind; only the few lines the two edge blocks touch are
        # copied by hand.  Otherwise (no library communicator, tiny blocks, single sweeps): one split pass into the
        # extended vectors.
        aos_first = (self.comm is not None and self.overlap and passes[0] > 1 and n_ext // S > 2 * (2 * T + passes[0])
                     and va.stride(1) == 1)
        if aos_first and min(passes) > 1 and int(e["plan"].info.dia_canon5) == 1 and e.get("c_route", True):
            # the whole call -- every pass's exchange, interior launch, ghost fill, edge launch and the next exchange -- enqueued
            # by ONE library call on this stream and a second one (csrc/gla_dist.hip: gla_dist_jacobi_temporal)
            m = T * S
            xb = e.get("xbufs")
            if xb is None or xb[0].dtype != dt:
                xb = e["xbufs"] = (torch.empty(6 * m, dtype=dt, device=dev), torch.empty(p.world * 6 * m, dtype=dt, device=dev))
            side = e.get("side_stream")
            if side is None:
                side = e["side_stream"] = torch.cuda.Stream(device=dev)
            x_last = torch.empty(n_ext, dtype=dt, device=dev)
            arr = (ctypes.c_int * len(passes))(*passes)
            rc = lib().gla_dist_jacobi_temporal(h, self.comm._h, code, ptr(va), va.stride(0), off, nl, ptr(d_ext), ptr(b_ext),
                                                ptr(cur), ptr(nxt), ptr(x_last), ptr(xb[0]), ptr(xb[1]), T, int(e["up"]),
                                                int(e["down"]), int(e.get("up_src", p.rank - 1)), int(e.get("down_src", p.rank + 1)),
                                                w, arr, len(passes), st, ctypes.c_void_p(side.cuda_stream))
            if rc == 0:
                EXCHANGES[0] += len(passes)
                self._ghost_pending = None
                return own(x_last).reshape(-1, 1)
            if rc != -5:
                check(rc)
            e["c_route"] = False                             # (the library declined this block: Python-driven route from now on)
        if aos_first:
            E = (2 * T + passes[0]) * S                      # rows the first / last edge block (2 T lines high) reads
            pairs = []
            for a, b_ in ((0, E), (n_ext - E, n_ext)):
                a, b_ = max(a, off), min(b_, off + nl)       # ... that are the rank's own
                if a < b_:
                    pairs += [(va[a - off:b_ - off, c_], vec[a:b_]) for c_, vec in enumerate((d_ext, b_ext, cur))]
            self._copy_batch(pairs)
        else:
            outs = (ctypes.c_void_p * 3)(own(d_ext).data_ptr(), own(b_ext).data_ptr(), own(cur).data_ptr())
            check(lib().gla_split_columns(code, nl, 3, ptr(va), va.stride(0) if nl > 1 else 3, outs, st))
        first = True
        prefetched = False
        spare = nxt
        for i, ts in enumerate(passes):
            if i == len(passes) - 1:                         # the last pass writes into a fresh buffer the caller keeps
                nxt = torch.empty(n_ext, dtype=dt, device=dev)
            # ghost lines of x every pass; those of A_ii and b ride along in the first exchange of the call
            vecs = [d_ext, b_ext, cur] if first else [cur]
            owns = [va[:, 0], va[:, 1], va[:, 2]] if (first and aos_first) else [own(v) for v in vecs]
            from_aos = first and aos_first
            first = False
            if ts == 1:                                      # one ordinary sweep of the extended block
                self._fill_ghosts(vecs, self._ghost_exchange(owns))
                check(lib().gla_jacobi_sweep(h, code, 1, ptr(d_ext), ptr(b_ext), 1, ptr(cur), 1, ptr(nxt), 1, w, 0, st))
            elif self.comm is not None and self.overlap:
                # ghost lines in flight on the communicator's stream while the interior line blocks run on this one; a
                # second stream waits for the gather, fills the ghost lines and runs the (short) first / last line block
                # CONCURRENTLY with the interior launch -- the two launches write disjoint lines of nxt
                main = torch.cuda.current_stream(dev)
                side = e.get("side_stream")
                if side is None:
                    side = e["side_stream"] = torch.cuda.Stream(device=dev)
                st_side = ctypes.c_void_p(side.cuda_stream)
                if not prefetched:
                    self._ghost_exchange(owns, begin_only=True, st=st)
                side.wait_stream(main)                        # cur / nxt as the previous pass left them
                if from_aos:
                    check(lib().gla_jacobi_fused_parts_aos(h, code, ptr(va), va.stride(0), off, nl, ptr(d_ext), ptr(b_ext),
                                                           ptr(nxt), w, ts, 1, T, st))
                    # The AoS interior launch WRITES the contiguous copies of A_ii and b (d_ext / b_ext).  On canonical 5-point
                    # plans it leaves the T + 2 ts lines either end -- everything the edge blocks read, hand-copied above --
                    # alone (gla_march5.hip: aos_wlo / aos_whi), so the two launches touch disjoint words.  The other stencil
                    # kernels write every line they own: there the edge blocks are ordered after the interior launch
                    if int(e["plan"].info.dia_canon5) != 1:
                        side.wait_stream(main)
                else:
                    check(lib().gla_jacobi_fused_parts(h, code, ptr(d_ext), ptr(b_ext), ptr(cur), ptr(nxt), w, ts, 1, T, st))
                # (everything below is ordered on the second stream by explicit stream pointers: library calls only)
                self._fill_ghosts(vecs, self._ghost_finish(st_side), st_side)
                check(lib().gla_jacobi_fused_parts(h, code, ptr(d_ext), ptr(b_ext), ptr(cur), ptr(nxt), w, ts, 2, T, st_side))
                # The edge blocks are 2 T lines high: they alone produce the T own lines either end that the neighbours
                # want for the NEXT pass, so its exchange starts here, on the second stream, while the interior launch of
                # this pass is still running (nothing of it is left on the critical path of the main stream).
                prefetched = i + 1 < len(passes) and passes[i + 1] > 1
                if prefetched:
                    self._ghost_exchange([own(nxt)], begin_only=True, st=st_side)
                main.wait_stream(side)
            else:
                self._fill_ghosts(vecs, self._ghost_exchange(owns))
                check(lib().gla_jacobi_fused(h, code, ptr(d_ext), ptr(b_ext), ptr(cur), ptr(nxt), w, ts, 0, st))
            cur, nxt = nxt, cur
        # cur = the result (fresh buffer); the two persistent x buffers stay with the driver
        e["buffers"] = (d_ext, b_ext, nxt, spare if len(passes) % 2 == 1 else buf[2])
        self._ghost_pending = None
        return own(cur).reshape(-1, 1)

    def _forward_temporal_k(self, n_iters, va, w):
        """K right-hand sides (vertex_attr [n_local, 3, K]) on the block extended by ghost lines: passes of up to five sweeps
        (canonical 9-point plans: three) -- the K-wide temporally blocked kernel of the single-GPU path, jacobi_marchk_kernel --
        with ONE exchange of the TS_MAX ghost lines per pass instead of one per sweep; the ghost lines of b and A_ii ride along
        in the first exchange.  Ghost lines are recomputed redundantly; own rows: the arithmetic of single sweeps, bit for bit.
        Returns None when the library declines (K not a multiple of the 16-byte vector, ...): exchange-per-sweep route."""
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        e, p = self.ext, self.part
        n_ext, off, nl = e["n_ext"], e["off"], p.n_local
        dev, dt, K = va.device, va.dtype, va.shape[2]
        vw = 16 // va.element_size()
        if K % vw or not int(e["plan"].info.dia_canon5) in (1, 9):
            return None
        tmax = 5 if int(e["plan"].info.dia_canon5) == 1 else 3
        bufs = e.setdefault("buffers_k", {})
        buf = bufs.get(K)
        if buf is None:                                      # persistent extended operands: A_ii, b, two x buffers, one scratch
            buf = bufs[K] = (torch.ones(n_ext, dtype=dt, device=dev), torch.zeros((n_ext, K), dtype=dt, device=dev),
                             torch.zeros((n_ext, K), dtype=dt, device=dev), torch.zeros((n_ext, K), dtype=dt, device=dev),
                             torch.zeros((n_ext, K), dtype=dt, device=dev))
        d_ext, b_ext, cur, nxt, tmp = buf
        own = lambda v: v[off:off + nl]                      # noqa: E731
        own(d_ext).copy_(va[:, 0, 0])
        va_full = self._ext_of(va, (3, K))                   # vertex_attr allocated by ext_tensor(3, K): read in place
        if va_full is None:
            own(b_ext).copy_(va[:, 1, :])
            own(cur).copy_(va[:, 2, :])
            b_op, ldb = b_ext, K
        else:
            b_op, ldb = va_full[:, 1, :], 3 * K              # (the library takes leading dimensions)
        kp = -(-n_iters // tmax)
        base, rem = divmod(n_iters, kp)
        code, st, h = dtype_code(dt), stream_ptr(dev), e["plan"]._h
        first = True
        scratch = [cur, nxt]
        for i in range(kp):
            ts = base + (1 if i < rem else 0)
            x_op, ldx = cur, K
            if first and va_full is not None:                # ghost lines of whole AoS rows (A_ii, b, x: one vector 3 K wide)
                vecs = [d_ext, va_full.reshape(n_ext, 3 * K)]
                x_op, ldx = va_full[:, 2, :], 3 * K
            else:
                vecs = [d_ext, b_ext, cur] if first else [cur]
            first = False
            self._fill_ghosts(vecs, self._ghost_exchange([own(v) for v in vecs]))
            if i == kp - 1:                                  # the last pass writes into a buffer the caller keeps (no clone)
                nxt = torch.empty((n_ext, K), dtype=dt, device=dev)
            rc = lib().gla_jacobi_sweeps(h, code, K, ptr(d_ext), ptr(b_op), ldb, ptr(x_op), ldx, ptr(nxt), ptr(tmp), K, w, ts, st)
            if rc == -5 and i == 0:
                self._ghost_pending = None
                return None
            check(rc)
            cur, nxt = nxt, cur
        bufs[K] = (d_ext, b_ext, scratch[0], scratch[1], tmp)
        self._ghost_pending = None
        return own(cur).unsqueeze(1)

    @classmethod
    def rehearsal_self_halo(cls, Nx, lines, device, comm=None, dtype=None):
        """A driver for ONE rank's block of `lines` grid lines x Nx points whose upper and lower neighbour are the rank itself (the
        grid is periodic in the line direction), over a real one-rank RCCL communicator of the library: everything a rank of an
        N-rank job does per forward() -- pack, ncclAllGather on the library's stream, interior line blocks, ghost fill + edge
        blocks on a second stream, join -- except waiting for other GPUs.  No multi-GPU hardware is needed: bench.py's
        `dist_route_one_gpu*` records, tools/bench_dist_config4_block.py and tests/test_gpu_dist.py use it.  Not a product
        path: a real job builds RowBlockJacobi(edgeij_pair, edge_attr, ...) per rank."""
        import types
        from . import UtilsGNN
        from .plan import CsrPlan
        T = cls.TS_MAX
        ei, ev = UtilsGNN.laplacianfun_torch(Nx, Ny=lines + 2 * T, device=device)
        if dtype is not None:
            ev = ev.to(dtype)
        plan = CsrPlan(ei, Nx * (lines + 2 * T))
        plan.set_values(ev)
        drv = cls.__new__(cls)
        drv.comm, drv.overlap, drv._ghost_pending, drv._lib_send = comm if comm is not None else LibComm(), True, None, {}
        drv.part = types.SimpleNamespace(world=1, rank=0, n_local=Nx * lines, group=None, n_in=Nx * lines, S=0)
        drv._sched = plan.jacobi_schedule
        drv.ext = dict(plan=plan, S=Nx, n_ext=Nx * (lines + 2 * T), off=Nx * T, up=True, down=True, slab=T * Nx,
                       dtype=ev.dtype, dev=torch.device(device), up_src=0, down_src=0)
        drv.mode = "temporally blocked, one-rank self-halo rehearsal"
        return drv

    def capture_forward(self, n_iters, vertex_attr, g):
        """forward(n_iters, vertex_attr, g) captured into a HIP graph (torch.cuda.CUDAGraph): returns run(vertex_attr) -> x;
        run.graph.replay() re-runs it on run.static_va in place (result in run.out).  Every rank must capture and replay alike.

        Both routes capture: the exchange-per-sweep loop (~6 enqueues per sweep: 29 us of host time per sweep eager, 17 us
        replayed, tools/graph_dist_loop.py) and -- round 5 -- the TEMPORALLY BLOCKED route (gla_dist_jacobi_temporal: per pass
        a pack launch, ncclAllGather on the library's stream, the interior launch, ghost fill + edge launch on the second stream,
        event fork / join: ~10 enqueues and 89 us of host time per pass, next to ~100 us of GPU time per pass on BASELINE
        config 4's block of one of 8 ranks, 1024 x 8192).  The call's A_ii flag word is one of the plan's capture words with a
        memset node in front (gla_jacobi_dg_slot), so replays and eager calls never share a word.  `g` must be a CPU tensor
        or already read (no device-to-host copy inside a capture); operands keep the captured shape."""
        dev = vertex_attr.device
        static_va = vertex_attr.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                        # warm-up off the capture: lazy allocations, code loads
            for _ in range(2):
                self.forward(n_iters, static_va, g)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.forward(n_iters, static_va, g)

        def run(va):
            static_va.copy_(va)
            graph.replay()
            return out.clone()
        run.graph, run.static_va, run.out = graph, static_va, out
        return run

    def capture_forward_native(self, n_iters, vertex_attr, g):
        """forward(n_iters, vertex_attr, g) of the temporally blocked route as ONE HIP graph built by the library itself
        (gla_dist_jacobi_temporal_graph: hipStreamBeginCapture on streams the graph owns -- no torch graph machinery, which is
        what a C++ host would call).  Returns run(vertex_attr) -> x [n_local, 1]; run.replay() launches the graph on torch's
        current stream with the operands as they stand (run.static_va in, run.out out).  The graph has its OWN extended work
        vectors: eager forward() calls and replays do not share buffers.  Every rank must build and replay alike.  Raises
        GlaError where the library declines (the block is not a canonical 5-point block, a pass of one sweep, ...)."""
        import ctypes
        from ._lib import check, dtype_code, lib, ptr, stream_ptr
        e, p = self.ext, self.part
        if e is None or self.comm is None or vertex_attr.dim() != 2 or vertex_attr.dtype != e["dtype"]:
            raise ValueError("capture_forward_native needs the temporally blocked route on the library's communicator (K = 1)")
        va = vertex_attr.clone() if vertex_attr.stride(1) == 1 else vertex_attr.contiguous().clone()
        n_ext, off, nl = e["n_ext"], e["off"], p.n_local
        dev, dt = va.device, va.dtype
        T, S = self.TS_MAX, e["S"]
        m = T * S
        passes = list(self._sched(int(n_iters), dt))
        bufs = [torch.ones(n_ext, dtype=dt, device=dev)] + [torch.zeros(n_ext, dtype=dt, device=dev) for _ in range(4)]
        d_ext, b_ext, xa, xb, x_last = bufs
        send, recv = torch.empty(6 * m, dtype=dt, device=dev), torch.empty(p.world * 6 * m, dtype=dt, device=dev)
        arr = (ctypes.c_int * len(passes))(*passes)
        h = ctypes.c_void_p()
        torch.cuda.synchronize(dev)
        check(lib().gla_dist_jacobi_temporal_graph(e["plan"]._h, self.comm._h, dtype_code(dt), ptr(va), va.stride(0), off, nl, ptr(d_ext),
                                                   ptr(b_ext), ptr(xa), ptr(xb), ptr(x_last), ptr(send), ptr(recv), T, int(e["up"]),
                                                   int(e["down"]), int(e.get("up_src", p.rank - 1)), int(e.get("down_src", p.rank + 1)),
                                                   scalar0(g), arr, len(passes), ctypes.byref(h)))
        out = x_last[off:off + nl].reshape(-1, 1)
        n_ex = len(passes)

        class _Run:
            static_va, graph_handle = va, h
            _keep = (bufs, send, recv, self)

            def replay(self_):
                check(lib().gla_graph_launch(h, stream_ptr(dev)))
                EXCHANGES[0] += n_ex

            def __call__(self_, v):
                va.copy_(v)
                self_.replay()
                return out.clone()

            def close(self_):
                if self_.graph_handle is not None:
                    torch.cuda.synchronize(dev)
                    lib().gla_graph_destroy(self_.graph_handle)
                    self_.graph_handle = None

            def __del__(self_):
                try:
                    self_.close()
                except Exception:                            # noqa: BLE001 -- interpreter shutdown
                    pass
        run = _Run()
        run.out = out
        return run

    def forward(self, n_iters, vertex_attr, g):
        p = self.part
        va = vertex_attr
        K = 1 if va.dim() == 2 else va.shape[2]
        if self.ext is not None and K == 1 and va.dim() == 2 and va.dtype == self.ext["dtype"] and int(n_iters) > 0:
            if va.stride(1) != 1:
                va = va.contiguous()
            return self._forward_temporal(int(n_iters), va, scalar0(g))
        if self.ext is not None and K > 1 and va.dim() == 3 and va.dtype == self.ext["dtype"] and int(n_iters) >= 2:
            out = self._forward_temporal_k(int(n_iters), va, scalar0(g))
            if out is not None:
                return out
        diag = (va[:, 0] if va.dim() == 2 else va[:, 0, 0]).contiguous()
        B = (va[:, 1].reshape(-1, 1) if va.dim() == 2 else va[:, 1, :]).contiguous()
        w = scalar0(g)
        cur = p.new_buffer(K, va.dtype, va.device)
        nxt = p.new_buffer(K, va.dtype, va.device)
        cur[:p.n_local] = va[:, 2].reshape(-1, 1) if va.dim() == 2 else va[:, 2, :]
        if self.comm is not None:
            # the whole loop -- pack, all-gather on the side stream, interior launch, event wait, boundary launch, per
            # sweep -- is enqueued by ONE library call (csrc/gla_dist.hip); nothing waits on the host
            from ._lib import dtype_code, lib, ptr, stream_ptr, check
            if self.plan.nnz and self.plan.val_dtype != va.dtype:
                raise ValueError(f"plan holds {self.plan.val_dtype} values, operands are {va.dtype}")
            check(lib().gla_dist_jacobi_sweeps(self.plan._h, self.comm._h, dtype_code(va.dtype), K, ptr(diag), ptr(B), K,
                                               ptr(cur), ptr(nxt), p.n_in, ptr(self.send_idx32), p.n_send, p.S,
                                               ptr(self._send_buf(cur)), w, int(n_iters), int(self.overlap),
                                               stream_ptr(va.device)))
            x = (cur if int(n_iters) % 2 == 0 else nxt)[:p.n_local]
            return x.clone() if va.dim() == 2 else x.unsqueeze(1).clone()
        # shapes are validated once (first sweep through the checked wrapper); the remaining sweeps call the C ABI
        # directly so that the per-sweep host cost stays far below the ~0.3 ms a sweep takes on the GPU
        from ._lib import dtype_code, lib, ptr, stream_ptr, check
        code, h, fn = dtype_code(va.dtype), self.plan._h, lib().gla_jacobi_sweep
        st = stream_ptr(va.device)
        d_p, b_p = ptr(diag), ptr(B)

        def raw(src, dst, which):
            check(fn(h, code, K, d_p, b_p, K, ptr(src), K, ptr(dst), K, w, which, st))

        for it in range(int(n_iters)):
            out = nxt[:p.n_local]
            if it == 0:
                self.run(cur, lambda which: self.plan.jacobi_sweep(diag, B, cur, out, w, which))
            else:
                self.run(cur, lambda which: raw(cur, out, which))
            cur, nxt = nxt, cur
        x = cur[:p.n_local]
        return x.clone() if va.dim() == 2 else x.unsqueeze(1).clone()

class RowBlockCheby(RowBlockOps):
    """ChebyRelaxGNN(deg).forward on a row block (ChebyGNN.py:324-353); returns (x, r, p, g_out) local rows.

    Two routes, as for RowBlockJacobi.  (1) One halo exchange per step (north_star's wording).  (2) `temporal=True` (default) on
    2-D stencil blocks: the fused pass of the single-GPU path (gla_cheby_fused / gla_cheby_fused_next: up to four steps in one
    pass over the data) on the block extended by TS_MAX ghost lines, ONE ghost-line exchange per pass -- b and x before the
    first pass, (r, x, p) before every continuation pass.  Ghost lines are recomputed redundantly; per row the arithmetic of
    the step kernels: bit-identical to route (1) and to the single-GPU module."""

    def __init__(self, edgeij_pair, edge_attr, n_global, row_lo, row_hi, group=None, overlap=True, col_blocks=None,
                 host_loop=False, temporal=True):
        super().__init__(edgeij_pair, edge_attr, n_global, row_lo, row_hi, group, overlap, col_blocks, host_loop)
        self.ext = None
        if temporal and col_blocks is None and self.part.world > 1 and edgeij_pair.is_cuda:
            self._setup_ghost(edgeij_pair, edge_attr, need_jacobi_passes=False)
        if self.ext is not None:
            self.mode = ("temporally blocked: one ghost-line exchange per fused pass of up to 4 Chebyshev steps, "
                         + ("libgla_hip RCCL all-gather (stream-ordered)" if self.comm is not None
                            else "torch.distributed all-gather (host-ordered)"))

    def _forward_ghost(self, deg, b, x0, g):
        from .ChebyGNN import _scalars, _g_out
        e, p = self.ext, self.part
        n_ext, off, nl, K = e["n_ext"], e["off"], p.n_local, b.shape[1]
        dev, dt = b.device, b.dtype
        plan = e["plan"]
        c, d, alphas, betas = _scalars(g, deg)
        bufs = e.setdefault("buffers_cheby", {})
        buf = bufs.get(K)
        if buf is None:                                      # persistent extended operands: b, two (x, r, p) sets
            buf = bufs[K] = [torch.zeros((n_ext, K), dtype=dt, device=dev) for _ in range(7)]
        b_ext, xa, ra, pa, xb, rb, pb = buf
        own = lambda v: v[off:off + nl]                      # noqa: E731
        b_full, x_full = self._ext_of(b, (K,)), self._ext_of(x0, (K,))      # operands allocated by ext_tensor(K): read in place
        if b_full is None:
            own(b_ext).copy_(b)
        else:
            b_ext = b_full
        x_in = xa                                            # the first pass's x operand (xa itself stays a scratch destination)
        if x_full is None:
            own(xa).copy_(x0)
        else:
            x_in = x_full
        self._fill_ghosts([b_ext, x_in], self._ghost_exchange([own(b_ext), own(x_in)]))
        # The LAST pass writes into extended buffers allocated for this call and the caller gets their own-row views: no
        # clone of x, r, p (three K-wide copies were 12.6 GB at 4096^2 x 16 -- more than the fused pass itself moves).  Every
        # ghost row of an output is written by the kernel before anything reads it: torch.empty is enough.
        fresh = lambda: [torch.empty((n_ext, K), dtype=dt, device=dev) for _ in range(3)]   # noqa: E731
        if deg <= 4:
            xb, rb, pb = fresh()
        done = 0
        for m in range(min(deg, 4), 1, -1):                  # the most steps this plan's fused kernel takes
            if plan.cheby_fused(b_ext, x_in, xb, rb, pb, [float(a) for a in alphas[:m]], [0.0] + [float(v) for v in betas[1:m]]):
                done = m
                break
        if done == 0:
            self._ghost_pending = None
            return None
        mmax = done
        X, R, P = xb, rb, pb                                 # the iterates after step `done`
        X2, R2, P2 = xa, ra, pa                              # the other set
        rest = deg - done
        sizes = []
        if rest >= 2:
            n_pass = -(-rest // mmax)
            base, extra = divmod(rest, n_pass)
            sizes = [base + 1] * extra + [base] * (n_pass - extra) if base >= 2 else [2] * (rest // 2) + [1] * (rest % 2)
        elif rest == 1:
            sizes = [1]
        owned = deg <= 4 and done == deg                     # (X, R, P) are this call's own buffers
        for i_sz, sz in enumerate(sizes):
            vecs = [R, X, P]
            self._fill_ghosts(vecs, self._ghost_exchange([own(v) for v in vecs]))
            if sz >= 2 and i_sz == len(sizes) - 1:           # the last pass: fresh destination set
                X2, R2, P2 = fresh()
            if sz >= 2 and plan.cheby_fused_next(R, X, P, X2, R2, P2, float(alphas[done - 1]),
                                                  [float(a) for a in alphas[done:done + sz]],
                                                  [float(v) for v in betas[done:done + sz]]):
                X, R, P, X2, R2, P2 = X2, R2, P2, X, R, P
                done += sz
                owned = i_sz == len(sizes) - 1
                continue
            owned = False
            for it in range(done + 1, done + sz + 1):        # step kernels on the extended block (R, X in place, P ping-pongs)
                if it > done + 1:
                    vecs = [R, X, P]
                    self._fill_ghosts(vecs, self._ghost_exchange([own(v) for v in vecs]))
                plan.cheby_steps(P, P2, R, X, [alphas[it - 2]], [alphas[it - 1]], [betas[it - 1]])
                P, P2 = P2, P
            done += sz
        cur = (X, R, P)
        self._ghost_pending = None
        if owned:
            return own(cur[0]), own(cur[1]), own(cur[2]), _g_out(c, d, alphas, betas, g)
        return own(cur[0]).clone(), own(cur[1]).clone(), own(cur[2]).clone(), _g_out(c, d, alphas, betas, g)

    def forward(self, deg, b, x0, g):
        p = self.part
        K = b.shape[1]
        if self.ext is not None and deg >= 2 and b.dtype == self.ext["dtype"] and b.is_cuda:
            out = self._forward_ghost(int(deg), b, x0, g)
            if out is not None:
                return out
        gh = g.detach().to("cpu")
        c, d = gh[0], gh[1]
        alpha = 1 / d
        beta = None
        xin = p.new_buffer(K, b.dtype, b.device)
        xin[:p.n_local] = x0
        r, x = torch.empty_like(b), torch.empty_like(b)
        pc, pn = p.new_buffer(K, b.dtype, b.device), p.new_buffer(K, b.dtype, b.device)
        if deg <= 0:
            return x0, None, None, g
        self.run(xin, lambda which: self.plan.cheby_first(b, xin, r, pc[:p.n_local], x, float(alpha), which))
        for it in range(2, deg + 1):
            beta = 0.5 * (c * alpha) ** 2 if it == 2 else ((c * alpha) / 2) ** 2
            alpha_new = 1 / (d - beta / alpha)
            a_old, a_new, bt = float(alpha), float(alpha_new), float(beta)
            src, dst = pc, pn
            self.run(src, lambda which: self.plan.cheby_next(src, dst[:p.n_local], r, x, a_old, a_new, bt, which))
            pc, pn = pn, pc
            alpha = alpha_new
        g_out = torch.hstack([c, d, alpha] if beta is None else [c, d, alpha, beta]).to(g.device)
        return x, r, pc[:p.n_local], g_out

class RowBlockPower(RowBlockOps):
    """PowerMethodGNN(num_iter).forward on a row block (PowerMethodGNN.py:296-334): halo exchange per SpMV plus
    an all-reduce of the K (resp. 2K) partial sums per iteration."""

    def forward(self, num_iter, b0):
        from .plan import col_dot, col_scale_rsqrt
        p = self.part
        K = b0.shape[1]
        buf = p.new_buffer(K, b0.dtype, b0.device)
        buf[:p.n_local] = b0
        y = torch.empty_like(b0)
        nrm = torch.zeros(K, dtype=b0.dtype, device=b0.device)
        ss, work, nrm_in = torch.zeros(K, dtype=b0.dtype, device=b0.device), None, None
        cur, nxt = buf, p.new_buffer(K, b0.dtype, b0.device)
        for _ in range(int(num_iter)):
            # fused step: gathers read cur (local + halo) ./ nrm_in, y lands in nxt's local part; the second-stage
            # reduction of the tile partials runs with the boundary launch, the all-reduce makes it global
            def launch(which, cur=cur, nxt=nxt):
                nonlocal work
                _, _, work = self.plan.power_step(cur, nrm=nrm_in, out=nxt[:p.n_local], which=which, work=work, sumsq=ss)
            self.run(cur, launch)
            self.all_reduce_sum(ss)
            nrm_in = torch.sqrt(ss)
            nrm = nrm_in.clone()
            cur, nxt = nxt, cur
        if int(num_iter) > 0:
            buf = p.new_buffer(K, b0.dtype, b0.device)
            buf[:p.n_local] = col_scale_rsqrt(cur[:p.n_local], ss)
        self.run(buf, lambda which: self.plan.spmm(buf, out=y, which=which))
        bl = buf[:p.n_local]
        sums = self.all_reduce_sum(torch.stack([col_dot(bl, y), col_dot(bl, bl)]))
        return bl.clone(), torch.stack([nrm, sums[0], sums[0] / sums[1]])
