"""GPU: the row-block multi-rank path end to end with the HIP kernels -- 2 ranks sharing cuda:0 (gloo carries
the halo through host memory here; on a multi-GPU node the same code runs over RCCL) must reproduce the
single-rank modules: bit-exact for Jacobi and Chebyshev (row sums are rank-independent), 1e-12 for the power
method (its global sums are re-associated by the all-reduce)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi, RowBlockCheby, RowBlockPower
        dev = torch.device("cuda:0")
        Nx, Ny, K = 96, 130, 2
        n = Nx * Ny
        lines = [0, 70, Ny] if world == 2 else [0, 30, 31, 90, Ny]   # uneven blocks (world 4: one single-line block)
        lo, hi = lines[rank] * Nx, lines[rank + 1] * Nx
        gen = torch.Generator().manual_seed(3)
        B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
        X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
        # single-rank references (whole grid) through the module surface
        ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, device=dev)
        diag = torch.full((n, K), -4.0, dtype=torch.float64, device=dev)
        g = torch.tensor([0.7], dtype=torch.float64)
        ea = torch.cat([ev, torch.zeros_like(ev)], 1)
        x_ref = gla.JacobiGNN.JacobiGNN()(6, torch.stack([diag, B, X], 1), ei, ea, g)[:, 0, :]
        gch = torch.tensor([-3.4, -4.0], dtype=torch.float64)
        v_ref, _, g_ref = gla.ChebyGNN.ChebyRelaxGNN(4, return_edge_attr=False)(torch.stack([B, X], 1), ei, ev, gch)
        _, _, p_ref = gla.PowerMethodGNN.PowerMethodGNN(5, return_edge_attr=False)(
            torch.stack([B, torch.zeros_like(B)], 1), ei, ea, torch.zeros(3, K, dtype=torch.float64), None)
        # this rank's block
        ei_l, ev_l = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, row_range=(lo, hi), device=dev)
        jac = RowBlockJacobi(ei_l, ev_l, n, lo, hi, overlap=overlap)
        x_loc = jac.forward(6, torch.stack([diag[lo:hi], B[lo:hi], X[lo:hi]], 1), g)[:, 0, :]
        assert torch.equal(x_loc, x_ref[lo:hi]), "jacobi"
        # K = 1 (the bench configuration): interior tiles may run on the diagonal (stencil) kernel, boundary tiles on
        # the CSR kernel -- together still the single-rank result, bit for bit
        x1_ref = gla.JacobiGNN.JacobiGNN()(6, torch.stack([diag[:, 0], B[:, 0], X[:, 0]], 1), ei, ea, g)
        x1_loc = jac.forward(6, torch.stack([diag[lo:hi, 0], B[lo:hi, 0], X[lo:hi, 0]], 1), g)
        assert torch.equal(x1_loc, x1_ref[lo:hi]), "jacobi K=1"
        # route: blocks of >= 5 grid lines take the temporally blocked route (one ghost-line exchange per 5 / 4 / 2 sweeps);
        # the world-4 partition has a one-line block, which sends every rank to the exchange-per-sweep route
        default_layout = not any(k in os.environ.get("GLA_OPTIONS", "") for k in ("dia=0", "compress=1", "jacobi_fuse="))
        if default_layout:                                     # (which route a block takes is a statement about the default layout)
            assert (jac.ext is not None) == (world == 2), (world, jac.ext is None)
            if jac.ext is not None and "sym=0" not in os.environ.get("GLA_OPTIONS", ""):
                # the extended block (own lines + ghost lines with their true rows) is a symmetric stencil matrix of its own:
                # its passes read 3 of the 5 diagonals like the single-GPU plan
                assert jac.ext["plan"].info.dia_sym == 3, jac.ext["plan"].info.dia_sym
        va1, va1_loc = torch.stack([diag[:, 0], B[:, 0], X[:, 0]], 1), torch.stack([diag[lo:hi, 0], B[lo:hi, 0], X[lo:hi, 0]], 1)
        for its in (1, 2, 4, 5, 7, 11):                        # 1, 2, 4, 5, 5 + 2 and 5 + 4 + 2 sweeps
            assert torch.equal(jac.forward(its, va1_loc, g), gla.JacobiGNN.JacobiGNN()(its, va1, ei, ea, g)[lo:hi]), its
        jac_ps = RowBlockJacobi(ei_l, ev_l, n, lo, hi, overlap=overlap, temporal=False)      # exchange per sweep
        assert jac_ps.ext is None and torch.equal(jac_ps.forward(6, va1_loc, g), x1_ref[lo:hi]), "jacobi per-sweep route"
        # the same routes on the 9-point Q1 FEM operator (random positive diagonal, 63 lines of 63 points)
        from gnn_applied_linear_algebra_amd.dist_amg import _row_slice
        m9 = 63
        n9 = m9 * m9
        ei9, ev9 = gla.UtilsGNN.heateqnfem2dfun([m9 + 1, m9 + 1], [1.0, 2.0], device=dev)
        cuts9 = [0, 30, m9] if world == 2 else [0, 9, 20, 41, m9]
        lo9, hi9 = cuts9[rank] * m9, cuts9[rank + 1] * m9
        va9 = torch.cat([torch.rand(n9, 1, generator=gen, dtype=torch.float64) + 2.0,
                         torch.rand(n9, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
        ea9 = torch.cat([ev9, torch.zeros_like(ev9)], 1)
        e9_l, v9_l = _row_slice(ei9, ev9, lo9, hi9)
        jac9 = RowBlockJacobi(e9_l, v9_l.reshape(-1, 1), n9, lo9, hi9, overlap=overlap)
        if default_layout:
            assert jac9.ext is not None, "FEM blocks of >= 8 lines take the temporally blocked route"
            if "sym=0" not in os.environ.get("GLA_OPTIONS", ""):
                assert jac9.ext["plan"].info.dia_sym == 15, jac9.ext["plan"].info.dia_sym
        for its in (3, 5, 9):
            assert torch.equal(jac9.forward(its, va9[lo9:hi9].contiguous(), g),
                               gla.JacobiGNN.JacobiGNN()(its, va9, ei9, ea9, g)[lo9:hi9]), ("fem", its)
        info = jac.plan.info                                   # the block's own rows are stencil rows, its halo rows not
        assert not default_layout or info.dia == 5 or hi - lo <= 2 * Nx, info.dia
        if info.dia:
            assert info.dia_row_lo == (Nx if rank > 0 else 0), (rank, info.dia_row_lo)
            assert info.dia_row_hi == (hi - lo) - (Nx if rank < world - 1 else 0), (rank, info.dia_row_hi)
        if overlap:                                            # a one-line block has boundary tiles only
            assert jac.plan.n_boundary > 0 and (jac.plan.n_interior > 0 or lines[rank + 1] - lines[rank] <= 2)
        ch = RowBlockCheby(ei_l, ev_l, n, lo, hi, overlap=overlap)
        x, r, p, g_out = ch.forward(4, B[lo:hi].contiguous(), X[lo:hi].contiguous(), gch)
        assert torch.equal(x, v_ref[lo:hi, 1]) and torch.equal(r, v_ref[lo:hi, 2]) and torch.equal(p, v_ref[lo:hi, 3])
        assert torch.equal(g_out, g_ref)
        # round 3: the ghost-line routes of the fused Chebyshev pass and of the K-wide Jacobi passes (one exchange per pass):
        # every degree / sweep count, against the single-rank modules, bit for bit; temporal=False keeps one exchange per step
        if default_layout:
            assert (ch.ext is not None) == (world == 2), (world, ch.ext is None)
        for deg in (2, 3, 5, 6, 9):
            vr, _, gr = gla.ChebyGNN.ChebyRelaxGNN(deg, return_edge_attr=False)(torch.stack([B, X], 1), ei, ev, gch)
            x, r, p, g_out = ch.forward(deg, B[lo:hi].contiguous(), X[lo:hi].contiguous(), gch)
            assert torch.equal(x, vr[lo:hi, 1]) and torch.equal(r, vr[lo:hi, 2]) and torch.equal(p, vr[lo:hi, 3]), ("cheby ghost", deg)
            assert torch.equal(g_out, gr), ("cheby g", deg)
        ch_ps = RowBlockCheby(ei_l, ev_l, n, lo, hi, overlap=overlap, temporal=False)
        x, r, p, g_out = ch_ps.forward(4, B[lo:hi].contiguous(), X[lo:hi].contiguous(), gch)
        assert ch_ps.ext is None and torch.equal(x, v_ref[lo:hi, 1]) and torch.equal(p, v_ref[lo:hi, 3]), "cheby per-step route"
        vaK, vaK_loc = torch.stack([diag, B, X], 1), torch.stack([diag[lo:hi], B[lo:hi], X[lo:hi]], 1)
        for its in (2, 3, 4, 7, 9):                           # passes of 2, 3, 4, 4 + 3, 3 + 3 + 3 sweeps
            assert torch.equal(jac.forward(its, vaK_loc, g), gla.JacobiGNN.JacobiGNN()(its, vaK, ei, ea, g)[lo:hi]), ("jacobi K", its)
        assert torch.equal(jac_ps.forward(5, vaK_loc, g), gla.JacobiGNN.JacobiGNN()(5, vaK, ei, ea, g)[lo:hi]), "jacobi K per-sweep"
        pw = RowBlockPower(ei_l, ev_l, n, lo, hi, overlap=overlap)
        _, g_pw = pw.forward(5, B[lo:hi].contiguous())
        torch.testing.assert_close(g_pw.cpu(), p_ref.cpu(), rtol=1e-12, atol=0)
        torch.cuda.synchronize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,world", [(False, 2), (True, 2), (True, 4)])
def test_two_ranks_one_gpu(overlap, world):
    """2 (and 4) ranks sharing the one GPU of the box drive the HIP kernels through the row-block drivers; results are
    bit-exact against the single-rank modules (SURVEY 8e equivalence test; gloo moves the halos through the host)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in sorted(res):
        assert status == "ok", f"rank {rank}: {status}"


def _oracle_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import c_oracle
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
        dev = torch.device("cuda:0")
        N, sweeps = 1024, 20
        n = N * N
        lines = N // world
        lo, hi = rank * lines * N, (rank + 1) * lines * N
        gen = torch.Generator().manual_seed(24601)
        b = torch.rand(n, 1, generator=gen, dtype=torch.float64)
        x = torch.rand(n, 1, generator=gen, dtype=torch.float64)
        diag = torch.full((n, 1), -4.0, dtype=torch.float64)
        ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
        A = c_oracle.Csr(ei.cpu().numpy(), ev.cpu().numpy(), n)
        want = torch.from_numpy(A.jacobi(diag.numpy(), b.numpy(), x.numpy(), 0.7, sweeps))
        del ei, ev
        ei_l, ev_l = gla.UtilsGNN.laplacianfun_torch(N, row_range=(lo, hi), device=dev)
        jac = RowBlockJacobi(ei_l, ev_l, n, lo, hi, overlap=True)
        default_layout = not any(k in os.environ.get("GLA_OPTIONS", "") for k in ("dia=0", "compress=1", "jacobi_fuse=", "jacobi_march="))
        if default_layout:
            assert jac.ext is not None and tuple(jac._sched(sweeps)) == (7, 7, 6), "the temporally blocked route, 7 + 7 + 6"
        va = torch.cat([diag, b, x], 1)[lo:hi].contiguous().to(dev)
        got = jac.forward(sweeps, va, torch.tensor([0.7], dtype=torch.float64))
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want[lo:hi]), f"rank {rank}: max |d| = {float((got.cpu() - want[lo:hi]).abs().max()):.3e}"
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_1024_20_sweeps_vs_c_oracle():
    """the multi-rank route against the ORACLE itself (the other tests of this file compare with the
    single-rank module): JacobiGNN's 20 sweeps on the 1024^2 Laplacian cut into two row blocks of 512 grid lines (2 gloo ranks
    sharing the GPU; on a multi-GPU node the same drivers run over RCCL), temporally blocked 7 + 7 + 6 with one ghost-line
    exchange per pass, every rank's rows bit-exact against the single-threaded C restatement (JacobiGNN.py:138-148)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in sorted(res):
        assert status == "ok", f"rank {rank}: {status}"


def _long_rows_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd.dist import RowBlockCheby, RowBlockJacobi
        from gnn_applied_linear_algebra_amd.dist_amg import _row_slice
        dev = torch.device("cuda:0")
        gen = torch.Generator().manual_seed(11)
        n = 5000
        # rows of 20 .. 400 nonzeros (many above GLA_WAVE_ROW = 64: summed by their wave), columns within +-1200 of the row,
        # a few 3-nonzero rows in between so that tiles mix both kinds
        lens = torch.randint(20, 401, (n,), generator=gen)
        lens[::7] = 3
        rows = torch.repeat_interleave(torch.arange(n), lens)
        cols = (rows + torch.randint(-1200, 1201, (rows.numel(),), generator=gen)).clamp_(0, n - 1)
        key = torch.unique(rows * n + cols)                    # sorted, duplicate-free
        rows, cols = key // n, key % n
        ev = (torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.3)
        ei = torch.stack([rows, cols]).to(dev)
        ev = ev.to(dev)
        diag = torch.rand(n, generator=gen, dtype=torch.float64).to(dev) + 150.0
        b = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
        x = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
        g = torch.tensor([0.7], dtype=torch.float64)
        va = torch.stack([diag, b, x], 1)
        ea = torch.cat([ev, torch.zeros_like(ev)], 1)
        want = gla.JacobiGNN.JacobiGNN()(4, va, ei, ea, g)
        gch = torch.tensor([-3.4, -150.0], dtype=torch.float64)
        vr, _, _ = gla.ChebyGNN.ChebyRelaxGNN(3, return_edge_attr=False)(torch.stack([b, x], 1), ei, ev, gch)
        cuts = [0, 1733, n] if world == 2 else [0, 901, 2777, n]      # blocks that start in the middle of single-rank tiles
        lo, hi = cuts[rank], cuts[rank + 1]
        e_l, v_l = _row_slice(ei, ev, lo, hi)
        for overlap in (False, True):
            jac = RowBlockJacobi(e_l, v_l.reshape(-1, 1), n, lo, hi, overlap=overlap)
            got = jac.forward(4, va[lo:hi].contiguous(), g)
            assert torch.equal(got, want[lo:hi]), ("jacobi, long rows", overlap, (got - want[lo:hi]).abs().max().item())
            ch = RowBlockCheby(e_l, v_l.reshape(-1, 1), n, lo, hi, overlap=overlap)
            xx, rr, pp, _ = ch.forward(3, b[lo:hi].reshape(-1, 1).contiguous(), x[lo:hi].reshape(-1, 1).contiguous(), gch)
            assert (torch.equal(xx[:, 0], vr[lo:hi, 1]) and torch.equal(rr[:, 0], vr[lo:hi, 2])
                    and torch.equal(pp[:, 0], vr[lo:hi, 3])), "cheby, long rows"
        torch.cuda.synchronize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_long_rows(world):
    """Rows of 64 nonzeros or more are summed by a wave (vector bin, option row_sum = 1): the association order must be a
    function of the row alone, not of the tile it shares -- a row-block partition, whose tiles start at the rank's first row,
    then reproduces the single-rank result bit for bit.  (Round 3 picked the lane count per tile: ADVICE r3.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_long_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in sorted(res):
        assert status == "ok", f"rank {rank}: {status}"


def test_world1_rowblock_equals_module():
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
    dev = torch.device("cuda:0")
    N = 50
    n = N * N
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
    gen = torch.Generator().manual_seed(9)
    va = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64), torch.rand(n, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
    g = torch.tensor([0.7], dtype=torch.float64)
    want = gla.JacobiGNN.JacobiGNN()(9, va, ei, torch.cat([ev, torch.zeros_like(ev)], 1), g)
    got = RowBlockJacobi(ei, ev, n, 0, n).forward(9, va, g)
    assert torch.equal(got, want)


def _nccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        n_local, S, K = 1000, 64, 2
        X = torch.zeros((n_local + S, K), dtype=torch.float64, device=dev)
        X[:n_local] = torch.arange(n_local * K, dtype=torch.float64, device=dev).reshape(n_local, K)
        idx = torch.arange(0, 2 * S, 2, device=dev)
        send = X[:n_local].index_select(0, idx)
        work = dist.all_gather_into_tensor(X[n_local:], send, async_op=True)     # same call shape as dist.py
        Y = X[:n_local] * 2                                                     # "interior" work meanwhile
        work.wait()
        torch.cuda.synchronize()
        assert torch.equal(X[n_local:], send) and torch.equal(Y, X[:n_local] * 2)
        t = torch.ones(3, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        dist.barrier()
        q.put("ok")
    except Exception:  # pragma: no cover
        import traceback
        q.put("fail: " + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_call_pattern_world1():
    """RCCL accepts the exact collective call shape dist.py uses (output = tail view of the X buffer, async)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    status = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", status


def _amg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd.dist_amg import RowBlockAMG
        dev = torch.device("cuda:0")
        m = 63
        ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
        n = m * m
        A = gla.VCycle._as_sparse(ei, ev, (n, n))
        V = gla.VCycle
        from gnn_applied_linear_algebra_amd import dist as gdist
        # (agglomerate_below: rows per rank under which a level is replicated.  0: only the coarsest level is -- every other
        #  level distributed, the ghost-line route wherever a rank holds >= 8 grid lines; 600: level 1 (1024 rows) and below
        #  run replicated on every rank)
        for kwargs, agg in ((dict(coarse_solver="dense"), 0), (dict(coarse_deg=8), 0), (dict(coarse_solver="dense"), 600),
                            (dict(coarse_deg=8, levels=4), 0)):
            kwargs = dict(kwargs)
            levels = kwargs.pop("levels", 3)
            H = V.AMGHierarchy(A, levels=levels, splitting=V.GridSplitting(m, m), **kwargs)
            D = RowBlockAMG(H, overlap=True, agglomerate_below=agg)
            assert D.n_dist == (1 if agg else levels - 1), (D.n_dist, agg)
            assert D.A_ops[0].ext is not None, "level 0: >= 8 grid lines per rank, the ghost-line route"
            lo0, hi0 = D.blocks[0][rank]
            assert lo0 % m == 0 and hi0 % m == 0, "row blocks of a stencil level are whole grid lines"
            gen = torch.Generator().manual_seed(5)
            b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
            x_ref = torch.zeros_like(b)
            x_loc = D.local(x_ref)
            b_loc = D.local(b)
            for _ in range(3):
                x_ref = H.cycle(b, x_ref)
                e0 = gdist.EXCHANGES[0]
                x_loc = D.cycle(b_loc, x_loc)
                n_ex = gdist.EXCHANGES[0] - e0
                assert torch.equal(x_loc, D.local(x_ref)), "distributed cycle differs from the single-rank cycle"
            # exchanges per cycle: 4 per ghost-line level (ghost lines of [b, x]; halo of r; halo of xc; ghost lines of x),
            # n_pre + n_post + 3 on a level without a ghost-line plan, + 1 gather at the hand-over to the replicated levels
            ghost_lv = sum(1 for o in D.A_ops if o.ext is not None)
            assert n_ex == 4 * ghost_lv + 9 * (D.n_dist - ghost_lv) + 1, (n_ex, ghost_lv, D.n_dist)
            r_loc = D.residual(b_loc, x_loc)
            assert torch.equal(r_loc, D.local(H.residual(b, x_ref)))
            rn = torch.sum(r_loc * r_loc).cpu()
            dist.all_reduce(rn)
            assert float(rn.sqrt()) < 0.5 * float(torch.norm(b))
        # a hierarchy replicated AS A WHOLE (level 0 already below the agglomeration threshold: n_dist = 0): the caller's
        # nonzero initial guess must reach the cycle exactly as AMGHierarchy.cycle(b, x) uses it
        H = V.AMGHierarchy(A, levels=3, splitting=V.GridSplitting(m, m), coarse_solver="dense")
        D = RowBlockAMG(H, overlap=True, agglomerate_below=1 << 20)
        assert D.n_dist == 0, D.n_dist
        gen = torch.Generator().manual_seed(6)
        b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
        x0 = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
        assert torch.equal(D.cycle(D.local(b), D.local(x0)), D.local(H.cycle(b, x0))), "n_dist = 0: initial guess dropped"
        assert torch.equal(D.residual(D.local(b), D.local(x0)), D.local(H.residual(b, x0))), "n_dist = 0: residual"
        torch.cuda.synchronize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_vcycle_two_ranks(world):
    """BASELINE config 5 shape with 2 (and 3: blocks that do not divide the levels evenly) ranks sharing the GPU:
    row-block distributed V-cycle == single-rank V-cycle, bit for bit (rectangular partitions for P / P^T, replicated
    coarsest solve)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_amg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in sorted(res):
        assert status == "ok", f"rank {rank}: {status}"


def _amg_one_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    keep = os.dup(1)
    os.dup2(2, 1)                                            # (RCCL's banner goes to stderr)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd import dist as gdist
        from gnn_applied_linear_algebra_amd.dist_amg import RowBlockAMG
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        gla._lib.check(gla._lib.lib().gla_set_option(b"jacobi_min_rows", 0))      # the ghost-line passes also on this small mesh
        V = gla.VCycle
        m = 127
        n = m * m
        ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
        A = V._as_sparse(ei, ev, (n, n))
        H = V.AMGHierarchy(A, levels=4, splitting=V.GridSplitting(m, m), coarse_solver="dense")
        RowBlockAMG.REHEARSE_ONE_RANK = True
        D = RowBlockAMG(H, overlap=True, agglomerate_below=2000)
        assert D.n_dist == 2 and all(o.ext is not None and o.comm is not None for o in D.A_ops), (D.n_dist, [o.ext is None for o in D.A_ops])
        gen = torch.Generator().manual_seed(2)
        b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
        xs, xd = torch.zeros_like(b), torch.zeros_like(b)
        for _ in range(3):
            xs = H.cycle(b, xs)
            e0 = gdist.EXCHANGES[0]
            xd = D.cycle(b, xd)
            assert torch.equal(xs, xd), "one-rank distributed cycle differs from AMGHierarchy.cycle"
        assert gdist.EXCHANGES[0] - e0 == 2 * D.n_dist          # (one rank: the halos of r and xc are empty, the hand-over gather is local)
        b2 = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)      # a different right-hand side: the copy cache must notice
        assert torch.equal(D.cycle(b2, xd), H.cycle(b2, xs))
        b2.mul_(0.5)                                                             # ... and an in-place change of the same tensor
        assert torch.equal(D.cycle(b2, xd), H.cycle(b2, xs))
        # ... and a change torch's version counter cannot see: the library's own out= entry points write through raw
        # pointers -- a solver loop that refreshes its residual tensor that way must not smooth against the previous one
        v0 = b2._version
        H.plans[0].residual(b, xs, out=b2)
        assert b2._version == v0, "(the premise of this check: a raw-pointer write does not bump the version counter)"
        assert torch.equal(D.cycle(b2, xd), H.cycle(b2, xs)), "stale right-hand side after an out= write"
        # the opt-in: the caller vouches for an unchanged b, the copy is skipped, same bits
        assert torch.equal(D.cycle(b2, xd, b_unchanged=True), H.cycle(b2, xs))
        run = D.capture_cycle()                                                 # the rank's cycle as one HIP graph, RCCL all-gathers included
        for rhs in (b, b2, b):
            xg = run(rhs, xd)
            assert torch.equal(xg, D.cycle(rhs, xd)) and torch.equal(xg, H.cycle(rhs, xs)), "captured cycle"
        torch.cuda.synchronize()
        q.put((0, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((0, "fail: " + traceback.format_exc()))
    finally:
        from gnn_applied_linear_algebra_amd.dist import LibComm
        LibComm.release_all()
        dist.destroy_process_group()
        os.dup2(keep, 1)


def test_distributed_vcycle_one_rank_rccl_and_graph_capture():
    """A process group of ONE rank over real RCCL: every distributed level on the ghost-line route with its all-gathers on
    libgla_hip's communicator (dist_amg.RowBlockAMG.REHEARSE_ONE_RANK), bit-identical to AMGHierarchy.cycle; the right-hand
    side copy cache follows new tensors and in-place updates; the whole cycle replays from a HIP graph bit-identically."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_amg_one_rank_worker, args=(_free_port(), q))
    pr.start()
    res = q.get(timeout=300)
    pr.join(timeout=60)
    assert res[1] == "ok", res[1]


def _self_halo_driver(gla, Nx, Ny, K, overlap, dev):
    """A one-rank 'row block' whose vertical couplings of the first / last two grid lines are redirected to HALO slots
    that the all-gather fills with copies of the very same rows: numerically the plain Laplacian, but every sweep goes
    through pack -> ncclAllGather on the library's side stream -> event wait -> boundary tiles (csrc/gla_dist.hip)."""
    import types
    from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockJacobi
    from gnn_applied_linear_algebra_amd.plan import CsrPlan
    n = Nx * Ny
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, device=dev)
    send_idx = torch.cat([torch.arange(0, 2 * Nx), torch.arange(n - 2 * Nx, n)]).to(dev)      # sorted
    S = int(send_idx.numel())
    row, col = ei[0], ei[1]
    vertical = (row - col).abs() == Nx
    in_send = torch.isin(col, send_idx)
    redirect = vertical & in_send
    col2 = col.clone()
    col2[redirect] = n + torch.searchsorted(send_idx, col[redirect])
    plan = CsrPlan(torch.stack([row, col2]), n, n + S)
    plan.set_values(ev)
    ni, nb = plan.classify_tiles(n)
    assert nb > 0
    drv = RowBlockJacobi.__new__(RowBlockJacobi)
    drv.plan, drv.overlap, drv._lib_send = plan, overlap, {}
    drv.comm = LibComm()                                   # single-process communicator: ncclCommInitRank(nranks = 1)
    drv.send_idx32 = send_idx.to(torch.int32)
    drv.part = types.SimpleNamespace(
        world=1, n_local=n, n_in=n, n_send=S, S=S,
        new_buffer=lambda K_, dt, dv: torch.zeros((n + S, K_), dtype=dt, device=dv))
    drv.ALL, drv.INTERIOR, drv.BOUNDARY = 0, 1, 2
    return drv, ei, ev


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("K", [1, 2])
def test_library_loop_self_halo_rccl(overlap, K):
    """gla_dist_jacobi_sweeps / gla_halo_begin / gla_halo_end with a REAL RCCL communicator (one rank): results are
    bit-identical to the single-GPU module, for odd and even sweep counts."""
    import gnn_applied_linear_algebra_amd as gla
    dev = torch.device("cuda:0")
    Nx, Ny = 128, 300
    n = Nx * Ny
    drv, ei, ev = _self_halo_driver(gla, Nx, Ny, K, overlap, dev)
    gen = torch.Generator().manual_seed(17)
    B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    diag = torch.full((n, K), -4.0, dtype=torch.float64, device=dev)
    g = torch.tensor([0.7], dtype=torch.float64)
    ea = torch.cat([ev, torch.zeros_like(ev)], 1)
    va = torch.stack([diag, B, X], 1) if K > 1 else torch.cat([diag, B, X], 1)
    jac = gla.JacobiGNN.JacobiGNN()
    for its in (1, 6, 7):
        want = jac(its, va, ei, ea, g)
        got = drv.forward(its, va, g)
        assert torch.equal(got, want), (its, (got - want).abs().max().item())
    # the two-call form used by the Chebyshev / power drivers: exchange + SpMM through RowBlockOps.run
    Xb = drv.part.new_buffer(K, torch.float64, dev)
    Xb[:n] = X
    Y = torch.empty(n, K, dtype=torch.float64, device=dev)
    drv.run(Xb, lambda which: drv.plan.spmm(Xb, out=Y, which=which))
    plan0 = gla.plan_for(ei, n, edge_attr=ev)
    assert torch.equal(Y, plan0.spmm(X))
    t = torch.arange(4, dtype=torch.float64, device=dev)
    assert torch.equal(drv.comm.all_reduce_sum(t.clone()), t)
    torch.cuda.synchronize()


def test_library_loop_host_cost_per_sweep():
    """Host time per sweep of the library-side loop (CPU clock around the enqueue of many sweeps on a tiny grid, so the
    GPU never throttles the host): the budget behind BASELINE config 4 at 8 GPUs is ~0.16 ms per sweep."""
    import time
    import gnn_applied_linear_algebra_amd as gla
    dev = torch.device("cuda:0")
    Nx, Ny = 64, 64
    n = Nx * Ny
    drv, ei, ev = _self_halo_driver(gla, Nx, Ny, 1, True, dev)
    va = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64), torch.rand(n, 2, dtype=torch.float64)], 1).to(dev)
    g = torch.tensor([0.7], dtype=torch.float64)
    drv.forward(10, va, g)
    torch.cuda.synchronize()
    sweeps = 400
    t0 = time.perf_counter()
    drv.forward(sweeps, va, g)
    t_host = (time.perf_counter() - t0) / sweeps
    torch.cuda.synchronize()
    print(f"library loop host cost: {t_host * 1e6:.1f} us per sweep (pack + all-gather + 2 launches + events)")
    assert t_host < 100e-6, t_host


def _amg_dist_setup_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_applied_linear_algebra_amd as gla
        from gnn_applied_linear_algebra_amd.dist_amg import RowBlockAMG, line_blocks, _row_slice
        dev = torch.device("cuda:0")
        V = gla.VCycle
        m = 63
        n = m * m
        ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 2.0], device=dev)
        lo, hi = line_blocks(n, world, m)[rank]                     # whole grid lines
        e_l, v_l = _row_slice(ei, ev, lo, hi)                       # all this rank is given: its rows of A
        for kwargs, agg in ((dict(coarse_solver="dense"), 0), (dict(coarse_deg=8), 0), (dict(coarse_solver="dense"), 600)):
            H = V.AMGHierarchy(V._as_sparse(ei, ev, (n, n)), levels=3, splitting=V.GridSplitting(m, m), **kwargs)
            D = RowBlockAMG.distributed_setup(e_l, v_l, n, levels=3, splitting=V.GridSplitting(m, m), agglomerate_below=agg,
                                              **kwargs)
            assert D.n_levels == len(H.A) == 3 and D.n_dist == (1 if agg else 2)
            assert D.A_ops[0].ext is not None
            for lv in range(D.n_dist):                              # every distributed level operator: this rank's rows, bit for bit
                h_ei, h_ea, _, h_dg = H.gnn[lv]
                blo, bhi = D.blocks[lv][rank]
                w_ei, w_ev = _row_slice(h_ei, h_ea, blo, bhi)
                part = D.A_ops[lv].part
                assert part.row_lo == blo and part.row_hi == bhi
                rowptr, col, _ = D.A_ops[lv].plan.arrays()
                assert int(rowptr[-1]) == w_ei.shape[1], (lv, int(rowptr[-1]), w_ei.shape[1])
                assert torch.equal(D.diag[lv], h_dg.reshape(-1)[blo:bhi])
            assert D.setup_nnz_local[0] == e_l.shape[1] < ei.shape[1]
            gen = torch.Generator().manual_seed(5)
            b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
            x_ref = torch.zeros_like(b)
            x_loc, b_loc = D.local(x_ref), D.local(b)
            for _ in range(3):
                x_ref = H.cycle(b, x_ref)
                x_loc = D.cycle(b_loc, x_loc)
                assert torch.equal(x_loc, D.local(x_ref)), "distributed-setup cycle differs from the single-rank cycle"
            assert torch.equal(D.residual(b_loc, x_loc), D.local(H.residual(b, x_ref)))
        torch.cuda.synchronize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_vcycle_setup(world):
    """BASELINE config 5's SETUP distributed by row blocks (nothing of nnz size replicated): each rank builds its rows
    of S, W, P, A P, P^T and A_c from its rows of A plus fetched halo rows; the resulting hierarchy and its cycles are
    bit-identical to the single-rank AMGHierarchy (2 and 3 ranks sharing the GPU, gloo carries the setup objects)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_amg_dist_setup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in sorted(res):
        assert status == "ok", f"rank {rank}: {status}"


def test_jacobi_fused_parts_and_library_all_gather():
    """gla_jacobi_fused: interior line blocks (part 1) + first / last line block (part 2) == all blocks (part 0) == 4
    single sweeps; and gla_comm_all_gather_begin / gla_halo_end on a real (one-rank) RCCL communicator -- the two
    pieces the temporally blocked multi-GPU route is made of (dist.RowBlockJacobi._forward_temporal)."""
    import ctypes
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd._lib import check, lib, ptr, stream_ptr, dtype_code
    from gnn_applied_linear_algebra_amd.dist import LibComm
    dev = torch.device("cuda:0")
    Nx, Ny = 300, 700
    n = Nx * Ny
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, device=dev)
    plan = gla.CsrPlan(ei, n)
    plan.set_values(ev)
    gen = torch.Generator().manual_seed(2)
    b = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
    x = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
    diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
    st, code = stream_ptr(dev), dtype_code(torch.float64)
    if plan.jacobi_schedule(2) != (2,):
        pytest.skip("no temporally blocked kernel under these GLA_OPTIONS")
    sizes = (8, 7, 6, 5, 4, 3, 2) if plan.jacobi_schedule(3) == (3,) else (5, 4, 2)     # march kernel / LDS-ring kernel
    for ts in sizes:
        try:
            check(lib().gla_set_option(b"jacobi_fuse", 0))
            want = plan.jacobi_sweeps(diag, b.reshape(-1, 1), x.reshape(-1, 1), 0.7, ts).reshape(-1)
        finally:
            check(lib().gla_set_option(b"jacobi_fuse", 1))
        for lines in (0, 40, 5):                               # 5 < 2 * ts: every block counts as boundary
            try:
                check(lib().gla_set_option(b"jacobi_lines", lines))
                whole, parts = torch.zeros_like(x), torch.zeros_like(x)
                check(lib().gla_jacobi_fused(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(whole), 0.7, ts, 0, st))
                check(lib().gla_jacobi_fused(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(parts), 0.7, ts, 1, st))
                mid = parts.clone()
                check(lib().gla_jacobi_fused(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(parts), 0.7, ts, 2, st))
            finally:
                check(lib().gla_set_option(b"jacobi_lines", 0))
            assert torch.equal(whole, want) and torch.equal(parts, want), (ts, lines)
            # the explicit form: 8 ghost lines at either end whatever the pass carries (what dist.RowBlockJacobi uses)
            parts8 = torch.zeros_like(x)
            check(lib().gla_jacobi_fused_parts(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(parts8), 0.7, ts, 1, 8, st))
            mid8 = parts8.clone()
            check(lib().gla_jacobi_fused_parts(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(parts8), 0.7, ts, 2, 8, st))
            assert torch.equal(parts8, want), (ts, lines, "ghost 8")
            # the interior launch wrote nothing into the first / last 8 + ts lines (it read no x from the first / last 8)
            assert bool((mid8[:(8 + ts) * Nx] == 0).all()) and bool((mid8[-(8 + ts) * Nx:] == 0).all()), (ts, lines)
            if lines == 40:                                    # the interior launch alone leaves the two edge blocks alone
                assert bool((mid[:2 * ts * Nx] == 0).all()) and bool((mid[-2 * ts * Nx:] == 0).all()) and bool((mid[2 * ts * Nx:40 * Nx] != 0).all())
    comm = LibComm()
    send = torch.arange(1000, dtype=torch.float64, device=dev)
    recv = torch.zeros(1000, dtype=torch.float64, device=dev)
    check(lib().gla_comm_all_gather_begin(comm._h, code, ptr(send), ptr(recv), 1000, st))
    check(lib().gla_halo_end(comm._h, st))
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    recv.zero_()                                               # the in-line form (the V-cycle levels' [b, x] ghost exchange)
    check(lib().gla_comm_all_gather(comm._h, code, ptr(send), ptr(recv), 1000, st))
    torch.cuda.synchronize()
    assert torch.equal(recv, send)


@pytest.mark.parametrize("Nx,Ny,up,down", [(300, 200, 1, 1), (70, 40, 1, 1), (1000, 64, 1, 1), (257, 49, 1, 1),
                                            (300, 120, 1, 0), (300, 120, 0, 1),
                                            (8192, 1024, 1, 1)])      # (BASELINE config 4's block of one of 8 ranks, at size)
def test_temporal_route_self_halo_rccl(Nx, Ny, up, down):
    """dist.RowBlockJacobi._forward_temporal with a REAL (one-rank) RCCL communicator: the rank's upper and lower neighbour
    are the rank itself (a grid periodic in the line direction), so every pass goes through pack -> ncclAllGather on the
    library's stream -> interior line blocks on the compute stream, ghost fill + first / last line block on a second
    stream -> join.  Reference: the same extended plan, one ordinary sweep at a time with the ghost lines refreshed by
    hand -- identical per-row arithmetic, so bit for bit."""
    import types
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd._lib import check, lib
    from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockJacobi
    dev = torch.device("cuda:0")
    T = RowBlockJacobi.TS_MAX
    # (up, down) = (1, 0) / (0, 1): the last / first rank of a partition -- one neighbour only, the block ends at the grid's edge
    n, n_ext, off, m = Nx * Ny, Nx * (Ny + (up + down) * T), Nx * T * up, Nx * T
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny + (up + down) * T, device=dev)
    plan = gla.CsrPlan(ei, n_ext)
    plan.set_values(ev)
    if plan.jacobi_schedule(2) != (2,):
        pytest.skip("no temporally blocked kernel under these GLA_OPTIONS")
    assert plan.info.dia_stride == Nx and max(plan.jacobi_schedule(4 * T)) <= T
    drv = RowBlockJacobi.__new__(RowBlockJacobi)
    drv.comm, drv.overlap, drv._ghost_pending = LibComm(), True, None
    drv.part = types.SimpleNamespace(world=1, rank=0, n_local=n, group=None)
    drv._sched = plan.jacobi_schedule
    drv.ext = dict(plan=plan, S=Nx, n_ext=n_ext, off=off, up=bool(up), down=bool(down), slab=T * Nx, dtype=torch.float64, dev=dev,
                   up_src=0, down_src=0)
    gen = torch.Generator().manual_seed(5)
    va = torch.rand(n, 3, generator=gen, dtype=torch.float64).to(dev)
    va[:, 0] += 3.0
    g = torch.tensor([0.7], dtype=torch.float64)

    def periodic(own):                                   # [own's last T lines | own | own's first T lines]
        return torch.cat(([own[-m:]] if up else []) + [own] + ([own[:m]] if down else []))
    # A_ii: random (the route's flag is raised by every row: the passes read the copies), the matrix diagonal (what the reference
    # passes, JacobiGNN.py:160-170: the flag stays clean, the passes neither load nor keep A_ii), and the diagonal with ONE row an
    # ulp off -- in a line that is a neighbour's ghost line, in the window the edge blocks read, in the interior
    cases = [("random", None)]
    if up and down:
        cases += [("diagonal", []), ("ghost line", [3]), ("edge window", [(2 * T + 1) * Nx + 5]), ("interior", [n // 2 + 7]),
                  ("last own row", [n - 1])]
    for a_name, off_rows in cases:
      if off_rows is not None:
        va[:, 0] = -4.0                                    # (laplacianfun_torch: the negative Laplacian, UtilsGNN.py:58-59)
        for r in off_rows:
            va[r, 0] = torch.nextafter(va[r, 0], torch.tensor(0.0, dtype=torch.float64, device=dev))
      d_e, b_e = periodic(va[:, 0].contiguous()), periodic(va[:, 1].contiguous())
      _temporal_route_case(plan, drv, va, g, d_e, b_e, periodic, up, down, n, n_ext, off, dev, a_name,
                           (2, 7, 9, 16, 20, 23) if off_rows is None else (7, 20))
    torch.cuda.synchronize()


def _temporal_route_case(plan, drv, va, g, d_e, b_e, periodic, up, down, n, n_ext, off, dev, a_name, its_list):
    from gnn_applied_linear_algebra_amd._lib import check, lib
    for its in its_list:
        x = va[:, 2].contiguous()
        try:
            check(lib().gla_set_option(b"jacobi_fuse", 0))
            for _ in range(its):
                x = plan.jacobi_sweeps(d_e, b_e.reshape(-1, 1), periodic(x).reshape(-1, 1), 0.7, 1).reshape(-1)[off:off + n]
        finally:
            check(lib().gla_set_option(b"jacobi_fuse", 1))
        # (overlap, staggered passes of the library-driven route -- option dist_stagger, csrc/gla_dist.hip: the interior launches
        # of consecutive passes depend on one another only; blocks too short for it take the plain form whatever the option says)
        for overlap, stagger in ((True, 1), (True, 0), (False, 1)):
            drv.overlap = overlap
            check(lib().gla_set_option(b"dist_stagger", stagger))
            # poison the persistent extended copies of A_ii and b: the first (AoS) pass must rebuild every row it or a later pass
            # reads -- the interior launch its lines, the hand copies the lines the edge blocks read -- with no overlap between
            # what the two concurrent launches write and read
            nan = float("nan")
            drv.ext["buffers"] = (torch.full((n_ext,), nan, dtype=torch.float64, device=dev), torch.full((n_ext,), nan, dtype=torch.float64, device=dev),
                                  torch.zeros(n_ext, dtype=torch.float64, device=dev), torch.zeros(n_ext, dtype=torch.float64, device=dev))
            got = drv.forward(its, va, g)
            if up and down:
                assert torch.equal(got.reshape(-1), x), (a_name, its, overlap, float((got.reshape(-1) - x).abs().max()))
            else:
                # One neighbour only (the first / last rank of a partition).  "My neighbour is myself" is then not a consistent
                # problem -- the neighbour's last line has nothing below it, my ghost copy of it has my first line -- so there is
                # no sweep-by-sweep reference; what is checked is that the three drivers of the route (the library's one-call
                # sequence, the Python-driven overlapped and sequential ones: the last is what the multi-rank gloo tests pin
                # against the single-rank module) fill and use a one-sided ghost block identically, bit for bit.
                drv.ext["c_route"] = False
                drv.ext["buffers"] = (torch.full((n_ext,), nan, dtype=torch.float64, device=dev), torch.full((n_ext,), nan, dtype=torch.float64, device=dev),
                                      torch.zeros(n_ext, dtype=torch.float64, device=dev), torch.zeros(n_ext, dtype=torch.float64, device=dev))
                got_py = drv.forward(its, va, g)
                drv.ext["c_route"] = True
                assert torch.equal(got, got_py), (its, overlap, "library-driven vs Python-driven route")
                if overlap and stagger:
                    first_route = got
                else:
                    assert torch.equal(got, first_route), (its, overlap, stagger, "overlapped vs sequential, staggered vs not")
            d_ext, b_ext = drv.ext["buffers"][0], drv.ext["buffers"][1]
            assert torch.equal(b_ext[off:off + n], va[:, 1]) and torch.equal(b_ext, b_e), (a_name, its, overlap, "b and its ghost lines")
            # the A_ii copy: always the lines the edge blocks read (hand copies + ghost lines); the interior only where some
            # A_ii is NOT the matrix diagonal (the library-driven route fills it after pass 0 then; the Python-driven one always)
            # (a single pass never reads the copy's interior; nor do later passes while every A_ii is the matrix diagonal)
            sched = plan.jacobi_schedule(its)
            win = 2 * drv.TS_MAX + sched[0]
            staggered = False
            if overlap and stagger and drv.ext.get("c_route", True) and len(sched) >= 2:
                # gla_dist.hip: edge heights H_0 = max(T + ts_0, 2 T), H_1 = H_0 + ts_0 + ts_1, H_i+1 = H_i + max(ts_i, ts_i+1);
                # A_ii windows of max(H_i + ts_i) lines
                Hs = [max(drv.TS_MAX + sched[0], 2 * drv.TS_MAX)]
                for a_, b_ in zip(sched[:-1], sched[1:]):
                    Hs.append(Hs[-1] + (a_ + b_ if len(Hs) == 1 else max(a_, b_)))
                if n_ext // drv.ext["S"] >= 2 * (Hs[-1] + sched[-1]) + 32:
                    staggered, win = True, max(h_ + t_ for h_, t_ in zip(Hs, sched))
            # staggered: each stream has its own flag word, and an interior launch whose flag is raised reads A_ii from the AoS
            # rows themselves -- the copy's interior is never needed, only the windows the edge blocks read
            full = a_name != "diagonal" and len(sched) >= 2 and not staggered
            if not bool(torch.isnan(d_ext).any()) or full:
                assert torch.equal(d_ext, d_e), (a_name, its, overlap, stagger, "A_ii copy")
            else:
                Er = win * drv.ext["S"]
                assert torch.equal(d_ext[:Er], d_e[:Er]) and torch.equal(d_ext[-Er:], d_e[-Er:]), (a_name, its, overlap, stagger, "A_ii windows")
        check(lib().gla_set_option(b"dist_stagger", 1))


def test_temporal_route_as_a_hip_graph_self_halo_rccl():
    """Round 5: the temporally blocked route -- gla_dist_jacobi_temporal: per pass a pack launch,
    ncclAllGather, the interior launch, ghost fill + edge launch on a second stream, fork / join, the A_ii flag -- captured into
    a HIP graph, by torch (RowBlockJacobi.capture_forward) and by the library itself (capture_forward_native ->
    gla_dist_jacobi_temporal_graph), over a real one-rank RCCL communicator whose neighbours are the rank itself.  Replays are
    bit-identical to the eager route (which the test above pins sweep by sweep) for A_ii = the diagonal (flag clean) and a
    random A_ii (flag raised in every replay), with eager calls in between (graph and eager calls never share a flag word:
    gla_jacobi_dg_slot hands captured calls words of their own with a memset node in front)."""
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
    dev = torch.device("cuda:0")
    Nx, lines = 300, 120
    drv = RowBlockJacobi.rehearsal_self_halo(Nx, lines, dev)
    if drv._sched(2) != (2,):
        pytest.skip("no temporally blocked kernel under these GLA_OPTIONS")
    n = Nx * lines
    gen = torch.Generator().manual_seed(9)
    g = torch.tensor([0.7], dtype=torch.float64)           # (a CPU tensor: no device-to-host read inside a capture)
    va_d = torch.rand(n, 3, generator=gen, dtype=torch.float64).to(dev)
    va_d[:, 0] = -4.0
    va_r = va_d.clone()
    va_r[:, 0] = torch.rand(n, generator=gen, dtype=torch.float64).to(dev) + 3.0
    try:
        for its in (20, 7):
            want_d, want_r = drv.forward(its, va_d, g).clone(), drv.forward(its, va_r, g).clone()
            assert not torch.equal(want_d, want_r)
            for kind in ("torch", "native"):
                run = drv.capture_forward(its, va_d, g) if kind == "torch" else drv.capture_forward_native(its, va_d, g)
                for rep in range(3):
                    assert torch.equal(run(va_d), want_d), (kind, its, rep, "A_ii = the diagonal")
                    assert torch.equal(run(va_r), want_r), (kind, its, rep, "A_ii random: the flag must be raised in THIS replay")
                    assert torch.equal(drv.forward(its, va_r if rep % 2 else va_d, g), want_r if rep % 2 else want_d), (kind, its, rep, "eager call between replays")
                assert torch.equal(run(va_d), want_d), (kind, its, "flag cleared again by the memset node")
                torch.cuda.synchronize()
                if kind == "native":
                    run.close()
                del run
    finally:
        torch.cuda.synchronize()
        gla._lib.lib().gla_comm_destroy(drv.comm._h)
        drv.comm._h = None


def test_k_wide_ghost_routes_self_halo_rccl():
    """The ghost-line routes of the K-wide Jacobi passes (RowBlockJacobi._forward_temporal_k) and of the fused Chebyshev pass
    (RowBlockCheby._forward_ghost) over a REAL one-rank RCCL communicator whose upper and lower neighbour are the rank itself
    (a grid periodic in the line direction): every pass goes through pack -> ncclAllGather on the library's stream -> ghost
    fill -> one temporally blocked pass.  Reference: the same periodic problem one sweep / one step at a time with the ghost
    lines refreshed by hand -- identical per-row arithmetic, so bit for bit."""
    import types
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd._lib import check, lib
    from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockCheby, RowBlockJacobi
    from gnn_applied_linear_algebra_amd.ChebyGNN import _scalars
    dev = torch.device("cuda:0")
    T = RowBlockJacobi.TS_MAX
    Nx, Ny, K = 96, 120, 4
    n, n_ext, off = Nx * Ny, Nx * (Ny + 2 * T), Nx * T
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny + 2 * T, device=dev)
    plan = gla.CsrPlan(ei, n_ext)
    plan.set_values(ev)
    if plan.jacobi_schedule(2) != (2,):
        pytest.skip("no temporally blocked kernel under these GLA_OPTIONS")
    comm = LibComm()
    gen = torch.Generator().manual_seed(8)

    def periodic(own):                                   # [own's last T lines | own | own's first T lines]
        return torch.cat([own[-off:], own, own[:off]])

    def driver(cls):
        drv = cls.__new__(cls)
        drv.comm, drv.overlap, drv._ghost_pending = comm, True, None
        drv.part = types.SimpleNamespace(world=1, rank=0, n_local=n, group=None, n_in=n, S=0)
        drv._sched = plan.jacobi_schedule
        drv.ext = dict(plan=plan, S=Nx, n_ext=n_ext, off=off, up=True, down=True, slab=T * Nx, dtype=torch.float64, dev=dev,
                       up_src=0, down_src=0)
        return drv

    # ---- Jacobi on K right-hand sides
    va = torch.rand(n, 3, K, generator=gen, dtype=torch.float64).to(dev)
    va[:, 0, :] = (torch.rand(n, 1, generator=gen, dtype=torch.float64) + 3.0).to(dev)
    g = torch.tensor([0.7], dtype=torch.float64)
    d_e, b_e = periodic(va[:, 0, 0].contiguous()), periodic(va[:, 1, :].contiguous())
    jac = driver(RowBlockJacobi)
    for its in (2, 4, 7, 9):
        x = va[:, 2, :].contiguous()
        try:
            check(lib().gla_set_option(b"jacobi_fuse", 0))
            for _ in range(its):
                x = plan.jacobi_sweeps(d_e, b_e, periodic(x), 0.7, 1)[off:off + n]
        finally:
            check(lib().gla_set_option(b"jacobi_fuse", 1))
        got = jac.forward(its, va, g)
        assert got.shape == (n, 1, K) and torch.equal(got[:, 0, :], x), ("jacobi K", its)
        va_e = jac.ext_tensor(3, K)                      # operand allocated with its ghost rows: read in place, same bits
        va_e.copy_(va)
        assert jac._ext_of(va_e, (3, K)) is not None and jac._ext_of(va, (3, K)) is None
        got_e = jac.forward(its, va_e, g)
        assert torch.equal(got_e, got) and torch.equal(va_e, va), ("jacobi K, ext operand", its)
    # ---- Chebyshev: fused passes with ghost exchanges vs one step at a time on the periodic extension
    b = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    x0 = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    gch = torch.tensor([-3.4, -4.0], dtype=torch.float64)
    ch = driver(RowBlockCheby)
    for deg in (2, 4, 5, 8, 9):
        c, d, alphas, betas = _scalars(gch, deg)
        xe, be = periodic(x0), periodic(b)
        re, pe, xo = torch.empty_like(xe), torch.empty_like(xe), torch.empty_like(xe)
        plan.cheby_first(be, xe, re, pe, xo, float(alphas[0]))
        x, r, p = xo[off:off + n].clone(), re[off:off + n].clone(), pe[off:off + n].clone()
        for it in range(2, deg + 1):
            pe_in, re_io, xe_io = periodic(p), periodic(r), periodic(x)
            pe_out = torch.empty_like(pe_in)
            plan.cheby_next(pe_in, pe_out, re_io, xe_io, float(alphas[it - 2]), float(alphas[it - 1]), float(betas[it - 1]))
            x, r, p = xe_io[off:off + n].clone(), re_io[off:off + n].clone(), pe_out[off:off + n].clone()
        gx, gr, gp, _ = ch.forward(deg, b, x0, gch)
        assert torch.equal(gx, x) and torch.equal(gr, r) and torch.equal(gp, p), ("cheby", deg)
        b_e, x_e = ch.ext_tensor(K), ch.ext_tensor(K)    # operands allocated with their ghost rows
        b_e.copy_(b)
        x_e.copy_(x0)
        ex, er, ep, _ = ch.forward(deg, b_e, x_e, gch)
        assert torch.equal(ex, x) and torch.equal(er, r) and torch.equal(ep, p), ("cheby, ext operands", deg)
        assert torch.equal(b_e, b) and torch.equal(x_e, x0), ("cheby: operands untouched", deg)
    torch.cuda.synchronize()
    lib().gla_comm_destroy(comm._h)
    comm._h = None
