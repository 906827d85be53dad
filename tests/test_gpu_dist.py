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
        info = jac.plan.info                                   # the block's own rows are stencil rows, its halo rows not
        assert info.dia == 5 or hi - lo <= 2 * Nx, info.dia
        if info.dia:
            assert info.dia_row_lo == (Nx if rank > 0 else 0), (rank, info.dia_row_lo)
            assert info.dia_row_hi == (hi - lo) - (Nx if rank < world - 1 else 0), (rank, info.dia_row_hi)
        if overlap:                                            # a one-line block has boundary tiles only
            assert jac.plan.n_boundary > 0 and (jac.plan.n_interior > 0 or lines[rank + 1] - lines[rank] <= 2)
        ch = RowBlockCheby(ei_l, ev_l, n, lo, hi, overlap=overlap)
        x, r, p, g_out = ch.forward(4, B[lo:hi].contiguous(), X[lo:hi].contiguous(), gch)
        assert torch.equal(x, v_ref[lo:hi, 1]) and torch.equal(r, v_ref[lo:hi, 2]) and torch.equal(p, v_ref[lo:hi, 3])
        assert torch.equal(g_out, g_ref)
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
        for kwargs in (dict(coarse_solver="dense"), dict(coarse_deg=8)):
            H = V.AMGHierarchy(A, levels=3, splitting=V.GridSplitting(m, m), **kwargs)
            D = RowBlockAMG(H, overlap=True)
            gen = torch.Generator().manual_seed(5)
            b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
            x_ref = torch.zeros_like(b)
            x_loc = D.local(x_ref)
            b_loc = D.local(b)
            for _ in range(3):
                x_ref = H.cycle(b, x_ref)
                x_loc = D.cycle(b_loc, x_loc)
                assert torch.equal(x_loc, D.local(x_ref)), "distributed cycle differs from the single-rank cycle"
            r_loc = D.residual(b_loc, x_loc)
            assert torch.equal(r_loc, D.local(H.residual(b, x_ref)))
            rn = torch.sum(r_loc * r_loc).cpu()
            dist.all_reduce(rn)
            assert float(rn.sqrt()) < 0.5 * float(torch.norm(b))
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
