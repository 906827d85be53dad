#!/usr/bin/env python3
"""Per-rank cost of the K-wide multi-GPU routes on ONE GPU: dist.RowBlockJacobi (K right-hand sides: passes of up to five
sweeps, one ghost-line exchange each) and dist.RowBlockCheby (fused passes of up to four steps, one exchange each) with a
real one-rank RCCL communicator whose upper and lower neighbour are the rank itself (grid periodic in the line direction),
against the single-GPU modules on the same block.  Usage: python tools/bench_dist_selfhalo_k.py [N=4096] [K=16] [sweeps=20] [deg=4]"""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd import dist as gdist
from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockCheby, RowBlockJacobi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
T = RowBlockJacobi.TS_MAX
n, n_ext, off = N * N, N * (N + 2 * T), N * T
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, Ny=N + 2 * T, device=dev)
plan = gla.CsrPlan(ei, n_ext); plan.set_values(ev)
del ei
comm = LibComm()


def driver(cls):
    drv = cls.__new__(cls)
    drv.comm, drv.overlap, drv._ghost_pending = comm, True, None
    drv.part = types.SimpleNamespace(world=1, rank=0, n_local=n, group=None, n_in=n, S=0)
    drv._sched = plan.jacobi_schedule
    drv._lib_send = {}
    drv.ext = dict(plan=plan, S=N, n_ext=n_ext, off=off, up=True, down=True, slab=T * N, dtype=torch.float64, dev=dev, up_src=0, down_src=0)
    return drv


gen = torch.Generator().manual_seed(5)
ei1, ev1 = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
ea1 = torch.cat([ev1, torch.zeros_like(ev1)], 1)


def tm(f, reps=6):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps


# ---- Jacobi on K right-hand sides
va = torch.rand(n, 3, K, generator=gen, dtype=torch.float64).to(dev)
va[:, 0, :] = (torch.rand(n, 1, generator=gen, dtype=torch.float64) + 3.0).to(dev)
g = torch.tensor([0.7], dtype=torch.float64)
jac1, jacd = gla.JacobiGNN.JacobiGNN(), driver(RowBlockJacobi)
ex0 = gdist.EXCHANGES[0]; jacd.forward(sweeps, va, g); n_ex = gdist.EXCHANGES[0] - ex0
t1 = min(tm(lambda: jac1(sweeps, va, ei1, ea1, g)) for _ in range(2))
td = min(tm(lambda: jacd.forward(sweeps, va, g)) for _ in range(2))
va_e = jacd.ext_tensor(3, K); va_e.copy_(va)          # vertex_attr allocated by the driver WITH its ghost rows: read in place
te = min(tm(lambda: jacd.forward(sweeps, va_e, g)) for _ in range(2))
print(f"JacobiGNN {sweeps} sweeps x {K} rhs, {N}^2: single-GPU module {t1:7.3f} ms; multi-GPU route {td:7.3f} ms ({n_ex} ghost-line exchanges) "
      f"= {t1 / td * 100:5.1f} % of the single-GPU rate; operands from drv.ext_tensor(): {te:7.3f} ms = {t1 / te * 100:5.1f} %")
del va, va_e
# ---- Chebyshev
vb = torch.rand(n, 2, K, generator=gen, dtype=torch.float64).to(dev)
gch = torch.tensor([-3.4, -4.0], dtype=torch.float64)
ch1, chd = gla.ChebyGNN.ChebyRelaxGNN(deg, return_edge_attr=False), driver(RowBlockCheby)
b, x0 = vb[:, 0, :].contiguous(), vb[:, 1, :].contiguous()
ex0 = gdist.EXCHANGES[0]; chd.forward(deg, b, x0, gch); n_ex = gdist.EXCHANGES[0] - ex0
t1 = min(tm(lambda: ch1(vb, ei1, ev1, gch)) for _ in range(2))
td = min(tm(lambda: chd.forward(deg, b, x0, gch)) for _ in range(2))
b_e, x_e = chd.ext_tensor(K), chd.ext_tensor(K); b_e.copy_(b); x_e.copy_(x0)
te = min(tm(lambda: chd.forward(deg, b_e, x_e, gch)) for _ in range(2))
print(f"ChebyRelaxGNN({deg}) x {K} rhs, {N}^2: single-GPU module {t1:7.3f} ms; multi-GPU route {td:7.3f} ms ({n_ex} ghost-line exchanges) "
      f"= {t1 / td * 100:5.1f} % of the single-GPU rate; operands from drv.ext_tensor(): {te:7.3f} ms = {t1 / te * 100:5.1f} %")
torch.cuda.synchronize()
