#!/usr/bin/env python3
"""Per-rank cost of the multi-GPU Jacobi route on ONE GPU: dist.RowBlockJacobi's temporally blocked forward with a real
(one-rank) RCCL communicator whose upper and lower neighbour are the rank itself (grid periodic in the line direction) --
pack, ncclAllGather on the library's stream, interior line blocks, ghost fill + first / last line block on a second stream --
against the single-GPU module on the same block.  The ratio is what weak scaling can reach before inter-GPU latency.
Usage: python tools/bench_dist_selfhalo.py [N=4096] [sweeps=20]"""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockJacobi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
T = RowBlockJacobi.TS_MAX
n, n_ext, off = N * N, N * (N + 2 * T), N * T
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, Ny=N + 2 * T, device=dev)
plan = gla.CsrPlan(ei, n_ext); plan.set_values(ev)
del ei
drv = RowBlockJacobi.__new__(RowBlockJacobi)
drv.comm, drv.overlap, drv._ghost_pending = LibComm(), True, None
drv.part = types.SimpleNamespace(world=1, rank=0, n_local=n, group=None)
drv._sched = plan.jacobi_schedule
drv.ext = dict(plan=plan, S=N, n_ext=n_ext, off=off, up=True, down=True, slab=T * N, dtype=torch.float64, dev=dev, up_src=0, down_src=0)
gen = torch.Generator().manual_seed(5)
va = torch.rand(n, 3, generator=gen, dtype=torch.float64).to(dev); va[:, 0] += 3.0
g = torch.tensor([0.7], dtype=torch.float64)
ei1, ev1 = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
ea1 = torch.cat([ev1, torch.zeros_like(ev1)], 1)
jac = gla.JacobiGNN.JacobiGNN()
def tm(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for a_name in ("A_ii = the matrix diagonal (-4: what the reference passes)", "A_ii random (the call's flag is raised: the passes read the copies)"):
    if a_name.startswith("A_ii = "):
        va[:, 0] = -4.0
    else:
        va[:, 0] = torch.rand(n, generator=gen, dtype=torch.float64).to(dev) + 3.0
    for _ in range(40):                                   # clocks
        jac(sweeps, va, ei1, ea1, g)
    res = {}
    for rnd in range(3):
        for name, f in (("single-GPU module", lambda: jac(sweeps, va, ei1, ea1, g)),
                        ("multi-GPU route, overlapped", lambda: (setattr(drv, "overlap", True), drv.forward(sweeps, va, g))),
                        ("multi-GPU route, sequential", lambda: (setattr(drv, "overlap", False), drv.forward(sweeps, va, g)))):
            res.setdefault(name, []).append(tm(f))
    base = min(res["single-GPU module"])
    print(a_name)
    for name, v in res.items():
        print(f"  {name:30s}: {min(v):7.3f} ms per {sweeps} sweeps = {sweeps / min(v) * 1e3:8.0f} sweeps/s  ({base / min(v) * 100:5.1f} % of the single-GPU rate)")
print("schedule", plan.jacobi_schedule(sweeps))
