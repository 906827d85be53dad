#!/usr/bin/env python3
"""BASELINE config 5 shape at scale: multi-level AMG V-cycle on the heateqnfem2dfun Q1 mesh (m x m interior).
Reports setup time (SOC, DI, P, Galerkin SpGEMM per level) and time per cycle.  Usage: bench_vcycle.py [m] [levels]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
splitting = sys.argv[3] if len(sys.argv) > 3 else None
sign = int(os.environ.get("GLA_SIGN", "1"))   # +1: SPD stiffness matrix (classical AMG convention); -1: the reference's negative convention
coarse_sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
coarse_deg = int(sys.argv[5]) if len(sys.argv) > 5 else None
coarse_solver = sys.argv[6] if len(sys.argv) > 6 and not sys.argv[6].startswith('--') else 'cheby'
dev = torch.device("cuda:0")
t0 = time.perf_counter()
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
n = m * m
A = gla.VCycle._as_sparse(ei, ev if sign > 0 else -ev, (n, n))
torch.cuda.synchronize()
t1 = time.perf_counter()
if splitting == 'grid':
    splitting = gla.VCycle.GridSplitting(m, m)
H = gla.VCycle.AMGHierarchy(A, levels=levels, splitting=splitting, coarse_sweeps=coarse_sweeps, min_coarse=500, coarse_deg=coarse_deg, coarse_solver=coarse_solver)
print('coarse bounds (c, d):', H.coarse_bounds)
torch.cuda.synchronize()
t2 = time.perf_counter()
del H                                               # second build: device allocations are warm (the first one pays
torch.cuda.synchronize()                            # the driver's first-touch cost of every large hipMalloc)
t2b = time.perf_counter()
if isinstance(splitting, gla.VCycle.GridSplitting):
    splitting = gla.VCycle.GridSplitting(m, m)      # the grid splitting walks down the levels: a fresh one per hierarchy
H = gla.VCycle.AMGHierarchy(A, levels=levels, splitting=splitting, coarse_sweeps=coarse_sweeps, min_coarse=500, coarse_deg=coarse_deg, coarse_solver=coarse_solver)
torch.cuda.synchronize()
t2c = time.perf_counter()
print(f"m={m} n={n} nnz={ei.shape[1]} build {1e3 * (t1 - t0):.1f} ms  setup({levels} levels) cold {1e3 * (t2 - t1):.1f} ms, "
      f"repeated {1e3 * (t2c - t2b):.1f} ms")
for lv, Al in enumerate(H.A):
    print(f"  level {lv}: n={Al.shape[0]} nnz={Al._nnz()}")
gen = torch.Generator().manual_seed(0)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.zeros_like(b)
r0 = float(torch.norm(H.residual(b, x)))
x = H.cycle(b, x)
torch.cuda.synchronize()
t3 = time.perf_counter()
ncyc = 5
for _ in range(ncyc):
    x = H.cycle(b, x)
torch.cuda.synchronize()
t4 = time.perf_counter()
r = float(torch.norm(H.residual(b, x)))
if "--graph" in sys.argv:
    run = H.capture_cycle()
    xg = torch.zeros_like(b)
    xe = torch.zeros_like(b)
    for _ in range(2):
        xg, xe = run(b, xg), H.cycle(b, xe)
    same = bool(torch.equal(xg, xe))
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    for _ in range(ncyc):
        xg = run(b, xg)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    print(f"captured cycle (HIP graph replay) {1e3 * (t6 - t5) / ncyc:.2f} ms  bit-identical to eager: {same}")
for a in sys.argv:
    if a.startswith("--rhs="):
        Kr = int(a.split("=")[1])
        Bk = torch.rand(n, Kr, generator=gen, dtype=torch.float64).to(dev)
        Xk = H.cycle(Bk, torch.zeros_like(Bk))
        torch.cuda.synchronize()
        t7 = time.perf_counter()
        for _ in range(3):
            Xk = H.cycle(Bk, Xk)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        print(f"cycle on {Kr} right-hand sides at once (SpMM smoothers / transfers): {1e3 * (t8 - t7) / 3:.2f} ms "
              f"= {1e3 * (t8 - t7) / 3 / Kr:.2f} ms per right-hand side")
        del Bk, Xk
print(f"cycle {1e3 * (t4 - t3) / ncyc:.2f} ms  residual {r0:.3e} -> {r:.3e} after {ncyc + 1} cycles  "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB (torch only)")
