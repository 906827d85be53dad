#!/usr/bin/env python3
"""Randomised stress of the temporally blocked Jacobi kernels: random grid shapes (also narrower than a strip, fewer lines
than a block), 5- / 9-point patterns with random holes and values, fp64 / fp32, random sweep counts, block heights,
sweeps-per-pass caps and 1 ... 16 right-hand sides -- gla_jacobi_sweeps, the AoS entry point and gla_jacobi_fused_residual against one ordinary sweep at a
time, bit for bit.  Usage: python tools/fuzz_jacobi_march.py [cases=300] [seed=0]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
check(lib().gla_set_option(b"jacobi_min_rows", 0))


def stencil(Nx, Ny, pts, holes):
    k = torch.arange(Nx * Ny)
    i, j = k % Nx, k // Nx
    rows, cols = [], []
    for a, b in sorted(pts, key=lambda ab: ab[0] * Nx + ab[1]):
        ok = (i + b >= 0) & (i + b < Nx) & (j + a >= 0) & (j + a < Ny)
        if holes and (a, b) != (0, 0):
            ok &= torch.rand(Nx * Ny, generator=gen) >= holes
        rows.append(k[ok]); cols.append(k[ok] + a * Nx + b)
    row, col = torch.cat(rows), torch.cat(cols)
    order = torch.argsort(row * (Nx * Ny) + col)
    ei = torch.stack([row[order], col[order]])
    return ei, torch.rand(ei.shape[1], 1, generator=gen, dtype=torch.float64) - 0.4


P5 = [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)]
P9 = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]
opts = ("jacobi_lines", "jacobi_march", "jacobi_ts_max", "jacobi_ts_max9")
bad = 0
for c in range(cases):
    Nx = rnd.choice([3, 5, 17, 64, 130, 241, 242, 243, 256, 300, 499, 700])
    Ny = rnd.choice([1, 2, 3, 7, 15, 16, 17, 33, 64, 150, 300])
    pts = rnd.choice([P5, P9])
    holes = rnd.choice([0.0, 0.0, 0.1, 0.5])
    dtype = rnd.choice([torch.float64, torch.float32])
    its = rnd.randint(1, 24)
    K = rnd.choice([1, 1, 1, 2, 4, 8, 16])
    if dtype == torch.float32 and K == 2:
        K = 4
    var = (rnd.choice([0, 0, 3, 9, 16, 40, 200]), rnd.choice([1, 1, 1, 0]), rnd.randint(2, 8), rnd.randint(2, 6))
    n = Nx * Ny
    ei, ev = stencil(Nx, Ny, pts, holes)
    plan = gla.CsrPlan(ei.to(dev), n)
    plan.set_values(ev.to(dtype).to(dev))
    b = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dtype).to(dev)
    x = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dtype).to(dev)
    diag = (torch.rand(n, generator=gen, dtype=torch.float64) + 3.0).to(dtype).to(dev)
    try:
        check(lib().gla_set_option(b"jacobi_fuse", 0))
        want = plan.jacobi_sweeps(diag, b, x, 0.7, its)
        want_r = plan.residual(b, want) if K == 1 else None
        check(lib().gla_set_option(b"jacobi_fuse", 1))
        for o, v in zip(opts, var):
            check(lib().gla_set_option(o.encode(), v))
        sched = plan.jacobi_schedule(its, dtype)
        got = plan.jacobi_sweeps(diag, b, x, 0.7, its)
        got_aos, xr = None, None
        if K == 1:
            va = torch.stack([diag, b.reshape(-1), x.reshape(-1)], 1)
            got_aos = plan.jacobi_sweeps_aos(va, 0.7, its)
            xr = plan.jacobi_sweeps_residual(diag, b, x, 0.7, its)
    finally:
        for o, v in zip(opts, (0, 1, 8, 5)):
            check(lib().gla_set_option(o.encode(), v))
    ok = torch.equal(got, want) and (got_aos is None or torch.equal(got_aos, want)) and \
        (xr is None or (torch.equal(xr[0], want) and torch.equal(xr[1], want_r)))
    if not ok:
        bad += 1
        print("MISMATCH", dict(K=K, Nx=Nx, Ny=Ny, pts=len(pts), holes=holes, dtype=str(dtype), its=its, var=var, sched=sched,
                               dia=int(plan.info.dia), canon=int(plan.info.dia_canon5), stride=int(plan.info.dia_stride),
                               sweeps=torch.equal(got, want), aos=None if got_aos is None else torch.equal(got_aos, want),
                               res=None if xr is None else (torch.equal(xr[0], want), torch.equal(xr[1], want_r))))
torch.cuda.synchronize()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
