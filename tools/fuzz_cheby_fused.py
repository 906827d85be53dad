#!/usr/bin/env python3
"""Randomised stress of the fused Chebyshev pass (gla_cheby_fused: cheby_march5_kernel / cheby_fused_dia2d_kernel) and of the
C-side step loop (gla_cheby_steps): random grid shapes, 5-point-type patterns with holes, K = 2 ... 16 right-hand sides,
degrees 1 ... 10, fp64 / fp32, through chebyshev_relax against one kernel per step (option jacobi_fuse = 0), bit for bit.
Usage: python tools/fuzz_cheby_fused.py [cases=200] [seed=0]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
from gnn_applied_linear_algebra_amd.ChebyGNN import chebyshev_relax
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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


PATTERNS = [[(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)], [(-1, 0), (0, -2), (0, 0), (0, 2), (1, 0)], [(-1, 1), (0, -1), (0, 0), (0, 1), (1, -1)],
            [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]]
bad = 0
for c in range(cases):
    Nx = rnd.choice([5, 17, 23, 24, 25, 64, 130, 300, 499])
    Ny = rnd.choice([1, 2, 3, 7, 8, 9, 33, 64, 150])
    pts = rnd.choice(PATTERNS)
    holes = rnd.choice([0.0, 0.0, 0.2])
    dtype = rnd.choice([torch.float64, torch.float32])
    K = rnd.choice([1, 2, 4, 8, 16, 6])
    deg = rnd.randint(1, 10)
    march = rnd.choice([1, 1, 0])
    lines = rnd.choice([0, 0, 9, 40])
    n = Nx * Ny
    ei, ev = stencil(Nx, Ny, pts, holes)
    plan = gla.CsrPlan(ei.to(dev), n)
    plan.set_values(ev.to(dtype).to(dev))
    b = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dtype).to(dev)
    x0 = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dtype).to(dev)
    g = torch.tensor([-3.4, -4.0], dtype=dtype)
    outs = []
    try:
        for fuse in (0, 1):
            check(lib().gla_set_option(b"jacobi_fuse", fuse))
            check(lib().gla_set_option(b"jacobi_march", march))
            check(lib().gla_set_option(b"jacobi_lines", lines))
            x, r, p, gathered, g_out = chebyshev_relax(plan, b, x0, g, deg, need_gathered=True)
            outs.append((x.clone(), r.clone(), p.clone(), gathered.clone(), g_out.clone()))
    finally:
        check(lib().gla_set_option(b"jacobi_fuse", 1)); check(lib().gla_set_option(b"jacobi_march", 1)); check(lib().gla_set_option(b"jacobi_lines", 0))
    ok = all(torch.equal(a_, b_) for a_, b_ in zip(outs[0], outs[1]))
    if not ok:
        bad += 1
        print("MISMATCH", dict(Nx=Nx, Ny=Ny, pts=len(pts), holes=holes, dtype=str(dtype), K=K, deg=deg, march=march, lines=lines,
                               eq=[bool(torch.equal(a_, b_)) for a_, b_ in zip(outs[0], outs[1])]))
torch.cuda.synchronize()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
