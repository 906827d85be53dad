#!/usr/bin/env python3
"""Randomised stress of the temporally blocked power-method kernels (gla_power_fused and the Rayleigh tail gla_power_rayleigh on
stencil plans): random grid shapes (also narrower than a strip, fewer lines than a block, ragged last line), 5- / 9-point
patterns with random holes, K in {1, 2, 4, 8, 16}, every pass size, random block heights -- against the step-by-step kernels
(plan.spmm, col_dot) to rounding.  Usage: python tools/fuzz_power.py [cases=200] [seed=0]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
from gnn_applied_linear_algebra_amd.plan import col_dot, col_scale_rsqrt
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
gen = torch.Generator().manual_seed(seed)
dev = torch.device("cuda:0")
check(lib().gla_set_option(b"jacobi_min_rows", 0))


def stencil(Nx, Ny, pts, holes, ragged):
    n = Nx * Ny - ragged
    k = torch.arange(n)
    i, j = k % Nx, k // Nx
    rows, cols = [], []
    for a, b in sorted(pts, key=lambda ab: ab[0] * Nx + ab[1]):
        ok = (i + b >= 0) & (i + b < Nx) & (j + a >= 0) & (k + a * Nx + b < n) & (k + a * Nx + b >= 0)
        if holes and (a, b) != (0, 0):
            ok &= torch.rand(n, generator=gen) >= holes
        rows.append(k[ok]); cols.append(k[ok] + a * Nx + b)
    row, col = torch.cat(rows), torch.cat(cols)
    order = torch.argsort(row * n + col)
    ei = torch.stack([row[order], col[order]])
    return ei, torch.rand(ei.shape[1], 1, generator=gen, dtype=torch.float64) - 0.4, n


bad = done = 0
for case in range(cases):
    nine = rnd.random() < 0.4
    Nx = rnd.choice([rnd.randint(3, 40), rnd.randint(40, 300), rnd.randint(250, 700)])
    Ny = rnd.choice([rnd.randint(2, 12), rnd.randint(12, 90)])
    pts = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)] if nine else [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)]
    holes = rnd.choice([0.0, 0.0, 0.15])
    ragged = rnd.choice([0, 0, rnd.randint(1, max(1, Nx - 1))]) if Ny > 2 else 0
    ei, ev, n = stencil(Nx, Ny, pts, holes, ragged)
    K = rnd.choice([1, 2, 4, 8, 16])
    plan = gla.CsrPlan(ei.to(dev), n)
    plan.set_values(ev.to(dev))
    tmax = plan.power_fused_max_iters(K, torch.float64)
    if tmax == 0:
        continue
    check(lib().gla_set_option(b"jacobi_lines", rnd.choice([0, 0, 7, 13, 40])))
    x = (torch.rand(n, K, generator=gen, dtype=torch.float64) - 0.3).to(dev)
    nrm = torch.sqrt(col_dot(x, x))
    m = rnd.randint(1, tmax)
    use_nrm = rnd.random() < 0.6
    y, ss = plan.power_fused(x, m, nrm=nrm if use_nrm else None)
    ref = x / nrm if use_nrm else x
    prev = ref
    for _ in range(m):
        prev, ref = ref, plan.spmm(ref)
    scale = float(ref.abs().max()) + 1e-300
    ok = (float((y - ref).abs().max()) <= 1e-10 * scale
          and torch.allclose(ss[1], col_dot(ref, ref), rtol=1e-10, atol=1e-300)
          and torch.allclose(ss[0], col_dot(prev, prev), rtol=1e-10, atol=1e-300))
    # the Rayleigh tail: b = y / ||y||, b^2, sum b (A b), sum b^2
    yy = (torch.rand(n, K, generator=gen, dtype=torch.float64) - 0.3).to(dev)
    ssy = col_dot(yy, yy)
    res = torch.zeros(n, 2, K, dtype=torch.float64, device=dev)
    got = plan.power_rayleigh(yy, torch.sqrt(ssy), res[:, 0, :], res[:, 1, :])
    if got is not None:
        b = col_scale_rsqrt(yy, ssy)
        ok = ok and torch.equal(res[:, 0, :], b) and torch.equal(res[:, 1, :], b * b)
        ok = ok and torch.allclose(got[0], col_dot(b, plan.spmm(b)), rtol=1e-9, atol=1e-14)
        ok = ok and torch.allclose(got[1], col_dot(b, b), rtol=1e-10, atol=0)
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, nine=nine, Nx=Nx, Ny=Ny, holes=holes, ragged=ragged, K=K, m=m, use_nrm=use_nrm, tail=got is not None))
check(lib().gla_set_option(b"jacobi_lines", 0))
print(f"{done} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
