#!/usr/bin/env python3
"""Jacobi sweeps on K right-hand sides: one sweep per launch (blocked SpMM kernel + EpiJacobi) vs the temporally blocked
passes of jacobi_marchk_kernel, through plan.jacobi_sweeps.  Usage: python tools/ab_jacobi_k.py [N=4096] [K=16] [sweeps=20] [fem]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
args = [a for a in sys.argv[1:] if a != "fem"]
N = int(args[0]) if args else 4096
K = int(args[1]) if len(args) > 1 else 16
SW = int(args[2]) if len(args) > 2 else 20
fem = "fem" in sys.argv
dev = torch.device("cuda:0")
ei, ev = (gla.UtilsGNN.heateqnfem2dfun([N + 1, N + 1], [1.0, 1.0], device=dev) if fem else gla.UtilsGNN.laplacianfun_torch(N, device=dev))
n, nnz = N * N, ei.shape[1]
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), 8.0 / 3.0 if fem else -4.0, dtype=torch.float64, device=dev)
Bj = 12 * nnz + 4 * (n + 1) + 8 * n + 24 * n * K
res, outs = {}, {}
for rnd in range(4):
    for name, fuse in (("single", 0), ("fused", 1)):
        check(lib().gla_set_option(b"jacobi_fuse", fuse))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = plan.jacobi_sweeps(diag, b, x, 0.7, SW); e1.record(); torch.cuda.synchronize()
        if rnd: res.setdefault(name, []).append(e0.elapsed_time(e1) / SW)
        outs[name] = out
check(lib().gla_set_option(b"jacobi_fuse", 1))
for name, v in res.items():
    t = sorted(v)[len(v) // 2]
    print(f"{'FEM 9-pt' if fem else 'Laplacian'} {N}^2 K={K} {SW} sweeps, {name:7s}: {t:7.3f} ms per sweep  {Bj / t / 1e6:8.1f} GB/s on the per-sweep byte model "
          f"({Bj / t / 8e7:5.1f} % of 8 TB/s)  bit-identical: {bool(torch.equal(outs[name], outs['single']))}")
