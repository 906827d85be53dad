#!/usr/bin/env python3
"""Per-op timing of the hot-path C-ABI calls on the N x N Laplacian: achieved GB/s against the algorithmic byte
model of SURVEY.md 8(d) / DESIGN.md section 3.  Usage: python tools/bench_ops.py [N] [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd.plan import col_dot, col_scale_rsqrt   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
KS = [int(k) for k in sys.argv[2:]] or [1, 16]
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n, nnz = N * N, ei.shape[1]
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
eio, evo = gla.UtilsGNN.remove_diag_entries(ei, ev)
nnz_od = eio.shape[1]
plan_od = gla.CsrPlan(eio, n)
plan_od.set_values(evo)
del ei, eio


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def report(name, us, nbytes):
    print(f"{name:38s} {us:9.1f} us   {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / us / 8e6 * 100:5.1f}% of 8 TB/s)", flush=True)


base = 12 * nnz + 4 * (n + 1)
gen = torch.Generator().manual_seed(0)
for K in KS:
    X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    Y, R, P, P2, Xo = (torch.empty_like(X) for _ in range(5))
    diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
    report(f"spmm K={K}", timeit(lambda: plan.spmm(X, out=Y)), base + 16 * n * K)
    report(f"residual K={K}", timeit(lambda: plan.residual(B, X, out=R)), base + 24 * n * K)
    report(f"jacobi_sweep K={K}", timeit(lambda: plan.jacobi_sweep(diag, B, X, Xo, 0.7)), base + 8 * n + 24 * n * K)
    report(f"cheby_first K={K}", timeit(lambda: plan.cheby_first(B, X, R, P, Xo, -0.25)), base + 40 * n * K)
    report(f"cheby_next K={K}", timeit(lambda: plan.cheby_next(P, P2, R, Xo, -0.25, -0.26, 0.1)), base + 48 * n * K)
    nrm = torch.sqrt(col_dot(X, X))
    _, _, wk = plan.power_step(X, nrm=nrm, out=Y)
    report(f"power_step (fused) K={K}", timeit(lambda: plan.power_step(X, nrm=nrm, out=Y, work=wk)), base + 16 * n * K)
    report(f"col_dot K={K}", timeit(lambda: col_dot(X, B)), 16 * n * K)
    ss = col_dot(X, X)
    report(f"col_scale_rsqrt K={K}", timeit(lambda: col_scale_rsqrt(X, ss)), 16 * n * K)
    report(f"edge_products K={K}", timeit(lambda: plan.edge_products(X)), 12 * nnz + 8 * nnz * K + 8 * n * K)
    del X, B, Y, R, P, P2, Xo
    torch.cuda.empty_cache()
report("soc_classic", timeit(lambda: plan_od.soc_classic(0.25, torch.float64)), 16 * nnz_od + 4 * (n + 1) + 8 * n)
cf = torch.zeros(n, dtype=torch.float64, device=dev)
cf[0::2] = 1
S = torch.ones(nnz_od, dtype=torch.float64, device=dev)
dg = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
report("direct_interp", timeit(lambda: plan_od.direct_interp(dg, cf, S)), 28 * nnz_od + 4 * (n + 1) + 32 * n)
report("soc_sa", timeit(lambda: plan_od.soc_sa(dg)), 20 * nnz_od + 4 * (n + 1) + 16 * n)
