#!/usr/bin/env python3
"""SpMV / Jacobi sweep on NON-stencil matrices (what csr_general runs on): (i) the level-2 Galerkin operator of the 4095^2 Q1 FEM
mesh under pmis_splitting, (ii) a 1 M-row graph with power-law row lengths (3 .. 3000 nonzeros), (iii) rows of 64 .. 1024.
Per matrix: us per launch, SURVEY 8(d) bytes (12 nnz + 4 (n + 1) + 16 n resp. + 32 n), GB/s, fraction of 8 TB/s, and the
largest relative deviation from the oracle's in-order row sums (C oracle, CPU).  Usage: bench_irregular.py [m=4095] [json]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402

dev = torch.device("cuda:0")
HBM = 8e12


def median_us(fn, reps=15):
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def measure(name, ei, ev, n, check=True):
    nnz = ei.shape[1]
    plan = gla.CsrPlan(ei, n)
    plan.set_values(ev)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    diag = (torch.rand(n, generator=gen, dtype=torch.float64) + 1.0).to(dev)
    y, xo = torch.empty_like(x), torch.empty_like(x)
    t_mv = median_us(lambda: plan.spmm(x, out=y))
    t_j = median_us(lambda: plan.jacobi_sweep(diag, b, x, xo, 0.7))
    B_mv = 12 * nnz + 4 * (n + 1) + 16 * n
    B_j = 12 * nnz + 4 * (n + 1) + 32 * n
    lens = torch.bincount(ei[0], minlength=n)
    rec = {"matrix": name, "rows": n, "nnz": nnz, "avg_row": round(nnz / n, 2), "max_row": int(lens.max()),
           "rows_ge_64": int((lens >= 64).sum()), "long_rows": int(plan.info.long_rows),
           "spmv_us": round(t_mv, 1), "spmv_GBps": round(B_mv / t_mv / 1e3, 1), "spmv_frac": round(B_mv / (t_mv * 1e-6) / HBM, 4),
           "jacobi_us": round(t_j, 1), "jacobi_GBps": round(B_j / t_j / 1e3, 1), "jacobi_frac": round(B_j / (t_j * 1e-6) / HBM, 4)}
    if check:
        import c_oracle
        A = c_oracle.Csr(ei.cpu().numpy(), ev.cpu().numpy().reshape(-1), n)
        want = A.spmm(x.cpu().numpy())
        got = plan.spmm(x).cpu().numpy()
        A.val = np.abs(A.val)                                  # sum_j |A_ij x_j|: the scale rounding errors of a row sum live on
        scale = np.maximum(A.spmm(np.abs(x.cpu().numpy())), 1e-300)
        rec["spmv_max_dev_over_row_abs_sum_vs_inorder_oracle"] = float(np.max(np.abs(got - want) / scale))
        rec["spmv_rows_bit_identical"] = float(np.mean(got == want))
    print(json.dumps(rec), flush=True)
    return rec


def power_law_graph(n, lo, hi, alpha, seed, window=0):
    gen = torch.Generator().manual_seed(seed)
    u = torch.rand(n, generator=gen, dtype=torch.float64)
    lens = (lo * (1 - u) ** (-1.0 / (alpha - 1.0))).clamp(max=hi).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n), lens)
    if window:                                                # columns within +- window of the row (a banded graph: cache-friendly gathers)
        cols = (rows + torch.randint(-window, window + 1, (rows.numel(),), generator=gen)).clamp(0, n - 1)
    else:
        cols = torch.randint(0, n, (rows.numel(),), generator=gen)
    ev = torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.5
    return torch.stack([rows, cols]).to(dev), ev.to(dev)


out = []
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
# (i) level-2 Galerkin operator of the m^2 FEM mesh under PMIS
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
A = gla.VCycle._as_sparse(ei, ev, (m * m, m * m))
H = gla.VCycle.AMGHierarchy(A, levels=3, splitting="pmis", min_coarse=500, coarse_deg=4)
for lv in (1, 2):
    Al = H.A[lv]
    eil, eal = gla.UtilsGNN.coo_to_gnn_input(Al)
    out.append(measure(f"level-{lv} Galerkin operator, {m}^2 Q1 FEM, PMIS", eil, eal, Al.shape[0], check=lv == 2))
del H, A, ei, ev
torch.cuda.empty_cache()
# (ii) power-law row lengths
ei, ev = power_law_graph(1_000_000, 3, 3000, 2.2, 11)
out.append(measure("1 M rows, power-law row lengths 3 .. 3000 (alpha 2.2), random columns", ei, ev, 1_000_000))
del ei, ev
ei, ev = power_law_graph(1_000_000, 3, 3000, 2.2, 11, window=4096)
out.append(measure("1 M rows, power-law row lengths 3 .. 3000 (alpha 2.2), columns within +-4096 of the row", ei, ev, 1_000_000))
del ei, ev
# (iii) medium rows only
gen = torch.Generator().manual_seed(3)
n = 200_000
lens = torch.randint(64, 1025, (n,), generator=gen)
rows = torch.repeat_interleave(torch.arange(n), lens)
cols = torch.randint(0, n, (rows.numel(),), generator=gen)
evm = torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.5
out.append(measure("200 k rows of 64 .. 1024 nonzeros, random columns", torch.stack([rows, cols]).to(dev), evm.to(dev), n))
cols = (rows + torch.randint(-4096, 4097, (rows.numel(),), generator=gen)).clamp(0, n - 1)
out.append(measure("200 k rows of 64 .. 1024 nonzeros, columns within +-4096 of the row", torch.stack([rows, cols]).to(dev), evm.to(dev), n))
if "--inorder" in sys.argv:
    gla._lib.check(gla._lib.lib().gla_set_option(b"row_sum", 0))
    out.append(measure("(row_sum = 0, one lane per row) 200 k rows of 64 .. 1024 nonzeros, columns within +-4096", torch.stack([rows, cols]).to(dev), evm.to(dev), n))
    gla._lib.check(gla._lib.lib().gla_set_option(b"row_sum", 1))
if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
    json.dump(out, open(sys.argv[2], "w"), indent=1)
