#!/usr/bin/env python3
"""A/B: gla_jacobi_sweeps with one sweep per pass (spmv_dia_kernel) vs two / five sweeps per pass
(jacobi_fused_dia2d_kernel, temporal blocking), same process, interleaved.
Usage: python tools/ab_jacobi_fuse.py [N=4096] [lines,lines,...] [fem]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd._lib import lib, check   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lines_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2][0].isdigit() else [0, 32, 64, 96]
fem = "fem" in sys.argv
dev = torch.device("cuda:0")
if fem:
    ei, ev = gla.UtilsGNN.heateqnfem2dfun([N + 1, N + 1], [1.0, 1.0], device=dev)
else:
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n, nnz = N * N, ei.shape[1]
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
gen = torch.Generator().manual_seed(24601)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), -4.0 if not fem else 8.0 / 3.0, dtype=torch.float64, device=dev)
B_J = 12 * nnz + 4 * (n + 1) + 32 * n
SW = 20
variants = [("single", 0, 0)] + [(f"x2 L={L}", 2, L) for L in lines_list] + [(f"x5 L={L}", 1, L) for L in lines_list]
res, outs = {}, {}
for rnd in range(7):
    for name, fuse, L in variants:
        check(lib().gla_set_option(b"jacobi_fuse", fuse))
        check(lib().gla_set_option(b"jacobi_lines", L))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = plan.jacobi_sweeps(diag, b, x, 0.7, SW)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / SW)
        outs[name] = out
check(lib().gla_set_option(b"jacobi_fuse", 1))
check(lib().gla_set_option(b"jacobi_lines", 0))
print(f"N={N} {'FEM 9-pt' if fem else 'Laplacian 5-pt'} dia={plan.info.dia}, {SW} sweeps per call")
for name, ts in res.items():
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:14s}: {med:7.1f} us per sweep  {1e6 / med:8.1f} sweeps/s  B_J/t = {B_J / med / 1e3:7.1f} GB/s "
          f"({B_J / med / 8e6 * 100:5.1f}% of 8 TB/s on SURVEY's per-sweep byte model)  bit-identical to single: "
          f"{bool(torch.equal(outs[name], outs['single']))}")
