#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: 20 Jacobi sweeps (gla_jacobi_sweeps) on the N x N Laplacian plus a calibration
copy of known size (cdna guide: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950 -- check it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
# the bench's step: JacobiGNN.forward through the module surface (AoS vertex attributes: gla_jacobi_sweeps_aos) -- the kernels and
# operand sources (A_ii from the matrix diagonal where vertex_attr[:, 0] repeats it) of the timed steps, three steps
va = torch.cat([diag.reshape(-1, 1), b, x], 1)
ea = torch.cat([ev, torch.zeros_like(ev)], 1)
g = torch.tensor([0.7], dtype=torch.float64, device=dev)
jac = gla.JacobiGNN.JacobiGNN()
ei2, _ = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
for _ in range(3):
    out = jac(20, va, ei2, ea, g)
torch.cuda.synchronize()
# calibration: elementwise add of two 512 MiB fp64 vectors (reads 1 GiB, writes 512 MiB, 16 B per lane)
a = torch.rand(64 * 1024 * 1024, dtype=torch.float64, device=dev)
c = torch.rand(64 * 1024 * 1024, dtype=torch.float64, device=dev)
d = a + c
torch.cuda.synchronize()
print("done", float(out[0, 0]), float(d[0]))
