#!/usr/bin/env python3
"""rocprofv3 --pmc workload: 5 launches each of the K = 16 SpMM / Jacobi / Chebyshev-next kernels at 4096^2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla  # noqa: E402

N, K = 4096, 16
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
del ei
gen = torch.Generator().manual_seed(1)
X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
Y, R, P2, Xo = (torch.empty_like(X) for _ in range(4))
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
for _ in range(5):
    plan.spmm(X, out=Y)
    plan.jacobi_sweep(diag, B, X, Xo, 0.7)
    plan.cheby_next(B, P2, R, Xo, -0.25, -0.26, 0.1)
torch.cuda.synchronize()
print("done")
