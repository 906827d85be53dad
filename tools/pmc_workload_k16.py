#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over BASELINE config 3's kernels: ChebyRelaxGNN(4) (one fused pass,
cheby_march5_kernel), PowerMethodGNN(10) (two power_march_kernel passes + the fused Rayleigh tail) and JacobiGNN(8 sweeps:
two jacobi_marchk_kernel passes) on the
4096^2 Laplacian with 16 right-hand sides, plus a calibration add of known volume (FETCH_SIZE x 2 on gfx950)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 16
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
gen = torch.Generator().manual_seed(1)
va = torch.rand(n, 2, K, generator=gen, dtype=torch.float64).to(dev)
cheb = gla.ChebyGNN.ChebyRelaxGNN(4, return_edge_attr=False)
powm = gla.PowerMethodGNN.PowerMethodGNN(10, return_edge_attr=False)
g2 = torch.tensor([-3.4, -4.0], dtype=torch.float64)
g3 = torch.zeros(3, K, dtype=torch.float64)
jac = gla.JacobiGNN.JacobiGNN()
va3 = torch.rand(n, 3, K, generator=gen, dtype=torch.float64).to(dev)
va3[:, 0, :] = -4.0
ea2 = torch.cat([ev, torch.zeros_like(ev)], 1)
gw = torch.tensor([0.7], dtype=torch.float64)
for _ in range(2):
    out = cheb(va, ei, ev, g2)
    outp = powm(va, ei, ev, g3)
    outj = jac(8, va3, ei, ea2, gw)
torch.cuda.synchronize()
a = torch.rand(64 * 1024 * 1024, dtype=torch.float64, device=dev)
c = torch.rand(64 * 1024 * 1024, dtype=torch.float64, device=dev)
d = a + c
torch.cuda.synchronize()
print("done", float(out[0][0, 1, 0]), float(outp[2][2, 0]), float(d[0]))
