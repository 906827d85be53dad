#!/usr/bin/env python3
"""A/B: ChebyRelaxGNN(deg) on K right-hand sides, one kernel per step (option jacobi_fuse=0) vs the temporally blocked
pass gla_cheby_fused (default), through the module surface.  Usage: python tools/ab_cheby_fused.py [N=4096] [K=16] [deg=4] [fem]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd._lib import lib, check   # noqa: E402

fem = "fem" in sys.argv
argv = [a for a in sys.argv if a != "fem"]
N = int(argv[1]) if len(argv) > 1 else 4096
K = int(argv[2]) if len(argv) > 2 else 16
deg = int(argv[3]) if len(argv) > 3 else 4
dev = torch.device("cuda:0")
ei, ev = (gla.UtilsGNN.heateqnfem2dfun([N + 1, N + 1], [1.0, 1.0], device=dev) if fem else gla.UtilsGNN.laplacianfun_torch(N, device=dev))
n, nnz = N * N, ei.shape[1]
gen = torch.Generator().manual_seed(24601)
va = torch.stack([torch.rand(n, K, generator=gen, dtype=torch.float64), torch.rand(n, K, generator=gen, dtype=torch.float64)], 1).to(dev)
g = torch.tensor([-3.4, -4.0], dtype=torch.float64)
cheb = gla.ChebyGNN.ChebyRelaxGNN(deg, return_edge_attr=False)
base = 12 * nnz + 4 * (n + 1)
B_alg = (base + 40 * n * K) + (deg - 1) * (base + 48 * n * K)
res, outs = {}, {}
for rnd in range(6):
    for name, fuse in (("step by step", 0), ("one pass", 1)):
        check(lib().gla_set_option(b"jacobi_fuse", fuse))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = cheb(va, ei, ev, g)[0]
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res.setdefault(name, []).append(e0.elapsed_time(e1))
        outs[name] = out
check(lib().gla_set_option(b"jacobi_fuse", 1))
print(f"ChebyRelaxGNN(deg={deg}).forward, {N}^2 {'FEM 9-point' if fem else 'Laplacian'}, K={K}, fp64; SURVEY bytes {B_alg / 1e9:.1f} GB")
for name, ts in res.items():
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:13s}: {med:8.3f} ms   {B_alg / med / 1e6:8.1f} GB/s on SURVEY's per-step byte model ({B_alg / med / 8e9 * 100:5.1f}% of 8 TB/s)"
          f"   bit-identical: {bool(torch.equal(outs[name], outs['step by step']))}")
