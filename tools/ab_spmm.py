#!/usr/bin/env python3
"""Tile-size sweep for the K-rhs kernels.  Usage: python tools/ab_spmm.py N K tile1,tile2,..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd._lib import lib, check   # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
tiles = [int(v) for v in sys.argv[3].split(",")]
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n, nnz = N * N, ei.shape[1]
plans = {}
for tnz in tiles:
    check(lib().gla_set_option(b"tile_nnz", tnz))
    p = gla.CsrPlan(ei, n)
    p.set_values(ev)
    plans[tnz] = p
gen = torch.Generator().manual_seed(0)
X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
Y, R, P, P2, Xo = (torch.empty_like(X) for _ in range(5))
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
base = 12 * nnz + 4 * (n + 1)
ops = {
    "spmm": (lambda p: p.spmm(X, out=Y), base + 16 * n * K),
    "jacobi": (lambda p: p.jacobi_sweep(diag, B, X, Xo, 0.7), base + 8 * n + 24 * n * K),
    "cheby_next": (lambda p: p.cheby_next(P, P2, R, Xo, -0.25, -0.26, 0.1), base + 48 * n * K),
}
res = {}
chunks = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [64]
dias = [1, 0] if "--dia-ab" in sys.argv else [1]
for rnd in range(6):
  for ch in chunks:
   for dia in dias:
    check(lib().gla_set_option(b"xcd_chunk", ch))
    check(lib().gla_set_option(b"dia", dia))
    for tnz in tiles:
        for name0, (fn, nb) in ops.items():
            name = f"{name0}/C{ch}" + ("" if len(dias) == 1 else f"/dia{dia}")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(plans[tnz])
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault((name, tnz), []).append(e0.elapsed_time(e1) * 1e3)
for (name, tnz), ts in sorted(res.items()):
    ts.sort()
    med = ts[len(ts) // 2]
    nb = ops[name.split('/')[0]][1]
    print(f"{name:16s} K={K} tile_nnz={plans[tnz].info.tile_nnz:5d}: {med:8.1f} us  {nb / med / 1e3:7.1f} GB/s ({nb / med / 8e6 * 100:5.1f}%)")
