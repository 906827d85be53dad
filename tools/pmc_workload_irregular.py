#!/usr/bin/env python3
"""Counter workload: five K = 1 SpMV launches (spmv_stream_vec2_kernel) on ONE irregular matrix of tools/bench_irregular.py.
Usage: pmc_workload_irregular.py <powerlaw_random|powerlaw_banded|medium_random|medium_banded|laplacian_csr>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_applied_linear_algebra_amd as gla          # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1]
gen = torch.Generator().manual_seed(11)
if kind.startswith("powerlaw"):
    n = 1_000_000
    u = torch.rand(n, generator=gen, dtype=torch.float64)
    lens = (3 * (1 - u) ** (-1.0 / 1.2)).clamp(max=3000).to(torch.int64)
elif kind.startswith("medium"):
    n = 200_000
    lens = torch.randint(64, 1025, (n,), generator=gen)
if kind == "laplacian_csr":
    gla._lib.check(gla._lib.lib().gla_set_option(b"dia", 0))
    ei, ev = gla.UtilsGNN.laplacianfun_torch(4096, device=dev)
    n = 4096 * 4096
else:
    rows = torch.repeat_interleave(torch.arange(n), lens)
    if kind.endswith("banded"):
        cols = (rows + torch.randint(-4096, 4097, (rows.numel(),), generator=gen)).clamp(0, n - 1)
    else:
        cols = torch.randint(0, n, (rows.numel(),), generator=gen)
    ev = (torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.5).to(dev)
    ei = torch.stack([rows, cols]).to(dev)
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
y = torch.empty_like(x)
for _ in range(5):
    plan.spmm(x, out=y)
torch.cuda.synchronize()
print(kind, "rows", n, "nnz", int(ei.shape[1]), "algorithmic bytes per SpMV", 12 * int(ei.shape[1]) + 4 * (n + 1) + 16 * n)
