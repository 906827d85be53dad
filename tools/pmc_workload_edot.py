#!/usr/bin/env python3
"""PMC workload: gla_spmm and gla_edge_dot (variants 0/7) with K = 20 on the N x N 5-point pattern, fp64 + fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
for dtype in (torch.float64, torch.float32):
    plan = gla.CsrPlan(ei, n)
    plan.set_values(ev.to(dtype))
    X = torch.rand(n, 20, dtype=dtype, device=dev)
    G = torch.rand(n, 20, dtype=dtype, device=dev)
    Y = torch.empty_like(X)
    for _ in range(3):
        plan.spmm(X, out=Y)
    for var in (0, 7):
        gla._lib.check(gla._lib.lib().gla_set_option(b"spmm_variant", var))
        for _ in range(3):
            plan.edge_dot(G, X)
    gla._lib.check(gla._lib.lib().gla_set_option(b"spmm_variant", 0))
    torch.cuda.synchronize()
print("done")
