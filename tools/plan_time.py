#!/usr/bin/env python3
"""Plan construction (gla_csr_create_from_coo) and gla_csr_set_values time at 4096^2, with and without the diagonal
layout.  Usage: python tools/plan_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(4096, device=dev)
n = 4096 * 4096
for dia in (1, 0):
    check(lib().gla_set_option(b"dia", dia))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = gla.CsrPlan(ei, n)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        p.set_values(ev)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        del p
    print(f"dia={dia}: plan creation {1e3*(t1-t0):.2f} ms, set_values {1e3*(t2-t1):.2f} ms")
