#!/usr/bin/env python3
"""Cost of the interior / boundary split of a temporally blocked pass (the multi-GPU route launches the interior line blocks
while the ghost lines are in flight, then the first and last line block): part 0 (all) vs part 1 + part 2, per launch.
Usage: python tools/ab_jacobi_parts.py [N=4096]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
out = torch.empty_like(x)
st, code = stream_ptr(dev), dtype_code(torch.float64)
res = {}
for rnd in range(6):
    for ts in (7, 6, 5):
        for name, parts in (("all", (0,)), ("interior", (1,)), ("boundary", (2,)), ("interior+boundary", (1, 2))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for p in parts:
                    check(lib().gla_jacobi_fused(plan._h, code, ptr(diag), ptr(b), ptr(x), ptr(out), 0.7, ts, p, st))
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((ts, name), []).append(e0.elapsed_time(e1) * 1e3 / 3)
for (ts, name), v in res.items():
    v.sort(); print(f"TS={ts} {name:18s}: {v[len(v) // 2]:7.1f} us")
