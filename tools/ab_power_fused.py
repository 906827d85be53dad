"""gla_power_fused (m power iterations per pass) vs gla_power_step (one per launch) on the 4096^2 Laplacian: us per pass and per
iteration for K = 16 and K = 1, then PowerMethodGNN(10) through the module surface.  Usage: python tools/ab_power_fused.py [N]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd.plan import col_dot
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)


def timeit(fn, reps=12):
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for K in (16, 1):
    x = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
    nrm = torch.sqrt(col_dot(x, x))
    y, ss, work = plan.power_step(x)
    t1 = timeit(lambda: plan.power_step(x, nrm=nrm, out=y, work=work))
    print(f"K={K:2d} power_step (one iteration per launch): {t1:8.1f} us")
    out = torch.empty_like(x)
    for m in range(1, plan.power_fused_max_iters(K, torch.float64) + 1):
        t = timeit(lambda: plan.power_fused(x, m, nrm=nrm, out=out))
        print(f"K={K:2d} power_fused m={m}: {t:8.1f} us per pass = {t / m:7.1f} us per iteration")
    ea2 = torch.cat([ev, torch.zeros_like(ev)], 1)
    pw = gla.PowerMethodGNN.PowerMethodGNN(10, return_edge_attr=False)
    va = torch.stack([x, torch.zeros_like(x)], 1) if K > 1 else torch.cat([x, torch.zeros_like(x)], 1)
    g = torch.zeros(3, K, dtype=torch.float64) if K > 1 else torch.zeros(3, dtype=torch.float64)
    t = timeit(lambda: pw(va, ei, ea2, g), 6)
    gla._lib.check(gla._lib.lib().gla_set_option(b"jacobi_fuse", 0))
    t0 = timeit(lambda: pw(va, ei, ea2, g), 6)
    gla._lib.check(gla._lib.lib().gla_set_option(b"jacobi_fuse", 1))
    print(f"K={K:2d} PowerMethodGNN(10) module: {t / 1e3:7.3f} ms temporally blocked, {t0 / 1e3:7.3f} ms one launch per iteration")
