import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
dev = torch.device("cuda:0")
check(lib().gla_set_option(b"jacobi_min_rows", 0))
for N in (256, 384, 512, 768, 1024, 1536, 2048, 3072):
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
    n = N * N
    plan = gla.CsrPlan(ei, n); plan.set_values(ev)
    gen = torch.Generator().manual_seed(1)
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
    res = {}
    for rnd in range(8):
        for name, fuse in (("single", 0), ("fused", 1)):
            check(lib().gla_set_option(b"jacobi_fuse", fuse))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): out = plan.jacobi_sweeps(diag, b, x, 0.7, 20)
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / 100)
    check(lib().gla_set_option(b"jacobi_fuse", 1))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"N={N:5d}: single {med['single']:7.1f} us per sweep, fused (7+7+6) {med['fused']:7.1f} us per sweep")
