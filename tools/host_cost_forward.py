import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
N = 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n, 1), -4.0, dtype=torch.float64, device=dev)
va = torch.cat([diag, b, x], 1)
ea = torch.cat([ev, torch.zeros_like(ev)], 1)
g = torch.tensor([0.7], dtype=torch.float64, device=dev)
jac = gla.JacobiGNN.JacobiGNN()
for _ in range(50):
    jac(20, va, ei, ea, g)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        y = jac(20, va, ei, ea, g)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e6 * (t1 - t0) / 100:.1f} us per forward; total {1e6 * (t2 - t0) / 100:.1f} us per forward")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    y = jac(20, va, ei, ea, g)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
