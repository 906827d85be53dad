"""The K-wide Jacobi pass alone (4096^2 x 16 right-hand sides, 5 sweeps per pass: jacobi_marchk_kernel<double, 8, 5, 5, 512>), a few
launches, for counter passes: rocprofv3 --pmc ... -- python3 tools/pmc_marchk_workload.py [N=4096] [launches=6]."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K = 16
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
dg = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
out, tmp = torch.empty_like(x), torch.empty_like(x)
st = stream_ptr(dev)
for _ in range(reps):
    check(lib().gla_jacobi_sweeps(plan._h, dtype_code(torch.float64), K, ptr(dg), ptr(b), K, ptr(x), K, ptr(out), ptr(tmp), K, 0.7, 5, st))
torch.cuda.synchronize()
