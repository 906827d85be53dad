"""A/B of the 7-sweep march pass by A_ii source: contiguous array / the plan's centre diagonal (diag = NULL), and of the module
step (gla_jacobi_sweeps_aos) with and without the A_ii flag flow.  Usage: python tools/ab_jacobi_dg.py [N=4096]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
out = torch.empty_like(x)
st = stream_ptr(dev)


def med(fn, reps=15):
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for rnd in range(2):
    for ts in (7, 6):
        m, mn = med(lambda: check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), ptr(diag), ptr(b), ptr(x), ptr(out), 0.7, ts, 0, st)))
        ref = out.clone()
        m2, mn2 = med(lambda: check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(out), 0.7, ts, 0, st)))
        print(f"TS={ts}: A_ii from the array {m:7.1f} us (min {mn:.1f});  from the centre diagonal {m2:7.1f} us (min {mn2:.1f});  identical: {bool(torch.equal(ref, out))}")
    va = torch.cat([diag.reshape(-1, 1), b, x], 1)
    ea = torch.cat([ev, torch.zeros_like(ev)], 1)
    g = torch.tensor([0.7], dtype=torch.float64, device=dev)
    jac = gla.JacobiGNN.JacobiGNN()
    for flag in (1, 0):
        check(lib().gla_set_option(b"jacobi_diag_flag", flag))
        m, mn = med(lambda: jac(20, va, ei, ea, g), 20)
        print(f"module step, 20 sweeps, jacobi_diag_flag={flag}: {m:7.1f} us (min {mn:.1f}) = {20e6 / m:8.0f} sweeps/s")
    check(lib().gla_set_option(b"jacobi_diag_flag", 1))
