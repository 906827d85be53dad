"""A/B of a library option on the headline shapes, same process, alternating: the 7- and 6-sweep march passes (A_ii = the centre
diagonal) and the module step (JacobiGNN.forward, 20 sweeps).  Usage: python tools/ab_jacobi_option.py [option=full_mask] [N=4096]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
name = (sys.argv[1] if len(sys.argv) > 1 else "full_mask").encode()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
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
print("plan:", {k: int(getattr(plan.info, k)) for k in ("dia", "dia_canon5", "dia_sym", "dia_full")})


def med(fn, reps=15):
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


va = torch.cat([diag.reshape(-1, 1), b, x], 1)
ea = torch.cat([ev, torch.zeros_like(ev)], 1)
g = torch.tensor([0.7], dtype=torch.float64, device=dev)
jac = gla.JacobiGNN.JacobiGNN()
for _ in range(60):                                   # clocks
    jac(20, va, ei, ea, g)
torch.cuda.synchronize()
ref = {}
for rnd in range(3):
    for v in (1, 0):
        check(lib().gla_set_option(name, v))
        line = f"{name.decode()}={v}:"
        for ts in (7, 6):
            m, mn = med(lambda: check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(out), 0.7, ts, 0, st)))
            same = torch.equal(ref.setdefault(ts, out.clone()), out)
            line += f"  TS={ts} {m:6.1f} us (min {mn:.1f}, identical {bool(same)})"
        m, mn = med(lambda: jac(20, va, ei, ea, g), 20)
        y = jac(20, va, ei, ea, g)
        same = torch.equal(ref.setdefault("step", y.clone()), y)
        print(line + f"  module step {m:6.1f} us (min {mn:.1f}) = {20e6 / m:7.0f} sweeps/s, identical {bool(same)}")
check(lib().gla_set_option(name, 1))
