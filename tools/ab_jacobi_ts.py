"""Time of ONE march pass by the sweeps it carries (A_ii = the centre coefficient: diag = NULL), each checked against single
sweeps.  Usage: python tools/ab_jacobi_ts.py [N=4096] [ts list, default 6,7,8,9,10]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tss = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "6,7,8,9,10").split(",")]
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


for _ in range(300):
    check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(out), 0.7, 7, 0, st))
torch.cuda.synchronize()
check(lib().gla_set_option(b"jacobi_fuse", 0))
refs, cur = {}, x
for k in range(1, max(tss) + 1):
    cur = plan.jacobi_sweeps(diag, b, cur, 0.7, 1)
    refs[k] = cur
check(lib().gla_set_option(b"jacobi_fuse", 1))
for rnd in range(3):
    for ts in tss:
        m, mn = med(lambda: check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(out), 0.7, ts, 0, st)))
        print(f"TS={ts:2d}: {m:7.1f} us (min {mn:.1f}) = {m / ts:5.1f} us per sweep; identical to {ts} single sweeps: {bool(torch.equal(out, refs[ts]))}")
