"""9- and 10-sweep instances of the plain DGC pass (GLA_M5_VARIANTS build): option m5_variant = 100 * ts + v makes a 7-sweep call
run ts = 9 / 10 levels (timing only).  v: 0 = M5Cfg's choice (hd 0, static set indices), 1 = hd 0 plain-int indices, 2 = hd 1, 3 = hd 3,
4 / 5 / 6 = prefetch 2 with hd 2 / 0 / 3.  Usage: GLA_LIB_PATH=.../libgla_hip_variants.so python tools/ab_m5_levels.py [N=4096]"""
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


run = lambda: check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(out), 0.7, 7, 0, st))
for _ in range(300):
    run()
torch.cuda.synchronize()
# references: 9 and 10 single... (7 + 2, 7 + 3 through the shipped passes)
def ref(k):
    t = torch.empty_like(x)
    check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(x), ptr(t), 0.7, 7, 0, st))
    o = torch.empty_like(x)
    check(lib().gla_jacobi_fused(plan._h, dtype_code(torch.float64), None, ptr(b), ptr(t), ptr(o), 0.7, k - 7, 0, st))
    return o
refs = {9: ref(9), 10: ref(10)}
for rnd in range(2):
    m, mn = med(run)
    print(f"7 sweeps, shipped                  : {m:7.1f} us (min {mn:.1f})  {m / 7:5.1f} us per sweep")
    for ts in (9, 10):
        for v in (0, 1, 2, 3, 4, 5, 6):
            if lib().gla_set_option(b"m5_variant", 100 * ts + v) != 0:
                continue
            try:
                m, mn = med(run)
            except Exception as e:
                print(ts, v, "failed", e); continue
            print(f"{ts:2d} sweeps, variant {v}               : {m:7.1f} us (min {mn:.1f})  {m / ts:5.1f} us per sweep  bit-identical to 7 + {ts - 7}: {bool(torch.equal(out, refs[ts]))}")
    check(lib().gla_set_option(b"m5_variant", 0))
