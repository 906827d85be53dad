"""Timing experiments on the K-wide Jacobi pass (jacobi_marchk_kernel<double, 8, 5, 5, 512>: 4096^2 x 16 right-hand sides, 5 sweeps
per pass) in a GLA_MK_VARIANTS build (libgla_hip_variants.so through GLA_LIB_PATH): option m5_variant 0 (shipped), 1 no barrier,
2 no output store, 3 the memory side alone (no levels), 4 the compute side alone (no global loads in the line step), 5 compute
side alone without barrier, 6 memory side alone without barrier, 7 no store + no loads.  Variants 1-7 compute WRONG results.
Usage: GLA_LIB_PATH=gnn_applied_linear_algebra_amd/libgla_hip_variants.so python tools/ab_mk_variants.py [N=4096]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
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
names = {0: "shipped", 1: "NO BARRIER", 2: "NO OUTPUT STORE", 3: "MEMORY SIDE ONLY (no levels)", 4: "COMPUTE SIDE ONLY (no loads)",
         5: "compute side only, no barrier", 6: "memory side only, no barrier", 7: "no store, no loads",
         8: "NO LDS traffic in the levels", 9: "no LDS traffic, no barrier", 10: "no LDS traffic, no loads", 11: "no LDS traffic, no loads, no barrier"}


def med(fn, reps=9):
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


run = lambda: check(lib().gla_jacobi_sweeps(plan._h, dtype_code(torch.float64), K, ptr(dg), ptr(b), K, ptr(x), K, ptr(out), ptr(tmp), K, 0.7, 5, st))
for _ in range(30):
    run()
torch.cuda.synchronize()
ref = out.clone()
for rnd in range(2):
    for v in range(12):
        if lib().gla_set_option(b"m5_variant", v) != 0:
            continue
        m, mn = med(run)
        print(f"variant {v} {names[v]:32s}: {m:8.1f} us (min {mn:.1f})  identical {bool(torch.equal(out, ref))}", flush=True)
check(lib().gla_set_option(b"m5_variant", 0))
