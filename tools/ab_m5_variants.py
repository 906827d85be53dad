"""Experimental instances of the 7-sweep DGC pass (GLA_M5_VARIANTS builds: libgla_hip_variants.so, picked up through GLA_LIB_PATH):
option m5_variant 0 (shipped) .. 8 (LDS read-ahead / prefetch depth, bit-identical) and 9 .. 15 (TIMING experiments with WRONG
results: no barrier, no output store, no s_setprio, the memory side alone, and combinations -- what each part of a line step
costs).  Usage: GLA_LIB_PATH=gnn_applied_linear_algebra_amd/libgla_hip_variants.so python tools/ab_m5_variants.py [N=4096]"""
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
names = {0: "shipped (hd2)", 1: "hd0", 2: "hd1", 3: "hd3", 4: "pf2 hd2", 5: "pf2 hd0", 6: "pf2 hd3", 7: "pf3 hd2", 8: "pf3 hd0",
         9: "NO BARRIER", 10: "NO OUTPUT STORE", 11: "no s_setprio", 12: "MEMORY SIDE ONLY (no levels)", 13: "no barrier, no store",
         14: "memory side only, no barrier", 15: "no barrier, no s_setprio", 16: "LINE-INTERLEAVED addressing (all arrays)",
         17: "line-interleaved, memory side only", 18: "line-interleaved, no barrier", 19: "COMPUTE SIDE ONLY (no loads in the step)",
         20: "compute side only, no barrier", 21: "compute side only, no store"}


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
ref = out.clone()
for rnd in range(3):
    for v in (0, 9, 12, 19, 20, 21):
        if lib().gla_set_option(b"m5_variant", v) != 0:
            continue
        m, mn = med(run)
        print(f"variant {v:2d} {names[v]:30s}: {m:7.1f} us (min {mn:.1f})  identical {bool(torch.equal(out, ref))}")
check(lib().gla_set_option(b"m5_variant", 0))
