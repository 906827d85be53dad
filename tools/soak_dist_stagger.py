"""Soak of the staggered multi-GPU passes (gla_dist_jacobi_temporal, option dist_stagger): many forward() calls on one block over a
one-rank RCCL communicator (neighbours = the rank itself), every result compared bit for bit with the plain form's (dist_stagger = 0),
for A_ii = the matrix diagonal (flags clean), a random A_ii (both flags raised) and one row an ulp off in the interior / in an edge
window (one flag raised).  Usage: python tools/soak_dist_stagger.py [Nx=8192] [lines=1024] [calls=300] [--f32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import check, lib
from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
_a = [a for a in sys.argv[1:] if not a.startswith("--")]
Nx = int(_a[0]) if len(_a) > 0 else 8192
lines = int(_a[1]) if len(_a) > 1 else 1024
calls = int(_a[2]) if len(_a) > 2 else 300
DT = torch.float32 if "--f32" in sys.argv else torch.float64
os.dup2(2, 1)
dev = torch.device("cuda:0")
drv = RowBlockJacobi.rehearsal_self_halo(Nx, lines, dev, dtype=DT) if DT != torch.float64 else RowBlockJacobi.rehearsal_self_halo(Nx, lines, dev)
n = Nx * lines
gen = torch.Generator().manual_seed(7)
g = torch.tensor([0.7], dtype=DT)
bad = 0
for case in ("diagonal", "random", "interior row off", "edge-window row off"):
    va = torch.rand(n, 3, generator=gen, dtype=torch.float64).to(DT).to(dev)
    va[:, 0] = -4.0
    if case == "random":
        va[:, 0] = (3.0 + torch.rand(n, generator=gen, dtype=torch.float64)).to(DT).to(dev)
    elif case == "interior row off":
        va[n // 2 + 11, 0] = torch.nextafter(va[n // 2 + 11, 0], torch.tensor(0.0, dtype=DT, device=dev))
    elif case == "edge-window row off":
        va[3 * Nx + 5, 0] = torch.nextafter(va[3 * Nx + 5, 0], torch.tensor(0.0, dtype=DT, device=dev))
    for sweeps in (20, 16, 9):
        check(lib().gla_set_option(b"dist_stagger", 0))
        want = drv.forward(sweeps, va, g).clone()
        check(lib().gla_set_option(b"dist_stagger", 1))
        miss = 0
        for i in range(calls):
            got = drv.forward(sweeps, va, g)
            if i % 7 == 0:                                   # (a changing input in between: buffers are reused across calls)
                va[:, 2].mul_(1.0)
            if not torch.equal(got, want):
                miss += 1
        torch.cuda.synchronize()
        bad += miss
        print(f"{case:22s} {sweeps:2d} sweeps: {calls} staggered calls, {miss} differ from the plain form", flush=True)
print("OK" if bad == 0 else f"FAILED: {bad} calls differ")
sys.exit(0 if bad == 0 else 1)
