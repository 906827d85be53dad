"""A/B of the 8-level V(3,3) cycle on the 4095^2 FEM mesh: A_ii taken from the matrix diagonal by the march passes (default) vs the
extracted vector streamed (AMGHierarchy._diag_from_matrix = False); and option sym on / off.  Usage: python tools/ab_vcycle_diag.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check
V = gla.VCycle
m = 4095
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
n = m * m
A = V._as_sparse(ei, ev, (n, n))
H = V.AMGHierarchy(A, levels=8, splitting=V.GridSplitting(m, m), coarse_solver="dense", min_coarse=500)
b = torch.rand(n, 1, dtype=torch.float64, device=dev)


def ms(reps=12):
    x = torch.zeros_like(b)
    for _ in range(3):
        x = H.cycle(b, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x = H.cycle(b, x)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, x


for rnd in range(3):
    V.AMGHierarchy._diag_from_matrix = True
    t1, x1 = ms()
    V.AMGHierarchy._diag_from_matrix = False
    t0, x0 = ms()
    V.AMGHierarchy._diag_from_matrix = True
    check(lib().gla_set_option(b"sym", 0))
    t2, x2 = ms()
    check(lib().gla_set_option(b"sym", 1))
    print(f"cycle: default {t1:.3f} ms | A_ii vector streamed {t0:.3f} ms | sym=0 {t2:.3f} ms | identical: {bool(torch.equal(x1, x0) and torch.equal(x1, x2))}")
