import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
m = 4095; dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
n = m * m
A = gla.VCycle._as_sparse(ei, ev, (n, n))
H = gla.VCycle.AMGHierarchy(A, levels=8, splitting=gla.VCycle.GridSplitting(m, m), coarse_sweeps=1, min_coarse=500, coarse_deg=4, coarse_solver='dense')
print("n_pre", H.n_pre, "n_post", H.n_post, "levels", [a.shape[0] for a in H.A])
def tm(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
tot = 0
for lvl in range(len(H.A) - 1):
    nl = H.A[lvl].shape[0]
    b = torch.rand(nl, 1, dtype=torch.float64, device=dev); x = torch.rand(nl, 1, dtype=torch.float64, device=dev)
    t_s = tm(lambda: H._smooth(lvl, H.n_pre, b, x)); t_r = tm(lambda: H.residual(b, x, lvl))
    r = H.residual(b, x, lvl)
    t_pt = tm(lambda: H.Pt[lvl].spmm(r)); rc = H.Pt[lvl].spmm(r)
    t_p = tm(lambda: H.P[lvl].spmm(rc.contiguous())); t_add = tm(lambda: x + x)
    t_z = tm(lambda: torch.zeros_like(rc))
    print(f"level {lvl} n={nl}: smooth({H.n_pre}) {t_s:7.1f} us  residual {t_r:7.1f}  restrict {t_pt:7.1f}  prolong {t_p:7.1f}  add {t_add:6.1f} zeros {t_z:6.1f}  sum(2 smooth) {2*t_s+t_r+t_pt+t_p+t_add+t_z:8.1f}")
    tot += 2*t_s+t_r+t_pt+t_p+t_add+t_z
b = torch.rand(n, 1, dtype=torch.float64, device=dev); x = torch.zeros_like(b)
print("sum of parts", tot, "us; whole cycle", tm(lambda: H.cycle(b, x), 10), "us")
t0 = time.perf_counter(); 
for _ in range(10): H.cycle(b, x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per cycle", (t1 - t0) * 1e5, "us; drained after", (t2 - t1) * 1e6, "us")
