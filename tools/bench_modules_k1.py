#!/usr/bin/env python3
"""The reference's OWN calls -- every module with ONE vector, exactly the tensors the reference's demos pass -- on the N x N
Laplacian through the module surface: time per forward and the bytes SURVEY 8(d) charges.  Usage: bench_modules_k1.py [N=4096]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n, nnz = N * N, ei.shape[1]
gen = torch.Generator().manual_seed(1)
r = lambda *s: torch.rand(*s, generator=gen, dtype=torch.float64).to(dev)   # noqa: E731


def tm(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps


base = 12 * nnz + 4 * (n + 1)
rows = []
x = r(n, 1)
mv = gla.MatVecGNN.MatVecGNN() if hasattr(gla.MatVecGNN, "MatVecGNN") else None
ea2 = torch.cat([ev, torch.zeros_like(ev)], 1)
batch = torch.zeros(n, device=dev)
ml = gla.MetaLayer(gla.MatVecGNN.EdgeUpdate(), gla.MatVecGNN.VertexUpdate(gla.MatVecGNN.edge_to_vertex_aggregation))
rows.append(("MatVecGNN MetaLayer (returns cat[x, y], cat[A, c])", tm(lambda: ml(x, ei, ev, None, batch)), base + 16 * n))
res = gla.GNNResidual.GNNResidual()
vb = r(n, 2)
rows.append(("GNNResidual.forward", tm(lambda: res(vb, ei, ev, None) if False else res(vb, ei, ea2, None)), base + 24 * n))
jac = gla.JacobiGNN.JacobiGNN()
va = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64, device=dev), r(n, 2)], 1)
g = torch.tensor([0.7], dtype=torch.float64)
rows.append(("JacobiGNN.forward, 20 sweeps", tm(lambda: jac(20, va, ei, ea2, g)), 20 * (base + 32 * n)))
gc = torch.tensor([-3.4, -4.0], dtype=torch.float64)
for deg in (3, 4):
    ch = gla.ChebyGNN.ChebyRelaxGNN(deg, return_edge_attr=False)
    rows.append((f"ChebyRelaxGNN({deg}).forward (no edge_attr)", tm(lambda: ch(vb, ei, ev, gc)), (base + 40 * n) + (deg - 1) * (base + 48 * n)))
ch = gla.ChebyGNN.ChebyRelaxGNN(3)
rows.append(("ChebyRelaxGNN(3).forward (with edge_attr [A, z])", tm(lambda: ch(vb, ei, ev, gc)), (base + 40 * n) + 2 * (base + 48 * n) + 28 * nnz))
pw = gla.PowerMethodGNN.PowerMethodGNN(10, return_edge_attr=False)
g3 = torch.zeros(3, dtype=torch.float64)
rows.append(("PowerMethodGNN(10).forward (no edge_attr)", tm(lambda: pw(vb, ei, ea2, g3)), 11 * (base + 16 * n) + 40 * n))
for name, t, nbytes in rows:
    print(f"{name:58s} {t:9.3f} ms   {nbytes / t / 1e6:9.1f} GB/s on the byte model ({nbytes / t / 8e7:6.1f} % of 8 TB/s)")
