#!/usr/bin/env python3
"""The reference's `__main__` self-checks, run against the MI355X path.

Each block mirrors the demo at the bottom of the corresponding reference module (file:line cited) with the imports
switched to `gnn_applied_linear_algebra_amd`, and prints the same diagnostic the reference prints.  Inputs are seeded
(the reference draws unseeded torch.rand).  Needs an MI355X; exits non-zero if a diagnostic is off.

    python examples/reference_demos.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_applied_linear_algebra_amd import MetaLayer                                           # noqa: E402
from gnn_applied_linear_algebra_amd.UtilsGNN import laplacianfun_torch, remove_diag_entries    # noqa: E402
from gnn_applied_linear_algebra_amd import (MatVecGNN, GNNResidual, JacobiGNN, ChebyGNN, PowerMethodGNN,   # noqa: E402
                                            SOCClassicGNN, DirectInterpGNN, MatrixWeightedNorm, VCycle)

torch.manual_seed(24601)
ok = True


def check(name, value, tol):
    global ok
    good = abs(value) <= tol
    ok &= good
    print(f"{name}: {value:.3e}  [{'ok' if good else 'FAIL'} <= {tol:g}]")


# ---- MatVecGNN.py:117-162 and GNNResidual.py:135-171 -------------------------------------------------------
edgeij_pair = torch.tensor([[0, 1], [1, 0], [1, 2], [2, 1], [0, 0], [1, 1], [2, 2]], dtype=torch.long).T
x = torch.tensor([[1], [10], [100]], dtype=torch.float)
A_ij = torch.tensor([1, 1, 2, 3, 10, 10, 10], dtype=torch.float).reshape(-1, 1)
batch = torch.zeros(x.size(0))
A = torch.sparse_coo_tensor(edgeij_pair, A_ij.flatten()).to_dense()
gnn_matvec = MetaLayer(MatVecGNN.EdgeUpdate(), MatVecGNN.VertexUpdate(MatVecGNN.edge_to_vertex_aggregation))
vertex_attr, _, _ = gnn_matvec(x, edgeij_pair, A_ij, None, batch=batch)
print("GNN Output", vertex_attr[:, 1].tolist(), " Real Ax", (A @ x).flatten().tolist())
check("matvec |y - Ax|", float(torch.norm(vertex_attr[:, 1:2] - A @ x)), 0.0)
r = GNNResidual.GNNResidual()(torch.cat([A @ x, x], 1), edgeij_pair, A_ij)
check("residual of b = Ax", float(torch.norm(r)), 0.0)

# ---- JacobiGNN.py:150-201 ----------------------------------------------------------------------------------
N, w = 5, 0.7
edgeij_pair, edge_attr_with_diag = laplacianfun_torch(N)
edgeij_pair, edge_attr_with_diag = edgeij_pair.cpu(), edge_attr_with_diag.cpu()
A = torch.sparse_coo_tensor(edgeij_pair, edge_attr_with_diag.flatten(), dtype=torch.float)
D = torch.diag(torch.diag(A.to_dense()))
edge_attr = torch.cat([edge_attr_with_diag, torch.zeros_like(edge_attr_with_diag)], 1).to(dtype=torch.float)
diag_vals = -4 * torch.ones((N * N, 1), dtype=torch.float)
x = torch.rand((N * N, 1), dtype=torch.float)
b = torch.rand((N * N, 1), dtype=torch.float)
vertex_attr = torch.cat([diag_vals, b, x], 1)
g = torch.tensor(w).reshape(-1)
gnn_jacobi = JacobiGNN.JacobiGNN()
x_gnn = gnn_jacobi(10, vertex_attr, edgeij_pair, edge_attr, g)
x_jac = x.clone()
for _ in range(10):
    x_jac = w * (D.inverse() @ b) + (torch.eye(N * N) - w * (D.inverse() @ A.to_dense())) @ x_jac
check("norm between traditional jacobi and GNN jacobi", float(torch.norm(x_gnn - x_jac)), 5e-6)
res = []
for i in range(100):
    vertex_attr, edge_attr, g = gnn_jacobi.iterate(vertex_attr, edgeij_pair, edge_attr, g)
    if (i + 1) % 50 == 0:
        res.append(float(torch.norm(b - A @ vertex_attr[:, 2].view(-1, 1))))
        print(f"iteration {i + 1}: norm of residual: {res[-1]:.3e}")
ok &= res[-1] < res[0]

# ---- ChebyGNN.py:355-412 -----------------------------------------------------------------------------------
edge_attr = edge_attr_with_diag.float()
x0, b = torch.rand((N * N, 1)), torch.rand((N * N, 1))
lambdaMax, lambdaMin = -7.46, -0.54
d, c = (lambdaMax + lambdaMin) / 2, (lambdaMax - lambdaMin) / 2
g = torch.tensor([c, d])
Ad = A.to_dense()
for deg in (1, 2, 3, 4, 8):
    va, _, _ = ChebyGNN.ChebyRelaxGNN(deg)(torch.cat([b, x0], 1), edgeij_pair, edge_attr, g)
    xt, rt = x0, b - Ad @ x0
    for i in range(deg):
        if i == 0:
            p, alpha = rt, 1. / d
        else:
            beta = 0.5 * (c * alpha) ** 2 if i == 1 else (c * alpha / 2) ** 2
            alpha = 1 / (d - (beta / alpha))
            p = rt + beta * p
        xt = xt + alpha * p
        rt = b - Ad @ xt
    check(f"degree {deg}: norm(x_gnn - x_trad)", float(torch.norm(va[:, 1:2] - xt)), 5e-6)

# ---- PowerMethodGNN.py:336-383 -----------------------------------------------------------------------------
A3 = torch.tensor([[1., 2., 0], [-2., 1., 2.], [1., 3., 1.]])
b3 = torch.rand((3, 1))
ei = torch.tensor([[i, j] for i in range(3) for j in range(3)]).T
ea = torch.tensor([[A3[i, j].item(), 0.] for i in range(3) for j in range(3)])
va, _, g = PowerMethodGNN.PowerMethodGNN(10)(torch.cat([b3, torch.zeros_like(b3)], 1), ei, ea, torch.zeros(3),
                                             torch.zeros(3))
b_k = b3
for _ in range(10):
    b_k = A3 @ b_k
    b_k = b_k / torch.linalg.norm(b_k)
lam = (b_k.T @ (A3 @ b_k) / (b_k.T @ b_k)).item()
print(f"Max lambda = {g[2].item():.6f} (gnn)  {lam:.6f} (usual way)")
check("power method rel difference", abs(lam - g[2].item()) / abs(lam), 1e-5)

# ---- SOCClassicGNN.py:149-186 and DirectInterpGNN.py:176-261 ------------------------------------------------
theta = 0.25
eio, evo = remove_diag_entries(edgeij_pair, edge_attr_with_diag.float())
S_ij = SOCClassicGNN.SOCClassicGNN(theta)(torch.zeros((N * N, 1)), eio, evo)
A_no_diag = Ad - torch.diag(torch.diag(Ad))
S_trad = torch.zeros((N * N, N * N))
for i in range(N * N):
    row_max = max(-A_no_diag[i, j].item() for j in range(N * N) if A_no_diag[i, j] != 0)
    S_trad[i, :] = -A_no_diag[i, :] / row_max
mism = int(torch.sum((torch.sparse_coo_tensor(eio, S_ij, (N * N, N * N)).to_dense() > 0) != (S_trad > theta)))
check("Number of differences in strong connections between GNN and traditional", mism, 0)
splitting = torch.zeros(N * N, 1)
splitting[0::2] = 1                                           # stand-in for pyamg CLJP (matlab/test_vcycle.m:65-67)
w_ij = DirectInterpGNN.DirectInterpGNN()(torch.hstack([diag_vals, splitting]), eio,
                                         torch.hstack([evo, (S_ij.reshape(-1, 1) > 0).float()]), None)
W = torch.eye(N * N) + torch.nan_to_num(torch.sparse_coo_tensor(eio, w_ij, (N * N, N * N)).to_dense())
keep = splitting.flatten() > 0
W_formula = torch.zeros(N * N, N * N)
for i in range(N * N):
    if keep[i]:
        W_formula[i, i] = 1.0
        continue
    num = torch.sum(Ad[i, :]) - Ad[i, i]
    den = Ad[i, i] * sum(Ad[i, k] for k in range(N * N) if S_trad[i, k] > theta and keep[k])
    W_formula[i, :] = (-Ad[i, :] * num) / den
check("norm of difference between Prolongation operators from GNN and from formula",
      float(torch.norm(W_formula[:, keep] - W[:, keep])), 1e-6)

# ---- MatrixWeightedNorm.py:165-210 -------------------------------------------------------------------------
n = 100
xw, Wm = torch.rand(n).reshape((-1, 1)), torch.rand(n, n)
eiw = torch.stack([torch.arange(n).repeat_interleave(n), torch.arange(n).repeat(n)])
M = MatrixWeightedNorm
gnn = MetaLayer(M.EdgeUpdate(), M.VertexUpdate(M.EdgeToVertexAggregation), M.GlobalUpdate(M.VertexToGlobalAggregation))
_, _, gw = gnn(xw, eiw, Wm.reshape(-1, 1), torch.zeros(1), batch=torch.zeros(n))
check("weighted norm difference", float(torch.abs(torch.sqrt(xw.t() @ Wm @ xw) - gw)), 1e-4)

# ---- VCycle.py:239-277 -------------------------------------------------------------------------------------
VCycle.configure(0.25, 4, N_=N)
xv, bv = torch.rand((N * N, 1)), torch.rand((N * N, 1))
print(f"original residual norm: {float(torch.norm(VCycle.runResidual(A, bv, xv))):.4e}")
for i in range(5):
    xv = VCycle.runVCycle(A, bv, xv, 3, 3, 5, True)
    rn = float(torch.norm(VCycle.runResidual(A, bv, xv)))
    print(f"Residual norm after iteration {i + 1}: {rn:.4e}")
check("V-cycle final residual", rn, 1e-5)

print("ALL OK" if ok else "SOME CHECKS FAILED")
sys.exit(0 if ok else 1)
