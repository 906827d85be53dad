#!/usr/bin/env python3
"""Train the learned Jacobi diagonal on the GPU path: the loop of the reference's
pytorch/TrainableJacobiDiag/train.py:98-108 (Adam, lr 1e-2, loss = mean Gelfand damping factor) on a small synthetic
batch of variable-coefficient 5-point operators.  Everything sparse runs in libgla_hip (gla_segment_stats for the
vertex statistics, gla_spmm / gla_edge_dot for the loss and its backward); the MLP and the optimiser are torch.

    python examples/train_learned_jacobi.py [grid=24] [graphs=8] [steps=30]
"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd.TrainableJacobiDiag import get_model, loss as L   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
n = N * N
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
xs, eis, eas, owner, mats = [], [], [], [], []
for g in range(G):
    s = (0.3 + 2.0 * torch.rand(n, generator=gen)).to(dev)                  # symmetric scaling: SPD, varying diagonal
    vals = (-ev.flatten().float()) * s[ei[0]] * s[ei[1]]
    off = ei[0] != ei[1]
    xs.append(vals[~off].reshape(-1, 1))                                       # vertex attribute = A_ii
    eis.append(ei[:, off] + g * n)
    eas.append(vals[off].reshape(-1, 1))                                       # edge attribute = A_ij
    owner.append(torch.full((n,), g, dtype=torch.long, device=dev))
    mats.append(gla.autograd.SparseMatrix(ei, vals, (n, n)))
batch = types.SimpleNamespace(x=torch.cat(xs), edgeij_pair=torch.cat(eis, 1), edge_attr=torch.cat(eas),
                              batch=torch.cat(owner), num_graphs=G, matrix=mats, coords=[None] * G)
torch.manual_seed(0)
model = get_model().to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-2)
with torch.no_grad():                                                          # plain weighted Jacobi for comparison
    base = sum(L.damping_factor(A, 2. / 3., x, exact=False) for A, x in zip(mats, xs)) / G
print(f"{G} graphs of {n} vertices; damping factor of 2/3-weighted Jacobi with the true diagonal: {float(base):.4f}")
for step in range(steps):
    opt.zero_grad()
    loss = L.loss_batch(model, batch)
    loss.backward()
    opt.step()
    if step % 5 == 0 or step == steps - 1:
        print(f"step {step:3d}  mean damping factor {loss.item():.4f}")
