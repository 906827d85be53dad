#!/usr/bin/env python3
"""torch.profiler breakdown of one damping-factor loss forward + backward (2048^2, m = 20, fp32): which kernels the
3 ms go to (library kernels vs torch elementwise / autograd glue).  Usage: python tools/prof_trainable_step.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd.TrainableJacobiDiag import loss as L
dev = torch.device("cuda:0")
N = 2048; m = 20; K = 3
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n = N * N
gen = torch.Generator(device=dev).manual_seed(5)
s = 0.5 + torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
vals = (-ev.flatten() * s[ei[0]] * s[ei[1]]).float()
A = gla.autograd.SparseMatrix(ei, vals, (n, n))
Y = torch.randn(n, m, generator=gen, device=dev, dtype=torch.float64).float()
Y = Y / torch.linalg.vector_norm(Y, dim=0)
diag = (4.0 * s * s).float().reshape(-1, 1).requires_grad_(True)
def step():
    diag.grad = None
    df = L.eigval_approx(K, m, L.build_error_matrix(A, diag, 2. / 3.), Y=Y)
    df.backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
