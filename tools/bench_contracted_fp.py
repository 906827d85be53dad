#!/usr/bin/env python3
"""What the bit-exact formulation of the headline step costs: JacobiGNN.forward, 20 sweeps, N x N Laplacian,
fp64, through whichever library GLA_LIB_PATH names -- run once with the default libgla_hip.so and once with
libgla_hip_contracted.so (csrc/build.sh: gla_march5.hip with -ffp-contract=fast -DGLA_CONTRACTED_FP: FMA row sums, the division
as ONE multiply-add with RN(1 / A_ii)); prints ms per step and the largest deviation from a plain torch fp64 iteration
(x + w (b - A x) / d with torch.sparse on the GPU) relative to max |x|.  NEVER the default; north_star's bar is 1e-6, the bit-exact
build meets the oracle bit for bit.  Usage: [GLA_LIB_PATH=...] python tools/bench_contracted_fp.py [N=4096] [sweeps=20] [--json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_applied_linear_algebra_amd as gla  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 4096
sweeps = int(args[1]) if len(args) > 1 else 20
dev = torch.device("cuda:0")
n = N * N
ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
gen = torch.Generator().manual_seed(24601)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
va = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64, device=dev), b, x], 1)
ea = torch.cat([ev, torch.zeros_like(ev)], 1)
g = torch.tensor([0.7], dtype=torch.float64, device=dev)
jac = gla.JacobiGNN.JacobiGNN()
out = jac(sweeps, va, ei, ea, g)
t_end = time.perf_counter() + 0.25
while time.perf_counter() < t_end:                      # clocks (profiles/r04/clock_ramp.txt)
    for _ in range(5):
        jac(sweeps, va, ei, ea, g)
    torch.cuda.synchronize()
steps = 50
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    jac(sweeps, va, ei, ea, g)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
# plain torch reference: CSR SpMV per sweep, fp64
crow = torch.zeros(n + 1, dtype=torch.int64, device=dev)
crow[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
A = torch.sparse_csr_tensor(crow, ei[1], ev.flatten(), (n, n))
xr = x.clone()
for _ in range(sweeps):
    xr = xr + (0.7 * (b - A @ xr)) / -4.0
dev_rel = float((out - xr).abs().max() / xr.abs().max())
# ... and a matrix whose products and quotients are NOT exact (the Laplacian's are: its coefficients are -4 and 1, so contraction and
# the reciprocal division change nothing there): D A D with a random diagonal D, entries a_ij (d_i d_j) -- d_i d_j commutes, so the
# value set stays symmetric bit for bit and the step runs the same kernels
d = (torch.rand(n, generator=gen, dtype=torch.float64) + 0.5).to(dev)
ev2 = (ev.flatten() * (d[ei[0]] * d[ei[1]])).reshape(-1, 1)
ea2 = torch.cat([ev2, torch.zeros_like(ev2)], 1)
diag2 = torch.zeros(n, dtype=torch.float64, device=dev)
on_d = ei[0] == ei[1]
diag2[ei[0][on_d]] = ev2.flatten()[on_d]
va2 = torch.cat([diag2.reshape(-1, 1), b, x], 1)
ei_b = ei.clone()                                        # (plans are cached per edge-index tensor)
out2 = jac(sweeps, va2, ei_b, ea2, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    jac(sweeps, va2, ei_b, ea2, g)
torch.cuda.synchronize()
ms2 = 1e3 * (time.perf_counter() - t0) / steps
A2 = torch.sparse_csr_tensor(crow, ei[1], ev2.flatten(), (n, n))
xr = x.clone()
for _ in range(sweeps):
    xr = xr + (0.7 * (b - A2 @ xr)) / diag2.reshape(-1, 1)
dev_rel2 = float((out2 - xr).abs().max() / xr.abs().max())
rec = {"library": os.path.basename(os.environ.get("GLA_LIB_PATH") or "libgla_hip.so"), "N": N, "sweeps": sweeps,
       "ms_per_step": round(ms, 4), "sweeps_per_s": round(sweeps / ms * 1e3, 1),
       "max_deviation_from_torch_fp64_iteration_rel_to_max_x": float(f"{dev_rel:.3e}"),
       "scaled_matrix_DAD": {"ms_per_step": round(ms2, 4), "max_deviation_from_torch_fp64_iteration_rel_to_max_x": float(f"{dev_rel2:.3e}")}}
print(json.dumps(rec) if "--json" in sys.argv else rec, flush=True)
