#!/usr/bin/env python3
"""Kernel lists of ONE V-cycle: AMGHierarchy.cycle (single GPU) or RowBlockAMG.cycle (one-rank RCCL rehearsal), for a kernel
trace: rocprofv3 --kernel-trace --stats -- python3 tools/trace_dist_vcycle.py single|dist [m=4095]."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd.dist_amg import RowBlockAMG
which = sys.argv[1] if len(sys.argv) > 1 else "dist"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4095
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
V = gla.VCycle
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
n = m * m
A = V._as_sparse(ei, ev, (n, n))
H = V.AMGHierarchy(A, levels=8, splitting=V.GridSplitting(m, m), coarse_solver="dense", min_coarse=500)
RowBlockAMG.REHEARSE_ONE_RANK = True
D = RowBlockAMG(H, overlap=True, agglomerate_below=524288) if which == "dist" else None
b = torch.rand(n, 1, dtype=torch.float64, device=dev)
x = torch.zeros_like(b)
fn = (lambda bb, xx: D.cycle(bb, xx, b_unchanged=True)) if which == "dist" else H.cycle
for _ in range(3):
    x = fn(b, x)
torch.cuda.synchronize()
a = torch.zeros(1 << 20, device=dev); a.add_(1.0); torch.cuda.synchronize()      # marker kernel: the 20 cycles after it are the ones to read
for _ in range(20):
    x = fn(b, x)
torch.cuda.synchronize()
