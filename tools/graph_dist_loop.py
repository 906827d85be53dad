#!/usr/bin/env python3
"""The exchange-per-sweep loop of the multi-GPU route (gla_dist_jacobi_sweeps: pack -> ncclAllGather on libgla_hip's side stream
-> interior tiles -> event wait -> boundary tiles, per sweep) captured into a HIP graph on a ONE-RANK RCCL communicator (the
self-halo block of tests/test_gpu_dist.py): host microseconds per sweep eager vs replayed, GPU time per sweep, and the replayed
result against the eager one bit for bit.  Usage: graph_dist_loop.py [Nx=1024] [Ny=1024] [sweeps=40]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd.dist import LibComm, RowBlockJacobi      # noqa: E402
from gnn_applied_linear_algebra_amd.plan import CsrPlan                      # noqa: E402

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Ny = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda:0")
n = Nx * Ny
ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, device=dev)
send_idx = torch.cat([torch.arange(0, 2 * Nx), torch.arange(n - 2 * Nx, n)]).to(dev)
S = int(send_idx.numel())
row, col = ei[0], ei[1]
redirect = ((row - col).abs() == Nx) & torch.isin(col, send_idx)
col2 = col.clone()
col2[redirect] = n + torch.searchsorted(send_idx, col[redirect])
plan = CsrPlan(torch.stack([row, col2]), n, n + S)
plan.set_values(ev)
plan.classify_tiles(n)
drv = RowBlockJacobi.__new__(RowBlockJacobi)
drv.plan, drv.overlap, drv._lib_send, drv.ext = plan, True, {}, None
drv.comm = LibComm()
drv.send_idx32 = send_idx.to(torch.int32)
drv.part = types.SimpleNamespace(world=1, n_local=n, n_in=n, n_send=S, S=S,
                                 new_buffer=lambda K_, dt, dv: torch.zeros((n + S, K_), dtype=dt, device=dv))
drv.ALL, drv.INTERIOR, drv.BOUNDARY = 0, 1, 2
gen = torch.Generator().manual_seed(1)
va = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64), torch.rand(n, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
g = torch.tensor([0.7], dtype=torch.float64)                    # (a CPU tensor: no device-to-host read inside the capture)


def host_and_gpu(fn, reps=5):
    hs, gs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        hs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        gs.append(e0.elapsed_time(e1) * 1e-3)
    hs.sort(); gs.sort()
    return hs[len(hs) // 2], gs[len(gs) // 2]


want = drv.forward(sweeps, va, g).clone()
h_e, g_e = host_and_gpu(lambda: drv.forward(sweeps, va, g))
print(f"{Nx} x {Ny} block, {sweeps} sweeps per call, one-rank RCCL communicator (self halo)")
print(f"eager:  host {h_e / sweeps * 1e6:7.2f} us per sweep to enqueue, GPU {g_e / sweeps * 1e6:7.2f} us per sweep")
try:
    run = drv.capture_forward(sweeps, va, g)
except Exception as exc:                                       # noqa: BLE001
    print(f"capture FAILED: {type(exc).__name__}: {str(exc)[:400]}")
    sys.exit(0)
out = run(va)
torch.cuda.synchronize()
same = bool(torch.equal(out, want))
graph = run.graph
h_g, g_g = host_and_gpu(graph.replay)
print(f"graph:  host {h_g / sweeps * 1e6:7.2f} us per sweep to replay,  GPU {g_g / sweeps * 1e6:7.2f} us per sweep; "
      f"replayed result bit-identical to the eager loop: {same}")
torch.cuda.synchronize()
gla._lib.lib().gla_comm_destroy(drv.comm._h)
drv.comm._h = None
