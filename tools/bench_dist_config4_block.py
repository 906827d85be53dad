#!/usr/bin/env python3
"""BASELINE config 4's REAL 8-rank block rehearsed on ONE GPU: one rank's share of the 8192 x 8192 Laplacian cut into 8 row
blocks is 1024 grid lines x 8192 points (8.4 M rows).  dist.RowBlockJacobi's temporally blocked forward (20 sweeps = passes
of 7 + 7 + 6, one ghost-line exchange per pass) runs on that block over a real one-rank RCCL communicator whose neighbours are
the rank itself (RowBlockJacobi.rehearsal_self_halo) -- everything a rank does per step except waiting for other GPUs --

  * eager: gla_dist_jacobi_temporal enqueues ~10 operations per pass from the host (89 us per pass, profiles/r03/
    dist_route_host_cost.txt) next to ~100 us of GPU time per pass at this block size: host-bound;
  * replayed from a HIP graph built by the library (RowBlockJacobi.capture_forward_native -> gla_dist_jacobi_temporal_graph: the
    pack launches, ncclAllGathers, interior / edge launches on two streams, the A_ii flag's memset node) -- bit-identical;

next to JacobiGNN.forward on the whole 8192 x 8192 matrix on this GPU.  bound_on_speedup = (8192^2 step) / (block step): what 8
GPUs can reach on config 4 before inter-GPU latency and load imbalance (north_star asks for >= 6).
Usage: python tools/bench_dist_config4_block.py [Nx=8192] [ranks=8] [sweeps=20] [--json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
as_json = "--json" in sys.argv
block_only = "--block-only" in sys.argv                        # (for kernel traces: no graph, no whole-matrix run)
Nx = int(args[0]) if len(args) > 0 else 8192
ranks = int(args[1]) if len(args) > 1 else 8
sweeps = int(args[2]) if len(args) > 2 else 20
keep = os.dup(1)
os.dup2(2, 1)                                                  # (RCCL's banner goes to stderr; stdout carries the report)
import gnn_applied_linear_algebra_amd as gla                   # noqa: E402
from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi  # noqa: E402

dev = torch.device("cuda:0")
lines = Nx // ranks


def host_and_gpu(fn, reps=30, warm=5):
    """median host seconds to enqueue one call (the stream is idle when it starts) and GPU seconds per call of a back-to-back run"""
    for _ in range(warm):
        fn()
    hs = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    hs.sort()
    return hs[len(hs) // 2], e0.elapsed_time(e1) * 1e-3 / reps, wall


gen = torch.Generator().manual_seed(24601)
n_blk = Nx * lines
va = torch.cat([torch.full((n_blk, 1), -4.0, dtype=torch.float64), torch.rand(n_blk, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
g = torch.tensor([0.7], dtype=torch.float64)                   # (a CPU tensor: no device-to-host read inside the capture)
def note(msg):
    print(f"bench_dist_config4_block: {msg}", file=sys.stderr, flush=True)


drv = RowBlockJacobi.rehearsal_self_halo(Nx, lines, dev)
comm = drv.comm
note("driver built")
sched = tuple(drv._sched(sweeps, torch.float64))
for _ in range(60):                                            # clocks (profiles/r04/clock_ramp.txt)
    drv.forward(sweeps, va, g)
torch.cuda.synchronize()
want = drv.forward(sweeps, va, g).clone()
h_e, g_e, w_e = host_and_gpu(lambda: drv.forward(sweeps, va, g))
note(f"eager timed: {w_e * 1e3:.3f} ms per step")
rec = {"what": f"one rank's block of BASELINE config 4 cut {ranks} ways ({lines} lines x {Nx} points = {n_blk} rows), {sweeps} sweeps, "
               f"schedule {list(sched)}, dist.RowBlockJacobi over a one-rank RCCL communicator (neighbours = the rank itself)",
       "eager_ms_per_step": round(w_e * 1e3, 4), "eager_host_us_per_step": round(h_e * 1e6, 1),
       "eager_host_us_per_pass": round(h_e * 1e6 / len(sched), 1)}
try:
    if block_only:
        raise RuntimeError("--block-only")
    run = drv.capture_forward_native(sweeps, va, g)
    note("captured")
    out = run(va)
    torch.cuda.synchronize()
    same = bool(torch.equal(out, want))
    # ... and against an eager call made AFTER the capture (the eager path must not trust what the capture left behind)
    same = same and bool(torch.equal(drv.forward(sweeps, va, g), want)) and bool(torch.equal(run(va), want))
    h_g, g_g, w_g = host_and_gpu(run.replay)
    h_r, g_r, w_r = host_and_gpu(lambda: run(va))
    rec.update({"graph_replay_ms_per_step": round(w_g * 1e3, 4), "graph_replay_host_us_per_step": round(h_g * 1e6, 1),
                "graph_run_incl_operand_copy_and_result_clone_ms": round(w_r * 1e3, 4), "graph_bit_identical_to_eager": same})
except Exception as exc:                                       # noqa: BLE001
    rec["graph_error"] = f"{type(exc).__name__}: {exc}"[:400]
    w_g = None
    run = None
note("graph timed" if w_g is not None else "no graph")
torch.cuda.synchronize()
del run, drv
torch.cuda.empty_cache()
# the whole matrix on this GPU through the module surface (what config4_1gpu of bench.py times)
free, _ = torch.cuda.mem_get_info()
if free > 40e9 and not block_only:
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, device=dev)
    n = Nx * Nx
    ea = torch.cat([ev, torch.zeros_like(ev)], 1)
    va1 = torch.cat([torch.full((n, 1), -4.0, dtype=torch.float64), torch.rand(n, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
    jac = gla.JacobiGNN.JacobiGNN()
    gd = g.to(dev)
    _, _, w_1 = host_and_gpu(lambda: jac(sweeps, va1, ei, ea, gd), reps=8, warm=3)
    rec["single_gpu_whole_matrix_ms_per_step"] = round(w_1 * 1e3, 4)
    best = min(w_e, w_g) if w_g is not None else w_e
    rec["bound_on_speedup_eager"] = round(w_1 / w_e, 3)
    if w_g is not None:
        rec["bound_on_speedup_graph"] = round(w_1 / w_g, 3)
    rec["bound_on_speedup"] = round(w_1 / best, 3)
    rec["bound_is"] = (f"(JacobiGNN.forward on the {Nx}^2 matrix on this GPU) / (one rank's block step, best of eager and graph replay): "
                       f"{ranks} ranks in parallel, before inter-GPU latency; {ranks} would be perfect")
torch.cuda.synchronize()
os.dup2(keep, 1)
if as_json:
    print(json.dumps(rec), flush=True)
else:
    for k, v in rec.items():
        print(f"{k}: {v}", flush=True)
gla._lib.lib().gla_comm_destroy(comm._h)                       # (before the interpreter tears the library down)
comm._h = None
