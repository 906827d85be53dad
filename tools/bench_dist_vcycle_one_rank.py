#!/usr/bin/env python3
"""Per-rank cost of the distributed V-cycle (dist_amg.RowBlockAMG) next to the single-GPU cycle (VCycle.AMGHierarchy) on the
same rows: a process group of ONE rank on one GPU, every distributed level on the ghost-line route (extended-block plans,
persistent operand buffers, gla_jacobi_fused_residual / gla_spmm_add passes) with its ghost-line exchanges as real RCCL
all-gathers on libgla_hip's communicator -- the rank has no neighbour, so what is measured is everything a rank does per
cycle except waiting for other GPUs.  Usage: bench_dist_vcycle_one_rank.py [m=4095] [levels=8] [agglomerate_below=65536]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd import dist as gdist          # noqa: E402
from gnn_applied_linear_algebra_amd.dist_amg import RowBlockAMG   # noqa: E402

as_json = "--json" in sys.argv
argv = [a for a in sys.argv if not a.startswith("--")]
m = int(argv[1]) if len(argv) > 1 else 4095
levels = int(argv[2]) if len(argv) > 2 else 8
agg = int(argv[3]) if len(argv) > 3 else 1 << 16
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
keep = os.dup(1)
os.dup2(2, 1)                                           # (RCCL prints a banner on stdout)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
V = gla.VCycle
ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
n = m * m
A = V._as_sparse(ei, ev, (n, n))
H = V.AMGHierarchy(A, levels=levels, splitting=V.GridSplitting(m, m), coarse_solver="dense", min_coarse=500)
RowBlockAMG.REHEARSE_ONE_RANK = True
D = RowBlockAMG(H, overlap=True, agglomerate_below=agg)
ghost_lv = sum(1 for o in D.A_ops if o.ext is not None)
gen = torch.Generator().manual_seed(0)
b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)


def timed(fn, x0, reps=8):
    x = fn(b, x0)
    x = fn(b, x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        x = fn(b, x)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return x, ts[len(ts) // 2], ts[0]


xs = torch.zeros_like(b)
xd = torch.zeros_like(b)
same = True
for _ in range(3):
    xs = H.cycle(b, xs)
    xd = D.cycle(b, xd)
    same = same and bool(torch.equal(xs, xd))
e0 = gdist.EXCHANGES[0]
D.cycle(b, xd)
n_ex = gdist.EXCHANGES[0] - e0
res = []
for rnd in range(3):                                    # alternating, so that box drift hits both alike
    _, ms_s, mn_s = timed(H.cycle, torch.zeros_like(b))
    # (a solver loop on ONE right-hand side: the caller vouches that b is unchanged, its copy into the extended block is skipped)
    _, ms_d, mn_d = timed(lambda bb, xx: D.cycle(bb, xx, b_unchanged=True), torch.zeros_like(b))
    res.append((ms_s, ms_d, mn_s, mn_d))
run = D.capture_cycle()
xg, xe = torch.zeros_like(b), torch.zeros_like(b)
same_g = True
for _ in range(3):
    xg, xe = run(b, xg), D.cycle(b, xe)
    same_g = same_g and bool(torch.equal(xg, xe))
_, ms_g, mn_g = timed(run, torch.zeros_like(b))
os.dup2(keep, 1)
if as_json:
    import json
    med = sorted(res)[len(res) // 2]
    print(json.dumps({"mesh": f"{m}^2 Q1 FEM", "rows": n, "level_rows": [int(a.shape[0]) for a in H.A], "distributed_levels": D.n_dist,
                      "ghost_line_levels": ghost_lv, "agglomerate_below_rows_per_rank": agg, "collectives_per_cycle_this_rank": n_ex,
                      "bit_identical_to_single_gpu_cycle": same, "single_gpu_ms_per_cycle": round(med[0], 3),
                      "one_rank_distributed_ms_per_cycle": round(med[1], 3), "ratio": round(med[1] / med[0], 3),
                      "hip_graph_replay_ms_per_cycle": round(ms_g, 3), "hip_graph_bit_identical": same_g}))
    gdist.LibComm.release_all()
    dist.destroy_process_group()
    sys.exit(0)
print(f"m={m} n={n} levels={[int(a.shape[0]) for a in H.A]}")
print(f"distributed levels {D.n_dist} (ghost-line route on {ghost_lv}), replicated from level {D.n_dist} on; "
      f"collectives per cycle on this rank: {n_ex}; iterates bit-identical to AMGHierarchy.cycle: {same}")
for ms_s, ms_d, mn_s, mn_d in res:
    print(f"single-GPU cycle {ms_s:.3f} ms (min {mn_s:.3f})   distributed cycle, one rank {ms_d:.3f} ms (min {mn_d:.3f})   ratio {ms_d / ms_s:.3f}")
print(f"distributed cycle replayed from a HIP graph (incl. operand copies in / result clone out): {ms_g:.3f} ms (min {mn_g:.3f}); "
      f"bit-identical to the eager cycle: {same_g}")
gdist.LibComm.release_all()
dist.destroy_process_group()
