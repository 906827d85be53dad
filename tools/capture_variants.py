#!/usr/bin/env python3
"""Which form of RowBlockJacobi's temporally blocked forward survives HIP-graph capture?  (torch.cuda.graph around the default
route -- gla_dist_jacobi_temporal on two streams + the library's communicator stream -- died inside capture_end on ROCm 7.0 /
torch 2.10 in round 5.)  Every variant runs in a child process of its own (a crash is a result, not the end of the run):
  A  the default route
  B  A without the A_ii flag (option jacobi_diag_flag=0: no memset node, no flag kernels)
  C  the Python-driven overlapped route (torch side stream; e["c_route"] = False)
  D  the sequential route (overlap = False: no second stream)
  E  A with the second stream = the capture stream itself (no parallel kernel branches in the graph)
  F  B + E
  N  the library's own capture (gla_dist_jacobi_temporal_graph: hipStreamBeginCapture on a stream of its own, no torch)
Usage: python tools/capture_variants.py [Nx=8192] [lines=1024] [sweeps=20] [variants=ABCDEFN]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    var, Nx, lines, sweeps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if var in "BF":
        os.environ["GLA_OPTIONS"] = (os.environ.get("GLA_OPTIONS", "") + ",jacobi_diag_flag=0").strip(",")
    import faulthandler
    faulthandler.enable()
    keep = os.dup(1)
    os.dup2(2, 1)
    import torch
    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(24601)
    n_blk = Nx * lines
    va = torch.cat([torch.full((n_blk, 1), -4.0, dtype=torch.float64), torch.rand(n_blk, 2, generator=gen, dtype=torch.float64)], 1).to(dev)
    g = torch.tensor([0.7], dtype=torch.float64)
    drv = RowBlockJacobi.rehearsal_self_halo(Nx, lines, dev)
    want = drv.forward(sweeps, va, g).clone()
    if var == "C":
        drv.ext["c_route"] = False
    if var == "D":
        drv.overlap = False
    if var in "EF":
        class _Cur:
            cuda_stream = property(lambda s: torch.cuda.current_stream(dev).cuda_stream)
        drv.ext["side_stream"] = _Cur()
    ok_eager = bool(torch.equal(drv.forward(sweeps, va, g), want))

    def timed(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    for _ in range(40):
        drv.forward(sweeps, va, g)
    t_e = timed(lambda: drv.forward(sweeps, va, g))
    if var == "N":
        run = drv.capture_forward_native(sweeps, va, g)
        replay = run.replay
    else:
        run = drv.capture_forward(sweeps, va, g)
        replay = run.graph.replay
    same = bool(torch.equal(run(va), want))
    t_g = timed(replay)
    same = same and bool(torch.equal(run(va), want)) and bool(torch.equal(drv.forward(sweeps, va, g), want))
    os.dup2(keep, 1)
    print(f"variant {var}: eager {t_e * 1e3:.4f} ms (bit-identical to the default route: {ok_eager}), graph replay {t_g * 1e3:.4f} ms, "
          f"replayed result bit-identical: {same}", flush=True)
    torch.cuda.synchronize()
    del run
    gla._lib.lib().gla_comm_destroy(drv.comm._h)
    drv.comm._h = None
    sys.exit(0)

args = sys.argv[1:]
Nx = int(args[0]) if len(args) > 0 else 8192
lines = int(args[1]) if len(args) > 1 else 1024
sweeps = int(args[2]) if len(args) > 2 else 20
variants = args[3] if len(args) > 3 else "ABCDEFN"
print(f"{lines} lines x {Nx} points, {sweeps} sweeps, one-rank RCCL self-halo")
for var in variants:
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", var, str(Nx), str(lines), str(sweeps)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    if p.returncode == 0:
        print(p.stdout.strip(), flush=True)
    else:
        tail = [ln for ln in p.stderr.splitlines() if ln.strip()][-12:]
        print(f"variant {var}: child exited with {p.returncode}\n    " + "\n    ".join(tail), flush=True)
