#!/usr/bin/env python3
"""bench.py -- JacobiGNN sweeps/s on the 2-D Laplacian (BASELINE.json metric), 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 4096] [--sweeps 20] [--no-cpu-baseline]

A "step" is one `JacobiGNN.forward(n_iters=20, ...)` call (BASELINE config 2: 20 sweeps, 4096 x 4096
Laplacian, fp64) through the drop-in nn.Module surface with all inputs resident in HBM: the per-call AoS
column split, the cached-plan lookup and the 20 fused CSR sweeps are inside the timed region; building the
graph / CSR plan (once per graph) is not.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): WEAK scaling.  Every rank owns a row block
of `n` grid lines x `n` vertices of the (n*N_gpus) x n grid Laplacian (at N=1 exactly config 2); after each
sweep the ranks exchange their boundary grid lines with one all-gather (gnn_applied_linear_algebra_amd/dist.py).
`value` = (rank-block sweeps done by all ranks) / wall time = N_gpus * sweeps/s of the coupled system.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes_jacobi(n, nnz):
    """12*nnz (val f64 + col i32) + 4*(n+1) (rowptr) + 32*n (read x, b, A_ii; write x)."""
    return 12 * nnz + 4 * (n + 1) + 32 * n


def cpu_baseline(n_grid, sweeps):
    """Reference formulation (oracle/reference_port.py: gather -> mul -> cat -> scatter_add, torch CPU) timed on
    the host cores: a bounded sample (`sweeps` sweeps) of the same 4096^2 workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import reference_port as rp
    ei, ev = rp.laplacianfun_torch(n_grid)
    n = n_grid * n_grid
    gen = torch.Generator().manual_seed(24601)
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64)
    x = torch.rand(n, 1, generator=gen, dtype=torch.float64)
    va = torch.cat([-4 * torch.ones(n, 1, dtype=torch.float64), b, x], 1)
    ea = torch.cat([ev, torch.zeros_like(ev)], 1)
    g = torch.tensor([0.7], dtype=torch.float64)
    rp.jacobi_forward(1, va, ei, ea, g)                        # warm-up
    t0 = time.perf_counter()
    rp.jacobi_forward(sweeps, va, ei, ea, g)
    dt = time.perf_counter() - t0
    # for scale: the same sweep with a plain CPU CSR SpMV (torch.sparse_csr, same threads),
    # so that the speed-up is not read off the reference's gather/concat overhead alone
    csr_value = None
    try:
        crow = torch.zeros(n + 1, dtype=torch.int64)
        crow[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
        A_csr = torch.sparse_csr_tensor(crow, ei[1], ev.flatten(), (n, n))
        xc = x.clone()
        xc = xc + 0.7 * (b - A_csr @ xc) / -4.0
        t1 = time.perf_counter()
        for _ in range(sweeps):
            xc = xc + 0.7 * (b - A_csr @ xc) / -4.0
        csr_value = sweeps / (time.perf_counter() - t1)
    except Exception:
        pass
    return {"value": sweeps / dt, "unit": "sweeps/s", "cores": torch.get_num_threads(), "kind": "port",
            "csr_spmv_sweeps_per_s": None if csr_value is None else round(csr_value, 3),
            "sample": f"{sweeps} sweeps of the {n_grid}x{n_grid} fp64 JacobiGNN in the reference's COO "
                      f"gather/cat/scatter_add formulation (oracle/reference_port.py), torch CPU, "
                      f"{torch.get_num_threads()} threads of {os.cpu_count()} logical cores",
            "ms_per_sweep": 1e3 * dt / sweeps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid lines per rank and vertices per line")
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=4)
    ap.add_argument("--no-overlap", action="store_true", help="multi-GPU: exchange, then compute (no overlap)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling instead of the default weak scaling: a FIXED (grid x grid) Laplacian "
                         "(BASELINE config 4: --grid 8192) is cut into N row blocks; reports scaling=strong")
    ap.add_argument("--no-optin", action="store_true", help="skip the side measurement of the opt-in packed stream")
    ap.add_argument("--no-dia", action="store_true",
                    help="keep the CSR kernels (compact index streams) instead of the diagonal layout for stencils")
    ap.add_argument("--no-col16", action="store_true",
                    help="read 4-byte columns (the plain CSR stream of SURVEY's byte model) instead of int16 "
                         "tile-relative columns")
    ap.add_argument("--compress", action="store_true",
                    help="opt-in packed matrix stream (int16 tile-relative columns + value dictionary, 3 B/nnz)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal of the N>1 path)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: all ranks share cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; before any HIP call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=dev)
            except TypeError:                                   # torch without the device_id argument
                dist.init_process_group(backend="nccl")
        else:
            dist.init_process_group(backend=args.backend)

    import gnn_applied_linear_algebra_amd as gla
    from gnn_applied_linear_algebra_amd import UtilsGNN
    if args.compress:
        gla._lib.check(gla._lib.lib().gla_set_option(b"compress", 1))
    if args.no_col16:
        gla._lib.check(gla._lib.lib().gla_set_option(b"col16", 0))
    if args.no_dia or args.no_col16:
        gla._lib.check(gla._lib.lib().gla_set_option(b"dia", 0))

    Nx = args.n
    if args.strong:
        if args.n % world:
            raise SystemExit("--strong needs --grid divisible by the number of GPUs")
        lines = args.n // world
    else:
        lines = args.n
    Ny = lines * world
    n_local = Nx * lines
    lo, hi = rank * n_local, (rank + 1) * n_local
    ei, ev = UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, row_range=(lo, hi), device=dev)
    nnz_local = ei.shape[1]
    gen = torch.Generator().manual_seed(24601 + rank)
    b = torch.rand(n_local, 1, generator=gen, dtype=torch.float64).to(dev)
    x0 = torch.rand(n_local, 1, generator=gen, dtype=torch.float64).to(dev)
    diag = torch.full((n_local, 1), -4.0, dtype=torch.float64, device=dev)
    vertex_attr = torch.cat([diag, b, x0], 1)
    edge_attr = torch.cat([ev, torch.zeros_like(ev)], 1)
    g = torch.tensor([0.7], dtype=torch.float64, device=dev)

    if world == 1:
        jac = gla.JacobiGNN.JacobiGNN()

        def step():
            return jac(args.sweeps, vertex_attr, ei, edge_attr, g)

        plan = gla.plan_for(ei, n_local, edge_attr=edge_attr)

        def kernel_only():
            return plan.jacobi_sweeps(diag.reshape(-1), b, x0, 0.7, args.sweeps)

        # HIP events around the sweep launches INSIDE the timed steps (recorded on the launch stream = torch's current
        # stream): the module calls plan.jacobi_sweeps, which is wrapped on this plan instance
        sweep_events = []
        _sweeps = plan.jacobi_sweeps

        def _timed_sweeps(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = _sweeps(*a, **k)
            e1.record()
            sweep_events.append((e0, e1))
            return out
        plan.jacobi_sweeps = _timed_sweeps
    else:
        from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
        drv = RowBlockJacobi(ei, edge_attr, n_global=Nx * Ny, row_lo=lo, row_hi=hi, overlap=not args.no_overlap)

        def step():
            return drv.forward(args.sweeps, vertex_attr, g)

        kernel_only = None
        sweep_events = []

    notes = []
    try:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
    except Exception as exc:                                    # keep the scaling run alive: serialise the exchange
        if world == 1 or args.no_overlap:
            raise
        notes.append(f"overlapped halo exchange failed ({type(exc).__name__}: {exc}); re-run without overlap")
        drv = RowBlockJacobi(ei, edge_attr, n_global=Nx * Ny, row_lo=lo, row_hi=hi, overlap=False)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    del sweep_events[:]                                      # keep only the events of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_sweeps = args.steps * args.sweeps
    ms_per_step = 1e3 * elapsed / args.steps
    value = (1 if args.strong else world) * total_sweeps / elapsed
    B = algorithmic_bytes_jacobi(n_local, nnz_local)

    # roofline of the dominant kernel (spmv_stream_vec2_kernel<double, EpiJacobi>): HIP events on the launch stream
    roof = None
    if rank == 0:
        if sweep_events:                                     # N = 1: events recorded inside the timed region
            t_launch = sum(e0.elapsed_time(e1) for e0, e1 in sweep_events) * 1e-3 / (len(sweep_events) * args.sweeps)
            plan.jacobi_sweeps = _sweeps
        elif kernel_only is not None:
            kernel_only()
            torch.cuda.synchronize()
            reps = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                kernel_only()
            e1.record()
            torch.cuda.synchronize()
            t_launch = e0.elapsed_time(e1) * 1e-3 / (reps * args.sweeps)
        else:
            t_launch = elapsed / total_sweeps                # per-rank sweep incl. halo exchange
        achieved = B / t_launch / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_jacobi.json")
        if world == 1 and os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if tj.get("n") == args.n and int(tj.get("col16", 0)) == int(plan.info.col16) \
                        and int(tj.get("dia", 0)) == int(plan.info.dia) and not bool(plan.info.packed):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "kernel": "spmv_stream_vec2_kernel<double,EpiJacobi>", "algorithmic_bytes_per_launch": B,
                "avg_launch_us": round(t_launch * 1e6, 2)}
        info = plan.info if world == 1 else None
        if info is not None and int(info.dia) > 0 and not bool(info.packed):
            # default layout for stencil matrices: values diagonal-major, NO index stream (a 2-byte row mask instead).
            # `achieved` / `frac` keep SURVEY 8(d)'s CSR byte model as prescribed; `layout_*` is what the kernel really
            # has to move: 8 D bytes of values + 2 per row + the vectors
            Bl = (8 * int(info.dia) + 2 + 32) * n_local
            roof.update({"kernel": "spmv_dia_kernel<double,EpiJacobi>", "layout": f"dia{int(info.dia)}",
                         "layout_bytes_per_launch": Bl, "layout_achieved": round(Bl / t_launch / 1e9, 1),
                         "layout_frac": round(Bl / t_launch / 8e12, 4)})
        elif info is not None and bool(info.col16) and not bool(info.packed):
            # default layout: 2-byte tile-relative columns.  `achieved` / `frac` keep SURVEY 8(d)'s CSR byte model
            # (12 B per nonzero, 4 B per row offset) as prescribed; `layout_*` is what the kernel really has to move
            # (10 B per nonzero, 2-byte tile-relative row offsets)
            Bl = 10 * nnz_local + 2 * (n_local + int(info.n_tiles)) + 32 * n_local
            roof.update({"layout_bytes_per_launch": Bl, "layout_achieved": round(Bl / t_launch / 1e9, 1),
                         "layout_frac": round(Bl / t_launch / 8e12, 4)})
        if args.compress and world == 1 and bool(plan.info.packed):
            # the packed layout moves 3 B per nonzero instead of the 12 B of SURVEY's CSR model: `achieved` keeps the
            # SURVEY byte model as prescribed (it can exceed what HBM delivers), `layout_*` is what is really moved
            Bl = 3 * nnz_local + 2 * (n_local + int(plan.info.n_tiles)) + 32 * n_local
            roof.update({"kernel": "spmv_packed_kernel<double,EpiJacobi>", "traffic": None,
                         "layout_bytes_per_launch": Bl, "layout_achieved": round(Bl / t_launch / 1e9, 1),
                         "layout_frac": round(Bl / t_launch / 8e12, 4)})

    # Side measurement (N = 1, default run only): the same step with the opt-in packed matrix stream
    # (gla_set_option("compress", 1): constant diagonals of a stencil are not streamed at all; otherwise int16 columns +
    # <= 256-entry value dictionary, 3 B per nonzero; bit-identical results).  Reported next to the headline, never as
    # `value`: it depends on the VALUES of the matrix (few distinct ones), not only on its structure.
    optin = None
    if rank == 0 and world == 1 and not args.compress and not args.no_col16 and not args.no_dia and not args.no_optin:
        check, L = gla._lib.check, gla._lib.lib()
        try:
            check(L.gla_set_option(b"compress", 1))
            ei_p = ei.clone()                                   # plans are cached per edge-index tensor: a fresh one
            plan_p = gla.plan_for(ei_p, n_local, edge_attr=edge_attr)
            if bool(plan_p.info.packed):
                ref_out = step()
                out_p = jac(args.sweeps, vertex_attr, ei_p, edge_attr, g)
                same = bool(torch.equal(ref_out, out_p))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    jac(args.sweeps, vertex_attr, ei_p, edge_attr, g)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                Bl = 3 * nnz_local + 2 * (n_local + int(plan_p.info.n_tiles)) + 32 * n_local
                mode = "packed CSR stream: int16 columns + value dictionary"
                if int(plan_p.info.packed) == 2:           # stencil with constant diagonals: only the row mask is left
                    Bl = (2 + 32) * n_local
                    mode = "constant-diagonal stencil: no matrix stream besides a 2-byte row mask"
                optin = {"option": "compress", "mode": mode, "value": round(total_sweeps / el, 2), "unit": "sweeps/s",
                         "us_per_sweep_whole_step": round(1e6 * el / total_sweeps, 2),
                         "bit_identical_to_default": same, "layout_bytes_per_launch": Bl}
            del plan_p, ei_p
        finally:
            check(L.gla_set_option(b"compress", 0))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.n, args.cpu_sweeps)
        cpu["value"] = round(cpu["value"], 4)
        cpu["ms_per_sweep"] = round(cpu["ms_per_sweep"], 1)

    if rank == 0:
        out = {
            "metric": "Jacobi-GNN sweeps/sec (2D Laplacian, fp64)", "value": round(value, 2), "unit": "sweeps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"JacobiGNN.forward, {args.sweeps} sweeps/step, 5-pt Laplacian, "
                                   f"{lines}x{Nx} vertices per GPU (global grid {Ny}x{Nx}), fp64, w=0.7"
                                   + ("" if world == 1 else ", row-block partition + RCCL halo all-gather"),
                       "N": Nx, "rows_per_gpu": n_local, "nnz_per_gpu": nnz_local, "sweeps_per_step": args.sweeps,
                       "achieved_GBps_per_gpu_whole_step": round(B * total_sweeps / elapsed / 1e9, 1)},
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if optin is not None:
            out["optin_packed_stream"] = optin
        if notes:
            out["notes"] = notes
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
