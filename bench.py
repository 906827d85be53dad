#!/usr/bin/env python3
"""bench.py -- JacobiGNN sweeps/s on the 2-D Laplacian (BASELINE.json metric), 1..8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 4096] [--sweeps 20] [--no-cpu-baseline] [--no-side]

A "step" is one `JacobiGNN.forward(n_iters=20, ...)` call (BASELINE config 2: 20 sweeps, 4096 x 4096
Laplacian, fp64) through the drop-in nn.Module surface with all inputs resident in HBM: the per-call AoS
column split, the cached-plan lookup and the 20 fused sweeps are inside the timed region; building the
graph / CSR plan (once per graph) is not.

`python bench.py --gpus N` works as typed: when no launcher has set WORLD_SIZE the process is only a PARENT -- it
imports neither torch nor the library, touches no GPU, starts N fresh rank processes (one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), relays rank 0's JSON line and
exits with the worst child status.  Launched by `torch.distributed.run` (WORLD_SIZE already set) it is a rank.

N > 1 (one rank per GPU, RCCL): the headline is WEAK scaling.  Every rank owns a row block of `n` grid lines x `n`
vertices of the (n*N_gpus) x n grid Laplacian (at N=1 exactly config 2); after each sweep the ranks exchange their
boundary grid lines with one all-gather (gnn_applied_linear_algebra_amd/dist.py).  `value` = (rank-block sweeps
done by all ranks) / wall time = N_gpus * sweeps/s of the coupled system.  The same line carries `config4_strong`:
BASELINE config 4 (FIXED 8192 x 8192 Laplacian cut into N row blocks) with its speed-up over the 1-GPU rate of the
same matrix measured in the same run.  A failure of the overlapped exchange is a failure of the run (non-zero exit).

N = 1 side records (never `value`): `one_sweep_per_pass` (the stencil kernel without temporal blocking), `csr_general`
(the CSR-stream kernel every non-stencil matrix gets),
`config3_k16` (BASELINE config 3: Chebyshev degree 4 + power method, 16 right-hand sides), `config4_1gpu`
(8192 x 8192 on one GPU), `optin_packed_stream`, `dist_route_one_gpu` (the N > 1 route rehearsed on this GPU with a one-rank
RCCL communicator).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK = 8e12          # B/s, MI355X_MICROARCH.md


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=4096, help="grid lines per rank and vertices per line")
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=4)
    ap.add_argument("--no-overlap", action="store_true", help="multi-GPU: exchange, then compute (no overlap)")
    ap.add_argument("--host-loop", action="store_true",
                    help="multi-GPU: drive every sweep from Python (torch.distributed all-gather) instead of the "
                         "library-side loop")
    ap.add_argument("--strong", action="store_true",
                    help="headline = strong scaling: a FIXED (grid x grid) Laplacian (BASELINE config 4: --grid 8192) "
                         "is cut into N row blocks; reports scaling=strong")
    ap.add_argument("--strong-grid", type=int, default=8192, help="grid of the config4_strong side record (N > 1)")
    ap.add_argument("--no-side", action="store_true", help="skip every side record (csr_general, config3_k16, ...)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE over tools/pmc_workload.py) that "
                         "measure roofline.traffic in this run; the figure of profiles/traffic_jacobi.json is then quoted, marked")
    ap.add_argument("--pmc-timeout", type=float, default=150.0, help="seconds per rocprofv3 counter pass")
    ap.add_argument("--no-optin", action="store_true", help="skip the side measurement of the opt-in packed stream")
    ap.add_argument("--no-dia", action="store_true",
                    help="keep the CSR kernels (compact index streams) instead of the diagonal layout for stencils")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="run the step untimed for this many milliseconds before the W warm-up steps (clock ramp of the card's "
                         "power management; 0: none)")
    ap.add_argument("--no-diag-flag", action="store_true",
                    help="gla_jacobi_sweeps_aos makes contiguous copies of A_ii and b as before round 4 (option jacobi_diag_flag=0) "
                         "instead of recognising that vertex_attr[:, 0] repeats the matrix diagonal")
    ap.add_argument("--no-col16", action="store_true",
                    help="read 4-byte columns (the plain CSR stream of SURVEY's byte model) instead of int16 "
                         "tile-relative columns")
    ap.add_argument("--compress", action="store_true",
                    help="opt-in packed matrix stream (int16 tile-relative columns + value dictionary, 3 B/nnz)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal of the N>1 path)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: all ranks share cuda:0 (needs --backend gloo)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="parent only: print the child command lines + environment as JSON and exit")
    ap.add_argument("--stack-dump-s", type=float, default=20.0,
                    help="N > 1: rank 0 dumps its Python stacks to stderr every so many seconds (0: never)")
    ap.add_argument("--child-timeout", type=float, default=1500.0, help="parent: seconds before the ranks are killed")
    return ap.parse_args(argv)


# =====================================================================================================================
# parent: start one fresh process per GPU.  No torch import, no HIP call in this process.
# =====================================================================================================================
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_specs(args, argv):
    port = int(os.environ.get("MASTER_PORT", 0)) or free_port()
    specs = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "GLA_BENCH_CHILD": "1"}
        if "OMP_NUM_THREADS" not in os.environ:
            # as torch.distributed.run does for several ranks per node: N ranks with a full set of intra-op CPU threads each
            # oversubscribe the host, and the setup-time host ops of the row-block drivers crawl (100 s instead of 2)
            env["OMP_NUM_THREADS"] = str(max(1, min(8, (os.cpu_count() or 8) // (2 * args.gpus))))
        specs.append({"cmd": [sys.executable, os.path.abspath(__file__)] + list(argv), "env": env})
    return specs


def launch(args, argv):
    """Spawn args.gpus rank processes, relay rank 0's stdout, return the worst exit status."""
    specs = child_specs(args, argv)
    if args.launch_dry_run:
        print(json.dumps({"children": specs, "torch_imported_in_parent": "torch" in sys.modules}))
        return 0
    procs = []
    for r, sp in enumerate(specs):
        env = dict(os.environ)
        env.update(sp["env"])
        procs.append(subprocess.Popen(sp["cmd"], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    deadline = time.time() + args.child_timeout
    rcs = [None] * len(procs)
    out0 = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = [rc for rc in rcs if rc not in (None, 0)]
        if failed or time.time() > deadline:              # one rank died (or the run hangs): stop the others by PID
            time.sleep(2.0)
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.terminate()
            for i, p in enumerate(procs):
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                rcs[i] = p.returncode
            if not failed:
                print("bench.py: ranks killed after --child-timeout", file=sys.stderr)
                rcs = [rc if rc else 124 for rc in rcs]
            break
        time.sleep(0.05)
    try:
        out0 = procs[0].stdout.read()
    except Exception:
        out0 = ""
    for line in (out0 or "").splitlines():                # stdout carries the JSON line only; anything else a library
        if line.lstrip().startswith("{"):                 # printed there (gloo's connection banner) goes to stderr
            sys.stdout.write(line + "\n")
        else:
            sys.stderr.write(line + "\n")
    sys.stdout.flush()
    worst = 0
    for rc in rcs:
        if rc:
            worst = rc if rc > 0 else 128 - rc
            break
    return worst


# =====================================================================================================================
# rank process
# =====================================================================================================================
def algorithmic_bytes_jacobi(n, nnz):
    """12*nnz (val f64 + col i32) + 4*(n+1) (rowptr) + 32*n (read x, b, A_ii; write x)."""
    return 12 * nnz + 4 * (n + 1) + 32 * n


def cpu_baseline(n_grid, sweeps):
    """Reference formulation (oracle/reference_port.py: gather -> mul -> cat -> scatter_add, torch CPU) timed on
    the host cores: a bounded sample (`sweeps` sweeps) of the same 4096^2 workload."""
    import torch
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
    except Exception:                                          # the optional extra baseline only
        pass
    return {"value": round(sweeps / dt, 4), "unit": "sweeps/s", "cores": torch.get_num_threads(), "kind": "port",
            "csr_spmv_sweeps_per_s": None if csr_value is None else round(csr_value, 3),
            "sample": f"{sweeps} sweeps of the {n_grid}x{n_grid} fp64 JacobiGNN in the reference's COO "
                      f"gather/cat/scatter_add formulation (oracle/reference_port.py), torch CPU, "
                      f"{torch.get_num_threads()} threads of {os.cpu_count()} logical cores",
            "ms_per_sweep": round(1e3 * dt / sweeps, 1)}


def measure_traffic(n_grid, timeout_s):
    """HBM bytes per launch of EVERY Jacobi pass of the module step from the PMC counters, measured NOW on this box: two child runs
    of `rocprofv3 --pmc <counter> -- python3 tools/pmc_workload.py N` (FETCH_SIZE, WRITE_SIZE: separate passes, KiB units,
    FETCH_SIZE x 2 on gfx950 -- MI355X_MICROARCH.md, HBM section; the factor is checked against the workload's calibration
    kernel of known volume), plus a third child run with --kernel-trace only (no counters) for the in-step duration of every
    pass instance.  Returns a dict (the dominant plain pass at top level, every instance under "instances"), or {"error": ...}.
    The children are started as ordinary subprocesses of this (GPU-initialised) process -- nothing is exec'ed in place -- and
    stopped by PID on a time-out."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="gla_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def kernel_key(name):
        """("jacobi", sweeps per launch, "plain" | "aos" | "res") of a Jacobi kernel name, ("calib", 0, "") of the calibration copy"""
        m = re.search(r"jacobi_march5_kernel<\w+,\s*(\d+),\s*(\d+)", name)
        if m:
            return ("jacobi", int(m.group(1)), ("plain", "aos", "res")[min(int(m.group(2)), 2)])
        m = re.search(r"jacobi_fused_dia2d_kernel<\w+,\s*\d+,\s*(\d+)(?:,\s*(\w+))?", name)
        if m:
            return ("jacobi", int(m.group(1)), "aos" if (m.group(2) or "") in ("true", "1") else "plain")
        if "EpiJacobi" in name:
            return ("jacobi", 1, "plain")
        if "CUDAFunctor_add" in name:
            return ("calib", 0, "")
        return None

    vals, inst = {}, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", None):
            d = os.path.join(tmp, ctr or "trace")
            cmd = [exe] + (["--pmc", ctr] if ctr else []) + ["--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), str(n_grid)]
            pr_ = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                rc = pr_.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                pr_.kill()
                pr_.wait()
                return {"error": f"rocprofv3 {'--pmc ' + ctr if ctr else '--kernel-trace'} pass timed out after {timeout_s:.0f} s"}
            if rc != 0:
                return {"error": f"rocprofv3 {'--pmc ' + ctr if ctr else '--kernel-trace'} pass exited with {rc}"}
            if ctr is None:                                  # durations of every pass instance inside the step (no counters attached)
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        key = kernel_key(r["Kernel_Name"])
                        if key and key[0] == "jacobi":
                            inst.setdefault(key, {}).setdefault("us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
                continue
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
            cover, per = {}, {}
            for r in rows:
                key = kernel_key(r["Kernel_Name"])
                if key:
                    per.setdefault(key, []).append(float(r["Counter_Value"]))
                    if key[0] == "jacobi" and key[2] == "plain":      # (the dominant kernel is picked among the plain passes)
                        cover[key] = cover.get(key, 0) + key[1]
            if not cover or ("calib", 0, "") not in per:
                return {"error": f"no Jacobi / calibration kernel rows in the {ctr} pass"}
            dom = max(cover, key=lambda k: (cover[k], k[1]))
            calib = sum(per[("calib", 0, "")]) / len(per[("calib", 0, "")])
            vals[ctr] = (dom, calib, len(per[dom]))
            for key, v in per.items():
                if key[0] == "jacobi":
                    inst.setdefault(key, {})[ctr] = sum(v) / len(v)
        dom_f, calib_f, nl = vals["FETCH_SIZE"]
        dom_w, calib_w, _ = vals["WRITE_SIZE"]
        if dom_f != dom_w:
            return {"error": "the two passes disagree on the dominant kernel"}
        fetch_factor = 2 * 512 * 1024 / calib_f          # the calibration kernel reads 2 x 512 MiB, writes 512 MiB
        write_factor = 512 * 1024 / calib_w
        if not (1.8 < fetch_factor < 2.2 and 0.9 < write_factor < 1.1):
            return {"error": f"calibration kernel off: fetch x{fetch_factor:.3f}, write x{write_factor:.3f}"}
        instances = []
        for key in sorted(inst, key=lambda k: (k[2] != "aos", -k[1])):
            v = inst[key]
            if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                continue
            rd_, wr_ = v["FETCH_SIZE"] * 1024 * 2.0, v["WRITE_SIZE"] * 1024
            us = sorted(v.get("us", []))
            instances.append({"sweeps_per_launch": key[1], "kind": key[2], "hbm_read_bytes": int(rd_), "hbm_write_bytes": int(wr_),
                              "hbm_bytes": int(rd_ + wr_), "in_step_us_kernel_trace": round(sum(us) / len(us), 2) if us else None,
                              "launches_traced": len(us)})
        d_ = inst[dom_f]
        rd, wr = d_["FETCH_SIZE"] * 1024 * 2.0, d_["WRITE_SIZE"] * 1024
        return {"sweeps_per_launch": dom_f[1], "launches_averaged": nl, "hbm_read_bytes_per_launch": int(rd),
                "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
                "calibration_fetch_factor_measured": round(fetch_factor, 4),
                "calibration_write_factor_measured": round(write_factor, 4), "instances": instances}
    except Exception as exc:                                 # (a measurement aid: never the reason a bench run fails)
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pass_row_bytes(D, n_sym, no_mask, kind, a_ii_streamed, flagged=True):
    """Bytes ONE ROW moves in one pass of the stencil Jacobi kernels with the layout the plan really has: 8 per diagonal that is
    read (D - n_sym: a symmetric value set reads its lower diagonals from their upper partners), 2 of row mask unless the plan
    is a full stencil (no_mask), and
      plain pass : b, x in, x out (+ A_ii unless it is taken from the centre coefficient)
      AoS pass   : the 24-byte AoS row [A_ii, b, x] in; x out and the contiguous copy of b (flag flow) / of A_ii and b out."""
    mat = 8 * (D - n_sym) + (0 if no_mask else 2)
    if kind == "aos":
        return mat + 24 + 8 + (8 if flagged else 16)
    return mat + 24 + (8 if a_ii_streamed else 0)


def step_layout(info, sched, dg_center, aos_first):
    """[(kind, sweeps, bytes per row)] of the passes behind one JacobiGNN.forward on a canonical 5-point march plan"""
    D = int(info.dia)
    march5 = int(info.dia_canon5) == 1
    n_sym = bin(int(info.dia_sym)).count("1") if march5 else 0
    no_mask = bool(n_sym and int(getattr(info, "dia_full", 0)))
    out = []
    for i, ts in enumerate(sched):
        kind = "aos" if (i == 0 and aos_first) else "plain"
        out.append((kind, ts, pass_row_bytes(D, n_sym, no_mask, kind, not dg_center, flagged=dg_center)))
    return out


def median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) % 2 else 0.5 * (xs[m - 1] + xs[m])


class Problem:
    """Synthetic inputs of SURVEY 8(d) for one rank's row block of the (Ny x Nx) grid Laplacian."""

    def __init__(self, gla, Nx, Ny, lo, hi, dev, seed):
        import torch
        from gnn_applied_linear_algebra_amd import UtilsGNN
        self.n_local = hi - lo
        self.ei, ev = UtilsGNN.laplacianfun_torch(Nx, Ny=Ny, row_range=(lo, hi), device=dev)
        self.nnz = self.ei.shape[1]
        gen = torch.Generator().manual_seed(seed)
        self.b = torch.rand(self.n_local, 1, generator=gen, dtype=torch.float64).to(dev)
        self.x0 = torch.rand(self.n_local, 1, generator=gen, dtype=torch.float64).to(dev)
        self.diag = torch.full((self.n_local, 1), -4.0, dtype=torch.float64, device=dev)
        self.vertex_attr = torch.cat([self.diag, self.b, self.x0], 1)
        self.edge_attr = torch.cat([ev, torch.zeros_like(ev)], 1)
        self.g = torch.tensor([0.7], dtype=torch.float64, device=dev)
        self.B = algorithmic_bytes_jacobi(self.n_local, self.nnz)


def timed_steps(step, steps, warmup, barrier=None):
    """W untimed steps, then exactly `steps` steps bracketed by barrier + synchronize; also per-step HIP-event times
    (recorded on the launch stream).  Returns (elapsed_s, [ms per step])."""
    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if barrier:
        barrier()
        torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if barrier:
        barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def single_gpu(args, gla, dev):
    import torch
    check, L = gla._lib.check, gla._lib.lib()
    Nx = args.n
    pr = Problem(gla, Nx, Nx, 0, Nx * Nx, dev, 24601)
    jac = gla.JacobiGNN.JacobiGNN()

    def step():
        return jac(args.sweeps, pr.vertex_attr, pr.ei, pr.edge_attr, pr.g)

    plan = gla.plan_for(pr.ei, pr.n_local, edge_attr=pr.edge_attr)
    # HIP events around the sweep launches INSIDE the timed steps (recorded on the launch stream = torch's current
    # stream): the module calls plan.jacobi_sweeps_aos / plan.jacobi_sweeps, which are wrapped on this plan instance
    sweep_events = []
    _sweeps, _sweeps_aos = plan.jacobi_sweeps, plan.jacobi_sweeps_aos

    def _timed(fn):
        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            if out is not None:                               # (the AoS entry returns None when it does not apply)
                sweep_events.append((e0, e1))
            return out
        return wrapped
    plan.jacobi_sweeps, plan.jacobi_sweeps_aos = _timed(_sweeps), _timed(_sweeps_aos)
    # Clock ramp: the card's power management takes ~100 ms of sustained load to bring the memory / fabric clocks up (the same
    # kernel is ~10 % slower in the first tens of milliseconds of a process: profiles/r04/clock_ramp.txt).  W = 5 warm-up steps
    # are 3.5 ms.  The step is therefore run untimed for --ramp-ms of wall time first (exactly what a production loop sees
    # after its first moments); the W warm-up steps and the K timed steps of the contract follow unchanged.
    no_ramp = None
    if args.ramp_ms > 0:
        # ... and the contract's protocol WITHOUT it first (W warm-up + K timed steps straight after the plan was built: what
        # BENCH_r01..r03 measured), reported beside the headline as `no_clock_ramp`
        el0, ms0 = timed_steps(step, args.steps, args.warmup)
        no_ramp = {"what": f"the same {args.steps} steps after {args.warmup} warm-up steps and NO clock ramp (the protocol of rounds 1-3), "
                           "measured in this process before the ramp",
                   "value": round(args.steps * args.sweeps / el0, 2), "unit": "sweeps/s", "ms_per_step": round(1e3 * el0 / args.steps, 4),
                   "ms_per_step_median": round(median(ms0), 4)}
        t_end = time.perf_counter() + args.ramp_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    del sweep_events[:]                                      # keep only the events of the timed steps
    elapsed, step_ms = timed_steps(step, args.steps, 0)
    plan.jacobi_sweeps, plan.jacobi_sweeps_aos = _sweeps, _sweeps_aos
    total_sweeps = args.steps * args.sweeps
    # launches behind one step: on 2-D stencil plans gla_jacobi_sweeps takes several sweeps in one pass over the data
    # (temporal blocking; 20 sweeps = passes of 7 + 7 + 6 on the 5-point Laplacian); the dominant kernel is the one that
    # covers most sweeps
    sched = plan.jacobi_schedule(args.sweeps)
    launches = len(sched)
    spl = max(set(sched), key=lambda ts: (ts * sched.count(ts), ts))      # sweeps per launch of the dominant kernel
    per_call = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in sweep_events]      # seconds per jacobi_sweeps call
    # (a) inside the timed steps the events bracket a whole library call (its launches are enqueued by one call); a launch's
    #     share is taken proportional to its sweeps: `in_step_bracket_us`
    t_bracket = sum(t * spl / args.sweeps for t in per_call) / len(per_call)
    # (b) the dominant kernel ALONE: one launch of `spl` sweeps (gla_jacobi_fused / gla_jacobi_sweep), HIP events on the launch
    #     stream around every single launch, 3 warm-up + 15 timed -- this is the duration `roofline` is built on
    from gnn_applied_linear_algebra_amd._lib import ptr, dtype_code, stream_ptr
    diag1, xo_buf = pr.diag.reshape(-1).contiguous(), torch.empty_like(pr.x0)
    # In the timed steps (gla_jacobi_sweeps_aos on a 5-point march plan) the passes after the first take A_ii from the matrix's
    # own centre coefficient when the first pass found vertex_attr[:, 0] to repeat the diagonal bit for bit -- no A_ii stream.
    # The kernel is timed alone in THAT configuration (diag = NULL: "A_ii is the plan's centre diagonal")
    info0 = plan.info
    on_d = pr.ei[0] == pr.ei[1]
    centre = torch.zeros(pr.n_local, dtype=torch.float64, device=dev)
    centre[pr.ei[0][on_d]] = pr.edge_attr[on_d, 0]
    dg_center = bool(int(info0.dia_canon5) == 1 and spl > 1 and plan.jacobi_schedule(3) == (3,) and min(sched) >= 2 and len(sched) >= 2
                     and not args.no_diag_flag and torch.equal(centre, diag1))
    del centre, on_d
    alone = []
    n_warm_alone = 3 + (60 if args.ramp_ms > 0 else 0)       # (the index computations above let the clocks drop again)
    for it in range(n_warm_alone + 15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if spl > 1:
            check(L.gla_jacobi_fused(plan._h, dtype_code(torch.float64), None if dg_center else ptr(diag1), ptr(pr.b), ptr(pr.x0),
                                     ptr(xo_buf), 0.7, spl, 0, stream_ptr(dev)))
        else:
            plan.jacobi_sweep(diag1, pr.b, pr.x0, xo_buf, 0.7)
        e1.record()
        if it >= n_warm_alone - 3:
            torch.cuda.synchronize()
        if it >= n_warm_alone:
            alone.append(e0.elapsed_time(e1) * 1e-3)
    t_launch, t_launch_med = sum(alone) / len(alone), median(alone)
    info = plan.info
    B_launch = pr.B * spl                                    # per-sweep bytes x the sweeps of one launch
    fused = spl > 1
    march = fused and bool(info.dia_canon5) and plan.jacobi_schedule(3) == (3,)      # (3 per pass: march kernel only)
    kname = ((f"jacobi_march{5 if int(info.dia_canon5) == 1 else 9}_kernel<double,{spl}>" if march else
              f"jacobi_fused_dia2d_kernel<double,{5 if int(info.dia) <= 5 else 9},{spl}>") if fused else
             ("spmv_dia_kernel<double,EpiJacobi>" if int(info.dia) > 0 and not bool(info.packed) else
              "spmv_stream_vec2_kernel<double,EpiJacobi>"))
    # bytes one launch has to move with this plan's layout (one pass over the matrix whatever the sweeps it carries)
    # (symmetric value sets, option "sym": the march kernel reads the lower diagonals from their upper partners -- 3 of a 5-point
    # matrix's 5 diagonals cross HBM)
    n_sym = bin(int(info.dia_sym)).count("1") if (march and int(info.dia_canon5) == 1) else 0
    no_mask = bool(n_sym and int(getattr(info, "dia_full", 0)))          # (every row has the mask of its grid position: not read)
    row_bytes = 8 * (int(info.dia) - n_sym) + (0 if no_mask else 2) + 32 - (8 if dg_center else 0)
    if int(info.dia) > 0 and not bool(info.packed):
        layout_name = (f"dia{int(info.dia)}" + (f", symmetric: {int(info.dia) - n_sym} diagonals read" if n_sym else "")
                       + (", A_ii = the centre coefficient (no A_ii stream)" if dg_center else "")
                       + (", no row masks (every row has its grid neighbours)" if no_mask else "")
                       + (f", {spl} sweeps per pass" if fused else ""))
        Bl = row_bytes * pr.n_local
    elif args.compress and bool(info.packed):
        layout_name, Bl = "packed stream", 3 * pr.nnz + 2 * (pr.n_local + int(info.n_tiles)) + 32 * pr.n_local
        kname = "spmv_packed_kernel<double,EpiJacobi>"
    elif bool(info.col16):
        layout_name, Bl = "csr, int16 tile-relative columns", 10 * pr.nnz + 2 * (pr.n_local + int(info.n_tiles)) + 32 * pr.n_local
    else:
        layout_name, Bl = "csr int32", pr.B
    # roofline.traffic: HBM bytes per launch by the PMC counters -- measured in THIS run when rocprofv3 is there (two child
    # passes over tools/pmc_workload.py), else the tracked figure of an earlier run of the same kernel and layout, marked so
    traffic, traffic_source, pmc = None, None, None
    default_layout_ = not (args.compress or args.no_col16 or args.no_dia)
    if default_layout_ and not args.no_pmc and not args.no_side:
        pmc = measure_traffic(args.n, args.pmc_timeout)
        if "error" not in pmc and pmc["sweeps_per_launch"] == spl:
            traffic = pmc["hbm_bytes_per_launch"]
            traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes over "
                              "tools/pmc_workload.py, FETCH_SIZE x 2 per the gfx950 note, calibration kernel checked)")
    tp = os.path.join(ROOT, "profiles", "traffic_jacobi.json")
    if traffic is None and os.path.exists(tp):
        tj = json.load(open(tp))
        if tj.get("n") == args.n and int(tj.get("dia", 0)) == int(info.dia) and int(tj.get("dia_sym", 0)) == n_sym and bool(tj.get("a_ii_from_centre", False)) == dg_center and bool(tj.get("no_mask", False)) == no_mask and not bool(info.packed) \
                and int(tj.get("sweeps_per_launch", 1)) == spl and (fused or int(tj.get("col16", 0)) == int(info.col16)):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/traffic_jacobi.json: PMC passes of an EARLIER run of this kernel and layout, not re-measured now"
    moved = traffic if traffic is not None else Bl
    roof = {"bound": "hbm", "achieved": round(moved / t_launch / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(moved / t_launch / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_source,
            "bytes_moved_per_launch": moved,
            "bytes_moved_is": "PMC traffic" if traffic is not None else f"layout bytes of one pass ({layout_name})",
            "unit_of_a_launch": "one pass over the matrix" + (f" carrying {spl} sweeps (temporal blocking)" if fused else " = one sweep"),
            "kernel": kname, "sweeps_per_launch": spl, "launches_per_step": launches, "schedule": list(sched),
            "avg_launch_us": round(t_launch * 1e6, 2), "median_launch_us": round(t_launch_med * 1e6, 2),
            "launch_timing": f"HIP events on the launch stream around single launches of this kernel alone, 15 after {n_warm_alone} warm-up",
            "in_step_bracket_us": round(t_bracket * 1e6, 2),
            "layout": layout_name, "layout_bytes_per_launch": Bl, "layout_frac": round(Bl / t_launch / HBM_PEAK, 4),
            # SURVEY 8(d)'s CSR byte model: per pass (the unit a launch processes) and per sweep x the sweeps of a launch.  The
            # latter exceeds what HBM delivers once several sweeps share a pass: a ratio, not a rate
            "algorithmic_bytes_per_sweep": pr.B, "pass_model_frac": round(pr.B / t_launch / HBM_PEAK, 4),
            "algorithmic_bytes_per_launch": B_launch, "model_ratio": round(B_launch / t_launch / HBM_PEAK, 4)}
    if pmc is not None:
        roof["pmc"] = pmc
    if traffic is not None:
        roof["traffic_over_layout"] = round(traffic / Bl, 4)
    # The WHOLE STEP: every pass behind one JacobiGNN.forward with its own bytes and time -- the dominant kernel above is the plain
    # 7-sweep pass, but the pass with the largest share of the step is the AoS first pass (it reads the 24-byte AoS rows and writes
    # the contiguous copy of b: 64 instead of 48 bytes per row).  bytes = PMC traffic of each instance where measured in this run
    # (else the layout's bytes), time = the contract's ms_per_step; per pass: the instance's average duration inside the step from
    # the --kernel-trace child pass.
    ms_step = 1e3 * elapsed / args.steps
    step_bytes, step_bytes_is = moved * launches, "dominant kernel's bytes x launches (no per-pass figures)"
    if fused and int(info.dia) > 0 and not bool(info.packed):
        lay = step_layout(info, sched, dg_center, aos_first=True)
        by = {(i_["kind"], i_["sweeps_per_launch"]): i_ for i_ in (pmc or {}).get("instances", [])} if pmc and "error" not in pmc else {}
        passes_rec, tot_pmc, tot_lay, all_pmc = [], 0, 0, True
        for kind, ts_, rb in lay:
            i_ = by.get((kind, ts_))
            rec_ = {"kernel": (f"jacobi_march{5 if int(info.dia_canon5) == 1 else 9}_kernel" if march else "jacobi_fused_dia2d_kernel")
                              + f"<double,{ts_}>" + (" AoS first pass (MODE 1)" if kind == "aos" else ""),
                    "sweeps": ts_, "layout_bytes_per_row": rb, "layout_bytes": rb * pr.n_local}
            tot_lay += rb * pr.n_local
            if i_ is not None:
                rec_.update({"hbm_bytes": i_["hbm_bytes"], "hbm_read_bytes": i_["hbm_read_bytes"], "hbm_write_bytes": i_["hbm_write_bytes"],
                             "in_step_us": i_["in_step_us_kernel_trace"],
                             "frac": None if not i_["in_step_us_kernel_trace"] else round(i_["hbm_bytes"] / (i_["in_step_us_kernel_trace"] * 1e-6) / HBM_PEAK, 4)})
                tot_pmc += i_["hbm_bytes"]
            else:
                all_pmc = False
            passes_rec.append(rec_)
        if all_pmc and passes_rec:
            step_bytes, step_bytes_is = tot_pmc, "PMC traffic of every pass of the step, measured in this run (rocprofv3 --pmc child passes)"
        else:
            step_bytes, step_bytes_is = tot_lay, "layout bytes of every pass of the step"
        roof["whole_step"] = {"bytes": int(step_bytes), "bytes_is": step_bytes_is, "layout_bytes": int(tot_lay), "ms_per_step": round(ms_step, 4),
                              "achieved": round(step_bytes / (ms_step * 1e-3) / 1e9, 1), "unit": "GB/s",
                              "frac": round(step_bytes / (ms_step * 1e-3) / HBM_PEAK, 4),
                              "in_step_us_is": "average duration of the instance over the launches of a rocprofv3 --kernel-trace child run of "
                                               "the same step (no counters attached)",
                              "passes": passes_rec}

    out = {
        "metric": "Jacobi-GNN sweeps/sec (2D Laplacian, fp64)", "value": round(total_sweeps / elapsed, 2),
        "unit": "sweeps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_step_median": round(median(step_ms), 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"JacobiGNN.forward, {args.sweeps} sweeps/step, 5-pt Laplacian, "
                               f"{Nx}x{Nx} vertices per GPU (global grid {Nx}x{Nx}), fp64, w=0.7",
                   "N": Nx, "rows_per_gpu": pr.n_local, "nnz_per_gpu": pr.nnz, "sweeps_per_step": args.sweeps,
                   "clock_ramp_ms": args.ramp_ms,
                   "bytes_moved_GBps_whole_step": round(step_bytes * args.steps / elapsed / 1e9, 1),
                   "bytes_moved_whole_step_is": step_bytes_is,
                   "model_equivalent_GBps_whole_step": round(pr.B * total_sweeps / elapsed / 1e9, 1)},
        "roofline": roof,
    }
    if no_ramp is not None:
        out["value_protocol"] = (f"{args.ramp_ms:.0f} ms of untimed steps (clock ramp of the card's power management), then the contract's "
                                 f"{args.warmup} warm-up + {args.steps} timed steps; `no_clock_ramp` = the same without the ramp")
        out["no_clock_ramp"] = no_ramp
    default_layout = not (args.compress or args.no_col16 or args.no_dia)
    side = default_layout and not args.no_side
    # ---- side: setup in front of the step ("COO -> CSR and AoS <-> SoA reported separately") ------------------
    if side:
        def timed_ms(fn, reps=3):
            ts_ = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                r_ = fn()
                torch.cuda.synchronize()
                ts_.append(1e3 * (time.perf_counter() - t0_))
                del r_
            return median(ts_)
        ei_s = pr.ei.clone()
        plans_ = []
        t_plan = timed_ms(lambda: plans_.append(gla.CsrPlan(ei_s, pr.n_local)) or 0)
        t_vals = timed_ms(lambda: plans_[-1].set_values(pr.edge_attr))
        t_graph = timed_ms(lambda: gla.UtilsGNN.laplacianfun_torch(Nx, device=dev))
        del plans_, ei_s
        out["setup"] = {
            "what": "once per graph, outside the timed step (the reference re-discovers the rows inside every scatter call and re-coalesces "
                    "per smoother call, VCycle.py:63,141,157); host wall time incl. a device synchronise, median of 3",
            "graph_builder_ms": round(t_graph, 3), "coo_to_plan_ms": round(t_plan, 3), "set_values_ms": round(t_vals, 3),
            "aos_to_soa": "no separate pass: the first pass of gla_jacobi_sweeps_aos reads the AoS rows itself (its 16 extra bytes per row "
                          "are in roofline.whole_step.passes[0])",
            "steps_to_amortise_plan_and_values": round((t_plan + t_vals) / ms_step, 1)}
        torch.cuda.empty_cache()

    # ---- side: the opt-in packed matrix stream (value-dependent; bit-identical) -------------------------------------
    if side and not args.no_optin:
        try:
            check(L.gla_set_option(b"compress", 1))
            ei_p = pr.ei.clone()                              # plans are cached per edge-index tensor: a fresh one
            plan_p = gla.plan_for(ei_p, pr.n_local, edge_attr=pr.edge_attr)
            if bool(plan_p.info.packed):
                same = bool(torch.equal(step(), jac(args.sweeps, pr.vertex_attr, ei_p, pr.edge_attr, pr.g)))
                el, _ = timed_steps(lambda: jac(args.sweeps, pr.vertex_attr, ei_p, pr.edge_attr, pr.g), args.steps, 1)
                Bl = 3 * pr.nnz + 2 * (pr.n_local + int(plan_p.info.n_tiles)) + 32 * pr.n_local
                mode = "packed CSR stream: int16 columns + value dictionary"
                if int(plan_p.info.packed) == 2:            # stencil with constant diagonals: only the row mask is left
                    Bl = (2 + 32) * pr.n_local
                    mode = "constant-diagonal stencil: no matrix stream besides a 2-byte row mask"
                out["optin_packed_stream"] = {
                    "option": "compress", "mode": mode, "value": round(total_sweeps / el, 2), "unit": "sweeps/s",
                    "us_per_sweep_whole_step": round(1e6 * el / total_sweeps, 2), "bit_identical_to_default": same,
                    "layout_bytes_per_launch": Bl}
            del plan_p, ei_p
        finally:
            check(L.gla_set_option(b"compress", 0))

    # ---- side: the stencil kernel with ONE sweep per pass (round 1's headline kernel; what a caller gets who needs
    #      every iterate, and what the multi-GPU path runs between halo exchanges) ---------------------------------
    if side and fused:
        try:
            check(L.gla_set_option(b"jacobi_fuse", 0))
            same = bool(torch.equal(step(), plan.jacobi_sweeps(pr.diag.reshape(-1), pr.b, pr.x0, 0.7, args.sweeps)))
            el, _ = timed_steps(step, args.steps, 1)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.jacobi_sweeps(pr.diag.reshape(-1), pr.b, pr.x0, 0.7, args.sweeps)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3 / args.sweeps)
            tl = median(ts)
            n_sym1 = bin(int(info.dia_sym)).count("1")     # spmv_dia_kernel reads the lower diagonals of a symmetric value set from their partners
            rb1 = 8 * (int(info.dia) - n_sym1) + 2 + 32
            Bl = rb1 * pr.n_local
            out["one_sweep_per_pass"] = {
                "what": "same step with option jacobi_fuse=0: spmv_dia_kernel<double,EpiJacobi>, one launch per sweep",
                "value": round(total_sweeps / el, 2), "unit": "sweeps/s", "launch_us": round(tl * 1e6, 2),
                "achieved": round(Bl / tl / 1e9, 1), "frac": round(Bl / tl / HBM_PEAK, 4),
                "bytes_moved_is": f"layout bytes of one pass ({rb1} B per row: {int(info.dia) - n_sym1} diagonals read, mask, A_ii, b, x in, x out)",
                "layout_bytes_per_launch": Bl,
                "pass_model_frac": round(pr.B / tl / HBM_PEAK, 4), "bit_identical_to_default": same}
        finally:
            check(L.gla_set_option(b"jacobi_fuse", 1))

    # ---- side: the same step on ALL FIVE diagonals (option sym = 0): what a matrix whose value set is not symmetric gets --------
    if side and n_sym:
        try:
            check(L.gla_set_option(b"sym", 0))
            ref_out = step()                                  # (sym is read at launch: this is the five-diagonal kernel)
            check(L.gla_set_option(b"sym", 1))
            same = bool(torch.equal(step(), ref_out))
            check(L.gla_set_option(b"sym", 0))
            el, _ = timed_steps(step, args.steps, 1)
            ts = []
            for it in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(L.gla_jacobi_fused(plan._h, dtype_code(torch.float64), None if dg_center else ptr(diag1), ptr(pr.b), ptr(pr.x0),
                                         ptr(xo_buf), 0.7, spl, 0, stream_ptr(dev)))
                e1.record()
                torch.cuda.synchronize()
                if it >= 2:
                    ts.append(e0.elapsed_time(e1) * 1e-3)
            tl = median(ts)
            Bg = (8 * int(info.dia) + 2 + 32 - (8 if dg_center else 0)) * pr.n_local
            out["unsymmetric_layout"] = {
                "what": "same step with option sym=0: every diagonal is read (what a plan whose VALUES are not symmetric bit for "
                        "bit runs on; the default recognises A_ij == A_ji at gla_csr_set_values and reads 3 of the 5 diagonals)",
                "value": round(total_sweeps / el, 2), "unit": "sweeps/s", "launch_us": round(tl * 1e6, 2),
                "layout_bytes_per_launch": Bg, "achieved": round(Bg / tl / 1e9, 1), "frac": round(Bg / tl / HBM_PEAK, 4),
                "bytes_moved_is": f"layout bytes of one pass ({8 * int(info.dia) + 34 - (8 if dg_center else 0)} B per row)",
                "bit_identical_to_default": same}
        finally:
            check(L.gla_set_option(b"sym", 1))

    # ---- side: the same step with contiguous A_ii / b copies made by the first pass (round 3's flow; what a caller whose
    #      vertex_attr[:, 0] is NOT the matrix diagonal pays, less the strided A_ii reads) ------------------------------------
    if side and dg_center:
        try:
            check(L.gla_set_option(b"jacobi_diag_flag", 0))
            same = bool(torch.equal(step(), jac(args.sweeps, pr.vertex_attr, pr.ei, pr.edge_attr, pr.g)))
            el, _ = timed_steps(step, args.steps, 1)
            out["a_ii_copied"] = {
                "what": "same step with option jacobi_diag_flag=0: the first pass writes contiguous copies of A_ii and b, every pass "
                        "streams A_ii (the default recognises that vertex_attr[:, 0] repeats the matrix diagonal bit for bit -- "
                        "checked inside the first pass -- and takes A_ii from the centre coefficient the later passes load anyway)",
                "value": round(total_sweeps / el, 2), "unit": "sweeps/s", "bit_identical_to_default": same}
        finally:
            check(L.gla_set_option(b"jacobi_diag_flag", 1))

    # ---- side: the general CSR-stream kernel (what every non-stencil matrix gets): same step with dia = 0 -----------
    if side:
        try:
            check(L.gla_set_option(b"dia", 0))
            ei_c = pr.ei.clone()
            plan_c = gla.plan_for(ei_c, pr.n_local, edge_attr=pr.edge_attr)
            same = bool(torch.equal(step(), jac(args.sweeps, pr.vertex_attr, ei_c, pr.edge_attr, pr.g)))
            el, _ = timed_steps(lambda: jac(args.sweeps, pr.vertex_attr, ei_c, pr.edge_attr, pr.g), args.steps, 1)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan_c.jacobi_sweeps(pr.diag.reshape(-1), pr.b, pr.x0, 0.7, args.sweeps)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3 / args.sweeps)
            tl = median(ts)
            Bl = 10 * pr.nnz + 2 * (pr.n_local + int(plan_c.info.n_tiles)) + 32 * pr.n_local
            out["csr_general"] = {
                "what": "same step with option dia=0: spmv_stream_vec2_kernel (CSR-adaptive stream tiles, int16 "
                        "tile-relative columns), the kernel non-stencil matrices run on",
                "value": round(total_sweeps / el, 2), "unit": "sweeps/s", "launch_us": round(tl * 1e6, 2),
                "achieved": round(Bl / tl / 1e9, 1), "frac": round(Bl / tl / HBM_PEAK, 4),
                "bytes_moved_is": "layout bytes of one pass (10 B per nonzero + 2 B per row + the vectors)",
                "layout_bytes_per_launch": Bl, "pass_model_frac": round(pr.B / tl / HBM_PEAK, 4),
                "bit_identical_to_default": same}
            del plan_c, ei_c
        finally:
            check(L.gla_set_option(b"dia", 1))

    # ---- side: BASELINE config 3 (Chebyshev degree 4 + power method 10 its, 16 right-hand sides, 4096^2) ------------
    if side:
        K = 16
        n, nnz = pr.n_local, pr.nnz
        gen = torch.Generator().manual_seed(24601)
        b16 = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
        x16 = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
        ev = pr.edge_attr[:, :1].contiguous()
        base = 12 * nnz + 4 * (n + 1)
        va = torch.stack([b16, x16], 1)                       # [n, 2, K]
        gc = torch.tensor([-3.4, -4.0], dtype=torch.float64)
        cheb = gla.ChebyGNN.ChebyRelaxGNN(4, return_edge_attr=False)
        el_c, ms_c = timed_steps(lambda: cheb(va, pr.ei, ev, gc), 5, 2)
        Bc = (base + 40 * n * K) + 3 * (base + 48 * n * K)
        mat = (8 * int(info.dia) + 2) * n if int(info.dia) > 0 else 10 * nnz + 2 * n      # the matrix as this plan stores it
        Mc = mat + 48 * n * K                                 # the fused pass: b, x in; x, r, p and the module's copy of b out
        del va
        pw = gla.PowerMethodGNN.PowerMethodGNN(10, return_edge_attr=False)
        vp = torch.stack([b16, torch.zeros_like(b16)], 1)
        g3 = torch.zeros(3, K, dtype=torch.float64)
        el_p, ms_p = timed_steps(lambda: pw(vp, pr.ei, pr.edge_attr, g3), 5, 2)
        Bp = 11 * (base + 16 * n * K) + 6 * 8 * n * K
        tmax_p = gla.plan_for(pr.ei, pr.n_local, edge_attr=pr.edge_attr).power_fused_max_iters(K, torch.float64)
        passes_p = -(-10 // tmax_p) if tmax_p >= 2 else 10   # temporally blocked: up to five iterations per pass (gla_power_fused)
        Mp = (passes_p + 1) * (mat + 16 * n * K) + 6 * 8 * n * K      # matrix + y in / out per pass (+ the Rayleigh pass), the tail's vectors
        tc, tpw = median(ms_c) * 1e-3, median(ms_p) * 1e-3
        # JacobiGNN on the same 16 right-hand sides (vertex_attr [n, 3, K]): temporally blocked passes of five sweeps
        vj = torch.stack([torch.full_like(b16, -4.0), b16, x16], 1)
        _, ms_j = timed_steps(lambda: jac(args.sweeps, vj, pr.ei, pr.edge_attr, pr.g), 3, 1)
        tj = median(ms_j) * 1e-3 / args.sweeps
        Bj16 = base + 8 * n + 24 * n * K
        passes16 = max(1, -(-args.sweeps // 5)) if bool(info.dia_canon5) else args.sweeps        # (gla_jacobi_sweeps, K > 1: passes of <= 5)
        Mj16 = (mat + 8 * n + 24 * n * K) * passes16 / args.sweeps                              # bytes moved per sweep
        del vj
        out["config3_k16"] = {
            "what": f"BASELINE config 3 through the module surface, {Nx}^2 Laplacian, K=16 right-hand sides, fp64",
            "note": "*_frac = bytes the kernels have to move with this plan's layout (matrix once per pass + the K-wide vectors in / "
                    "out) / time / 8 TB/s; *_model_ratio = SURVEY 8(d)'s per-step byte model / time / 8 TB/s, above 1 where "
                    "several steps share a pass (ChebyRelaxGNN(4): ONE temporally blocked pass, gla_cheby_fused)",
            "cheby_deg4_ms": round(tc * 1e3, 3), "cheby_bytes_moved": Mc, "cheby_GBps": round(Mc / tc / 1e9, 1),
            "cheby_frac": round(Mc / tc / HBM_PEAK, 4), "cheby_model_ratio": round(Bc / tc / HBM_PEAK, 4),
            "power_10its_ms": round(tpw * 1e3, 3), "power_passes": passes_p, "power_bytes_moved": Mp, "power_GBps": round(Mp / tpw / 1e9, 1),
            "power_frac": round(Mp / tpw / HBM_PEAK, 4), "power_model_ratio": round(Bp / tpw / HBM_PEAK, 4),
            "jacobi_ms_per_sweep": round(tj * 1e3, 4), "jacobi_bytes_moved_per_sweep": int(Mj16),
            "jacobi_GBps": round(Mj16 / tj / 1e9, 1), "jacobi_frac": round(Mj16 / tj / HBM_PEAK, 4),
            "jacobi_model_ratio": round(Bj16 / tj / HBM_PEAK, 4)}
        del vp, b16, x16
        torch.cuda.empty_cache()

    # ---- side: BASELINE config 4's matrix on ONE GPU (8192^2) -------------------------------------------------------
    if side and args.n == 4096:
        free, _ = torch.cuda.mem_get_info()
        if free > 40e9:
            p4 = Problem(gla, 8192, 8192, 0, 8192 * 8192, dev, 24601)
            el4, ms4 = timed_steps(lambda: jac(args.sweeps, p4.vertex_attr, p4.ei, p4.edge_attr, p4.g), 3, 1)
            t4 = median(ms4) * 1e-3 / args.sweeps
            plan4 = gla.plan_for(p4.ei, p4.n_local, edge_attr=p4.edge_attr)
            lay4 = step_layout(plan4.info, plan4.jacobi_schedule(args.sweeps), dg_center, aos_first=True)
            m4 = sum(rb for _, _, rb in lay4) * p4.n_local / args.sweeps          # bytes moved per sweep with the layout the passes really read
            out["config4_1gpu"] = {"what": "JacobiGNN.forward, 20 sweeps, 8192^2 Laplacian fp64 on one GPU",
                                   "value": round(1.0 / t4, 2), "unit": "sweeps/s", "us_per_sweep": round(t4 * 1e6, 1),
                                   "ms_per_step": round(t4 * args.sweeps * 1e3, 4),
                                   "achieved": round(m4 / t4 / 1e9, 1), "frac": round(m4 / t4 / HBM_PEAK, 4),
                                   "bytes_moved_is": "layout bytes: passes of " + " + ".join(f"{rb} ({k}, {ts_} sweeps)" for k, ts_, rb in lay4)
                                                     + f" B per row per {args.sweeps} sweeps",
                                   "model_ratio": round(p4.B / t4 / HBM_PEAK, 4)}
            del plan4
            del p4
            torch.cuda.empty_cache()

        # the multi-GPU route on this ONE GPU: dist.RowBlockJacobi's temporally blocked forward with a real one-rank RCCL
        # communicator whose upper and lower neighbour are the rank itself (periodic in the line direction): pack,
        # ncclAllGather on the library's stream, interior line blocks, ghost fill + edge blocks on a second stream.  Its
        # ratio to `value` is what weak scaling can reach before inter-GPU latency (tools/bench_dist_selfhalo.py)
        try:
            from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
            Nn = args.n
            drv = RowBlockJacobi.rehearsal_self_halo(Nn, Nn, dev)
            comm_, plan_e = drv.comm, drv.ext["plan"]
            va_d = pr.vertex_attr
            eld, msd = timed_steps(lambda: drv.forward(args.sweeps, va_d, pr.g), 20, 3)
            td = median(msd) * 1e-3
            out["dist_route_one_gpu"] = {
                "what": "dist.RowBlockJacobi forward (the N > 1 route: one ghost-line exchange per pass over a real one-rank "
                        "RCCL communicator, neighbours = the rank itself), same block size and sweeps as the headline",
                "value": round(args.sweeps / td, 2), "unit": "sweeps/s", "ms_per_step": round(td * 1e3, 4),
                "fraction_of_headline_step": round((median(step_ms) * 1e-3) / td, 4)}
            # ... and the K = 16 ghost-line routes of config 3 the same way (one exchange per pass of five Jacobi sweeps / per
            # fused Chebyshev pass), plain operands and operands allocated with their ghost rows (drv.ext_tensor: read in place)
            if args.n == 4096:
                try:
                    from gnn_applied_linear_algebra_amd.dist import RowBlockCheby
                    K16 = 16

                    def k_driver(cls):
                        d_ = cls.__new__(cls)
                        d_.__dict__.update(drv.__dict__)
                        d_._lib_send, d_._ghost_pending, d_.ext = {}, None, dict(drv.ext)
                        return d_
                    genk = torch.Generator().manual_seed(7)
                    vj = torch.rand(Nn * Nn, 3, K16, generator=genk, dtype=torch.float64).to(dev)
                    vj[:, 0, :] = -4.0
                    jd = k_driver(RowBlockJacobi)
                    _, m1 = timed_steps(lambda: jd.forward(args.sweeps, vj, pr.g), 4, 1)
                    vje = jd.ext_tensor(3, K16)
                    vje.copy_(vj)
                    _, m2 = timed_steps(lambda: jd.forward(args.sweeps, vje, pr.g), 4, 1)
                    jac_m = gla.JacobiGNN.JacobiGNN()
                    _, m0 = timed_steps(lambda: jac_m(args.sweeps, vj, pr.ei, pr.edge_attr, pr.g), 4, 1)
                    rec = {"what": "the K = 16 ghost-line routes of dist.py over a one-rank RCCL communicator (neighbours = the rank "
                                   "itself), 4096^2 rows per rank; *_ext = operands allocated with their ghost rows (drv.ext_tensor)",
                           "jacobi_single_gpu_ms": round(median(m0), 3), "jacobi_route_ms": round(median(m1), 3),
                           "jacobi_route_ext_ms": round(median(m2), 3)}
                    del vj, vje, jd
                    torch.cuda.empty_cache()
                    bk = torch.rand(Nn * Nn, K16, generator=genk, dtype=torch.float64).to(dev)
                    xk = torch.rand(Nn * Nn, K16, generator=genk, dtype=torch.float64).to(dev)
                    gch = torch.tensor([-3.4, -4.0], dtype=torch.float64)
                    cd = k_driver(RowBlockCheby)
                    _, c1 = timed_steps(lambda: cd.forward(4, bk, xk, gch), 4, 1)
                    bke, xke = cd.ext_tensor(K16), cd.ext_tensor(K16)
                    bke.copy_(bk)
                    xke.copy_(xk)
                    _, c2 = timed_steps(lambda: cd.forward(4, bke, xke, gch), 4, 1)
                    rec.update({"cheby_deg4_route_ms": round(median(c1), 3), "cheby_deg4_route_ext_ms": round(median(c2), 3),
                                "cheby_deg4_single_gpu_ms": out.get("config3_k16", {}).get("cheby_deg4_ms")})
                    out["dist_route_one_gpu_k16"] = rec
                    del bk, xk, bke, xke, cd
                except Exception as exc:
                    out["dist_route_one_gpu_k16"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            del drv, plan_e
            torch.cuda.synchronize()
            from gnn_applied_linear_algebra_amd._lib import lib as _lib
            _lib().gla_comm_destroy(comm_._h)
            comm_._h = None
            torch.cuda.empty_cache()
        except Exception as exc:                             # (a side record only: e.g. no RCCL library on the box)
            out["dist_route_one_gpu"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # ---- side: BASELINE config 5 (AMG V-cycle on the heateqnfem2dfun Q1 mesh, 4095^2 = 16.8 M unknowns) on this GPU ----------
    if side and args.n == 4096:
        try:
            out["config5_vcycle"] = side_config5(gla, dev, 4095)
        except Exception as exc:
            out["config5_vcycle"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        torch.cuda.empty_cache()
        # ---- side: config 5 is "8 x MI355X": a rank's share of the distributed cycle rehearsed on this GPU (child process: it
        # needs a one-rank RCCL process group of its own)
        out["dist_vcycle_one_rank"] = side_dist_vcycle(120.0)
        # ---- side: BASELINE config 4's REAL 8-rank block (1024 lines x 8192 points) over a one-rank RCCL communicator, eager and
        # replayed from a HIP graph, next to the whole 8192^2 matrix on this GPU (child process: tools/bench_dist_config4_block.py)
        c4_tool = [os.path.join(ROOT, "tools", "bench_dist_config4_block.py"), "8192", "8", str(args.sweeps), "--json"]
        out["dist_route_one_gpu_config4_block"] = side_child_json(c4_tool, 240.0)
        # (A/B in the same line: the passes NOT staggered -- one edge height, the main stream joins the second one every pass;
        # option dist_stagger = 0)
        c4_plain = side_child_json(c4_tool + ["--block-only"], 120.0, extra_env={"GLA_OPTIONS": "dist_stagger=0"})
        if isinstance(out["dist_route_one_gpu_config4_block"], dict) and isinstance(c4_plain, dict):
            out["dist_route_one_gpu_config4_block"]["eager_ms_per_step_one_join_per_pass"] = c4_plain.get("eager_ms_per_step", c4_plain.get("error"))
        # ---- side: the price of bit-exactness, measured once (never the default): the same step through the
        # opt-in build variant libgla_hip_contracted.so (FMA contraction in the row sums, division as one multiply-add with
        # RN(1 / A_ii)) next to the default library, each in a child process of its own (tools/bench_contracted_fp.py)
        lib_c = os.path.join(ROOT, "gnn_applied_linear_algebra_amd", "libgla_hip_contracted.so")
        if os.path.exists(lib_c):
            tool = os.path.join(ROOT, "tools", "bench_contracted_fp.py")
            rec_d = side_child_json([tool, str(args.n), str(args.sweeps), "--json"], 120.0)
            rec_c = side_child_json([tool, str(args.n), str(args.sweeps), "--json"], 120.0, extra_env={"GLA_LIB_PATH": lib_c})
            out["contracted_fp"] = {
                "what": "opt-in BUILD VARIANT, not the product: gla_march5.hip compiled with -ffp-contract=fast -DGLA_CONTRACTED_FP (row sums "
                        "as FMA chains, x + num * RN(1 / A_ii) instead of the correctly rounded quotient); deviation = largest "
                        "|x - x_ref| / max |x_ref| after the step against a plain torch fp64 iteration (north_star's bar: 1e-6; the "
                        "default build is bit-exact against the oracle)",
                "default_build": rec_d, "contracted_build": rec_c,
                "speedup": (round(rec_d["ms_per_step"] / rec_c["ms_per_step"], 4)
                            if "ms_per_step" in rec_d and "ms_per_step" in rec_c else None),
                "speedup_scaled_matrix_DAD": (round(rec_d["scaled_matrix_DAD"]["ms_per_step"] / rec_c["scaled_matrix_DAD"]["ms_per_step"], 4)
                                              if "scaled_matrix_DAD" in rec_d and "scaled_matrix_DAD" in rec_c else None),
                "note": "the Laplacian's products and quotients are exact (coefficients -4 and 1): there the two builds agree bit for bit; "
                        "D A D with a random diagonal D (entries a_ij (d_i d_j): still a symmetric value set) is the case where they differ"}
        # ---- side: non-stencil matrices through the CSR-adaptive kernels (stream / vector / long bins) ----------------------
        try:
            out["csr_irregular"] = side_irregular(gla, dev)
        except Exception as exc:
            out["csr_irregular"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        torch.cuda.empty_cache()

    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.n, args.cpu_sweeps)
    emit(out)


def _median_us(fn, reps=10, warm=2):
    import torch
    ts = []
    for i in range(reps + warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return median(ts)


def side_config5(gla, dev, m):
    """BASELINE config 5 on one GPU: V(3,3) cycles of AMGHierarchy on the m x m Q1 FEM mesh of heateqnfem2dfun (grid
    coarsening): (a) 8 levels with a dense coarsest solve, (b) BASELINE's 3-level shape with a Chebyshev coarsest solve that
    converges (degree 768 on the 1024^2 level; the degree-4 one the reference's two-grid script uses does not reduce the
    residual on a level this large).  Setup time, ms per cycle, residual reduction per cycle."""
    import torch
    ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
    n = m * m
    A = gla.VCycle._as_sparse(ei, ev, (n, n))
    gen = torch.Generator().manual_seed(0)
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
    rec = {"what": f"AMGHierarchy V(3,3) cycle, Q1 FEM mesh {m}^2 = {n} unknowns, {ei.shape[1]} nonzeros, fp64, grid coarsening"}
    for key, kw, ncyc in (("levels8_dense_coarsest", dict(levels=8, coarse_solver="dense"), 6),
                          ("levels3_cheby768_coarsest", dict(levels=3, coarse_solver="cheby", coarse_deg=768), 4)):
        def build():
            return gla.VCycle.AMGHierarchy(A, splitting=gla.VCycle.GridSplitting(m, m), min_coarse=500, **kw)
        H = build()                                          # (first build pays the first touch of every large allocation)
        torch.cuda.synchronize()
        del H
        t0 = time.perf_counter()
        H = build()
        torch.cuda.synchronize()
        setup_ms = 1e3 * (time.perf_counter() - t0)
        x = torch.zeros_like(b)
        hist = [float(torch.norm(H.residual(b, x)))]
        for _ in range(ncyc):
            x = H.cycle(b, x)
            hist.append(float(torch.norm(H.residual(b, x))))
        xs = [x]

        def one():
            xs[0] = H.cycle(b, xs[0])
        t_us = _median_us(one, reps=5, warm=1)
        rates = [hist[i + 1] / hist[i] for i in range(len(hist) - 1)]
        rec[key] = {"levels": len(H.A), "level_rows": [int(Al.shape[0]) for Al in H.A], "setup_ms": round(setup_ms, 1),
                    "ms_per_cycle": round(t_us * 1e-3, 3), "residual_history": [float(f"{h:.4e}") for h in hist],
                    "residual_reduction_per_cycle": [round(r, 4) for r in rates]}
        del H, x, xs
        torch.cuda.empty_cache()
    return rec


def _row_bytes(drv, blocked, sweeps=None):
    """Bytes a 5-point row of a row-block driver moves per STEP of `sweeps` sweeps divided by the passes it takes (i.e. the average
    per pass), from the layout its passes really read: the temporally blocked route runs the single-GPU path's kernels on the
    extended block (step_layout: symmetric value sets read 3 of 5 diagonals, full stencils no masks, the first pass reads the
    AoS rows, the later ones take A_ii from the centre coefficient where the call's flag stays clean -- the bench's A_ii IS the
    diagonal); the exchange-per-sweep route streams 5 diagonals, the mask, A_ii, b, x in, x out."""
    ext = getattr(drv, "ext", None)
    if blocked and ext is not None and ext.get("plan") is not None and sweeps:
        info = ext["plan"].info
        sched = drv._sched(sweeps)
        flag_flow = bool(int(info.dia_canon5) == 1 and bin(int(info.dia_sym)).count("1") and int(getattr(info, "dia_full", 0))
                         and min(sched) >= 2 and len(sched) >= 2)
        lay = step_layout(info, sched, flag_flow, aos_first=True)
        return sum(rb for _, _, rb in lay) / len(lay)
    return 8 * 5 + 2 + 32


def side_child_json(argv, timeout_s, extra_env=None):
    """Run a tool as a child process (its own RCCL process group / communicator) and return the JSON object it prints last."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() + 7) % 400))
    env.update(extra_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        pr_ = subprocess.Popen([sys.executable] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        try:
            so, _ = pr_.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            pr_.kill()
            pr_.wait()
            return {"error": f"timed out after {timeout_s:.0f} s"}
        lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if pr_.returncode == 0 and lines else {"error": f"child exited with {pr_.returncode}"}
    except Exception as exc:                                 # (a side record: never the reason a bench run fails)
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def side_dist_vcycle(timeout_s):
    """dist_amg.RowBlockAMG's cycle driven by ONE rank over real RCCL next to AMGHierarchy.cycle on the same rows
    (tools/bench_dist_vcycle_one_rank.py, run as a child process with its own process group): every distributed level on the
    ghost-line route, its all-gathers on libgla_hip's communicator, no neighbour data -- what a rank's cycle costs before
    inter-GPU latency.  (a) 4095^2 with the agglomeration threshold of 8 ranks: a whole GPU's worth of rows; (b) 1447^2 =
    2.1 M rows, the level-0 block of one of 8 ranks: the eager cycle is host-bound there, the HIP-graph replay is not."""
    recs = {"what": "a rank's share of the distributed V(3,3) cycle on ONE GPU over a one-rank RCCL group vs the single-GPU "
                    "cycle on the same rows; no multi-GPU hardware was reachable: NOT a scaling measurement"}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    for key, argv in (("mesh4095_threshold_of_8_ranks", ["4095", "8", "524288"]), ("mesh1447_block_of_one_of_8_ranks", ["1447", "7", "65536"])):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_dist_vcycle_one_rank.py"), *argv, "--json"]
        try:
            pr_ = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            try:
                so, _ = pr_.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                pr_.kill()
                pr_.wait()
                recs[key] = {"error": f"timed out after {timeout_s:.0f} s"}
                continue
            lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
            recs[key] = json.loads(lines[-1]) if pr_.returncode == 0 and lines else {"error": f"child exited with {pr_.returncode}"}
        except Exception as exc:                             # (a side record: never the reason a bench run fails)
            recs[key] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    return recs


def side_irregular(gla, dev):
    """SpMV, one Jacobi sweep and a 16-column SpMM on matrices that are NOT stencils (the CSR-adaptive kernels: stream tiles, the vector bin for
    tiles whose rows average >= 32 nonzeros, the long bin): us per launch, GB/s and fraction of 8 TB/s on SURVEY 8(d)'s bytes
    (12 nnz + 4 (n + 1) + 16 n / + 32 n; the x gather counted once)."""
    import torch
    check, L = gla._lib.check, gla._lib.lib()

    def measure(name, ei, ev, n):
        nnz = ei.shape[1]
        plan = gla.CsrPlan(ei, n)
        plan.set_values(ev)
        gen = torch.Generator().manual_seed(5)
        x = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
        bb = torch.rand(n, 1, generator=gen, dtype=torch.float64).to(dev)
        dg = (torch.rand(n, generator=gen, dtype=torch.float64) + 1.0).to(dev)
        y, xo = torch.empty_like(x), torch.empty_like(x)
        t_mv = _median_us(lambda: plan.spmm(x, out=y))
        t_j = _median_us(lambda: plan.jacobi_sweep(dg, bb, x, xo, 0.7))
        B_mv, B_j = 12 * nnz + 4 * (n + 1) + 16 * n, 12 * nnz + 4 * (n + 1) + 32 * n
        lens = torch.bincount(ei[0], minlength=n)
        # K = 16 right-hand sides through the same plan (spmm_stream_vec2_kernel; rows of >= 64 nonzeros: the K-wide vector bin)
        K16 = 16
        x16 = torch.rand(n, K16, generator=gen, dtype=torch.float64).to(dev)
        y16 = torch.empty_like(x16)
        t_mm = _median_us(lambda: plan.spmm(x16, out=y16))
        B_mm = 12 * nnz + 4 * (n + 1) + 16 * n * K16
        check(L.gla_set_option(b"row_sum", 0))
        try:
            t_mm0 = _median_us(lambda: plan.spmm(x16, out=y16))
        finally:
            check(L.gla_set_option(b"row_sum", 1))
        return {"matrix": name, "rows": n, "nnz": nnz, "avg_row": round(nnz / n, 2), "max_row": int(lens.max()),
                "spmv_us": round(t_mv, 1), "spmv_GBps": round(B_mv / t_mv / 1e3, 1), "spmv_frac": round(B_mv / (t_mv * 1e-6) / HBM_PEAK, 4),
                "jacobi_us": round(t_j, 1), "jacobi_GBps": round(B_j / t_j / 1e3, 1), "jacobi_frac": round(B_j / (t_j * 1e-6) / HBM_PEAK, 4),
                "spmm_k16_us": round(t_mm, 1), "spmm_k16_GBps": round(B_mm / t_mm / 1e3, 1),
                "spmm_k16_frac": round(B_mm / (t_mm * 1e-6) / HBM_PEAK, 4), "spmm_k16_us_one_lane_group_per_row": round(t_mm0, 1)}

    recs = []
    m = 4095
    ei, ev = gla.UtilsGNN.heateqnfem2dfun([m + 1, m + 1], [1.0, 1.0], device=dev)
    A = gla.VCycle._as_sparse(ei, ev, (m * m, m * m))
    H = gla.VCycle.AMGHierarchy(A, levels=3, splitting="pmis", min_coarse=500, coarse_deg=4)
    for lv in (1, 2):
        eil, eal = gla.UtilsGNN.coo_to_gnn_input(H.A[lv])
        recs.append(measure(f"level-{lv} Galerkin operator of the {m}^2 Q1 FEM mesh under pmis_splitting", eil, eal, H.A[lv].shape[0]))
    del H, A, ei, ev
    torch.cuda.empty_cache()
    gen = torch.Generator().manual_seed(11)
    n = 1_000_000
    u = torch.rand(n, generator=gen, dtype=torch.float64)
    lens = (3 * (1 - u) ** (-1.0 / 1.2)).clamp(max=3000).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n), lens)
    evp = torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.5
    for label, cols in (("random columns (the x gather is L2 / Infinity-Cache traffic, not in the byte model)",
                         torch.randint(0, n, (rows.numel(),), generator=gen)),
                        ("columns within +-4096 of the row",
                         (rows + torch.randint(-4096, 4097, (rows.numel(),), generator=gen)).clamp(0, n - 1))):
        recs.append(measure(f"1 M rows, power-law row lengths 3 .. 3000 (alpha 2.2), {label}", torch.stack([rows, cols]).to(dev), evp.to(dev), n))
    n = 200_000
    lens = torch.randint(64, 1025, (n,), generator=gen)
    rows = torch.repeat_interleave(torch.arange(n), lens)
    cols = (rows + torch.randint(-4096, 4097, (rows.numel(),), generator=gen)).clamp(0, n - 1)
    evm = torch.rand(rows.numel(), 1, generator=gen, dtype=torch.float64) - 0.5
    recs.append(measure("200 k rows of 64 .. 1024 nonzeros (vector bin), columns within +-4096 of the row", torch.stack([rows, cols]).to(dev), evm.to(dev), n))
    return {"what": "non-stencil matrices through the CSR-adaptive K = 1 kernels (row_sum = 1: vector bin on)", "matrices": recs}


def multi_gpu(args, gla, dev, world, rank):
    import torch
    import torch.distributed as dist
    from gnn_applied_linear_algebra_amd.dist import RowBlockJacobi
    ranks_seen = dist.get_world_size()
    on_dev = args.backend == "nccl"
    t_start = time.time()

    def log(msg):                                            # progress on stderr (rank 0): a long run is not a hung run
        if rank == 0:
            print(f"bench.py [{time.time() - t_start:7.1f} s] {msg}", file=sys.stderr, flush=True)

    def barrier():
        dist.barrier()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev if on_dev else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    setup_log = []

    def run_block(Nx, lines, steps, warmup):
        """Row block of `lines` grid lines per rank of the (lines*world) x Nx grid; returns (elapsed, problem, drv)."""
        Ny = lines * world
        n_local = Nx * lines
        torch.cuda.synchronize()
        t_s0 = time.perf_counter()
        pr = Problem(gla, Nx, Ny, rank * n_local, (rank + 1) * n_local, dev, 24601 + rank)
        torch.cuda.synchronize()
        t_s1 = time.perf_counter()
        log(f"{Ny} x {Nx} grid, {lines} lines per rank: local problem built")
        drv = RowBlockJacobi(pr.ei, pr.edge_attr, n_global=Nx * Ny, row_lo=rank * n_local, row_hi=(rank + 1) * n_local,
                             overlap=not args.no_overlap, host_loop=args.host_loop)
        torch.cuda.synchronize()
        t_s2 = time.perf_counter()
        log(f"row-block driver ready ({drv.mode})")
        # setup in front of the timed steps, slowest rank ("put setup_s per block into the N > 1 line")
        setup_log.append({"block": f"{lines} lines x {Nx} points per rank", "problem_s": round(max_over_ranks(t_s1 - t_s0), 3),
                          "row_block_driver_s": round(max_over_ranks(t_s2 - t_s1), 3)})
        if args.ramp_ms > 0:
            # clock ramp as in the single-GPU line (profiles/r04/clock_ramp.txt).  Every forward() is a collective: the number
            # of ramp steps is agreed (from the slowest rank's time for two steps), never a per-rank clock
            barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                drv.forward(args.sweeps, pr.vertex_attr, pr.g)
            torch.cuda.synchronize()
            est = max(max_over_ranks(time.perf_counter() - t0) / 2, 1e-5)
            for _ in range(min(400, int(args.ramp_ms * 1e-3 / est))):
                drv.forward(args.sweeps, pr.vertex_attr, pr.g)
            torch.cuda.synchronize()
        elapsed, step_ms = timed_steps(lambda: drv.forward(args.sweeps, pr.vertex_attr, pr.g), steps, warmup, barrier)
        log(f"{steps} steps timed: {1e3 * elapsed / steps:.3f} ms per step")
        return max_over_ranks(elapsed), pr, drv

    Nx = args.n
    if args.strong and args.n % world:
        raise SystemExit("--strong needs --grid divisible by the number of GPUs")
    lines = args.n // world if args.strong else args.n
    elapsed, pr, drv = run_block(Nx, lines, args.steps, args.warmup)
    total_sweeps = args.steps * args.sweeps
    t_sweep = elapsed / total_sweeps
    # bytes a rank moves per step: one pass over its block (74 B per row: values, mask, b, A_ii, x in, x out) per launch
    sched_fn = getattr(drv, "_sched", None)
    passes = len(sched_fn(args.sweeps)) if sched_fn is not None and drv.mode.startswith("temporally blocked") else args.sweeps
    row_b = _row_bytes(drv, passes < args.sweeps, args.sweeps)
    moved_sweep = row_b * pr.n_local * passes / args.sweeps
    roof = {"bound": "hbm", "achieved": round(moved_sweep / t_sweep / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(moved_sweep / t_sweep / HBM_PEAK, 4), "traffic": None,
            "bytes_moved_is": f"layout bytes per rank: {passes} passes of {row_b:.1f} B per row on average per {args.sweeps} sweeps",
            "kernel": "per-rank step incl. halo exchange (interior + boundary launches), per sweep",
            "passes_per_step": passes, "algorithmic_bytes_per_sweep": pr.B,
            "model_ratio": round(pr.B / t_sweep / HBM_PEAK, 4), "avg_launch_us": round(t_sweep * 1e6 * args.sweeps / passes, 2)}
    out = {
        "metric": "Jacobi-GNN sweeps/sec (2D Laplacian, fp64)",
        "value": round((1 if args.strong else world) * total_sweeps / elapsed, 2), "unit": "sweeps/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"JacobiGNN.forward, {args.sweeps} sweeps/step, 5-pt Laplacian, {lines}x{Nx} vertices "
                               f"per GPU (global grid {lines * world}x{Nx}), fp64, w=0.7, row-block partition + RCCL "
                               f"halo all-gather ({drv.mode})",
                   "N": Nx, "rows_per_gpu": pr.n_local, "nnz_per_gpu": pr.nnz, "sweeps_per_step": args.sweeps,
                   "backend": args.backend, "exchange": drv.mode,
                   "aggregate_bytes_moved_GBps": round(world * moved_sweep * total_sweeps / elapsed / 1e9, 1),
                   "aggregate_model_equivalent_GBps": round(world * pr.B * total_sweeps / elapsed / 1e9, 1)},
        "roofline": roof,
    }
    del pr, drv
    # (the allocator's cache is KEPT: the strong-scaling block below is no larger than this one, and fresh hipMallocs of
    #  several GB -- first touch, page tables -- cost seconds, tens of seconds when ranks share a GPU in a rehearsal)

    # ---- BASELINE config 4: FIXED strong_grid^2 Laplacian cut into `world` row blocks, and the 1-GPU rate of the
    #      same matrix measured by rank 0 alone in this run (the other ranks wait at the barrier) -------------------
    G = args.strong_grid
    if not args.no_side and not args.strong and G % world == 0:
        steps4 = max(3, min(args.steps, 10))
        el4, p4, d4 = run_block(G, G // world, steps4, 2)
        t4 = el4 / (steps4 * args.sweeps)
        B4_global = algorithmic_bytes_jacobi(G * G, 5 * G * G - 4 * G)
        rec = {"what": f"BASELINE config 4: JacobiGNN {args.sweeps} sweeps/step on the FIXED {G}x{G} Laplacian, "
                       f"{G // world} grid lines per GPU, halo all-gather per sweep ({d4.mode})",
               "scaling": "strong", "n_gpus": world, "ranks_seen": ranks_seen, "value": round(1.0 / t4, 2),
               "unit": "sweeps/s", "us_per_sweep": round(t4 * 1e6, 2),
               "model_equivalent_GBps_per_gpu": round(p4.B / t4 / 1e9, 1), "model_ratio_per_gpu": round(p4.B / t4 / HBM_PEAK, 4),
               "aggregate_model_equivalent_GBps": round(B4_global / t4 / 1e9, 1)}
        sf4 = getattr(d4, "_sched", None)
        ps4 = len(sf4(args.sweeps)) if sf4 is not None and d4.mode.startswith("temporally blocked") else args.sweeps
        m4 = _row_bytes(d4, ps4 < args.sweeps, args.sweeps) * p4.n_local * ps4 / args.sweeps
        rec.update({"bytes_moved_GBps_per_gpu": round(m4 / t4 / 1e9, 1), "frac_per_gpu": round(m4 / t4 / HBM_PEAK, 4),
                    "passes_per_step": ps4})
        del p4, d4
        t1 = None
        if rank == 0 and not args.single_device:
            jac = gla.JacobiGNN.JacobiGNN()
            p1 = Problem(gla, G, G, 0, G * G, dev, 24601)
            _, ms1 = timed_steps(lambda: jac(args.sweeps, p1.vertex_attr, p1.ei, p1.edge_attr, p1.g), 3, 1)
            t1 = median(ms1) * 1e-3 / args.sweeps
            log(f"{G} x {G} on rank 0 alone: {t1 * 1e6:.1f} us per sweep")
            del p1
            torch.cuda.empty_cache()
        barrier()
        if rank == 0 and t1 is not None:
            rec.update({"one_gpu_sweeps_per_s": round(1.0 / t1, 2), "speedup_vs_one_gpu": round(t1 / t4, 3),
                        "parallel_efficiency": round(t1 / t4 / world, 3)})
        out["config4_strong"] = rec
    out["setup_s"] = {"what": "seconds in front of the timed steps per block, slowest rank: the synthetic problem (graph builder + operands) and "
                              "the row-block driver (partition, block plan, ghost rows of the matrix, extended plan, communicator)",
                      "blocks": setup_log}
    if rank == 0:
        emit(out)


_JSON_FD = None
_STACK_FILE = None


def emit(obj):
    """The ONE JSON line, written to the process's original stdout."""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def worker(args):
    global _JSON_FD
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; before any HIP call
    # stdout carries the JSON line and nothing else: libraries print there too (RCCL's version banner on the first
    # communicator, gloo's connection banner), so file descriptor 1 is pointed at stderr for the life of the rank and
    # the line goes to a duplicate of the original
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # a slow phase must say WHERE it is: rank 0 of a multi-rank run dumps its Python stacks to stderr every --stack-dump-s seconds
    # (faulthandler: a watchdog thread, no cost to the run); round 4's fresh-box rehearsal sat 27-98 s in setup phases nobody
    # could name afterwards
    if args.stack_dump_s > 0 and world > 1 and (rank == 0 or os.environ.get("GLA_BENCH_STACK_DIR")):
        import faulthandler
        sdir = os.environ.get("GLA_BENCH_STACK_DIR")        # (every rank into a file of its own: who waits for whom)
        global _STACK_FILE
        _STACK_FILE = open(os.path.join(sdir, f"stacks_rank{rank}.txt"), "w") if sdir else sys.stderr
        faulthandler.dump_traceback_later(args.stack_dump_s, repeat=True, file=_STACK_FILE)
    import torch
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
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
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")

    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // (2 * world))))
    import gnn_applied_linear_algebra_amd as gla
    check, L = gla._lib.check, gla._lib.lib()
    if args.compress:
        check(L.gla_set_option(b"compress", 1))
    if args.no_col16:
        check(L.gla_set_option(b"col16", 0))
    if args.no_dia or args.no_col16:
        check(L.gla_set_option(b"dia", 0))
    if args.no_diag_flag:
        check(L.gla_set_option(b"jacobi_diag_flag", 0))
    if world == 1:
        single_gpu(args, gla, dev)
        return
    try:
        multi_gpu(args, gla, dev, world, rank)
    except BaseException:
        # a failed rank must not wait for its peers in destroy_process_group (they may sit in a collective): report
        # and leave at once; the parent (or torchrun) stops the other ranks and the run exits non-zero
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    from gnn_applied_linear_algebra_amd.dist import LibComm
    torch.cuda.synchronize()
    LibComm.release_all()
    dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or args.launch_dry_run):
        sys.exit(launch(args, argv))                            # parent: never touches torch or the GPU
    if env_world is not None and int(env_world) != args.gpus:
        args.gpus = int(env_world)                              # the launcher's world size wins
    worker(args)


if __name__ == "__main__":
    main()
