#!/usr/bin/env python3
"""Why the 4-rank SINGLE-DEVICE rehearsals of `bench.py --gpus 4` sat 27-104 s in setup phases at random places (rounds 4, 5).
The per-rank stack dumps of round 5 (profiles/r05/gloo4_single_device_stall.txt) show all four ranks inside the same kind of call
with the GPU 100 % busy: torch.unique / boolean-mask indexing (hipCUB sort, select) or CsrPlan creation (rocPRIM scan / radix
sort) -- single-pass kernels with DECOUPLED LOOK-BACK, whose workgroups spin on flags their predecessors write.  With several
PROCESSES time-sharing one GPU a spinning workgroup can hold the card while the workgroup it waits for is switched out.
This tool measures exactly that and nothing of this repository: N processes on cuda:0 each run a loop of look-back kernels
(torch.unique, torch.sort, masked selection, cumsum on 2^24 elements) and, for contrast, of plain elementwise kernels, and report
the median and the maximum time per call.  Usage: python tools/time_shared_gpu_lookback.py [N=4] [seconds=20]"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    rank, secs = int(sys.argv[2]), float(sys.argv[3])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(rank)
    x = torch.randint(0, 1 << 20, (1 << 24,), device=dev, generator=g)
    f = torch.rand(1 << 24, device=dev, generator=g, dtype=torch.float64)
    ops = {"unique (sort + look-back select)": lambda: torch.unique(x),
           "sort (onesweep radix sort)": lambda: torch.sort(x),
           "masked select (look-back scan)": lambda: f[f > 0.5],
           "cumsum (look-back scan)": lambda: torch.cumsum(f, 0),
           "elementwise mul + add (no inter-workgroup wait)": lambda: f * 2.0 + 1.0}
    for fn in ops.values():
        fn()
    torch.cuda.synchronize()
    res = {}
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        for name, fn in ops.items():
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(time.perf_counter() - t0)
    lines = []
    for name, ts in res.items():
        ts.sort()
        lines.append(f"    {name:50s} calls {len(ts):5d}  median {1e3 * ts[len(ts) // 2]:9.2f} ms  max {1e3 * ts[-1]:10.2f} ms")
    print(f"  rank {rank}\n" + "\n".join(lines), flush=True)
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
print(f"{N} process(es) sharing cuda:0, {secs:.0f} s each, 2^24 elements per call")
t0 = time.perf_counter()
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(secs)], stdout=subprocess.PIPE, text=True)
         for r in range(N)]
for p in procs:
    so, _ = p.communicate(timeout=900)
    print(so, end="")
print(f"wall {time.perf_counter() - t0:.1f} s")
