#!/usr/bin/env python3
"""What does a fresh device allocation cost when SEVERAL processes share one GPU?  (Round 4's 4-rank single-device rehearsals of
`bench.py --gpus 4` sat 27-98 s in setup phases at random places; the stack dumps of round 5 show the rank waiting for the GPU
inside whatever torch op came next.)  N processes (children of this script) each allocate, touch, release and re-allocate
tensors of 0.25 .. 4 GiB on cuda:0 at the same time and report the host time of every step.
Usage: python tools/time_shared_gpu_malloc.py [N=4]     (N = 1: the same sequence alone)"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    rank = int(sys.argv[2])
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    out = []
    t_all = time.perf_counter()
    for gib in (0.25, 1.0, 2.0, 4.0, 2.0, 4.0):
        n = int(gib * (1 << 30)) // 8
        t0 = time.perf_counter()
        x = torch.empty(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        x.fill_(1.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        x.mul_(2.0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        del x
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        out.append(f"{gib:5.2f} GiB: alloc {1e3 * (t1 - t0):9.1f} ms  first touch {1e3 * (t2 - t1):9.1f} ms  second pass {1e3 * (t3 - t2):8.1f} ms  "
                   f"free {1e3 * (t4 - t3):9.1f} ms")
    print(f"rank {rank} (total {time.perf_counter() - t_all:6.1f} s)\n  " + "\n  ".join(out), flush=True)
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r)], stdout=subprocess.PIPE, text=True) for r in range(N)]
print(f"{N} process(es) on cuda:0, each: empty -> fill -> mul -> del + empty_cache, sizes 0.25, 1, 2, 4, 2, 4 GiB")
for p in procs:
    so, _ = p.communicate(timeout=600)
    print(so, end="")
