#!/usr/bin/env python3
"""The interior / edge split of a multi-GPU pass on config 4's block (1024 + 16 lines x 8192 points): the interior launch
(part 1), the two edge line blocks (part 2) and both at once on two streams, each timed with HIP events.
Usage: python tools/time_edge_blocks.py [Nx=8192] [lines=1024] [ts=7]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code, stream_ptr
import ctypes
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ts = int(sys.argv[3]) if len(sys.argv) > 3 else 7
T = 8
dev = torch.device("cuda:0")
ei, ev = gla.UtilsGNN.laplacianfun_torch(Nx, Ny=lines + 2 * T, device=dev)
n = Nx * (lines + 2 * T)
plan = gla.CsrPlan(ei, n); plan.set_values(ev)
gen = torch.Generator().manual_seed(1)
b = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
x = torch.rand(n, generator=gen, dtype=torch.float64).to(dev)
out = torch.empty_like(x)
code = dtype_code(torch.float64)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)


def launch(part, st):
    check(lib().gla_jacobi_fused_parts(plan._h, code, None, ptr(b), ptr(x), ptr(out), 0.7, ts, part, T, ctypes.c_void_p(st.cuda_stream)))


def med(fn, reps=20):
    v = []
    for i in range(reps + 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(main); fn(); e1.record(main); torch.cuda.synchronize()
        if i >= 5: v.append(e0.elapsed_time(e1) * 1e3)
    v.sort(); return v[len(v) // 2]


def both():
    side.wait_stream(main)
    launch(1, main)
    launch(2, side)
    main.wait_stream(side)


def both_edge_first():
    side.wait_stream(main)
    launch(2, side)
    launch(1, main)
    main.wait_stream(side)


for _ in range(50): launch(0, main)
print(f"{lines}+16 lines x {Nx} points, {ts} sweeps per pass (A_ii = the centre coefficient)")
print(f"whole block, uniform line blocks (part 0): {med(lambda: launch(0, main)):7.1f} us")
print(f"interior line blocks only        (part 1): {med(lambda: launch(1, main)):7.1f} us")
for prio in (0, 1, 0, 1):
    check(lib().gla_set_option(b"edge_prio", prio))
    t2 = med(lambda: launch(2, main))
    tb, tf = med(both), med(both_edge_first)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    side.wait_stream(main); e0.record(side); launch(1, main); launch(2, side); e1.record(side); e2.record(main); torch.cuda.synchronize()
    print(f"option edge_prio = {prio}: edge blocks alone {t2:6.1f} us | interior on main || edge on a second stream {tb:6.1f} us "
          f"(edge first: {tf:6.1f}) | concurrent: the edge launch took {e0.elapsed_time(e1) * 1e3:6.1f} us, the interior {e0.elapsed_time(e2) * 1e3:6.1f} us")
check(lib().gla_set_option(b"edge_prio", 1))
