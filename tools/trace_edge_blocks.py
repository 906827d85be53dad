#!/usr/bin/env python3
"""Where and when do the workgroups of the edge launch run beside the interior launch?  (GLA_M5_VARIANTS build with
gla_debug_m5_stamps: every workgroup records its start / end wall clock, its CU (HW_ID) and XCD.)
Usage: GLA_LIB_PATH=.../libgla_hip_variants.so python tools/trace_edge_blocks.py [Nx=8192] [lines=1024] [ts=7]"""
import ctypes, os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla
from gnn_applied_linear_algebra_amd._lib import lib, check, ptr, dtype_code
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
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
L = lib()
dbg = ctypes.CDLL(gla._lib.LIB_PATH).gla_debug_m5_stamps
dbg.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
s_int = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)
s_edge = torch.zeros(4 * 1024, dtype=torch.int64, device=dev)


def launch(part, st):
    check(L.gla_jacobi_fused_parts(plan._h, code, None, ptr(b), ptr(x), ptr(out), 0.7, ts, part, T, ctypes.c_void_p(st.cuda_stream)))


for _ in range(30): launch(0, main)
torch.cuda.synchronize()
for prio in (1, 0):
    check(L.gla_set_option(b"edge_prio", prio))
    for mode in ("edge alone", "interior || edge"):
        dbg(s_int.data_ptr(), s_edge.data_ptr())
        s_int.zero_(); s_edge.zero_()
        torch.cuda.synchronize()
        if mode == "edge alone":
            launch(2, main)
        else:
            side.wait_stream(main); launch(1, main); launch(2, side); main.wait_stream(side)
        torch.cuda.synchronize()
        dbg(None, None)
        for name, buf in (("interior", s_int), ("edge", s_edge)):
            a = buf.cpu().reshape(-1, 4)
            a = a[a[:, 1] != 0]
            if a.shape[0] == 0:
                continue
            t0 = int(torch.cat([s_int.cpu().reshape(-1, 4), s_edge.cpu().reshape(-1, 4)])[:, 0][torch.cat([s_int.cpu().reshape(-1, 4), s_edge.cpu().reshape(-1, 4)])[:, 1] != 0].min())
            start, end = (a[:, 0] - t0).double() / 100.0, (a[:, 1] - t0).double() / 100.0       # us
            hw, xcc = a[:, 2] & 0xffffffff, (a[:, 2] >> 32) & 0xf
            cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
            cuid = [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(a.shape[0])]
            dur = end - start
            print(f"edge_prio={prio} [{mode}] {name:8s}: {a.shape[0]:4d} workgroups  start {float(start.min()):6.1f} .. {float(start.max()):6.1f} us  "
                  f"end {float(end.min()):6.1f} .. {float(end.max()):6.1f}  duration min {float(dur.min()):6.1f} median {float(dur.median()):6.1f} max {float(dur.max()):6.1f}  "
                  f"per XCD {sorted(collections.Counter(int(v) for v in xcc).items())}  distinct CUs {len(set(cuid))}")
            real = dur > 3.0
            per = collections.defaultdict(list)
            for i in range(a.shape[0]):
                if bool(real[i]):
                    per[int(xcc[i])].append(round(float(start[i]), 1))
            print("   real workgroups per XCD (count: start times of the last three):", {k: (len(v), sorted(v)[-3:]) for k, v in sorted(per.items())})
            if name == "edge" and mode != "edge alone":
                ia = s_int.cpu().reshape(-1, 4); ia = ia[ia[:, 1] != 0]
                ihw, ixcc = ia[:, 2] & 0xffffffff, (ia[:, 2] >> 32) & 0xf
                icu = collections.Counter((int(ixcc[i]), int((ihw[i] >> 13) & 7), int((ihw[i] >> 12) & 1), int((ihw[i] >> 8) & 0xf)) for i in range(ia.shape[0]))
                shared = [icu.get(c, 0) for c in cuid]
                late = [(round(float(start[i]), 1), round(float(dur[i]), 1), cuid[i], shared[i]) for i in torch.argsort(start, descending=True)[:12].tolist()]
                print("   interior workgroups on the CUs of the edge workgroups:", sorted(collections.Counter(shared).items()),
                      "; interior workgroups per CU overall:", sorted(collections.Counter(icu.values()).items()))
                print("   the 12 edge workgroups that started last (start us, duration us, (xcd, se, sh, cu), interior wgs on that CU):", late)
check(L.gla_set_option(b"edge_prio", 1))
