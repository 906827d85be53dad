#!/usr/bin/env python3
"""A/B of the K > 1 kernels on a stencil plan, same process, interleaved rounds:
   csr   = CSR row-tile kernel (spmm_stream_vec2_kernel; gla_set_option("spmm_variant", 9))
   d2d4  = 2-D blocked LDS-window kernel, 4-line blocks (default, spmm_dia2d_kernel<..., BJ = 4>)
   d2d8  = the same with 8-line blocks (spmm_variant 10)
   mfma  = (--mfma, fp64 K = 16) 4-line blocks with the products on v_mfma_f64_16x16x4_f64 (spmm_variant 12; an
           experiment: not bit-identical, compared at 1e-13; no reducing epilogue, so power_step is skipped for it)
Outputs are compared bit for bit.  Usage: python tools/ab_spmm_dia2d.py [N=4096] [K=16] [fem]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd._lib import lib, check   # noqa: E402
from gnn_applied_linear_algebra_amd.plan import col_dot   # noqa: E402

args = [a for a in sys.argv[1:] if a.isdigit()]
N = int(args[0]) if args else 4096
K = int(args[1]) if len(args) > 1 else 16
dev = torch.device("cuda:0")
if "fem" in sys.argv:
    ei, ev = gla.UtilsGNN.heateqnfem2dfun([N + 1, N + 1], [1.0, 1.0], device=dev)
else:
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
n, nnz = N * N, ei.shape[1]
plan = gla.CsrPlan(ei, n)
plan.set_values(ev)
print(f"N={N} K={K} nnz={nnz} dia={plan.info.dia}")
gen = torch.Generator().manual_seed(0)
X = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
B = torch.rand(n, K, generator=gen, dtype=torch.float64).to(dev)
diag = torch.full((n,), -4.0, dtype=torch.float64, device=dev)
base = 12 * nnz + 4 * (n + 1)
nrm = torch.sqrt(col_dot(X, X))
work = [None]


def mk():
    return torch.empty_like(X)


def run_spmm(out):
    plan.spmm(X, out=out[0])


def run_jacobi(out):
    plan.jacobi_sweep(diag, B, X, out[0], 0.7)


def run_cheby(out):
    out[1].copy_(B)
    out[2].copy_(X)
    plan.cheby_next(X, out[0], out[1], out[2], -0.25, -0.26, 0.1)


def run_power(out):
    _, ss, work[0] = plan.power_step(X, nrm=nrm, out=out[0], work=work[0], sumsq=out[3])


ops = {"spmm": (run_spmm, base + 16 * n * K), "jacobi": (run_jacobi, base + 8 * n + 24 * n * K),
       "cheby_next": (run_cheby, base + 48 * n * K), "power_step": (run_power, base + 16 * n * K)}
variants = {"csr": 9, "d2d4": 0, "d2d8": 10}
if "--mfma" in sys.argv and K == 16:
    variants["mfma"] = 12
outs = {v: [mk(), mk(), mk(), torch.empty(K, dtype=torch.float64, device=dev)] for v in variants}
res = {}
for rnd in range(7):
    for name, (fn, nb) in ops.items():
        for v, code in variants.items():
            check(lib().gla_set_option(b"spmm_variant", code))
            if name == "cheby_next":                       # the in-place operands are reset outside the timed part
                outs[v][1].copy_(B)
                outs[v][2].copy_(X)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.cheby_next(X, outs[v][0], outs[v][1], outs[v][2], -0.25, -0.26, 0.1)
                e1.record()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(outs[v])
                e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault((name, v), []).append(e0.elapsed_time(e1) * 1e3)
        exact = [v for v in variants if not v.startswith("mfma")]
        same = all(torch.equal(outs["csr"][i], outs[v][i]) for v in exact for i in range(3 if name == "cheby_next" else 1))
        if rnd == 1:
            print(f"{name}: outputs bit-identical across {exact}: {same}")
            for mv in [v for v in variants if v.startswith("mfma")]:
                if name == "power_step":
                    continue
                rng = range(3 if name == "cheby_next" else 1)
                err = max(float(((outs[mv][i] - outs["csr"][i]).abs().max() / outs["csr"][i].abs().max())) for i in rng)
                nbits = sum(int((outs[mv][i] != outs["csr"][i]).sum()) for i in rng)
                print(f"{name}: {mv} max abs diff / max |value| vs csr {err:.2e}, elements that differ: {nbits}")
            if name == "power_step":
                rel = max(float(((outs[v][3] - outs['csr'][3]).abs() / outs['csr'][3].abs()).max()) for v in variants)
                print(f"power_step: sumsq max rel diff vs csr: {rel:.2e}")
check(lib().gla_set_option(b"spmm_variant", 0))
for (name, v), ts in sorted(res.items()):
    ts.sort()
    med = ts[len(ts) // 2]
    nb = ops[name][1]
    print(f"{name:12s} {v:5s} K={K}: {med:8.1f} us  {nb / med / 1e3:7.1f} GB/s ({nb / med / 8e6 * 100:5.1f}% of 8 TB/s)")
