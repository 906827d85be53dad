#!/usr/bin/env python3
"""Learned-Jacobi tier timing: the Gelfand damping-factor loss (K = 3 products with m = 20
test vectors) forward + backward on the N x N variable-coefficient 5-point operator, per-kernel achieved GB/s, and the
same step on the host cores with torch sparse autograd (the reference's formulation, loss.py:92-128).
Usage: python tools/bench_trainable.py [N] [--cpu]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_applied_linear_algebra_amd as gla          # noqa: E402
from gnn_applied_linear_algebra_amd.TrainableJacobiDiag import loss as L, TrainableJacobiGNN as TJ   # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 2048
dev = torch.device("cuda:0")
m, K = 20, 3


def timeit(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def report(name, us, nbytes=None):
    extra = "" if nbytes is None else f"   {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / us / 8e6 * 100:5.1f}% of 8 TB/s)"
    print(f"{name:46s} {us:10.1f} us{extra}", flush=True)


for dtype, w in ((torch.float32, 4), (torch.float64, 8)):
    print(f"--- N={N}  dtype={dtype}  m={m}  K={K}")
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=dev)
    n, nnz = N * N, ei.shape[1]
    gen = torch.Generator(device=dev).manual_seed(5)
    s = 0.5 + torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
    vals = (-ev.flatten() * s[ei[0]] * s[ei[1]]).to(dtype)
    A = gla.autograd.SparseMatrix(ei, vals, (n, n))
    Y = torch.randn(n, m, generator=gen, device=dev, dtype=torch.float64).to(dtype)
    Y = Y / torch.linalg.vector_norm(Y, dim=0)
    diag = (4.0 * s * s).to(dtype).reshape(-1, 1).requires_grad_(True)

    # model inference pieces
    eio, evo = gla.UtilsGNN.remove_diag_entries(ei, vals.reshape(-1, 1))
    plan_od = gla.plan_for(eio, n)
    report("segment_stats (min/mean/sum/max, 1 pass)", timeit(lambda: plan_od.segment_stats(evo)),
           w * eio.shape[1] + 4 * (n + 1) + 4 * w * n)
    model = TJ.get_model().to(dev).to(dtype)
    va = (4.0 * s * s).to(dtype).reshape(-1, 1)
    report("model forward, autograd on (stats + torch MLP)", timeit(lambda: model(va, eio, evo, batch=None)))
    with torch.no_grad():
        report("model inference, no_grad (stats + gla_mlp_relu3)", timeit(lambda: model(va, eio, evo, batch=None)))

    # kernels of the loss
    T = L.build_error_matrix(A, diag.detach(), 2. / 3.)
    plan = gla.plan_for(T.indices(), n)
    plan.set_values_cached(T.values())
    Z = torch.empty_like(Y)
    base = (w + 4) * nnz + 4 * (n + 1)
    report(f"gla_spmm K={m} (forward product)", timeit(lambda: plan.spmm(Y, out=Z)), base + 2 * w * n * m)
    G = torch.randn_like(Y)
    report(f"gla_edge_dot K={m} (dL/dvalues)", timeit(lambda: plan.edge_dot(G, Y)),
           4 * nnz + 4 * (n + 1) + 2 * w * n * m + w * nnz)
    for var, label in ((7, "scalar-slice"),):
        gla._lib.check(gla._lib.lib().gla_set_option(b"spmm_variant", var))
        report(f"   ({label} variant of gla_edge_dot)", timeit(lambda: plan.edge_dot(G, Y)),
               4 * nnz + 4 * (n + 1) + 2 * w * n * m + w * nnz)
    gla._lib.check(gla._lib.lib().gla_set_option(b"spmm_variant", 0))
    pt = gla.autograd._transposed_plan(plan, T.indices())
    pt.set_values_cached(T.values())
    report(f"gla_spmm K={m} on A^T plan (dL/dX)", timeit(lambda: pt.spmm(G, out=Z)), base + 2 * w * n * m)

    def step():
        diag.grad = None
        df = L.eigval_approx(K, m, L.build_error_matrix(A, diag, 2. / 3.), Y=Y)
        df.backward()
        return df

    report("build_error_matrix (incl. pattern coalesce)", timeit(lambda: L.build_error_matrix(A, diag.detach(), 2. / 3.)))
    report("damping-factor loss forward + backward", timeit(step))
    df = step()
    print(f"    df = {df.item():.6f}   |grad|_max = {diag.grad.abs().max().item():.3e}")

    del ei, ev, A, T, plan, pt, Y, Z, G
    torch.cuda.empty_cache()

if "--cpu" in sys.argv:
    # The reference's formulation on the host cores.  torch's sparse autograd materialises a dense n x n product for
    # dL/dvalues (grad @ Y^T, then masked), so this leg only fits small graphs: N_cpu = 96 (n = 9216, 340 MB dense).
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import reference_port as rp
    Nc = 96
    ei, ev = gla.UtilsGNN.laplacianfun_torch(Nc, device=dev)
    n = Nc * Nc
    gen = torch.Generator(device=dev).manual_seed(5)
    s = 0.5 + torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
    vals = (-ev.flatten() * s[ei[0]] * s[ei[1]]).float()
    Y = torch.randn(n, m, generator=gen, device=dev, dtype=torch.float64).float()
    Y = Y / torch.linalg.vector_norm(Y, dim=0)
    diag = (4.0 * s * s).float().reshape(-1, 1).requires_grad_(True)
    A = gla.autograd.SparseMatrix(ei, vals, (n, n))

    def step_small():
        diag.grad = None
        df = L.eigval_approx(K, m, L.build_error_matrix(A, diag, 2. / 3.), Y=Y)
        df.backward()
        return df

    us = timeit(step_small)
    df = step_small()
    ei_c, vals_c, Y_c = ei.cpu(), vals.cpu(), Y.cpu()
    ts = []
    for _ in range(3):
        d_c = diag.detach().cpu().requires_grad_(True)
        t0 = time.perf_counter()
        df_c = rp.trainable_eigval_approx(K, rp.trainable_error_matrix(ei_c, vals_c, n, d_c, 2. / 3.), Y_c)
        df_c.backward()
        ts.append(time.perf_counter() - t0)
    print(f"--- N={Nc} fp32 loss forward + backward: GPU {us:.0f} us;  host cores ({torch.get_num_threads()} torch "
          f"threads), reference formulation: {min(ts) * 1e3:.1f} ms;  df {df.item():.6f} vs {df_c.item():.6f}, "
          f"grad max rel diff {((d_c.grad - diag.grad.cpu()).abs().max() / d_c.grad.abs().max()).item():.2e}")
