#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference modules (TEST INFRASTRUCTURE).

Runs only in the build container (needs /root/reference).  The fixtures hold numeric
inputs and the reference's outputs, nothing else; seeds are fixed here because the
reference's demos draw unseeded torch.rand.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 24601                         # matlab/test_jacobi.m:52 uses rng(24601,'twister')
SIZES = (3, 4, 5, 8, 16)


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def rand(gen, *shape, dtype=torch.float64):
    return torch.rand(*shape, generator=gen, dtype=dtype)


def checkerboard(n):
    s = torch.zeros(n, dtype=torch.int64)
    s[0::2] = 1
    return s


def main():
    ref = ref_harness.load_reference()
    os.makedirs(OUT, exist_ok=True)
    U = ref.UtilsGNN
    out = {}

    # ---- hand-written known-answer cases from the reference's own __main__ blocks ------
    ei3 = torch.tensor([[0, 1], [1, 0], [1, 2], [2, 1], [0, 0], [1, 1], [2, 2]], dtype=torch.long).T
    A3 = torch.tensor([1, 1, 2, 3, 10, 10, 10], dtype=torch.float64).reshape(-1, 1)   # MatVecGNN.py:118-137
    x3 = torch.tensor([[1.0], [10.0], [100.0]], dtype=torch.float64)
    x3b = torch.tensor([[1.0, 7], [10, 70], [100, 700]], dtype=torch.float64)
    mv = ref.MetaLayer(ref.MatVecGNN.EdgeUpdate(), ref.MatVecGNN.VertexUpdate(ref.MatVecGNN.edge_to_vertex_aggregation))
    for tag, xx in (("k1", x3), ("k2", x3b)):
        va, ea, _ = mv(xx, ei3, A3, None, batch=torch.zeros(3))
        out[f"kat3_matvec_{tag}_x"] = npy(xx)
        out[f"kat3_matvec_{tag}_va"] = npy(va)
        out[f"kat3_matvec_{tag}_ea"] = npy(ea)
    out["kat3_ei"] = npy(ei3)
    out["kat3_A"] = npy(A3)
    b3 = torch.sparse_coo_tensor(ei3, A3.flatten()).to_dense() @ x3
    out["kat3_residual_r"] = npy(ref.GNNResidual.GNNResidual()(torch.cat([b3, x3], 1), ei3, A3))
    out["kat3_residual_b"] = npy(b3)

    for N in SIZES + (64,):
        ei, ev = U.laplacianfun_torch(N)
        out[f"lap{N}_ei"] = npy(ei).astype(np.int32)
        out[f"lap{N}_ev"] = npy(ev)
        eio, evo = U.remove_diag_entries(ei, ev)
        out[f"lap{N}_od_ei"] = npy(eio).astype(np.int32)
        out[f"lap{N}_od_ev"] = npy(evo)

    for N in SIZES:
        gen = torch.Generator().manual_seed(SEED + N)
        n = N * N
        ei, ev = U.laplacianfun_torch(N)
        eio, evo = U.remove_diag_entries(ei, ev)
        nnz = ei.shape[1]
        b = rand(gen, n, 1)
        x = rand(gen, n, 1)
        X16 = rand(gen, n, 16)
        out[f"N{N}_b"], out[f"N{N}_x"], out[f"N{N}_X16"] = npy(b), npy(x), npy(X16)
        batch = torch.zeros(n)

        # unsorted edge order (random permutation) exercises the stable COO->CSR path
        perm = torch.randperm(nnz, generator=gen)
        out[f"N{N}_perm"] = npy(perm).astype(np.int32)

        # M: MatVec k in {1,2,16}
        for k, X in ((1, x), (2, X16[:, :2].contiguous()), (16, X16)):
            va, ea, _ = mv(X, ei, ev, None, batch=batch)
            out[f"N{N}_matvec_k{k}_va"] = npy(va)
            if k <= 2:
                out[f"N{N}_matvec_k{k}_ea"] = npy(ea)
        va, ea, _ = mv(x, ei[:, perm], ev[perm], None, batch=batch)
        out[f"N{N}_matvec_perm_va"] = npy(va)
        out[f"N{N}_matvec_perm_ea"] = npy(ea)

        # R
        out[f"N{N}_residual_r"] = npy(ref.GNNResidual.GNNResidual()(torch.cat([b, x], 1), ei, ev))

        # J: w as fp32 tensor (the demo's dtype, JacobiGNN.py:170) and as fp64
        diag = -4 * torch.ones((n, 1), dtype=torch.float64)
        va0 = torch.cat([diag, b, x], 1)
        ea0 = torch.cat([ev, torch.zeros_like(ev)], 1)
        jac = ref.JacobiGNN.JacobiGNN()
        for wtag, g in (("w32", torch.tensor(0.7).reshape(-1)), ("w64", torch.tensor(0.7, dtype=torch.float64).reshape(-1))):
            for it in (1, 10, 20):
                out[f"N{N}_jacobi_{wtag}_it{it}_x"] = npy(jac(it, va0, ei, ea0, g))
        va1, ea1, g1 = jac.iterate(va0, ei, ea0, torch.tensor(0.7, dtype=torch.float64).reshape(-1))
        out[f"N{N}_jacobi_iterate_va"], out[f"N{N}_jacobi_iterate_ea"] = npy(va1), npy(ea1)
        out[f"N{N}_jacobi_perm_it10_x"] = npy(jac(10, va0, ei[:, perm], ea0[perm], torch.tensor(0.7, dtype=torch.float64).reshape(-1)))

        # CH: demo eigen-bounds (ChebyGNN.py:365-368) in fp64
        lmax, lmin = -7.46, -0.54
        gch = torch.tensor([(lmax - lmin) / 2, (lmax + lmin) / 2], dtype=torch.float64)
        out[f"N{N}_cheby_g0"] = npy(gch)
        for deg in (0, 1, 2, 3, 4, 8):
            va, ea, g = ref.ChebyGNN.ChebyRelaxGNN(deg)(torch.cat([b, x], 1), ei, ev, gch)
            out[f"N{N}_cheby_deg{deg}_va"] = npy(va)
            out[f"N{N}_cheby_deg{deg}_ea"] = npy(ea)
            out[f"N{N}_cheby_deg{deg}_g"] = npy(g)

        # P
        for it in (1, 10):
            va, ea, g = ref.PowerMethodGNN.PowerMethodGNN(it)(
                torch.cat([b, torch.zeros_like(b)], 1), ei, torch.cat([ev, torch.zeros_like(ev)], 1),
                torch.zeros(3, dtype=torch.float64), batch)
            out[f"N{N}_power_it{it}_va"] = npy(va)
            out[f"N{N}_power_it{it}_ea"] = npy(ea)
            out[f"N{N}_power_it{it}_g"] = npy(g)

        # SO: theta 0.25 on the Laplacian and on a randomly weighted off-diagonal (so flags vary)
        soc = ref.SOCClassicGNN.SOCClassicGNN(0.25)
        out[f"N{N}_soc_S"] = npy(soc(torch.zeros((n, 1), dtype=torch.float64), eio, evo))
        evr = -(rand(gen, eio.shape[1], 1) + 0.01) * 2
        evr[::7] *= -1                       # a few positive off-diagonals
        out[f"N{N}_socr_ev"] = npy(evr)
        out[f"N{N}_socr_S"] = npy(soc(torch.zeros((n, 1), dtype=torch.float64), eio, evr))

        # D: checkerboard splitting (matlab/test_direct_interpolation.m:63-65)
        split = checkerboard(n)
        Sflag = (soc(torch.zeros((n, 1), dtype=torch.float64), eio, evr).reshape(-1, 1) > 0).to(torch.float64)
        for tag, ea_in in (("lap", torch.hstack([evo, torch.ones_like(evo)])), ("rnd", torch.hstack([evr, Sflag]))):
            w = ref.DirectInterpGNN.DirectInterpGNN()(torch.hstack([diag, split.reshape(-1, 1).to(torch.float64)]), eio, ea_in, None)
            out[f"N{N}_di_{tag}_w"] = npy(w)
        out[f"N{N}_split"] = npy(split)

        # next tier: SA SOC and matrix-weighted norm
        _, ea, _ = ref.MetaLayer(ref.SOCSAGNN.EdgeUpdate())(diag, eio, evo, batch=batch)
        out[f"N{N}_socsa_ea"] = npy(ea)
        M = ref.MatrixWeightedNorm
        _, _, gn = ref.MetaLayer(M.EdgeUpdate(), M.VertexUpdate(M.EdgeToVertexAggregation),
                                 M.GlobalUpdate(M.VertexToGlobalAggregation))(x, ei, -ev, torch.zeros(1), batch=batch)
        out[f"N{N}_wnorm"] = npy(gn)

    # ---- V-cycle script (fp32, N=5, checkerboard stand-in splitting) -------------------
    g = ref_harness.run_vcycle_script(SEED, torch)
    torch.manual_seed(SEED)
    x0 = torch.rand((25, 1), dtype=torch.float).reshape(-1, 1)      # same draws as VCycle.py:245-246
    b0 = torch.rand((25, 1), dtype=torch.float).reshape(-1, 1)
    assert torch.equal(b0, g["b"])
    xs, hist = [], []
    xk = x0
    for _ in range(5):
        xk = g["runVCycle"](g["A"], b0, xk, 3, 3, 5, True)
        xs.append(npy(xk))
        hist.append(float(torch.norm(g["runResidual"](g["A"], b0, xk))))
    assert torch.equal(xk, g["x"])
    out["vcycle_x0"], out["vcycle_b"] = npy(x0), npy(b0)
    out["vcycle_xs"] = np.stack(xs)
    out["vcycle_resnorm"] = np.array(hist)

    np.savez_compressed(os.path.join(OUT, "reference_vectors.npz"), **out)
    print(f"wrote {len(out)} arrays to {OUT}/reference_vectors.npz "
          f"({os.path.getsize(os.path.join(OUT, 'reference_vectors.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
