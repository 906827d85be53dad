"""CPU: oracle/reference_port.py must reproduce the reference's golden vectors BIT-EXACTLY.

The goldens were produced by the unmodified reference modules (oracle/gen_golden.py); the port
uses the same torch CPU primitives in the same order, so equality is exact, not approximate.
"""
import os

import numpy as np
import pytest
import torch

import reference_port as rp
from conftest import GOLDEN_SIZES


def t(a, dtype=None):
    x = torch.from_numpy(np.array(a))
    return x.to(dtype) if dtype is not None else x


def eq(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_array_equal(a, np.asarray(b))


def lap(golden, N, od=False):
    tag = f"lap{N}_od" if od else f"lap{N}"
    return t(golden[tag + "_ei"], torch.long), t(golden[tag + "_ev"])


@pytest.mark.parametrize("N", GOLDEN_SIZES + (64,))
def test_builders(golden, N):
    ei, ev = rp.laplacianfun_torch(N)
    assert ei.dtype == torch.long and ev.dtype == torch.float64 and ev.shape == (5 * N * N - 4 * N, 1)
    eq(ei, golden[f"lap{N}_ei"])
    eq(ev, golden[f"lap{N}_ev"])
    eio, evo = rp.remove_diag_entries(ei, ev)
    eq(eio, golden[f"lap{N}_od_ei"])
    eq(evo, golden[f"lap{N}_od_ev"])


def test_kat_3x3(golden):
    ei, A = t(golden["kat3_ei"]), t(golden["kat3_A"])
    for tag in ("k1", "k2"):
        va, ea, _ = rp.matvec_layer(t(golden[f"kat3_matvec_{tag}_x"]), ei, A)
        eq(va, golden[f"kat3_matvec_{tag}_va"])
        eq(ea, golden[f"kat3_matvec_{tag}_ea"])
    # the reference's printed known answer, MatVecGNN.py:118-162
    eq(rp.matvec_layer(t(golden["kat3_matvec_k1_x"]), ei, A)[0][:, 1], [20.0, 301.0, 1030.0])
    r = rp.residual_forward(torch.cat([t(golden["kat3_residual_b"]), t(golden["kat3_matvec_k1_x"])], 1), ei, A)
    eq(r, golden["kat3_residual_r"])
    eq(r, np.zeros((3, 1)))


@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_matvec_residual(golden, N):
    ei, ev = lap(golden, N)
    x, b, X16 = t(golden[f"N{N}_x"]), t(golden[f"N{N}_b"]), t(golden[f"N{N}_X16"])
    for k, X in ((1, x), (2, X16[:, :2].contiguous()), (16, X16)):
        va, ea, _ = rp.matvec_layer(X, ei, ev)
        eq(va, golden[f"N{N}_matvec_k{k}_va"])
        if k <= 2:
            eq(ea, golden[f"N{N}_matvec_k{k}_ea"])
    perm = t(golden[f"N{N}_perm"], torch.long)
    va, ea, _ = rp.matvec_layer(x, ei[:, perm], ev[perm])
    eq(va, golden[f"N{N}_matvec_perm_va"])
    eq(ea, golden[f"N{N}_matvec_perm_ea"])
    eq(rp.residual_forward(torch.cat([b, x], 1), ei, ev), golden[f"N{N}_residual_r"])


@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_jacobi(golden, N):
    ei, ev = lap(golden, N)
    n = N * N
    va0 = torch.cat([-4 * torch.ones((n, 1), dtype=torch.float64), t(golden[f"N{N}_b"]), t(golden[f"N{N}_x"])], 1)
    ea0 = torch.cat([ev, torch.zeros_like(ev)], 1)
    w64 = torch.tensor(0.7, dtype=torch.float64).reshape(-1)
    for wtag, g in (("w32", torch.tensor(0.7).reshape(-1)), ("w64", w64)):
        for it in (1, 10, 20):
            eq(rp.jacobi_forward(it, va0, ei, ea0, g), golden[f"N{N}_jacobi_{wtag}_it{it}_x"])
    va1, ea1, _ = rp.jacobi_iterate(va0, ei, ea0, w64)
    eq(va1, golden[f"N{N}_jacobi_iterate_va"])
    eq(ea1, golden[f"N{N}_jacobi_iterate_ea"])
    perm = t(golden[f"N{N}_perm"], torch.long)
    eq(rp.jacobi_forward(10, va0, ei[:, perm], ea0[perm], w64), golden[f"N{N}_jacobi_perm_it10_x"])


@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_cheby(golden, N):
    ei, ev = lap(golden, N)
    va0 = torch.cat([t(golden[f"N{N}_b"]), t(golden[f"N{N}_x"])], 1)
    for deg in (0, 1, 2, 3, 4, 8):
        va, ea, g = rp.cheby_forward(deg, va0, ei, ev, t(golden[f"N{N}_cheby_g0"]))
        eq(va, golden[f"N{N}_cheby_deg{deg}_va"])
        eq(ea, golden[f"N{N}_cheby_deg{deg}_ea"])
        eq(g, golden[f"N{N}_cheby_deg{deg}_g"])


@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_power(golden, N):
    ei, ev = lap(golden, N)
    b = t(golden[f"N{N}_b"])
    for it in (1, 10):
        va, ea, g = rp.power_forward(it, torch.cat([b, torch.zeros_like(b)], 1), ei,
                                     torch.cat([ev, torch.zeros_like(ev)], 1), torch.zeros(3, dtype=torch.float64))
        eq(va, golden[f"N{N}_power_it{it}_va"])
        eq(ea, golden[f"N{N}_power_it{it}_ea"])
        eq(g, golden[f"N{N}_power_it{it}_g"])


@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_soc_di(golden, N):
    eio, evo = lap(golden, N, od=True)
    n = N * N
    z = torch.zeros((n, 1), dtype=torch.float64)
    eq(rp.soc_classic_forward(0.25, z, eio, evo), golden[f"N{N}_soc_S"])
    # SURVEY sanity value: every off-diagonal of the Laplacian gives S = 1/1 - 0.25
    assert np.all(golden[f"N{N}_soc_S"] == 0.75)
    evr = t(golden[f"N{N}_socr_ev"])
    S = rp.soc_classic_forward(0.25, z, eio, evr)
    eq(S, golden[f"N{N}_socr_S"])
    diag = -4 * torch.ones((n, 1), dtype=torch.float64)
    split = t(golden[f"N{N}_split"]).reshape(-1, 1).to(torch.float64)
    Sflag = (S.reshape(-1, 1) > 0).to(torch.float64)
    for tag, ea_in in (("lap", torch.hstack([evo, torch.ones_like(evo)])), ("rnd", torch.hstack([evr, Sflag]))):
        w = rp.direct_interp_forward(torch.hstack([diag, split]), eio, ea_in)
        eq(w, golden[f"N{N}_di_{tag}_w"])          # assert_array_equal treats NaN == NaN
    _, ea, _ = rp.socsa_layer(diag, eio, evo)
    eq(ea, golden[f"N{N}_socsa_ea"])
    ei, ev = lap(golden, N)
    _, _, gn = rp.matrix_weighted_norm_layer(t(golden[f"N{N}_x"]), ei, -ev, torch.zeros(1))
    eq(gn, golden[f"N{N}_wnorm"])


def test_di_known_answer(golden):
    """N=5 checkerboard -> F-row weights all 0.25, 40 NaNs, all on C rows."""
    w = golden["N5_di_lap_w"]
    rows = golden["lap5_od_ei"][0]
    split = golden["N5_split"]
    assert np.isnan(w).sum() == 40 and np.all(split[rows[np.isnan(w)]] == 1)
    assert np.all(w[split[rows] == 0] == 0.25)


def test_vcycle_script(golden):
    """The fp32 two-grid script (VCycle.py:239-277) with the stand-in splitting, 5 cycles."""
    ei, ev = rp.laplacianfun_torch(5)
    A = torch.sparse_coo_tensor(ei, ev.flatten(), dtype=torch.float)
    b, x = t(golden["vcycle_b"]), t(golden["vcycle_x0"])
    split = torch.zeros(25, dtype=torch.int64)
    split[0::2] = 1
    for k in range(5):
        x = rp.vcycle(A, b, x, split)
        eq(x, golden["vcycle_xs"][k])
    res = golden["vcycle_resnorm"]
    assert res[0] < 0.2 and res[-1] < 1e-5        # converges like the script's printed history


def test_fem_known_answer():
    """heateqnfem2dfun.m:105-110: interior stencil (1/6a)[4(2a^2+2), 2(-2a^2+1) x, 2(a^2-2) y, (-a^2-1) corners]."""
    for alpha, m in ((1.0, 6), (4.0, 5)):
        ei, ev = rp.heateqnfem2dfun([m + 1, m + 1], [1.0, alpha], [2, 2])
        n = m * m
        assert ei.shape[1] == (3 * m - 2) ** 2
        A = torch.sparse_coo_tensor(ei, ev.flatten(), (n, n)).to_dense()
        k = (m // 2) * m + m // 2
        a2, s = alpha * alpha, 1 / (6 * alpha)
        np.testing.assert_allclose(A[k, k], s * 4 * (2 * a2 + 2), rtol=1e-14)
        np.testing.assert_allclose(A[k, k + 1], s * 2 * (-2 * a2 + 1), rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(A[k, k + m], s * 2 * (a2 - 2), rtol=1e-14)
        np.testing.assert_allclose(A[k, k + m + 1], s * (-a2 - 1), rtol=1e-14)
        assert torch.equal(A, A.T)
    ei, ev = rp.heateqnfem2dfun([7, 7], [1.0, 1.0])
    A = torch.sparse_coo_tensor(ei, ev.flatten(), (36, 36)).to_dense()
    np.testing.assert_allclose(A[14, 14], 8 / 3, rtol=1e-14)      # alpha = 1 -> 8/3, -1/3 x 8
    np.testing.assert_allclose(A[14, [7, 8, 9, 13, 15, 19, 20, 21]], -1 / 3, rtol=1e-14)


# ---------------------------------------------------------------------------------------------------
# the C restatement (oracle/gla_oracle.c) against the same golden vectors: bit-exact
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", GOLDEN_SIZES)
def test_c_oracle_golden(golden, N):
    import c_oracle
    n = N * N
    ei, ev = golden[f"lap{N}_ei"].astype(np.int64), golden[f"lap{N}_ev"]
    perm = golden[f"N{N}_perm"].astype(np.int64)
    b, x, X16 = golden[f"N{N}_b"], golden[f"N{N}_x"], golden[f"N{N}_X16"]
    for e_i, e_v, tag in ((ei, ev, ""), (ei[:, perm], ev[perm], "perm_")):
        A = c_oracle.Csr(e_i, e_v, n)
        eq(A.spmm(x), golden[f"N{N}_matvec_{tag or 'k1_'}va"][:, 1:])
    A = c_oracle.Csr(ei, ev, n)
    eq(A.spmm(X16), golden[f"N{N}_matvec_k16_va"][:, 16:])
    diag = -4 * np.ones(n)
    for it in (1, 10, 20):
        eq(A.jacobi(diag, b, x, 0.7, it), golden[f"N{N}_jacobi_w64_it{it}_x"])
        eq(A.jacobi(diag, b, x, float(np.float32(0.7)), it), golden[f"N{N}_jacobi_w32_it{it}_x"])
    c, d = golden[f"N{N}_cheby_g0"]
    for deg in (1, 2, 3, 4, 8):
        xo, r, p, g = A.cheby(b, x, c, d, deg)
        va = golden[f"N{N}_cheby_deg{deg}_va"]
        eq(xo, va[:, 1]); eq(r, va[:, 2]); eq(p, va[:, 3])
        eq(g[:len(golden[f"N{N}_cheby_deg{deg}_g"])], golden[f"N{N}_cheby_deg{deg}_g"])
    eio = golden[f"lap{N}_od_ei"].astype(np.int64)
    Aod = c_oracle.Csr(eio, golden[f"N{N}_socr_ev"], n)
    v, S = Aod.soc_classic(0.25)
    eq(S, golden[f"N{N}_socr_S"])
    w = Aod.direct_interp((S > 0).astype(np.float64), diag, golden[f"N{N}_split"].astype(np.float64))
    eq(w, golden[f"N{N}_di_rnd_w"])


def test_fem_bc1_structure():
    """bcs=[1,1] (heateqnfem2dfun.m:143-161): boundary nodes are identity rows/columns, interior block = bcs [2,2]."""
    m = 5
    ei, ev = rp.heateqnfem2dfun([m + 1, m + 1], [1.0, 2.0], [1, 1])
    nx = m + 2
    A = torch.sparse_coo_tensor(ei, ev.flatten(), (nx * nx, nx * nx)).to_dense()
    k = torch.arange(nx * nx)
    bnd = (k % nx == 0) | (k % nx == m + 1) | (k // nx == 0) | (k // nx == m + 1)
    assert torch.equal(A[bnd][:, bnd], torch.eye(int(bnd.sum()), dtype=torch.float64))
    assert A[bnd][:, ~bnd].abs().sum() == 0 and A[~bnd][:, bnd].abs().sum() == 0
    ei2, ev2 = rp.heateqnfem2dfun([m + 1, m + 1], [1.0, 2.0], [2, 2])
    assert torch.equal(A[~bnd][:, ~bnd], torch.sparse_coo_tensor(ei2, ev2.flatten(), (m * m, m * m)).to_dense())


# ---------------------------------------------------------------------------------------------------
# fp64 V-cycle fixtures (the reference's own runVCycle driven in double precision, oracle/gen_golden_vcycle64.py)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vc64():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vcycle64_vectors.npz"))


@pytest.mark.parametrize("N", [5, 8, 16])
def test_vcycle_fp64_oracle_reproduces_reference(vc64, N):
    """5 two-grid cycles in fp64: the oracle restatement against the unmodified reference, iterate by iterate.
    Everything is the same arithmetic in the same order except torch's sparse products, so the bar is 1e-12."""
    ei, ev = rp.laplacianfun_torch(N)
    A = torch.sparse_coo_tensor(ei, ev.flatten(), dtype=torch.float64)
    b, x = torch.from_numpy(vc64[f"N{N}_b"]), torch.from_numpy(vc64[f"N{N}_x0"])
    split = torch.zeros(N * N, dtype=torch.int64)
    split[0::2] = 1
    saved = torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float64)
        for k in range(5):
            x = rp.vcycle(A, b, x, split)
            np.testing.assert_allclose(x.numpy(), vc64[f"N{N}_xs"][k], rtol=1e-12, atol=1e-14)
        ei_c, ea_c = rp.coo_to_gnn_input(A)
        res = float(torch.norm(rp.residual_forward(torch.cat([b, x], 1), ei_c, ea_c)))
    finally:
        torch.set_default_dtype(saved)
    np.testing.assert_allclose(res, vc64[f"N{N}_resnorm"][4], rtol=1e-9)


@pytest.mark.parametrize("k", [1, 2, 16])
def test_matvec_config1_size(vc64, k):
    """BASELINE config 1 at its stated size: MatVecGNN on the 64 x 64 Laplacian (MatVecGNN.py:146-150), bit-exact."""
    ei, ev = rp.laplacianfun_torch(64)
    X = torch.from_numpy(vc64["mv64_X16"][:, :k]).contiguous()
    va, ea, _ = rp.matvec_layer(X, ei, ev)
    np.testing.assert_array_equal(va.numpy(), vc64[f"mv64_k{k}_va"])
    if k <= 2:
        np.testing.assert_array_equal(ea.numpy(), vc64[f"mv64_k{k}_ea"])


# ---------------------------------------------------------------------------------------------------
# U4 (heateqnfem2dfun, MATLAB only): an INDEPENDENT known answer for the whole matrix, boundary rows included
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cx,cy,hx,hy", [(6, 5, 1.0, 1.0), (9, 7, 1.0, 4.0), (4, 12, 0.5, 0.125), (33, 17, 2.0, 3.0)])
def test_heateqnfem2dfun_equals_tensor_product_stiffness(cx, cy, hx, hy):
    """The element matrix of matlab/heateqnfem2dfun.m:71-100, (1/6a)[2a^2+2, -2a^2+1, a^2-2, -1-a^2] with a = hy/hx,
    is the Q1 stiffness matrix of -Laplace on hx x hy rectangles, and a Q1 stiffness matrix on a tensor-product mesh
    IS K = M_y (x) K_x + K_y (x) M_x with the 1-D matrices K = tridiag(-1, 2, -1)/h, M = tridiag(1, 4, 1) h/6 (vertex
    k = j*nx + i).  Eliminating the Dirichlet boundary (BC type 2, :164-168) keeps the interior blocks of the 1-D matrices.
    This pins EVERY row of the restated builder -- rows next to the boundary too, which the file's own known answer
    (interior stencil, :105-110) does not -- through an assembly that shares no code with it; BC type 1 (:143-161) is then
    the same interior block plus identity rows."""
    import scipy.sparse as sp

    def tri(n, lo, di):
        return sp.diags([lo * np.ones(n - 1), di * np.ones(n), lo * np.ones(n - 1)], [-1, 0, 1], format="csr")

    a = hy / hx                                                    # only the ratio enters (2-D stiffness is scale-free)
    mx, my = cx - 1, cy - 1
    Kx, Mx = tri(mx, -1.0, 2.0), tri(mx, 1.0, 4.0) / 6.0           # hx = 1
    Ky, My = tri(my, -1.0, 2.0) / a, tri(my, 1.0, 4.0) * (a / 6.0)  # hy = a
    K_ref = (sp.kron(My, Kx, format="csr") + sp.kron(Ky, Mx, format="csr")).tocsr()
    K_ref.sum_duplicates()
    K_ref.eliminate_zeros()
    ei, ev = rp.heateqnfem2dfun([cx, cy], [hx, hy], bcs=(2, 2))
    n = mx * my
    K = sp.csr_matrix((ev.numpy().ravel(), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))
    assert K.nnz == (3 * mx - 2) * (3 * my - 2) == K_ref.nnz        # same pattern: 9-point, cut at the boundary
    assert (abs(K - K_ref) > 1e-14 * abs(K_ref).max()).nnz == 0
    assert abs(K - K.T).max() == 0.0                                # exactly symmetric
    # BC type 1: all (cx+1)(cy+1) vertices; boundary rows / columns are identity, the interior block is the matrix above
    ei1, ev1 = rp.heateqnfem2dfun([cx, cy], [hx, hy], bcs=(1, 1))
    nx, N = cx + 1, (cx + 1) * (cy + 1)
    K1 = sp.csr_matrix((ev1.numpy().ravel(), (ei1[0].numpy(), ei1[1].numpy())), shape=(N, N))
    k = np.arange(N)
    bnd = (k % nx == 0) | (k % nx == cx) | (k // nx == 0) | (k // nx == cy)
    assert abs(K1[~bnd][:, ~bnd] - K).max() == 0.0
    assert abs(K1[bnd][:, bnd] - sp.identity(int(bnd.sum()))).max() == 0.0
    assert K1[bnd][:, ~bnd].nnz == 0 and K1[~bnd][:, bnd].nnz == 0
