"""CPU, build container only: oracle/reference_port.py against the UNMODIFIED reference modules imported live
(oracle/ref_harness.py) on fresh seeds -- skipped on the GPU box, where /root/reference does not exist."""
import pytest
import torch

import ref_harness
import reference_port as rp

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_harness.load_reference()


@pytest.mark.parametrize("N,seed", [(6, 1), (11, 2)])
def test_port_equals_reference_live(ref, N, seed):
    gen = torch.Generator().manual_seed(seed)
    ei, ev = ref.UtilsGNN.laplacianfun_torch(N)
    ei_p, ev_p = rp.laplacianfun_torch(N)
    assert torch.equal(ei, ei_p) and torch.equal(ev, ev_p)
    n = N * N
    b = torch.rand(n, 1, generator=gen, dtype=torch.float64)
    x = torch.rand(n, 1, generator=gen, dtype=torch.float64)
    diag = -4 * torch.ones(n, 1, dtype=torch.float64)
    ea2 = torch.cat([ev, torch.zeros_like(ev)], 1)
    g = torch.tensor([0.7], dtype=torch.float64)
    assert torch.equal(ref.JacobiGNN.JacobiGNN()(13, torch.cat([diag, b, x], 1), ei, ea2, g),
                       rp.jacobi_forward(13, torch.cat([diag, b, x], 1), ei, ea2, g))
    gch = torch.tensor([-3.4, -4.0], dtype=torch.float64)
    for deg in (1, 5):
        a = ref.ChebyGNN.ChebyRelaxGNN(deg)(torch.cat([b, x], 1), ei, ev, gch)
        bb = rp.cheby_forward(deg, torch.cat([b, x], 1), ei, ev, gch)
        assert all(torch.equal(u, v) for u, v in zip(a, bb))
    a = ref.PowerMethodGNN.PowerMethodGNN(4)(torch.cat([b, torch.zeros_like(b)], 1), ei, ea2,
                                             torch.zeros(3, dtype=torch.float64), torch.zeros(n))
    bb = rp.power_forward(4, torch.cat([b, torch.zeros_like(b)], 1), ei, ea2, torch.zeros(3, dtype=torch.float64))
    assert all(torch.equal(u, v) for u, v in zip(a, bb))
    eio, evo = ref.UtilsGNN.remove_diag_entries(ei, ev)
    evr = -(torch.rand(eio.shape[1], 1, generator=gen, dtype=torch.float64) + 0.05)
    S = ref.SOCClassicGNN.SOCClassicGNN(0.3)(torch.zeros(n, 1, dtype=torch.float64), eio, evr)
    assert torch.equal(S, rp.soc_classic_forward(0.3, torch.zeros(n, 1, dtype=torch.float64), eio, evr))
    split = torch.zeros(n, 1, dtype=torch.float64)
    split[0::2] = 1
    ea = torch.hstack([evr, (S.reshape(-1, 1) > 0).to(torch.float64)])
    w = ref.DirectInterpGNN.DirectInterpGNN()(torch.hstack([diag, split]), eio, ea, None)
    w_p = rp.direct_interp_forward(torch.hstack([diag, split]), eio, ea)
    assert torch.equal(torch.nan_to_num(w, nan=7.0), torch.nan_to_num(w_p, nan=7.0))
    assert torch.equal(ref.GNNResidual.GNNResidual()(torch.cat([b, x], 1), ei, ev),
                       rp.residual_forward(torch.cat([b, x], 1), ei, ev))


@pytest.mark.parametrize("N,seed", [(5, 3), (9, 4)])
def test_trainable_port_equals_reference_live(ref, N, seed):
    """reference_port.trainable_* against the unmodified TrainableJacobiDiag modules on fresh seeds (fp32)."""
    import importlib
    import sys
    tdir = "/root/reference/pytorch/TrainableJacobiDiag"
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    TJ = importlib.import_module("TrainableJacobiGNN")
    L = importlib.import_module("loss")
    gen = torch.Generator().manual_seed(seed)
    ei, ev = ref.UtilsGNN.laplacianfun_torch(N)
    n = N * N
    s = 0.5 + torch.rand(n, generator=gen)
    A = (-torch.sparse_coo_tensor(ei, ev.flatten()).to_dense().float()) * s.reshape(-1, 1) * s.reshape(1, -1)
    mask = (A != 0) & ~torch.eye(n, dtype=torch.bool)
    eio = mask.nonzero().T.contiguous()
    ea = A[eio[0], eio[1]].reshape(-1, 1)
    va = torch.diag(A).reshape(-1, 1).clone()
    assert torch.equal(TJ.edge_to_vertex_aggregation(eio, ea, n), rp.trainable_edge_stats(eio, ea, n))
    torch.manual_seed(seed)
    gnn = TJ.get_model()
    d_ref, _, _ = gnn(va, eio, ea, batch=torch.zeros(n))
    torch.testing.assert_close(rp.trainable_model(gnn.state_dict(), va, eio, ea), d_ref, rtol=2e-6, atol=0)
    As = A.to_sparse().coalesce()
    d = d_ref.detach()
    T_ref = L.build_error_matrix(As, d, 2. / 3.)
    T_p = rp.trainable_error_matrix(As.indices(), As.values(), n, d, 2. / 3.)
    assert torch.equal(T_ref.indices(), T_p.indices()) and torch.equal(T_ref.values(), T_p.values())
    torch.manual_seed(seed + 10)
    Y = L.get_random_on_sphere(n, 20)
    torch.manual_seed(seed + 10)
    df_ref = L.eigval_approx(3, 20, T_ref, method='uniform')
    assert torch.equal(df_ref, rp.trainable_eigval_approx(3, T_p, Y))
    if int(n ** 0.5) ** 2 == n:
        torch.manual_seed(seed + 20)
        Yh = L.get_random_high_freq(n, 20, None)
        torch.manual_seed(seed + 20)
        assert torch.equal(Yh, rp.trainable_high_freq(n, 20))
