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
