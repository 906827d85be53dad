"""GPU parity of the learned-Jacobi tier through the C ABI (gla_segment_stats, gla_spmm,
gla_edge_dot) and the TrainableJacobiDiag mirror:
  * golden vectors of the unmodified reference (tests/golden/trainable_vectors.npz): the 4-statistic aggregation
    BIT-EXACT; the fp32 model output, error matrix, Gelfand damping factor, its gradient and a whole loss_batch step
    within the fp32 tolerances written below (the reference's torch.sparse CPU product and this library order the
    fp32 additions of the 20-column products / gradients differently);
  * the CPU oracle (reference_port.trainable_*) on seeded larger graphs in fp64 at 1e-12;
  * size-independent property at 1024^2 x 20 columns: the adjoint identity <G, A X> = <A^T G, X> = sum_e a_e dot_e.
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("demo4", "lap4", "var4", "lap8", "var8")
gla = rp = TJ = L = DEV = None


@pytest.fixture(scope="module", autouse=True)
def _setup():
    global gla, rp, TJ, L, DEV
    import gnn_applied_linear_algebra_amd as _gla
    import gnn_applied_linear_algebra_amd.TrainableJacobiDiag as _tjd
    import reference_port as _rp
    assert torch.cuda.is_available()
    gla, rp, DEV = _gla, _rp, torch.device("cuda:0")
    TJ, L = _tjd.TrainableJacobiGNN, _tjd.loss
    yield


@pytest.fixture(scope="module")
def tg():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "trainable_vectors.npz")))


def T(a):
    return torch.from_numpy(np.asarray(a))


def load_case(tg, name):
    va, ei, ea = T(tg[f"{name}_va"]), T(tg[f"{name}_ei"]).long(), T(tg[f"{name}_ea"])
    params = {k[len(name) + 7:]: T(v) for k, v in tg.items() if k.startswith(f"{name}_param_")}
    return va, ei, ea, params


def model_with(params):
    gnn = TJ.get_model()
    gnn.load_state_dict(params)
    return gnn.to(DEV)


@pytest.mark.parametrize("name", CASES)
def test_golden_aggregation_and_model(tg, name):
    va, ei, ea, params = load_case(tg, name)
    n = va.shape[0]
    agg = TJ.edge_to_vertex_aggregation(ei.to(DEV), ea.to(DEV), n)
    assert agg.is_cuda
    np.testing.assert_array_equal(agg.cpu().numpy(), tg[f"{name}_agg"])            # min / mean / sum / max: bit-exact
    agg_cpu_in = TJ.edge_to_vertex_aggregation(ei, ea, n)                          # CPU operands come back on the CPU
    assert not agg_cpu_in.is_cuda and torch.equal(agg_cpu_in, agg.cpu())
    out, _, _ = model_with(params)(va.to(DEV), ei.to(DEV), ea.to(DEV), batch=torch.zeros(n))
    np.testing.assert_allclose(out.detach().cpu().numpy(), tg[f"{name}_model_out"], rtol=1e-5, atol=0)


@pytest.mark.parametrize("name", CASES)
def test_golden_error_matrix_and_damping_factor(tg, name):
    A = T(tg[f"{name}_A"]).to_sparse().coalesce()
    d = T(tg[f"{name}_model_out"]).to(DEV).requires_grad_(True)
    Tm = L.build_error_matrix(A, d, 2. / 3.)
    np.testing.assert_array_equal(Tm.indices().cpu().numpy(), tg[f"{name}_T_ind"])
    np.testing.assert_allclose(Tm.values().detach().cpu().numpy(), tg[f"{name}_T_val"], rtol=1e-6, atol=0)
    df = L.eigval_approx(3, 20, Tm, Y=T(tg[f"{name}_Y_uniform"]))
    df.backward()
    np.testing.assert_allclose(df.item(), tg[f"{name}_df_uniform"], rtol=1e-5)
    g_ref = tg[f"{name}_df_uniform_grad"]
    np.testing.assert_allclose(d.grad.cpu().numpy(), g_ref, rtol=2e-3, atol=1e-5 * np.abs(g_ref).max())
    # exact damping factor (dense eigen-solve, as in the reference)
    np.testing.assert_allclose(float(L.damping_factor(A, 2. / 3., d.detach().cpu(), exact=True)),
                               tg[f"{name}_df_exact"], rtol=1e-4)


@pytest.mark.parametrize("name", CASES)
def test_golden_test_vectors(tg, name):
    n = tg[f"{name}_va"].shape[0]
    torch.manual_seed(77)
    np.testing.assert_array_equal(L.get_random_high_freq(n, 20, None).numpy(), tg[f"{name}_Y_highfreq"])
    torch.manual_seed(78)
    np.testing.assert_array_equal(L.get_random_on_sphere(n, 20).numpy(), tg[f"{name}_Y_uniform"])


@pytest.mark.parametrize("name", CASES)
def test_golden_training_step(tg, name):
    """loss_batch + backward: the reference's training step (train.py:98-108) on a one-graph batch."""
    va, ei, ea, params = load_case(tg, name)
    n = va.shape[0]
    A = T(tg[f"{name}_A"]).to_sparse().coalesce()
    gnn = model_with(params)
    batch = types.SimpleNamespace(x=va.to(DEV), edgeij_pair=ei.to(DEV), edge_attr=ea.to(DEV),
                                  batch=torch.zeros(n, dtype=torch.long, device=DEV), num_graphs=1,
                                  matrix=[gla.autograd.SparseMatrix.from_torch(A, DEV)], coords=[None])
    torch.manual_seed(79)
    loss = L.loss_batch(gnn, batch)
    loss.backward()
    np.testing.assert_allclose(loss.item(), tg[f"{name}_loss"], rtol=2e-5)
    for k, p in gnn.named_parameters():
        want = tg[f"{name}_lossgrad_{k}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=5e-3, atol=2e-5 * max(np.abs(want).max(), 1e-30))


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-5), (torch.float64, 1e-13)])
def test_fused_inference_matches_torch_mlp(dtype, rtol):
    """gla_mlp_relu3 (no_grad path of VertexUpdate) against the torch Sequential it replaces, and against golden."""
    torch.manual_seed(11)
    gnn = TJ.get_model().to(DEV).to(dtype)
    n = 70001
    gen = torch.Generator().manual_seed(12)
    x = (torch.randn(n, 5, generator=gen, dtype=torch.float64) * 3).to(dtype).to(DEV)
    want = gnn.node_model.vertex_update(x)
    with torch.no_grad():
        got = TJ._fused_inference(gnn.node_model.vertex_update, x)
    assert got is not None and got.shape == (n, 1)
    np.testing.assert_allclose(got.cpu().numpy(), want.detach().cpu().numpy(), rtol=rtol, atol=0)
    # other widths, incl. the limits, and negative weights so that the ReLUs bite
    for dims in ((1, 1, 1), (8, 512, 32), (3, 17, 5)):
        seq = torch.nn.Sequential(torch.nn.Linear(dims[0], dims[1]), torch.nn.ReLU(), torch.nn.Linear(dims[1], dims[2]),
                                  torch.nn.ReLU(), torch.nn.Linear(dims[2], 1)).to(DEV).to(dtype)
        xs = torch.randn(1000, dims[0], generator=gen, dtype=torch.float64).to(dtype).to(DEV)
        with torch.no_grad():
            got = TJ._fused_inference(seq, xs)
            want = seq(xs)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=50 * rtol, atol=50 * rtol)
    assert TJ._fused_inference(torch.nn.Sequential(torch.nn.Linear(5, 1)), x) is None      # other shapes: torch path


def test_model_no_grad_uses_fused_kernel_and_matches_golden(tg):
    for name in CASES:
        va, ei, ea, params = load_case(tg, name)
        gnn = model_with(params)
        with torch.no_grad():
            out, _, _ = gnn(va.to(DEV), ei.to(DEV), ea.to(DEV), batch=None)
        np.testing.assert_allclose(out.cpu().numpy(), tg[f"{name}_model_out"], rtol=1e-5, atol=0)


def test_golden_training_run(tg):
    """Four Adam steps of the reference's training loop (train.py:98-108) on a two-graph batch: the loss trajectory
    and the trained model's output follow the reference (fp32; steps amplify rounding differences: 1e-3 / 1e-2)."""
    xs, eis, eas, bvec, mats = [], [], [], [], []
    off = 0
    for gi, name in enumerate(("var8", "lap8")):
        va, ei, ea, params = load_case(tg, name)
        xs.append(va); eis.append(ei + off); eas.append(ea)
        bvec.append(torch.full((va.shape[0],), gi, dtype=torch.long))
        mats.append(T(tg[f"{name}_A"]).to_sparse().coalesce())              # torch sparse tensors, as Data.py holds
        off += va.shape[0]
    batch = types.SimpleNamespace(x=torch.cat(xs).to(DEV), edgeij_pair=torch.cat(eis, 1).to(DEV),
                                  edge_attr=torch.cat(eas).to(DEV), batch=torch.cat(bvec).to(DEV), num_graphs=2,
                                  matrix=mats, coords=[None, None])
    gnn = model_with(params)                                               # the seeded initial weights (same for all)
    opt = torch.optim.Adam(gnn.parameters(), lr=1e-2)
    losses = []
    for step in range(4):
        opt.zero_grad()
        torch.manual_seed(500 + step)
        loss = L.loss_batch(gnn, batch)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, tg["train_losses"], rtol=1e-3)
    assert losses[-1] < 0.5 * losses[0]
    with torch.no_grad():
        out = gnn(batch.x, batch.edgeij_pair, batch.edge_attr, [], batch.batch)[0]
    np.testing.assert_allclose(out.cpu().numpy(), tg["train_final_out"], rtol=1e-2)


def random_graph(n, n_cols, deg, seed, sort=True):
    gen = torch.Generator().manual_seed(seed)
    rows = torch.randint(0, n, (n * deg,), generator=gen)
    rows[rows % 7 == 3] = 0 if n < 8 else 5                     # a long row and many empty rows
    cols = torch.randint(0, n_cols, (n * deg,), generator=gen)
    key = torch.unique(rows * n_cols + cols)                    # coalesced pattern
    ei = torch.stack([key // n_cols, key % n_cols])
    if not sort:
        ei = ei[:, torch.randperm(ei.shape[1], generator=gen)]
    return ei, gen


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("sort", [True, False])
def test_segment_stats_vs_oracle(dtype, sort):
    n = 5000
    ei, gen = random_graph(n, n, 4, 5, sort)
    src = torch.randn(ei.shape[1], 3, generator=gen, dtype=torch.float64).to(dtype)
    want = rp.trainable_edge_stats(ei, src, n)
    plan = gla.CsrPlan(ei.to(DEV), n)
    got = plan.segment_stats(src.to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    # and it equals the four separate reductions of gla_segment_reduce
    four = torch.cat([plan.segment_reduce(src.to(DEV), r) for r in ("min", "mean", "sum", "max")], 1)
    assert torch.equal(got, four)


@pytest.mark.parametrize("K", [1, 3, 4, 8, 20, 40, 130])
@pytest.mark.parametrize("sort", [True, False])
def test_sparse_mm_autograd_vs_oracle(K, sort):
    """Forward and both gradients of Y = A X against torch's CPU sparse autograd, rectangular A, fp64."""
    n, m = 3000, 2100
    ei, gen = random_graph(n, m, 5, 11 + K, sort)
    vals = torch.randn(ei.shape[1], generator=gen, dtype=torch.float64)
    X = torch.randn(m, K, generator=gen, dtype=torch.float64)
    W = torch.randn(n, K, generator=gen, dtype=torch.float64)
    v_ref, X_ref = vals.clone().requires_grad_(True), X.clone().requires_grad_(True)
    Y_ref = torch.sparse.mm(torch.sparse_coo_tensor(ei, v_ref, (n, m)), X_ref)
    (Y_ref * W).sum().backward()
    v, Xd = vals.to(DEV).requires_grad_(True), X.to(DEV).requires_grad_(True)
    Y = gla.autograd.sparse_mm(ei.to(DEV), v, Xd, n, m)
    (Y * W.to(DEV)).sum().backward()
    np.testing.assert_allclose(Y.detach().cpu().numpy(), Y_ref.detach().numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(v.grad.cpu().numpy(), v_ref.grad.numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(Xd.grad.cpu().numpy(), X_ref.grad.numpy(), rtol=1e-12, atol=1e-13)
    # plan-level entry point: edge order of an unsorted list is the caller's
    dots = gla.plan_for(ei.to(DEV), n, m).edge_dot(W.to(DEV), X.to(DEV))
    want = (W[ei[0]] * X[ei[1]]).sum(1)
    np.testing.assert_allclose(dots.cpu().numpy(), want.numpy(), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("K", [4, 8, 12, 20, 64, 68])
def test_edge_dot_fp32_lane_groups(K):
    """fp32 takes other lane-group widths (4 values per 16-byte vector) than fp64: every width against an fp64 result."""
    n, m = 2500, 1900
    ei, gen = random_graph(n, m, 6, 40 + K, sort=True)
    G = torch.randn(n, K, generator=gen, dtype=torch.float64)
    X = torch.randn(m, K, generator=gen, dtype=torch.float64)
    want = (G[ei[0]] * X[ei[1]]).sum(1)
    got = gla.plan_for(ei.to(DEV), n, m).edge_dot(G.float().to(DEV), X.float().to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5 * K ** 0.5)


def test_damping_factor_vs_oracle_fp64():
    """Gelfand estimate and its gradient on a 40x40-vertex variable-coefficient operator, fp64, vs the oracle."""
    N = 40
    ei, ev = rp.laplacianfun_torch(N)
    n = N * N
    gen = torch.Generator().manual_seed(3)
    s = 0.5 + torch.rand(n, generator=gen, dtype=torch.float64)
    vals = -ev.flatten() * s[ei[0]] * s[ei[1]]
    diag0 = (4.0 * s * s) * (0.8 + 0.4 * torch.rand(n, generator=gen, dtype=torch.float64))
    Y = torch.randn(n, 20, generator=gen, dtype=torch.float64)
    Y = Y / torch.linalg.vector_norm(Y, dim=0)
    d_ref = diag0.clone().reshape(-1, 1).requires_grad_(True)
    df_ref = rp.trainable_eigval_approx(3, rp.trainable_error_matrix(ei, vals, n, d_ref, 2. / 3.), Y)
    df_ref.backward()
    d = diag0.reshape(-1, 1).to(DEV).requires_grad_(True)
    A = gla.autograd.SparseMatrix(ei.to(DEV), vals.to(DEV), (n, n))
    Tm = L.build_error_matrix(A, d, 2. / 3.)
    T_ref = rp.trainable_error_matrix(ei, vals, n, diag0.reshape(-1, 1), 2. / 3.)
    np.testing.assert_array_equal(Tm.indices().cpu().numpy(), T_ref.indices().numpy())
    np.testing.assert_array_equal(Tm.values().detach().cpu().numpy(), T_ref.values().numpy())
    df = L.eigval_approx(3, 20, Tm, Y=Y)
    df.backward()
    np.testing.assert_allclose(df.item(), df_ref.item(), rtol=1e-12)
    np.testing.assert_allclose(d.grad.cpu().numpy(), d_ref.grad.numpy(), rtol=1e-10, atol=1e-16)


def test_adjoint_identity_full_size():
    """1024^2 vertices, 20 columns (the loss's m): <G, A X> = <A^T G, X> = sum_e a_e * edge_dot_e."""
    N, K = 1024, 20
    ei, ev = gla.UtilsGNN.laplacianfun_torch(N, device=DEV)
    n = N * N
    gen = torch.Generator(device=DEV).manual_seed(9)
    vals = (ev.flatten() * (0.5 + torch.rand(ei.shape[1], generator=gen, device=DEV, dtype=torch.float64)))
    X = torch.randn(n, K, generator=gen, device=DEV, dtype=torch.float64).requires_grad_(True)
    G = torch.randn(n, K, generator=gen, device=DEV, dtype=torch.float64)
    v = vals.clone().requires_grad_(True)
    Y = gla.autograd.sparse_mm(ei, v, X, n)
    s1 = (G * Y).sum()
    s1.backward()
    s2 = (X.grad * X.detach()).sum()
    s3 = (v.grad * vals).sum()
    np.testing.assert_allclose(s2.item(), s1.item(), rtol=1e-11)
    np.testing.assert_allclose(s3.item(), s1.item(), rtol=1e-11)
