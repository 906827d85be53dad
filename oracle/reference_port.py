"""CPU restatement ("port") of the reference's sparse message-passing hot path.

TEST INFRASTRUCTURE ONLY -- this file is the parity ORACLE, not the product.  Only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it; the
product package (gnn_applied_linear_algebra_amd) never does and fails loudly when
its HIP library is missing.

It restates, with plain torch CPU ops and in the reference's own formulation
(COO edge list -> gather x[row], x[col] -> elementwise edge update -> torch.cat ->
edge-order sequential scatter-add -> vertex update -> torch.cat), what the modules
under /root/reference/pytorch compute.  Each function cites the file:line it
follows.  The third-party pieces the reference calls but does not vendor are
restated from their published semantics:
  * torch_scatter.scatter (unpinned, pytorch/requirements.txt:2): out = zeros(dim_size);
    sum accumulates in edge order; max/min give 0 on empty segments.
  * torch_geometric.nn.MetaLayer (unpinned, requirements.txt:4): edge(x[row], x[col],
    edge_attr, u, batch[row]) -> node(x, ei, edge_attr, u, batch) -> global(...).
  * torch_geometric.utils.remove_self_loops: order-preserving mask row != col.

PINNING: this port is checked (tests/test_oracle_golden.py) against tests/golden/*.npz,
which oracle/gen_golden.py produced by importing the unmodified reference modules in
the build container (oracle/ref_harness.py), and live against the reference whenever
/root/reference is present (tests/test_oracle_vs_reference.py).  Parity pinned; the
only unpinned third-party piece is pyamg's randomised CLJP splitting, which is an
INPUT here (see oracle/standins/pyamg/classical/split.py).
"""
import numpy as np
import torch


# --------------------------------------------------------------------------------------
# third-party primitives (restated)
# --------------------------------------------------------------------------------------
def scatter_sum(src, index, n):
    """torch_scatter.scatter(..., reduce='sum') along dim 0 (call sites SURVEY.md 2.3 SC1-SC5,7,8)."""
    shape = (n,) + tuple(src.shape[1:])
    idx = index if src.dim() == 1 else index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)


def scatter_max(src, index, n):
    """torch_scatter.scatter(..., reduce='max'); empty segments -> 0 (SOCClassicGNN.py:69)."""
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index if src.dim() == 1 else index.view(-1, 1).expand_as(src)
    return out.scatter_reduce_(0, idx, src, "amax", include_self=False)


def meta_layer(edge_fn, node_fn, global_fn, x, ei, edge_attr, u, batch):
    """PyG MetaLayer.forward semantics."""
    row, col = ei[0], ei[1]
    if edge_fn is not None:
        edge_attr = edge_fn(x[row], x[col], edge_attr, u, batch[row])
    if node_fn is not None:
        x = node_fn(x, ei, edge_attr, u, batch)
    if global_fn is not None:
        u = global_fn(x, ei, edge_attr, u, batch)
    return x, edge_attr, u


def _batch(n):
    return torch.zeros(n)


# --------------------------------------------------------------------------------------
# U1-U4 builders
# --------------------------------------------------------------------------------------
def laplacianfun_torch(N, Ny=None):
    """pytorch/UtilsGNN.py:53-67.  kron(I,T)+kron(T,I) with T=tridiag(1,-2,1): diag -4, the
    four grid neighbours +1, vertex k = j*N + i; scipy's COO of the CSR sum is sorted
    row-major with ascending columns.  Built here directly (no scipy).  `Ny` (extension,
    default N) gives the Ny x N rectangular grid used for the weak-scaling row blocks."""
    Ny = N if Ny is None else Ny
    k = np.arange(N * Ny, dtype=np.int64)
    i, j = k % N, k // N
    cand_col = np.stack([k - N, k - 1, k, k + 1, k + N], axis=1)
    valid = np.stack([j > 0, i > 0, np.ones_like(k, bool), i < N - 1, j < Ny - 1], axis=1)
    vals = np.broadcast_to(np.array([1.0, 1.0, -4.0, 1.0, 1.0]), cand_col.shape)
    rows = np.broadcast_to(k[:, None], cand_col.shape)
    edge_index = torch.tensor(np.array([rows[valid], cand_col[valid]]), dtype=torch.long)
    edge_val = torch.tensor(vals[valid].copy())[:, None]
    return edge_index, edge_val


def remove_diag_entries(edge_index, edge_val):
    """pytorch/UtilsGNN.py:69-72 -> PyG remove_self_loops."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], edge_val[mask]


def coo_to_gnn_input(A):
    """pytorch/UtilsGNN.py:74-78 (torch sparse COO coalesce = sort by (row, col), sum duplicates)."""
    A = A.coalesce()
    return A.indices(), A.values().reshape(-1, 1)


def heateqnfem2dfun(num_cells, h_all, bcs=(2, 2)):
    """matlab/heateqnfem2dfun.m:52-172 restated (MATLAB cannot run here; pinned by the file's own
    known answer, the interior stencil at :105-110).  Q1 stiffness on a cx x cy cell grid with
    stretch alpha = hy/hx; BC type 2 (eliminate boundary, keep interior) on both axes is the
    only mode the GNN drivers use.  Returns (edge_index int64 [2,nnz], edge_val f64 [nnz,1]) in
    coalesced (row-major) order for the (cx-1)*(cy-1) interior unknowns."""
    cx, cy = int(num_cells[0]), int(num_cells[1])
    alpha = float(h_all[1]) / float(h_all[0])
    a2 = alpha ** 2
    s = 1 / 6 / alpha                                                          # (1/6/alpha), :94
    # element matrix entries (heateqnfem2dfun.m:71-100): self, x-neighbour, y-neighbour, corner
    e_self, e_x, e_y, e_c = s * (2 * a2 + 2), s * (-2 * a2 + 1), s * (a2 - 2), s * (-1 - a2)
    loc = np.array([[e_self, e_x, e_c, e_y],
                    [e_x, e_self, e_y, e_c],
                    [e_c, e_y, e_self, e_x],
                    [e_y, e_c, e_x, e_self]])
    nx, ny = cx + 1, cy + 1
    xid, yid = np.meshgrid(np.arange(cx), np.arange(cy), indexing="xy")
    base = (nx * yid + xid).ravel()
    gidx = np.stack([base, base + 1, base + nx + 1, base + nx], axis=1)      # :114-125 (0-based)
    rows = np.repeat(gidx, 4, axis=1).ravel()
    cols = np.tile(gidx, (1, 4)).ravel()
    vals = np.tile(loc.ravel(), gidx.shape[0])
    ix, iy = rows % nx, rows // nx
    jx, jy = cols % nx, cols // nx
    interior = (ix > 0) & (ix < cx) & (iy > 0) & (iy < cy) & (jx > 0) & (jx < cx) & (jy > 0) & (jy < cy)
    if tuple(bcs) == (2, 2):                                                   # :164-168 keep interior only
        mx = cx - 1
        r = (iy[interior] - 1) * mx + (ix[interior] - 1)
        c = (jy[interior] - 1) * mx + (jx[interior] - 1)
        n = mx * (cy - 1)
        A = torch.sparse_coo_tensor(torch.tensor(np.array([r, c])), torch.tensor(vals[interior]), (n, n))
        return coo_to_gnn_input(A)
    if tuple(bcs) == (1, 1):                                                   # :143-161 zero rows/cols + identity
        n = nx * ny
        k = np.arange(n)
        bnd = (k % nx == 0) | (k % nx == cx) | (k // nx == 0) | (k // nx == cy)
        r = np.concatenate([rows[interior], k[bnd]])
        c = np.concatenate([cols[interior], k[bnd]])
        v = np.concatenate([vals[interior], np.ones(int(bnd.sum()))])
        A = torch.sparse_coo_tensor(torch.tensor(np.array([r, c])), torch.tensor(v), (n, n))
        return coo_to_gnn_input(A)
    raise NotImplementedError("bcs must be [1,1] or [2,2] (mixed modes are rejected by the reference, :60-63)")


# --------------------------------------------------------------------------------------
# M: MatVec   (pytorch/MatVecGNN.py)
# --------------------------------------------------------------------------------------
def matvec_layer(x, ei, A_ij, batch=None):
    """MetaLayer(EdgeUpdate(), VertexUpdate(agg)) of MatVecGNN.py:146-150.
    Returns (cat[x, y] [n,2k], cat[A, c] [nnz,1+k], None)."""
    batch = _batch(x.shape[0]) if batch is None else batch

    def edge(vi, vj, ea, g, b):                      # MatVecGNN.py:66-84
        return torch.cat([ea, ea * vj], 1)

    def node(va, ei_, ea, g, b):                     # MatVecGNN.py:95-114, agg :43-62
        c = ea[:, 1:1 + va.shape[1]]
        return torch.cat([va, scatter_sum(c, ei_[0], va.shape[0])], 1)

    return meta_layer(edge, node, None, x, ei, A_ij, None, batch)


# --------------------------------------------------------------------------------------
# R: Residual   (pytorch/GNNResidual.py)
# --------------------------------------------------------------------------------------
def residual_layer(vertex_attr, ei, edge_attr, batch=None):
    batch = _batch(vertex_attr.shape[0]) if batch is None else batch

    def edge(vi, vj, ea, g, b):                      # GNNResidual.py:66-86
        return torch.cat([ea, ea * vj[:, 1].view(-1, 1)], 1)

    def node(va, ei_, ea, g, b):                     # GNNResidual.py:97-118
        cbar = scatter_sum(ea[:, 1].view(-1, 1), ei_[0], va.shape[0])
        return torch.cat([va[:, 0].view(-1, 1), va[:, 1].view(-1, 1), va[:, 0].view(-1, 1) - cbar], 1)

    return meta_layer(edge, node, None, vertex_attr, ei, edge_attr, None, batch)


def residual_forward(vertex_attr, ei, edge_attr, batch=None):
    """GNNResidual.forward, GNNResidual.py:126-132 -> r [n,1]."""
    va, _, _ = residual_layer(vertex_attr, ei, edge_attr, batch)
    return va[:, 2].view(-1, 1)


# --------------------------------------------------------------------------------------
# J: Jacobi   (pytorch/JacobiGNN.py)
# --------------------------------------------------------------------------------------
def jacobi_iterate(vertex_attr, ei, edge_attr, g, batch=None):
    """JacobiGNN.iterate, JacobiGNN.py:130-135: one MetaLayer(EdgeUpdate :73-89, VertexUpdate :100-123)."""
    batch = _batch(vertex_attr.shape[0]) if batch is None else batch

    def edge(vi, vj, ea, g_, b):
        A = ea[:, 0].view(-1, 1)
        return torch.cat([A, A * vj[:, 2].view(-1, 1)], 1)

    def node(va, ei_, ea, g_, b):
        A_ii, b_i, x_i = va[:, 0].view(-1, 1), va[:, 1].view(-1, 1), va[:, 2].view(-1, 1)
        cbar = scatter_sum(ea[:, 1], ei_[0], va.shape[0]).reshape(-1, 1)      # :52-69
        x_i = x_i + g_[0] * (b_i - cbar) / A_ii                              # :119
        return torch.cat([A_ii, b_i, x_i], 1)

    return meta_layer(edge, node, None, vertex_attr, ei, edge_attr, g, batch)


def jacobi_forward(n_iters, vertex_attr, ei, edge_attr, g, batch=None):
    """JacobiGNN.forward, JacobiGNN.py:138-148 -> x [n,1]."""
    for _ in range(n_iters):
        vertex_attr, edge_attr, g = jacobi_iterate(vertex_attr, ei, edge_attr, g, batch)
    return vertex_attr[:, 2].reshape(-1, 1)


# --------------------------------------------------------------------------------------
# CH: Chebyshev   (pytorch/ChebyGNN.py)
# --------------------------------------------------------------------------------------
def cheby_forward(deg, vertex_attr, ei, edge_attr, g, batch=None):
    """ChebyRelaxGNN(deg).forward, ChebyGNN.py:287-353 (layer wiring :291-321)."""
    n = vertex_attr.shape[0]
    batch = _batch(n) if batch is None else batch

    def agg(ei_, ea):                                                          # :73-89
        return scatter_sum(ea[:, 1], ei_[0], n).view(-1, 1)

    def e_first(vi, vj, ea, g_, b):                                            # :49-70
        return torch.cat([ea, ea * vj[:, 1].view(-1, 1)], 1)

    def v_first(va, ei_, ea, g_, b):                                           # :91-121
        b_i, x_i = va[:, 0].view(-1, 1), va[:, 1].view(-1, 1)
        return torch.cat([b_i, x_i, b_i - agg(ei_, ea)], 1)

    def g_first(va, ei_, ea, g_, b):                                           # :123-139
        return torch.hstack([g_[0], g_[1], 1 / g_[1]])

    def v_first2(va, ei_, ea, g_, b):                                          # :141-163
        b_i, x_i, r_i = (va[:, k].view(-1, 1) for k in range(3))
        return torch.cat([b_i, x_i + g_[2] * r_i, r_i, r_i], 1)

    def e_next(vi, vj, ea, g_, b):                                             # :166-183
        A = ea[:, 0].view(-1, 1)
        return torch.cat([A, A * vj[:, 3].view(-1, 1)], 1)

    def v_next(va, ei_, ea, g_, b):                                            # :186-216
        b_i, x_i, r_i, p_i = (va[:, k].view(-1, 1) for k in range(4))
        return torch.cat([b_i, x_i, r_i - g_[2] * agg(ei_, ea), p_i], 1)

    def v_next2(va, ei_, ea, g_, b):                                           # :219-243
        b_i, x_i, r_i, p_i = (va[:, k].view(-1, 1) for k in range(4))
        p_i = r_i + g_[3] * p_i
        return torch.cat([b_i, x_i + g_[2] * p_i, r_i, p_i], 1)

    def g_second(va, ei_, ea, g_, b):                                          # :246-265
        c, d, alpha = g_[0], g_[1], g_[2]
        beta = 0.5 * (c * alpha) ** 2
        return torch.hstack([c, d, 1 / (d - beta / alpha), beta])

    def g_later(va, ei_, ea, g_, b):                                           # :269-285
        c, d, alpha = g_[0], g_[1], g_[2]
        beta = ((c * alpha) / 2) ** 2
        return torch.hstack([c, d, 1 / (d - beta / alpha), beta])

    layers = []
    if deg > 0:
        layers += [(e_first, v_first, g_first), (None, v_first2, None)]
    if deg > 1:
        layers += [(e_next, v_next, g_second), (None, v_next2, None)]
    for _ in range(deg - 2):
        layers += [(e_next, v_next, g_later), (None, v_next2, None)]
    for e, v, gl in layers:
        vertex_attr, edge_attr, g = meta_layer(e, v, gl, vertex_attr, ei, edge_attr, g, batch)
    return vertex_attr, edge_attr, g


# --------------------------------------------------------------------------------------
# P: power method   (pytorch/PowerMethodGNN.py)
# --------------------------------------------------------------------------------------
def power_forward(num_iter, vertex_attr, ei, edge_attr, g, batch=None):
    """PowerMethodGNN(num_iter).forward, PowerMethodGNN.py:296-334."""
    n = vertex_attr.shape[0]
    batch = _batch(n) if batch is None else batch

    def agg(ei_, ea):                                                          # :64-83
        return scatter_sum(ea[:, 1], ei_[0], n).reshape(-1, 1)

    def vsum(va):                                                              # :50-61
        return torch.sum(va[:, 1])

    def e1(vi, vj, ea, g_, b):                                                 # :86-106
        A = ea[:, 0].view(-1, 1)
        return torch.cat([A, A * vj[:, 0].view(-1, 1)], 1)

    def v_sq(va, ei_, ea, g_, b):                                              # :109-126
        b_i = va[:, 0].view(-1, 1)
        return torch.cat([b_i, b_i * b_i], 1)

    def v_it1(va, ei_, ea, g_, b):                                             # :129-158
        return torch.cat([agg(ei_, ea), va[:, 1].view(-1, 1)], 1)

    def g_it2(va, ei_, ea, g_, b):                                             # :160-185
        return torch.tensor([torch.sqrt(vsum(va)), g_[1], g_[2]])

    def v_it3(va, ei_, ea, g_, b):                                             # :187-207
        return torch.cat([va[:, 0].view(-1, 1) / g_[0], va[:, 1].view(-1, 1)], 1)

    def v_ray1(va, ei_, ea, g_, b):                                            # :209-237
        b_i = va[:, 0].view(-1, 1)
        return torch.cat([b_i, b_i * agg(ei_, ea)], 1)

    def g_ray1(va, ei_, ea, g_, b):                                            # :239-266
        return torch.tensor([g_[0], vsum(va), g_[2]])

    def g_ray2(va, ei_, ea, g_, b):                                            # :268-294
        return torch.tensor([g_[0], g_[1], g_[1] / vsum(va)])

    layers = [(e1, v_it1, None), (None, v_sq, g_it2), (None, v_it3, None)] * num_iter
    layers += [(e1, v_ray1, g_ray1), (None, v_sq, g_ray2)]
    for e, v, gl in layers:
        vertex_attr, edge_attr, g = meta_layer(e, v, gl, vertex_attr, ei, edge_attr, g, batch)
    return vertex_attr, edge_attr, g


# --------------------------------------------------------------------------------------
# SO: classical strength of connection   (pytorch/SOCClassicGNN.py)
# --------------------------------------------------------------------------------------
def soc_classic_forward(theta, vertex_attr, ei, edge_attr, batch=None, return_v=False):
    """SOCClassicGNN(theta).forward, SOCClassicGNN.py:131-147 -> S_ij [nnz_od] (1-D)."""
    n = vertex_attr.shape[0]
    batch = _batch(n) if batch is None else batch

    def v1(va, ei_, ea, g_, b):                                                # :74-102, agg :50-72
        return scatter_max(-1 * ea[:, 0], ei_[0], n).reshape(-1, 1)

    def e2(vi, vj, ea, g_, b):                                                 # :104-129
        A = ea[:, 0].view(-1, 1)
        return torch.cat([A, torch.relu(-1 * A / vi[:, 0].view(-1, 1) - theta)], 1)

    va, ea, g = meta_layer(None, v1, None, vertex_attr, ei, edge_attr, None, batch)
    va, ea, g = meta_layer(e2, None, None, va, ei, ea, g, batch)
    return (ea[:, 1], va) if return_v else ea[:, 1]


# --------------------------------------------------------------------------------------
# D: direct interpolation   (pytorch/DirectInterpGNN.py)
# --------------------------------------------------------------------------------------
def direct_interp_forward(vertex_attr, ei, edge_attr, g=None, batch=None):
    """DirectInterpGNN.forward, DirectInterpGNN.py:155-174 -> w_ij [nnz_od] (1-D).
    C rows evaluate (1-1)*(-A*alpha) with alpha = +-inf/NaN -> NaN, as in the reference."""
    n = vertex_attr.shape[0]
    batch = _batch(n) if batch is None else batch

    def e1(vi, vj, ea, g_, b):                                                 # :50-69
        return torch.cat([ea, vj[:, 1].view(-1, 1)], 1)

    def v1(va, ei_, ea, g_, b):                                                # :99-131, agg :71-97
        num = scatter_sum(ea[:, 0], ei_[0], n)
        den = scatter_sum(ea[:, 0] * ea[:, 1] * ea[:, 2], ei_[0], n)
        gamma = (num / den).reshape(-1, 1)
        A_ii, C_i = va[:, 0].view(-1, 1), va[:, 1].view(-1, 1)
        return torch.cat([A_ii, C_i, (1 / A_ii) * gamma], 1)

    def e2(vi, vj, ea, g_, b):                                                 # :133-152
        A, S = ea[:, 0].view(-1, 1), ea[:, 1].view(-1, 1)
        w = (1 - vi[:, 1].view(-1, 1)) * (-A * vi[:, 2].view(-1, 1))
        return torch.cat([A, S, w], 1)

    va, ea, _ = meta_layer(e1, v1, None, vertex_attr, ei, edge_attr, None, batch)
    va, ea, _ = meta_layer(e2, None, None, va, ei, ea, None, batch)
    return ea[:, 2]


# --------------------------------------------------------------------------------------
# next-tier: SA strength of connection, matrix-weighted norm
# --------------------------------------------------------------------------------------
def socsa_layer(vertex_attr, ei, edge_attr, batch=None):
    """MetaLayer(SOCSAGNN.EdgeUpdate()), SOCSAGNN.py:49-71: S_ij = A_ij^2 / (A_ii A_jj)."""
    batch = _batch(vertex_attr.shape[0]) if batch is None else batch

    def e(vi, vj, ea, g_, b):
        return torch.cat([ea, (ea * ea) / (vi * vj)], 1)

    return meta_layer(e, None, None, vertex_attr, ei, edge_attr, None, batch)


def matrix_weighted_norm_layer(node_attr, ei, edge_attr, g=None, batch=None):
    """MetaLayer(EdgeUpdate, VertexUpdate, GlobalUpdate) of MatrixWeightedNorm.py:49-161."""
    n = node_attr.shape[0]
    batch = _batch(n) if batch is None else batch

    def e(vi, vj, ea, g_, b):                                                  # :49-66
        return torch.cat([ea, ea * vj], 1)

    def v(va, ei_, ea, g_, b):                                                 # :88-122, agg :68-86
        return torch.cat([va, va * scatter_sum(ea[:, 1], ei_[0], n).reshape(-1, 1)], 1)

    def gl(va, ei_, ea, g_, b):                                                # :124-161
        return torch.sqrt(torch.sum(va[:, 1]))

    return meta_layer(e, v, gl, node_attr, ei, edge_attr, g, batch)


# --------------------------------------------------------------------------------------
# V: two-grid V-cycle glue   (pytorch/VCycle.py)
# --------------------------------------------------------------------------------------
def vcycle_interp_P(A, S_flags, splitting, diag_vals):
    """runDirectInterp, VCycle.py:94-137, with the C/F splitting passed in as data."""
    ei, ea = coo_to_gnn_input(A)
    ei, ea = remove_diag_entries(ei, ea)
    ea = torch.hstack([ea, S_flags])
    va = torch.hstack([diag_vals, splitting.reshape(-1, 1)])
    w = direct_interp_forward(va, ei, ea)
    n = A.shape[0]
    W = torch.eye(n, dtype=A.dtype) + torch.sparse_coo_tensor(ei, w, (n, n), dtype=A.dtype)   # script: torch.float
    W = W.to_dense()
    return W[:, splitting.flatten() > 0].to_sparse()


def vcycle(A, b, x, splitting, n_pre=3, n_post=3, cheb_deg=4, theta=0.25, w=0.7):
    """runVCycle(use_jacobi=True), VCycle.py:175-237.  The script is fp32 (`dtype=torch.float`, :259); here every
    such dtype follows A, and the scalar tensors (`torch.tensor(w)`, `torch.tensor([c, d])`) follow torch's default
    dtype exactly as in the script -- so an fp32 A under the fp32 default reproduces the script, and an fp64 A under
    torch.set_default_dtype(float64) reproduces the reference's functions driven in double precision
    (oracle/gen_golden_vcycle64.py).  N-agnostic only through diag_vals = -4 (VCycle.py:117,165 hard-code it)."""
    n = A.shape[0]
    diag = -4 * torch.ones((n, 1), dtype=A.dtype)

    def run_jacobi(k, x_):                                                     # :156-173
        ei, ea = coo_to_gnn_input(A)
        ea = torch.cat([ea, torch.zeros_like(ea)], 1)
        return jacobi_forward(k, torch.cat([diag, b, x_], 1), ei, ea, torch.tensor(w).reshape(-1))

    def run_cheby(A_, b_, x_, c, d):                                           # :139-154
        ei, ea = coo_to_gnn_input(A_)
        va, _, _ = cheby_forward(cheb_deg, torch.cat([b_, x_], 1), ei, ea, torch.tensor([c, d]))
        return va[:, 1].reshape(-1, 1)

    def run_residual(x_):                                                      # :58-70
        ei, ea = coo_to_gnn_input(A)
        return residual_forward(torch.cat([b, x_], 1), ei, ea)

    x = run_jacobi(n_pre, x)
    ei, ea = remove_diag_entries(*coo_to_gnn_input(A))                         # runSOC :72-92
    S = soc_classic_forward(theta, torch.zeros((n, 1), dtype=A.dtype), ei, ea).reshape(-1, 1) > 0
    P = vcycle_interp_P(A, S, splitting, diag)
    Ac = P.t() @ (A @ P)                                                       # :209
    r = run_residual(x)
    rc = P.t() @ r                                                             # :215
    xc = run_cheby(Ac, rc, torch.zeros_like(rc), -3.4, -4.0)                   # :221-223
    x = x + P @ xc                                                             # :226
    return run_jacobi(n_post, x)


def pmis_splitting(edgeij_pair_od, strong, n, max_rounds=64):
    """CPU restatement (torch ops) of the library's deterministic PMIS rule, gla_pmis_split (csrc/gla_sparse.hip).
    NOT a reference function: the reference calls pyamg's randomised CLJP (VCycle.py:114), whose output nothing in
    the reference pins (parity unpinned), so the rule is the build's own and this is its checker.
    weight_i = (number of vertices that strongly depend on i) << 52 | fmix32(i + 1) << 20 | (i & 0xFFFFF)."""
    strong = strong.reshape(-1).to(torch.bool)
    row, col = edgeij_pair_od[0][strong], edgeij_pair_od[1][strong]
    lam = torch.bincount(col, minlength=n)
    idx = torch.arange(n, dtype=torch.int64)
    h = (idx + 1) & 0xFFFFFFFF                                       # murmur3 fmix32: a fixed pseudo-random order
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    w = (lam << 52) | (h << 20) | (idx & 0xFFFFF)
    state = torch.zeros(n, dtype=torch.int64)                        # 0 undecided, 1 C, -1 F
    touched = torch.zeros(n, dtype=torch.bool)
    touched[row] = True
    touched[col] = True
    state[~touched] = 1
    a = torch.cat([row, col])                                        # symmetrised strong graph a -> b
    b = torch.cat([col, row])
    for _ in range(max_rounds):
        und = state == 0
        if not bool(und.any()):
            break
        live = und[a] & und[b]
        m = torch.full((n,), -1, dtype=torch.int64)
        m.scatter_reduce_(0, a[live], w[b[live]], "amax", include_self=True)
        new_c = und & (w > m)
        state[new_c] = 1
        dep = (state[row] == 0) & (state[col] == 1)                  # undecided i strongly depends on a C point
        state[row[dep]] = -1
    state[state == 0] = 1
    return (state == 1).to(torch.int64)


# ---------------------------------------------------------------------------------------------------
# learned Jacobi diagonal: pytorch/TrainableJacobiDiag/{TrainableJacobiGNN,loss}.py.
# Pinned by tests/golden/trainable_vectors.npz (oracle/gen_golden_trainable.py, unmodified reference).
# ---------------------------------------------------------------------------------------------------
def trainable_edge_stats(ei, edge_attr, n):
    """TrainableJacobiGNN.py:53-70  cat(min, mean, sum, max) of the edge attributes per sending vertex."""
    idx = ei[0].reshape(-1, 1).expand_as(edge_attr)
    zeros = torch.zeros((n, edge_attr.shape[1]), dtype=edge_attr.dtype)
    s = zeros.clone().scatter_add_(0, idx, edge_attr)
    cnt = zeros.clone().scatter_add_(0, idx, torch.ones_like(edge_attr))
    mn = zeros.clone().scatter_reduce_(0, idx, edge_attr, "amin", include_self=False)
    mx = zeros.clone().scatter_reduce_(0, idx, edge_attr, "amax", include_self=False)
    return torch.cat((mn, s / cnt.clamp(min=1), s, mx), 1)


def trainable_mlp(params, x):
    """TrainableJacobiGNN.py:86-91  Linear(5,50) ReLU Linear(50,20) ReLU Linear(20,1); params = the state dict."""
    for k in (0, 2, 4):
        x = x @ params[f"node_model.vertex_update.{k}.weight"].T + params[f"node_model.vertex_update.{k}.bias"]
        if k != 4:
            x = torch.relu(x)
    return x


def trainable_model(params, vertex_attr, ei, edge_attr):
    """TrainableJacobiGNN.py:93-111  MLP(cat(vertex_attr, edge statistics))."""
    return trainable_mlp(params, torch.cat([vertex_attr, trainable_edge_stats(ei, edge_attr, vertex_attr.shape[0])], 1))


def trainable_error_matrix(A_ind, A_val, n, diag, omega):
    """loss.py:92-102  T = I - omega D^-1 A, coalesced torch sparse tensor (A coalesced)."""
    T_val = -omega * (1 / diag[A_ind[0, :]].reshape(-1)) * A_val
    eye = torch.arange(n)
    T = torch.sparse_coo_tensor(torch.hstack((A_ind, torch.stack((eye, eye)))),
                                torch.hstack((T_val, torch.ones(n, dtype=T_val.dtype))), (n, n))
    return T.coalesce()


def trainable_eigval_approx(K, T, Y):
    """loss.py:105-128  max_j ||T^K y_j||^(1/K) for given normalised test vectors."""
    for _ in range(K):
        Y = T @ Y
    return torch.max(torch.linalg.vector_norm(Y, dim=0)) ** (1 / K)


def trainable_high_freq(N, m):
    """loss.py:139-170 with xy=None (consumes torch's global CPU generator exactly like the reference)."""
    import math
    n = int(math.sqrt(N))
    xx = torch.zeros(n * n, 1)
    yy = torch.zeros(n * n, 1)
    for j in range(n):
        for i in range(n):
            me = (j - 1) * n + i
            xx[me] = (i + 1) / (n + 1)
            yy[me] = (j + 1) / (n + 1)
    Y = torch.zeros(N, m)
    k = 0
    while k < m:
        thetax, thetay = (n - 1) * torch.rand(2) + 1
        if thetax > n / 2 or thetay > n / 2:
            Y[:, k] = (torch.sin(thetax * torch.pi * xx) * torch.sin(thetay * torch.pi * yy)).squeeze()
            k += 1
    return Y @ torch.diagflat(1 / torch.linalg.vector_norm(Y, dim=0))
