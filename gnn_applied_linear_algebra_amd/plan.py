"""CsrPlan: Python handle of a libgla_hip CSR plan + shape-checked wrappers of the C-ABI hot-path calls.

A plan is the once-per-graph replacement for what `torch_scatter.scatter(src, edgeij_pair[0], ...)`
re-derives on every layer call in the reference: the row structure of the COO edge list.
Plans are cached on the edge-index tensor (see `plan_for`) so that the nn.Module surface, which receives
`edgeij_pair` / `edge_attr` on every forward like the reference's, only pays for the conversion once.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, stream_ptr

ALL_TILES, INTERIOR_TILES, BOUNDARY_TILES = 0, 1, 2


class _DeviceGuard:
    """libgla_hip launches on the calling thread's current HIP device: make it the operands' device for the call
    (a no-op in the usual one-process-per-GPU setting, where torch.cuda.set_device has been called once)."""

    def __init__(self, device):
        self.idx, self.prev = device.index, None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if self.idx is not None and cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


def _on_device(fn):
    def wrapped(self, *a, **k):
        with _DeviceGuard(self.device):
            return fn(self, *a, **k)
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


def _dense(t, rows, name):
    """Validate a row-major [rows, K] (or [rows]) GPU operand; return (2-D view, K, leading dimension)."""
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor")
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.dim() != 2 or t.shape[0] != rows:
        raise ValueError(f"{name} must have shape [{rows}, K], got {tuple(t.shape)}")
    K = t.shape[1]
    if K < 1:
        raise ValueError(f"{name} needs at least one column")
    if K > 1 and t.stride(1) != 1:
        raise ValueError(f"{name} must be row-major (unit stride along K)")
    ld = t.stride(0) if rows > 1 else K
    if ld < K:
        raise ValueError(f"{name}: leading dimension {ld} < K={K}")
    return t, K, ld


class CsrPlan:
    """Owns a gla_csr_t.  Not copyable; freed on garbage collection."""

    def __init__(self, edgeij_pair, n_rows, n_cols=None):
        _lib.require_gpu()
        if edgeij_pair.dim() != 2 or edgeij_pair.shape[0] != 2:
            raise ValueError(f"edgeij_pair must be [2, nnz], got {tuple(edgeij_pair.shape)}")
        if edgeij_pair.dtype != torch.int64:
            raise TypeError("edgeij_pair must be int64 (torch.long), as in the reference")
        if not edgeij_pair.is_cuda:
            raise ValueError("edgeij_pair must be a GPU tensor")
        ei = edgeij_pair.contiguous()
        n_cols = n_rows if n_cols is None else n_cols
        self.device = ei.device
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(ei.shape[1])
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().gla_csr_create_from_coo(self.n_rows, self.n_cols, self.nnz, ptr(ei[0]), ptr(ei[1]),
                                                stream_ptr(self.device), ctypes.byref(self._h)))
        info = _lib.CsrInfo()
        check(lib().gla_csr_info(self._h, ctypes.byref(info)))
        self.info = info
        self.sorted_input = bool(info.sorted_input)
        self.val_dtype = None
        self._val_key = None
        self._val_ref = None
        self.n_interior = self.n_boundary = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib._lib is not None:
            _lib._lib.gla_csr_destroy(h)
            self._h = None

    # ---- values ---------------------------------------------------------------------------------
    @_on_device
    def set_values(self, edge_val):
        """edge_val: [nnz] or [nnz, n_e] (column 0 holds A_ij, as in the reference's edge_attr)."""
        if edge_val.shape[0] != self.nnz:
            raise ValueError(f"edge values have {edge_val.shape[0]} rows, plan has {self.nnz} edges")
        if not edge_val.is_cuda:
            raise ValueError("edge values must be a GPU tensor")
        stride = edge_val.stride(0) if edge_val.dim() > 1 or self.nnz > 1 else 1
        if edge_val.dim() == 2 and edge_val.shape[1] > 1 and edge_val.stride(1) != 1:
            edge_val = edge_val[:, 0].contiguous()
            stride = 1
        check(lib().gla_csr_set_values(self._h, ptr(edge_val), max(int(stride), 1), dtype_code(edge_val.dtype),
                                       stream_ptr(self.device)))
        self.val_dtype = edge_val.dtype
        check(lib().gla_csr_info(self._h, ctypes.byref(self.info)))          # `packed` may have changed

    def set_values_cached(self, edge_attr):
        """Refresh the CSR-ordered values unless `edge_attr` is the very tensor object they were gathered from
        and it has not been written since (identity via weakref: a data_ptr alone can be recycled by the
        allocator for a different tensor)."""
        key = (edge_attr._version, tuple(edge_attr.shape), tuple(edge_attr.stride()), edge_attr.dtype)
        ref = self._val_ref() if self._val_ref is not None else None
        if ref is not edge_attr or key != self._val_key:
            self.set_values(edge_attr)
            self._val_key = key
            self._val_ref = weakref.ref(edge_attr)
            _PLANS_WITH_VALUES.add(weakref.ref(self))

    @_on_device
    def arrays(self):
        """(rowptr, col, perm) as int32 GPU tensors -- for the CSR-indexing parity tests."""
        rowptr = torch.empty(self.n_rows + 1, dtype=torch.int32, device=self.device)
        col = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        perm = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        check(lib().gla_csr_export(self._h, ptr(rowptr), ptr(col), ptr(perm), stream_ptr(self.device)))
        return rowptr, col, perm

    @_on_device
    def classify_tiles(self, col_limit):
        ni, nb = ctypes.c_int32(), ctypes.c_int32()
        check(lib().gla_csr_classify_tiles(self._h, int(col_limit), stream_ptr(self.device), ctypes.byref(ni),
                                           ctypes.byref(nb)))
        self.n_interior, self.n_boundary = ni.value, nb.value
        return ni.value, nb.value

    # ---- hot path -------------------------------------------------------------------------------
    def _need(self, dtype):
        if self.nnz and self.val_dtype != dtype:
            raise _lib.GlaError(f"plan holds values of dtype {self.val_dtype}, operands are {dtype}")

    @_on_device
    def spmm(self, X, out=None, which=ALL_TILES):
        X, K, ldx = _dense(X, self.n_cols, "X")
        self._need(X.dtype)
        Y = torch.empty((self.n_rows, K), dtype=X.dtype, device=self.device) if out is None else out
        Y, Ky, ldy = _dense(Y, self.n_rows, "Y")
        assert Ky == K and Y.dtype == X.dtype
        check(lib().gla_spmm(self._h, dtype_code(X.dtype), K, ptr(X), ldx, ptr(Y), ldy, which,
                             stream_ptr(self.device)))
        return Y

    @_on_device
    def spmm_add(self, Z, X, out=None, which=ALL_TILES):
        """Z + A X in one pass (gla_spmm_add); out may be Z itself."""
        X, K, ldx = _dense(X, self.n_cols, "X")
        Z, Kz, ldz = _dense(Z, self.n_rows, "Z")
        self._need(X.dtype)
        Y = torch.empty((self.n_rows, K), dtype=X.dtype, device=self.device) if out is None else out
        Y, Ky, ldy = _dense(Y, self.n_rows, "Y")
        assert Kz == K and Ky == K and Z.dtype == X.dtype == Y.dtype
        check(lib().gla_spmm_add(self._h, dtype_code(X.dtype), K, ptr(Z), ldz, ptr(X), ldx, ptr(Y), ldy, which,
                                 stream_ptr(self.device)))
        return Y

    @_on_device
    def residual(self, B, X, out=None, which=ALL_TILES):
        X, K, ldx = _dense(X, self.n_cols, "X")
        B, Kb, ldb = _dense(B, self.n_rows, "B")
        self._need(X.dtype)
        R = torch.empty((self.n_rows, K), dtype=X.dtype, device=self.device) if out is None else out
        R, Kr, ldr = _dense(R, self.n_rows, "R")
        assert Kb == K and Kr == K and B.dtype == X.dtype == R.dtype
        check(lib().gla_residual(self._h, dtype_code(X.dtype), K, ptr(B), ldb, ptr(X), ldx, ptr(R), ldr, which,
                                 stream_ptr(self.device)))
        return R

    @_on_device
    def jacobi_sweep(self, diag, B, X, Xout, w, which=ALL_TILES):
        X, K, ldx = _dense(X, self.n_cols, "X")
        B, Kb, ldb = _dense(B, self.n_rows, "B")
        Xout, Ko, ldxo = _dense(Xout, self.n_rows, "Xout")
        self._need(X.dtype)
        assert diag.numel() == self.n_rows and diag.is_contiguous() and diag.is_cuda
        assert Kb == K and Ko == K and B.dtype == X.dtype == Xout.dtype == diag.dtype
        assert X.data_ptr() != Xout.data_ptr()
        check(lib().gla_jacobi_sweep(self._h, dtype_code(X.dtype), K, ptr(diag), ptr(B), ldb, ptr(X), ldx,
                                     ptr(Xout), ldxo, float(w), which, stream_ptr(self.device)))
        return Xout

    def _diag_from_matrix(self, K, n_iters, dtype, contiguous):
        """The caller vouches that its `diag` IS this plan's matrix diagonal (an AMG hierarchy extracts it from the matrix): on
        canonical 5- / 9-point plans whose schedule for n_iters is march passes only the library is then told diag = NULL and the
        passes take A_ii from the centre coefficient they load anyway (no A_ii stream; same division, bit-identical)."""
        if K != 1 or not contiguous or int(self.info.dia_canon5) not in (1, 9):
            return False
        sched = self.jacobi_schedule(int(n_iters), dtype)
        return len(sched) > 0 and min(sched) >= 2

    @_on_device
    def jacobi_sweeps(self, diag, B, X0, w, n_iters, diag_is_matrix_diagonal=False, out=None):
        """n_iters fused sweeps; returns a new [n, K] tensor, X0 untouched (JacobiGNN.py:138-148).  out: a contiguous buffer of at
        least n x K elements of X0's dtype on this device that receives the result instead (dist_amg.py: the prolongation's operand)."""
        assert self.n_rows == self.n_cols
        X0, K, ldx0 = _dense(X0, self.n_cols, "X0")
        B, Kb, ldb = _dense(B, self.n_rows, "B")
        self._need(X0.dtype)
        assert diag.numel() == self.n_rows and diag.is_contiguous() and diag.is_cuda
        assert Kb == K and B.dtype == X0.dtype == diag.dtype
        if n_iters <= 0:
            return X0.clone()
        if out is not None and out.is_contiguous() and out.dtype == X0.dtype and out.device == X0.device and out.numel() >= self.n_rows * K \
                and out.data_ptr() != X0.data_ptr():
            Xout = out.reshape(-1)[:self.n_rows * K].reshape(self.n_rows, K)
        else:
            Xout = torch.empty((self.n_rows, K), dtype=X0.dtype, device=self.device)
        Xtmp = torch.empty_like(Xout) if n_iters > 1 else None
        no_stream = diag_is_matrix_diagonal and self._diag_from_matrix(K, n_iters, X0.dtype, ldb == 1 and ldx0 == 1)

        def run(dg):
            return lib().gla_jacobi_sweeps(self._h, dtype_code(X0.dtype), K, dg, ptr(B), ldb, ptr(X0), ldx0, ptr(Xout), ptr(Xtmp), K,
                                           float(w), int(n_iters), stream_ptr(self.device))
        rc = run(None if no_stream else ptr(diag))
        if rc == -1 and no_stream:                           # the library would not do without the array (option jacobi_march = 0: the
            rc = run(ptr(diag))                              # canonical plans run the generic kernels); refused BEFORE anything was enqueued
        check(rc)
        return Xout

    @_on_device
    def jacobi_sweeps_aos(self, vertex_attr, w, n_iters):
        """JacobiGNN.forward from the AoS tensor vertex_attr [n, 3] = [A_ii, b, x] without a separate column split;
        returns x after n_iters sweeps as [n, 1], or None when the plan / count has no fused first pass (the caller
        splits the columns and uses jacobi_sweeps)."""
        va = vertex_attr
        assert self.n_rows == self.n_cols and va.dim() == 2 and va.shape == (self.n_rows, 3) and va.is_cuda
        if va.stride(1) != 1 or self.n_rows == 0 or int(n_iters) < 2:
            return None
        self._need(va.dtype)
        if self.jacobi_schedule(n_iters, va.dtype)[0] < 2:
            return None
        Xout = torch.empty((self.n_rows, 1), dtype=va.dtype, device=self.device)
        Xtmp = torch.empty_like(Xout)
        dwork, bwork = torch.empty_like(Xout), torch.empty_like(Xout)
        check(lib().gla_jacobi_sweeps_aos(self._h, dtype_code(va.dtype), ptr(va), va.stride(0), ptr(dwork), ptr(bwork),
                                          ptr(Xout), ptr(Xtmp), float(w), int(n_iters), stream_ptr(self.device)))
        return Xout

    @_on_device
    def jacobi_sweeps_residual(self, diag, B, X0, w, n_iters, diag_is_matrix_diagonal=False, r_out=None):
        """(x after n_iters sweeps, r = B - A x) with the residual formed inside the last temporally blocked pass
        (gla_jacobi_fused_residual), or None when the plan / count does not qualify (the caller runs jacobi_sweeps and
        residual).  Single contiguous rhs."""
        from ._lib import GlaError
        n_iters = int(n_iters)
        if self.n_rows != self.n_cols or n_iters < 1 or self.n_rows == 0:
            return None
        X0, K, _ = _dense(X0, self.n_cols, "X0")
        B, Kb, _ = _dense(B, self.n_rows, "B")
        if K != 1 or Kb != 1 or not (X0.is_contiguous() and B.is_contiguous() and diag.is_contiguous()):
            return None
        self._need(X0.dtype)
        info = self.info
        cap = 7 if int(info.dia_canon5) == 1 else (5 if int(info.dia_canon5) == 9 else 0)
        sched = self.jacobi_schedule(n_iters, X0.dtype)
        last = sched[-1] if sched else 0
        if cap == 0 or last > cap or (n_iters > 1 and max(sched) < 2):
            return None
        x = X0
        if n_iters > last:                                   # the earlier passes of the schedule as usual
            x = self.jacobi_sweeps(diag, B, X0, w, n_iters - last, diag_is_matrix_diagonal=diag_is_matrix_diagonal)
        Xout = torch.empty((self.n_rows, 1), dtype=X0.dtype, device=self.device)
        if r_out is not None and r_out.is_contiguous() and r_out.dtype == X0.dtype and r_out.numel() >= self.n_rows and r_out.device == Xout.device:
            R = r_out.reshape(-1)[:self.n_rows].reshape(-1, 1)      # (the caller's buffer: e.g. the restriction's operand, dist_amg.py)
        else:
            R = torch.empty_like(Xout)
        try:
            rc = lib().gla_jacobi_fused_residual(self._h, dtype_code(X0.dtype), None if diag_is_matrix_diagonal else ptr(diag),
                                                 ptr(B), ptr(x), ptr(Xout), ptr(R), float(w), last, stream_ptr(self.device))
            if rc == -1 and diag_is_matrix_diagonal:         # (as in jacobi_sweeps: the plan's kernels want the array)
                rc = lib().gla_jacobi_fused_residual(self._h, dtype_code(X0.dtype), ptr(diag), ptr(B), ptr(x), ptr(Xout), ptr(R),
                                                     float(w), last, stream_ptr(self.device))
            check(rc)
        except GlaError as err:
            if err.code == -5:                               # GLA_ERR_UNSUPPORTED (small matrix, options)
                return None
            raise
        return Xout, R

    def jacobi_schedule(self, n_iters, dtype=torch.float64):
        """Sweeps per launch, in launch order, that gla_jacobi_sweeps uses for n_iters sweeps on one contiguous rhs:
        (7, 7, 6) for 20 sweeps on a canonical 5-point plan, (5, 5, 5, 5) on other 2-D stencils, (1,) * 20 unfused."""
        counts, sweeps = (ctypes.c_int * 4)(), (ctypes.c_int * 4)()
        check(lib().gla_jacobi_sweeps_schedule(self._h, dtype_code(dtype), int(n_iters), counts, sweeps))
        return tuple(ts for c, ts in zip(counts, sweeps) for _ in range(c))

    @_on_device
    def cheby_first(self, B, X, R, P, Xout, alpha, which=ALL_TILES):
        X, K, ldx = _dense(X, self.n_cols, "X")
        B, _, ldb = _dense(B, self.n_rows, "B")
        R, _, ldr = _dense(R, self.n_rows, "R")
        P, _, ldp = _dense(P, self.n_rows, "P")
        Xout, _, ldxo = _dense(Xout, self.n_rows, "Xout")
        self._need(X.dtype)
        assert B.shape[1] == R.shape[1] == P.shape[1] == Xout.shape[1] == K
        assert B.dtype == R.dtype == P.dtype == Xout.dtype == X.dtype
        assert X.data_ptr() != Xout.data_ptr()
        check(lib().gla_cheby_first(self._h, dtype_code(X.dtype), K, ptr(B), ldb, ptr(X), ldx, ptr(R), ldr, ptr(P),
                                    ldp, ptr(Xout), ldxo, float(alpha), which, stream_ptr(self.device)))

    @_on_device
    def cheby_next(self, P, Pout, R, X, alpha_old, alpha_new, beta, which=ALL_TILES):
        P, K, ldp = _dense(P, self.n_cols, "P")
        Pout, _, ldpo = _dense(Pout, self.n_rows, "Pout")
        R, _, ldr = _dense(R, self.n_rows, "R")
        X, _, ldx = _dense(X, self.n_rows, "X")
        self._need(P.dtype)
        assert Pout.shape[1] == R.shape[1] == X.shape[1] == K
        assert Pout.dtype == R.dtype == X.dtype == P.dtype
        assert P.data_ptr() != Pout.data_ptr()
        check(lib().gla_cheby_next(self._h, dtype_code(P.dtype), K, ptr(P), ldp, ptr(Pout), ldpo, ptr(R), ldr,
                                   ptr(X), ldx, float(alpha_old), float(alpha_new), float(beta), which,
                                   stream_ptr(self.device)))

    @_on_device
    def cheby_steps(self, P0, P1, R, X, alpha_old, alpha_new, beta):
        """len(alpha_new) consecutive cheby_next steps enqueued by ONE library call (gla_cheby_steps): step i gathers from
        P0 (i even) / P1 (i odd) and writes the other; R, X in place."""
        import ctypes
        k = len(alpha_new)
        if k == 0:
            return
        P0, K, ldp0 = _dense(P0, self.n_cols, "P0")
        P1, _, ldp1 = _dense(P1, self.n_rows, "P1")
        R, _, ldr = _dense(R, self.n_rows, "R")
        X, _, ldx = _dense(X, self.n_rows, "X")
        self._need(P0.dtype)
        assert P1.shape[1] == R.shape[1] == X.shape[1] == K and P1.dtype == R.dtype == X.dtype == P0.dtype
        assert P0.data_ptr() != P1.data_ptr() and len(alpha_old) == k and len(beta) == k
        arr = lambda v: (ctypes.c_double * k)(*[float(t) for t in v])      # noqa: E731
        check(lib().gla_cheby_steps(self._h, dtype_code(P0.dtype), K, k, ptr(P0), ldp0, ptr(P1), ldp1, ptr(R), ldr, ptr(X), ldx,
                                    arr(alpha_old), arr(alpha_new), arr(beta), stream_ptr(self.device)))

    @_on_device
    def cheby_fused(self, B, X0, Xout, R, P, alphas, betas, Pprev=None, Bcopy=None):
        """Steps 1 .. len(alphas) (2 .. 4) of the Chebyshev relaxation in ONE pass (gla_cheby_fused).  Returns False when
        the plan / operands do not qualify (the caller runs cheby_first / cheby_next step by step), True otherwise."""
        X0, K, ldx0 = _dense(X0, self.n_cols, "X0")
        B, _, ldb = _dense(B, self.n_rows, "B")
        Xout, _, ldxo = _dense(Xout, self.n_rows, "Xout")
        R, _, ldr = _dense(R, self.n_rows, "R")
        P, _, ldp = _dense(P, self.n_rows, "P")
        self._need(X0.dtype)
        assert B.shape[1] == R.shape[1] == P.shape[1] == Xout.shape[1] == K
        assert B.dtype == R.dtype == P.dtype == Xout.dtype == X0.dtype and X0.data_ptr() != Xout.data_ptr()
        ldpp = 0
        if Pprev is not None:
            Pprev, _, ldpp = _dense(Pprev, self.n_rows, "Pprev")
        ldbc = 0
        if Bcopy is not None:
            Bcopy, _, ldbc = _dense(Bcopy, self.n_rows, "Bcopy")
        deg = len(alphas)
        a = (ctypes.c_double * deg)(*[float(v) for v in alphas])
        bt = (ctypes.c_double * deg)(*[float(v) for v in betas])
        rc = lib().gla_cheby_fused(self._h, dtype_code(X0.dtype), K, deg, ptr(B), ldb, ptr(X0), ldx0, ptr(Xout), ldxo,
                                   ptr(R), ldr, ptr(P), ldp, ptr(Pprev), ldpp, ptr(Bcopy), ldbc, a, bt,
                                   stream_ptr(self.device))
        if rc == -5:
            return False
        check(rc)
        return True

    @_on_device
    def cheby_fused_next(self, Rin, Xin, Pin, Xout, R, P, alpha_prev, alphas, betas, Pprev=None):
        """Steps k + 1 .. k + len(alphas) of the Chebyshev relaxation in ONE pass, from (r, x, p) after step k
        (gla_cheby_fused_next; out of place).  False when the plan / operands do not qualify."""
        Pin, K, ldpi = _dense(Pin, self.n_cols, "Pin")
        Rin, _, ldri = _dense(Rin, self.n_rows, "Rin")
        Xin, _, ldxi = _dense(Xin, self.n_rows, "Xin")
        Xout, _, ldxo = _dense(Xout, self.n_rows, "Xout")
        R, _, ldr = _dense(R, self.n_rows, "R")
        P, _, ldp = _dense(P, self.n_rows, "P")
        self._need(Pin.dtype)
        assert Rin.shape[1] == Xin.shape[1] == Xout.shape[1] == R.shape[1] == P.shape[1] == K
        assert len({t.dtype for t in (Rin, Xin, Pin, Xout, R, P)}) == 1
        assert Xin.data_ptr() != Xout.data_ptr() and Rin.data_ptr() != R.data_ptr() and Pin.data_ptr() != P.data_ptr()
        ldpp = 0
        if Pprev is not None:
            Pprev, _, ldpp = _dense(Pprev, self.n_rows, "Pprev")
        deg = len(alphas)
        a = (ctypes.c_double * deg)(*[float(v) for v in alphas])
        bt = (ctypes.c_double * deg)(*[float(v) for v in betas])
        rc = lib().gla_cheby_fused_next(self._h, dtype_code(Pin.dtype), K, deg, ptr(Rin), ldri, ptr(Xin), ldxi, ptr(Pin), ldpi,
                                        ptr(Xout), ldxo, ptr(R), ldr, ptr(P), ldp, ptr(Pprev), ldpp, float(alpha_prev), a, bt,
                                        stream_ptr(self.device))
        if rc == -5:
            return False
        check(rc)
        return True

    @_on_device
    def power_step(self, X, nrm=None, out=None, which=ALL_TILES, work=None, sumsq=None):
        """Fused power-method step: Y = A (X ./ nrm), sumsq[k] = sum_i Y[i,k]^2 (PowerMethodGNN.py:86-207).
        Returns (Y, sumsq, work); raises GlaError(-5) when the plan / K has no fused kernel."""
        X, K, ldx = _dense(X, self.n_cols, "X")
        self._need(X.dtype)
        code = dtype_code(X.dtype)
        Y = torch.empty((self.n_rows, K), dtype=X.dtype, device=self.device) if out is None else out
        Y, Ky, ldy = _dense(Y, self.n_rows, "Y")
        assert Ky == K and Y.dtype == X.dtype
        if nrm is not None:
            assert nrm.numel() == K and nrm.dtype == X.dtype and nrm.is_cuda and nrm.is_contiguous()
        if work is None:
            work = torch.empty(lib().gla_power_work_bytes(self._h, K, code), dtype=torch.uint8, device=self.device)
        if sumsq is None:
            sumsq = torch.empty(K, dtype=X.dtype, device=self.device)
        rnrm = None if nrm is None else torch.reciprocal(nrm)
        check(lib().gla_power_step(self._h, code, K, ptr(X), ldx, ptr(nrm), ptr(rnrm), ptr(Y), ldy, ptr(sumsq),
                                   ptr(work), which, stream_ptr(self.device)))
        return Y, sumsq, work

    def power_fused_max_iters(self, K, dtype):
        """Power iterations gla_power_fused takes per pass on this plan with K right-hand sides (0: use power_step)."""
        if self.val_dtype != dtype:
            return 0
        return int(lib().gla_power_fused_max_iters(self._h, dtype_code(dtype), int(K)))

    @_on_device
    def power_fused(self, X, iters, nrm=None, out=None):
        """`iters` power iterations in ONE pass (gla_power_fused): Y = A^iters (X ./ nrm) and the sums of squares of the last two
        levels, sumsq [2, K]: sumsq[0] = sum_i (A^(iters-1) (X ./ nrm))[i]^2, sumsq[1] = sum_i Y[i]^2.  X and Y must not alias."""
        X, K, ldx = _dense(X, self.n_cols, "X")
        self._need(X.dtype)
        code = dtype_code(X.dtype)
        Y = torch.empty((self.n_rows, K), dtype=X.dtype, device=self.device) if out is None else out
        Y, Ky, ldy = _dense(Y, self.n_rows, "Y")
        assert Ky == K and Y.dtype == X.dtype
        if nrm is not None:
            assert nrm.numel() == K and nrm.dtype == X.dtype and nrm.is_cuda and nrm.is_contiguous()
        nbytes = int(lib().gla_power_fused_work_bytes(self._h, code, K, int(iters)))
        if nbytes <= 0:
            raise _lib.GlaError("gla_power_fused: no kernel for this plan / K / iteration count (error -5)")
        work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        sumsq = torch.empty((2, K), dtype=X.dtype, device=self.device)
        check(lib().gla_power_fused(self._h, code, K, ptr(X), ldx, ptr(nrm), ptr(Y), ldy, ptr(sumsq), ptr(work), int(iters),
                                    stream_ptr(self.device)))
        return Y, sumsq

    @_on_device
    def power_rayleigh(self, Y, nrm, Bout, B2out):
        """The tail of the power method in one pass (gla_power_rayleigh): Bout = Y ./ nrm, B2out = Bout^2; returns
        (n_A, sum b^2) [K] each, or None when the plan / K has no fused kernel (the caller runs the unfused tail)."""
        Y, K, ldy = _dense(Y, self.n_cols, "Y")
        Bout, Kb, ldb = _dense(Bout, self.n_rows, "Bout")
        B2out, Kb2, ldb2 = _dense(B2out, self.n_rows, "B2out")
        if self.n_rows != self.n_cols or Kb != K or Kb2 != K or Bout.dtype != Y.dtype or B2out.dtype != Y.dtype:
            return None
        self._need(Y.dtype)
        code = dtype_code(Y.dtype)
        assert nrm.numel() == K and nrm.dtype == Y.dtype and nrm.is_cuda and nrm.is_contiguous()
        work = torch.empty(2 * lib().gla_power_work_bytes(self._h, K, code), dtype=torch.uint8, device=self.device)
        n_A = torch.empty(K, dtype=Y.dtype, device=self.device)
        bb = torch.empty(K, dtype=Y.dtype, device=self.device)
        rnrm = torch.reciprocal(nrm)
        rc = lib().gla_power_rayleigh(self._h, code, K, ptr(Y), ldy, ptr(nrm), ptr(rnrm), ptr(Bout), ldb, ptr(B2out), ldb2,
                                      ptr(n_A), ptr(bb), ptr(work), stream_ptr(self.device))
        if rc == -5:
            return None
        check(rc)
        return n_A, bb

    @_on_device
    def segment_reduce(self, src, reduce="sum"):
        """torch_scatter.scatter(src, row, dim=0, dim_size=n_rows, reduce=...) on the plan's row structure; src is in
        EDGE order, [nnz] or [nnz, K]; returns [n_rows] resp. [n_rows, K]."""
        op = {"sum": 0, "add": 0, "max": 1, "min": 2, "mean": 3}[reduce]
        one_d = src.dim() == 1
        if src.shape[0] != self.nnz:
            raise ValueError(f"src has {src.shape[0]} rows, plan has {self.nnz} edges")
        width = 1
        for dsz in src.shape[1:]:
            width *= dsz
        s2 = src.reshape(self.nnz, width)
        if s2.shape[1] > 1 and s2.stride(1) != 1 or not s2.is_cuda:
            s2 = s2.contiguous().to(self.device)
        K = s2.shape[1]
        ld = s2.stride(0) if self.nnz > 1 else K
        out = torch.empty((self.n_rows, K), dtype=s2.dtype, device=self.device)
        check(lib().gla_segment_reduce(self._h, dtype_code(s2.dtype), K, ptr(s2), max(ld, K), op, ptr(out), K,
                                       stream_ptr(self.device)))
        return out.reshape(-1) if one_d else out

    @_on_device
    def segment_stats(self, src):
        """cat(min, mean, sum, max) of src over each row's edges, one pass: src [nnz] or [nnz, K] in EDGE order ->
        [n_rows, 4K] (TrainableJacobiGNN.py:53-70)."""
        if src.shape[0] != self.nnz:
            raise ValueError(f"src has {src.shape[0]} rows, plan has {self.nnz} edges")
        width = 1
        for dsz in src.shape[1:]:
            width *= dsz
        s2 = src.reshape(self.nnz, width)
        if s2.shape[1] > 1 and s2.stride(1) != 1 or not s2.is_cuda:
            s2 = s2.contiguous().to(self.device)
        K = s2.shape[1]
        ld = s2.stride(0) if self.nnz > 1 else K
        out = torch.empty((self.n_rows, 4 * K), dtype=s2.dtype, device=self.device)
        check(lib().gla_segment_stats(self._h, dtype_code(s2.dtype), K, ptr(s2), max(ld, K), ptr(out), 4 * K,
                                      stream_ptr(self.device)))
        return out

    @_on_device
    def edge_dot(self, G, X):
        """out[e] = sum_k G[row_e, k] * X[col_e, k] in EDGE order, [nnz] (dL/dvalues of Y = A X for G = dL/dY)."""
        G, K, ldg = _dense(G, self.n_rows, "G")
        X, Kx, ldx = _dense(X, self.n_cols, "X")
        if Kx != K or X.dtype != G.dtype:
            raise ValueError("G and X must have the same number of columns and dtype")
        out = torch.empty(self.nnz, dtype=G.dtype, device=self.device)
        check(lib().gla_edge_dot(self._h, dtype_code(G.dtype), K, ptr(G), ldg, ptr(X), ldx, ptr(out),
                                 stream_ptr(self.device)))
        return out

    @_on_device
    def edge_products(self, X):
        """c_e = A_e * X[col_e] in EDGE order, [nnz, K]."""
        X, K, ldx = _dense(X, self.n_cols, "X")
        self._need(X.dtype)
        C = torch.empty((self.nnz, K), dtype=X.dtype, device=self.device)
        check(lib().gla_edge_products(self._h, dtype_code(X.dtype), K, ptr(X), ldx, ptr(C), K,
                                      stream_ptr(self.device)))
        return C

    @_on_device
    def edge_attr(self, X):
        """[A_e, A_e * X[col_e]] in EDGE order, [nnz, 1 + K]: the edge attributes an EdgeUpdate returns, one pass."""
        X, K, ldx = _dense(X, self.n_cols, "X")
        self._need(X.dtype)
        out = torch.empty((self.nnz, 1 + K), dtype=X.dtype, device=self.device)
        check(lib().gla_edge_attr(self._h, dtype_code(X.dtype), K, ptr(X), ldx, ptr(out), 1 + K,
                                  stream_ptr(self.device)))
        return out

    @_on_device
    def soc_classic(self, theta, dtype):
        self._need(dtype)
        v = torch.empty((self.n_rows, 1), dtype=dtype, device=self.device)
        S = torch.empty((self.nnz,), dtype=dtype, device=self.device)
        check(lib().gla_soc_classic(self._h, dtype_code(dtype), float(theta), ptr(v), ptr(S),
                                    stream_ptr(self.device)))
        return v, S

    @_on_device
    def direct_interp(self, diag, cf, S, want_alpha=False):
        dtype = diag.dtype
        self._need(dtype)
        for name, t, n in (("diag", diag, self.n_rows), ("cf", cf, self.n_cols), ("S", S, self.nnz)):
            if t.numel() != n or not t.is_contiguous() or t.dtype != dtype or not t.is_cuda:
                raise ValueError(f"{name} must be a contiguous {dtype} GPU vector of {n} elements")
        w = torch.empty((self.nnz,), dtype=dtype, device=self.device)
        alpha = torch.empty((self.n_rows,), dtype=dtype, device=self.device) if want_alpha else None
        check(lib().gla_direct_interp(self._h, dtype_code(dtype), ptr(diag), ptr(cf), ptr(S), ptr(w), ptr(alpha),
                                      stream_ptr(self.device)))
        return (w, alpha) if want_alpha else w

    @_on_device
    def soc_sa(self, diag):
        dtype = diag.dtype
        self._need(dtype)
        if diag.numel() != self.n_rows or not diag.is_contiguous() or not diag.is_cuda:
            raise ValueError("diag must be a contiguous GPU vector of n_rows elements")
        S = torch.empty((self.nnz,), dtype=dtype, device=self.device)
        check(lib().gla_soc_sa(self._h, dtype_code(dtype), ptr(diag), ptr(S), stream_ptr(self.device)))
        return S


# ---- dense helpers ------------------------------------------------------------------------------
def col_dot(X, Y):
    """Column sums of X .* Y -> [K] on the device (PowerMethodGNN.py:50-61 vertex_to_global_aggregation)."""
    n = X.shape[0]
    X, K, ldx = _dense(X, n, "X")
    Y, Ky, ldy = _dense(Y, n, "Y")
    assert K == Ky and X.dtype == Y.dtype
    code = dtype_code(X.dtype)
    work = torch.empty(lib().gla_dot_work_bytes(n, K, code), dtype=torch.uint8, device=X.device)
    out = torch.empty(K, dtype=X.dtype, device=X.device)
    with _DeviceGuard(X.device):
        check(lib().gla_col_dot(code, n, K, ptr(X), ldx, ptr(Y), ldy, ptr(out), ptr(work), stream_ptr(X.device)))
    return out


def col_scale_rsqrt(X, sumsq, out=None):
    """X[:, k] / sqrt(sumsq[k]) (PowerMethodGNN.py:160-207); `out`: row-major [n, K] destination (may be strided)."""
    n = X.shape[0]
    X, K, ldx = _dense(X, n, "X")
    assert sumsq.numel() == K and sumsq.dtype == X.dtype and sumsq.is_cuda
    Y = torch.empty((n, K), dtype=X.dtype, device=X.device) if out is None else out
    Y2, Ky, ldy = _dense(Y, n, "out")
    assert Ky == K and Y2.dtype == X.dtype
    with _DeviceGuard(X.device):
        check(lib().gla_col_scale_rsqrt(dtype_code(X.dtype), n, K, ptr(X), ldx, ptr(sumsq), ptr(Y2), ldy,
                                        stream_ptr(X.device)))
    return Y


def invalidate(*tensors):
    """Drop everything this package cached ON the given tensors: the CSR plan of an `edgeij_pair`, the CSR-ordered copy
    of an `edge_attr`'s values (the plan re-gathers them on the next call), the GPU copy of a CPU operand, the remembered
    relaxation weight of a GPU-resident `g`.  The caches
    are keyed on tensor identity plus torch's in-place version counter, which does NOT see writes made through a raw
    pointer, a NumPy view sharing the memory, or `.data`: after such a write call this (or pass a fresh tensor)."""
    for t in tensors:
        cached_copy = getattr(t, "_gla_dev_copy", None)
        if cached_copy is not None:
            invalidate(cached_copy[1])
        for attr in ("_gla_plan", "_gla_dev_copy", "_gla_gnn_input", "_gla_scalar0"):
            if hasattr(t, attr):
                try:
                    delattr(t, attr)
                except Exception:
                    pass
    for ref in list(_PLANS_WITH_VALUES):
        plan = ref()
        if plan is None:
            _PLANS_WITH_VALUES.discard(ref)
        elif plan._val_ref is not None and any(plan._val_ref() is t for t in tensors):
            plan._val_key = plan._val_ref = None


_PLANS_WITH_VALUES = set()


# ---- plan cache ---------------------------------------------------------------------------------
def plan_for(edgeij_pair, n_rows, n_cols=None, edge_attr=None):
    """Plan cached on the edge-index tensor object itself (lifetime = the tensor's); rebuilt if the tensor
    was modified in place (version counter) or is used with another shape."""
    n_cols = n_rows if n_cols is None else n_cols
    key = (edgeij_pair.data_ptr(), edgeij_pair._version, tuple(edgeij_pair.shape), int(n_rows), int(n_cols))
    cached = getattr(edgeij_pair, "_gla_plan", None)
    if cached is None or cached[0] != key:
        plan = CsrPlan(edgeij_pair, n_rows, n_cols)
        try:
            edgeij_pair._gla_plan = (key, plan)
        except Exception:
            pass
    else:
        plan = cached[1]
    if edge_attr is not None:
        plan.set_values_cached(edge_attr)
    return plan
