"""ctypes/numpy binding of oracle/libgla_oracle.so (TEST INFRASTRUCTURE ONLY, see gla_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgla_oracle.so")
_lib = None

_p = ctypes.c_void_p
_i64 = ctypes.c_int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "gla_oracle.c")):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Csr:
    """Stable COO -> CSR of an edge list (numpy int64 [2, nnz]) with fp64 values [nnz]."""

    def __init__(self, ei, ev, n):
        ei = np.ascontiguousarray(ei, dtype=np.int64)
        self.n, self.nnz = int(n), ei.shape[1]
        self.rowptr = np.zeros(self.n + 1, dtype=np.int64)
        self.col = np.zeros(self.nnz, dtype=np.int32)
        self.perm = np.zeros(self.nnz, dtype=np.int64)
        rc = lib().oracle_coo_to_csr(_i64(self.n), _i64(self.nnz), _ptr(ei[0].copy()), _ptr(ei[1].copy()),
                                     _ptr(self.rowptr), _ptr(self.col), _ptr(self.perm))
        if rc:
            raise ValueError(f"oracle_coo_to_csr failed: {rc}")
        self.val = np.ascontiguousarray(np.asarray(ev, dtype=np.float64).reshape(self.nnz, -1)[:, 0][self.perm])

    def spmm(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(self.n, -1)
        K = X.shape[1]
        Y = np.empty_like(X)
        lib().oracle_spmm_f64(_i64(self.n), _ptr(self.rowptr), _ptr(self.col), _ptr(self.val), ctypes.c_int(K),
                              _ptr(X), _i64(K), _ptr(Y), _i64(K))
        return Y

    def jacobi(self, diag, b, x, w, n_iters):
        x = np.array(x, dtype=np.float64).reshape(-1).copy()
        rc = lib().oracle_jacobi_f64(_i64(self.n), _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                     _ptr(np.ascontiguousarray(diag, dtype=np.float64).reshape(-1)),
                                     _ptr(np.ascontiguousarray(b, dtype=np.float64).reshape(-1)), _ptr(x),
                                     ctypes.c_double(w), ctypes.c_int(n_iters))
        assert rc == 0
        return x.reshape(-1, 1)

    def cheby(self, b, x, c, d, deg):
        x = np.array(x, dtype=np.float64).reshape(-1).copy()
        r, p, g = np.zeros_like(x), np.zeros_like(x), np.zeros(4)
        rc = lib().oracle_cheby_f64(_i64(self.n), _ptr(self.rowptr), _ptr(self.col), _ptr(self.val),
                                    _ptr(np.ascontiguousarray(b, dtype=np.float64).reshape(-1)), _ptr(x), _ptr(r),
                                    _ptr(p), ctypes.c_double(c), ctypes.c_double(d), ctypes.c_int(deg), _ptr(g))
        assert rc == 0
        return x, r, p, g

    def soc_classic(self, theta):
        v, S = np.zeros(self.n), np.zeros(self.nnz)
        lib().oracle_soc_classic_f64(_i64(self.n), _ptr(self.rowptr), _ptr(self.val), ctypes.c_double(theta),
                                     _ptr(v), _ptr(S))
        out = np.empty(self.nnz)
        out[self.perm] = S                                  # back to edge order
        return v, out

    def direct_interp(self, S_edge, diag, cf):
        S = np.ascontiguousarray(np.asarray(S_edge, dtype=np.float64).reshape(-1)[self.perm])
        w = np.zeros(self.nnz)
        lib().oracle_direct_interp_f64(_i64(self.n), _ptr(self.rowptr), _ptr(self.col), _ptr(self.val), _ptr(S),
                                       _ptr(np.ascontiguousarray(diag, dtype=np.float64).reshape(-1)),
                                       _ptr(np.ascontiguousarray(cf, dtype=np.float64).reshape(-1)), _ptr(w))
        out = np.empty(self.nnz)
        out[self.perm] = w
        return out
