/*
 * gla_oracle.c -- plain-C CPU restatement of the reference's hot path in CSR form.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle for sizes where the torch port in reference_port.py is too slow,
 * e.g. 20 Jacobi sweeps at 4096^2).  Never linked or loaded by the product package.
 *
 * Every routine follows the arithmetic of the reference module it cites, with the aggregation written the way
 * torch_scatter's CPU scatter-sum executes it: rows are summed sequentially in EDGE order starting from 0.0,
 * products and sums are rounded separately (compile with -ffp-contract=off).  It is pinned bit-for-bit against
 * the golden vectors of the unmodified reference in tests/test_oracle_golden.py::test_c_oracle_*.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* Stable counting sort of a COO edge list by row: rowptr[n+1], perm[nnz] (CSR position -> edge id).
 * = the index side of torch_scatter.scatter(src, edgeij_pair[0], dim_size=n). */
int oracle_coo_to_csr(int64_t n, int64_t nnz, const int64_t* row, const int64_t* col,
                      int64_t* rowptr, int32_t* col_csr, int64_t* perm) {
  memset(rowptr, 0, (size_t)(n + 1) * sizeof(int64_t));
  for (int64_t e = 0; e < nnz; ++e) {
    if (row[e] < 0 || row[e] >= n) return -2;
    rowptr[row[e] + 1]++;
  }
  for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  int64_t* next = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  if (!next) return -1;
  memcpy(next, rowptr, (size_t)n * sizeof(int64_t));
  for (int64_t e = 0; e < nnz; ++e) {
    int64_t p = next[row[e]]++;
    perm[p] = e;
    col_csr[p] = (int32_t)col[e];
  }
  free(next);
  return 0;
}

/* y = A x for K right-hand sides, row-major.  MatVecGNN.py:43-114. */
void oracle_spmm_f64(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val, int K,
                     const double* X, int64_t ldx, double* Y, int64_t ldy) {
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        double c = val[p] * X[(int64_t)col[p] * ldx + k];      /* EdgeUpdate: c_ij = A_ij * x_j  (:80) */
        acc = acc + c;                                         /* scatter-sum in edge order      (:60) */
      }
      Y[i * ldy + k] = acc;
    }
}

/* n_iters weighted-Jacobi sweeps, x <- x + (w*(b - A x))/A_ii.  JacobiGNN.py:52-148 (update :119).
 * x is updated through a scratch buffer (Jacobi, not Gauss-Seidel). */
int oracle_jacobi_f64(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val,
                      const double* diag, const double* b, double* x, double w, int n_iters) {
  double* xn = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  if (!xn) return -1;
  for (int it = 0; it < n_iters; ++it) {
    for (int64_t i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        double c = val[p] * x[col[p]];
        acc = acc + c;
      }
      double t = w * (b[i] - acc);
      xn[i] = x[i] + t / diag[i];
    }
    memcpy(x, xn, (size_t)n * sizeof(double));
  }
  free(xn);
  return 0;
}

/* Chebyshev relaxation of degree deg; x, r, p are [n]; g_out[4] = {c, d, alpha, beta}.
 * ChebyGNN.py:49-353 (alpha/beta recurrences :137, :262-263, :282-283; alpha timing SURVEY.md 3.2). */
int oracle_cheby_f64(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val, const double* b,
                     double* x, double* r, double* p, double c, double d, int deg, double* g_out) {
  double alpha = 0.0, beta = 0.0;
  double* z = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  if (!z) return -1;
  for (int it = 1; it <= deg; ++it) {
    const double* src = (it == 1) ? x : p;
    for (int64_t i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
        double t = val[q] * src[col[q]];
        acc = acc + t;
      }
      z[i] = acc;
    }
    if (it == 1) {
      alpha = 1 / d;
      for (int64_t i = 0; i < n; ++i) {
        r[i] = b[i] - z[i];
        p[i] = r[i];
        double t = alpha * p[i];
        x[i] = x[i] + t;
      }
    } else {
      double alpha_old = alpha;
      if (it == 2) { double ca = c * alpha; beta = 0.5 * (ca * ca); }
      else { double h = (c * alpha) / 2; beta = h * h; }
      alpha = 1 / (d - beta / alpha);
      for (int64_t i = 0; i < n; ++i) {
        double t = alpha_old * z[i];
        r[i] = r[i] - t;
        double u = beta * p[i];
        p[i] = r[i] + u;
        double v = alpha * p[i];
        x[i] = x[i] + v;
      }
    }
  }
  g_out[0] = c; g_out[1] = d; g_out[2] = alpha; g_out[3] = beta;
  free(z);
  return 0;
}

/* Classical strength of connection on the off-diagonal CSR: v_i = max_j(-A_ij) (0 if the row is empty),
 * S_p = relu(-A_p / v_i - theta) in CSR order.  SOCClassicGNN.py:50-129. */
void oracle_soc_classic_f64(int64_t n, const int64_t* rowptr, const double* val, double theta, double* v,
                            double* S) {
  for (int64_t i = 0; i < n; ++i) {
    double m = 0.0;
    if (rowptr[i] < rowptr[i + 1]) {
      m = -1 * val[rowptr[i]];
      for (int64_t p = rowptr[i] + 1; p < rowptr[i + 1]; ++p) {
        double a = -1 * val[p];
        if (a > m || a != a) m = a;
      }
    }
    v[i] = m;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
      double q = (-1 * val[p]) / m - theta;
      S[p] = (q != q) ? q : (q > 0.0 ? q : 0.0);
    }
  }
}

/* Direct interpolation weights in CSR order.  DirectInterpGNN.py:50-152. */
void oracle_direct_interp_f64(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val,
                              const double* S, const double* diag, const double* cf, double* w) {
  for (int64_t i = 0; i < n; ++i) {
    double num = 0.0, den = 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
      num = num + val[p];
      double t = (val[p] * S[p]) * cf[col[p]];
      den = den + t;
    }
    double gamma = num / den;
    double alpha = (1 / diag[i]) * gamma;
    double omc = 1 - cf[i];
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) w[p] = omc * ((-1 * val[p]) * alpha);
  }
}
