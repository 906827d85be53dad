// A host WITHOUT torch or Python on the C ABI of include/gla.h: build the N x N 5-point Laplacian on the device,
// run n_sweeps fused Jacobi sweeps (JacobiGNN.forward, pytorch/JacobiGNN.py:138-148) and check the result bit for
// bit against a plain host loop written in the reference's operation order (edge-order row sums, separate
// multiply / add, x + (w*(b - Ax))/A_ii).
//
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -Iinclude examples/c_host_jacobi.cpp \
//         -Lgnn_applied_linear_algebra_amd -lgla_hip -Wl,-rpath,$PWD/gnn_applied_linear_algebra_amd -o c_host_jacobi
//   ./c_host_jacobi 512 10
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gla.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define GLA_OK_(x) do { int rc_ = (x); if (rc_ != 0) { \
  std::fprintf(stderr, "%s:%d gla error %d: %s\n", __FILE__, __LINE__, rc_, gla_error_string(rc_)); return 3; } } while (0)

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? std::atoll(argv[1]) : 512;
  const int sweeps = argc > 2 ? std::atoi(argv[2]) : 10;
  const double w = 0.7;
  const int64_t n = N * N;
  const int64_t nnz = gla_laplacian2d_nnz(N, N, 0, n);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  int64_t *d_row, *d_col;
  double *d_val, *d_diag, *d_b, *d_x, *d_out, *d_tmp;
  HIP_OK(hipMalloc(&d_row, nnz * sizeof(int64_t)));
  HIP_OK(hipMalloc(&d_col, nnz * sizeof(int64_t)));
  HIP_OK(hipMalloc(&d_val, nnz * sizeof(double)));
  GLA_OK_(gla_laplacian2d_coo(N, N, 0, n, d_row, d_col, d_val, GLA_F64, st));

  gla_csr_t A = nullptr;
  GLA_OK_(gla_csr_create_from_coo(n, n, nnz, d_row, d_col, st, &A));
  GLA_OK_(gla_csr_set_values(A, d_val, 1, GLA_F64, st));

  std::vector<double> diag(n, -4.0), b(n), x(n);
  uint64_t s = 24601;                                   // the reference's seed (matlab/test_jacobi.m:52), own LCG
  auto next = [&s]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; };
  for (int64_t i = 0; i < n; ++i) { b[i] = next(); x[i] = next(); }
  HIP_OK(hipMalloc(&d_diag, n * sizeof(double)));
  HIP_OK(hipMalloc(&d_b, n * sizeof(double)));
  HIP_OK(hipMalloc(&d_x, n * sizeof(double)));
  HIP_OK(hipMalloc(&d_out, n * sizeof(double)));
  HIP_OK(hipMalloc(&d_tmp, n * sizeof(double)));
  HIP_OK(hipMemcpyAsync(d_diag, diag.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(d_b, b.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(d_x, x.data(), n * sizeof(double), hipMemcpyHostToDevice, st));

  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  GLA_OK_(gla_jacobi_sweeps(A, GLA_F64, 1, d_diag, d_b, 1, d_x, 1, d_out, d_tmp, 1, w, sweeps, st));   // warm-up
  HIP_OK(hipEventRecord(e0, st));
  GLA_OK_(gla_jacobi_sweeps(A, GLA_F64, 1, d_diag, d_b, 1, d_x, 1, d_out, d_tmp, 1, w, sweeps, st));
  HIP_OK(hipEventRecord(e1, st));
  std::vector<double> got(n);
  HIP_OK(hipMemcpyAsync(got.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));

  // host restatement: vertex k = j*N + i, neighbours in ascending column order k-N, k-1, k, k+1, k+N
  std::vector<double> cur(x), nxt(n);
  for (int it = 0; it < sweeps; ++it) {
    for (int64_t k = 0; k < n; ++k) {
      const int64_t i = k % N, j = k / N;
      double acc = 0.0;
      if (j > 0) acc = acc + 1.0 * cur[k - N];
      if (i > 0) acc = acc + 1.0 * cur[k - 1];
      acc = acc + (-4.0) * cur[k];
      if (i < N - 1) acc = acc + 1.0 * cur[k + 1];
      if (j < N - 1) acc = acc + 1.0 * cur[k + N];
      nxt[k] = cur[k] + (w * (b[k] - acc)) / diag[k];
    }
    cur.swap(nxt);
  }
  int64_t bad = 0;
  for (int64_t k = 0; k < n; ++k) bad += std::memcmp(&cur[k], &got[k], sizeof(double)) != 0;
  gla_csr_info_t info;
  GLA_OK_(gla_csr_info(A, &info));
  std::printf("N=%lld n=%lld nnz=%lld tiles=%d col16=%d diagonals=%d  %d sweeps in %.3f ms (%.1f us/sweep)  "
              "mismatching values: %lld\n",
              (long long)N, (long long)n, (long long)nnz, info.n_tiles, info.col16, info.dia, sweeps, ms,
              1e3 * ms / sweeps, (long long)bad);
  gla_csr_destroy(A);
  for (void* q : {(void*)d_row, (void*)d_col, (void*)d_val, (void*)d_diag, (void*)d_b, (void*)d_x, (void*)d_out,
                  (void*)d_tmp})
    (void)hipFree(q);
  std::puts(bad == 0 ? "C-ABI host check: OK (bit-exact)" : "C-ABI host check: FAILED");
  return bad == 0 ? 0 : 1;
}
