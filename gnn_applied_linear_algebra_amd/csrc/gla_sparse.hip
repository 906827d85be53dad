// libgla_hip.so -- V-cycle SETUP on the device (gfx950): COO coalesce and sparse-sparse product.
//
// The reference builds its coarse operator with torch's CPU sparse algebra every cycle:
//   Ac = P.t() @ (A @ P)            pytorch/VCycle.py:209
//   A.coalesce()                    pytorch/UtilsGNN.py:74-78 (coo_to_gnn_input), VCycle.py:63,77,96,141,157
// Both are restated here as expand -> sort -> compress (ESC) passes:
//   expand    one thread per nonzero of A (CSR order) emits A_ik * B_kj for the row k of B, keyed (i, j);
//   sort      stable LSD radix sort of the 64-bit keys (rocPRIM), so duplicates keep their emission order;
//   compress  run heads are flagged and scanned, one lane per output entry adds its run SEQUENTIALLY, so the
//             result is deterministic and its summation order is defined (A's CSR order, then B's).
// These are setup-time, HBM-bound integer/byte passes (no MFMA); they are not on the per-sweep hot path.
#include <cstring>
#include "gla_internal.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

inline unsigned grid_for(int64_t work, int64_t per_block = GLA_BLOCK, int64_t cap = 256 * 16) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// products emitted per nonzero of A = length of the matching row of B
__global__ __launch_bounds__(GLA_BLOCK) void spgemm_count_kernel(const int32_t* __restrict__ colA, int64_t nnzA,
                                                                 const int32_t* __restrict__ rowptrB,
                                                                 int64_t* __restrict__ cnt) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnzA; p += (int64_t)gridDim.x * GLA_BLOCK) {
    const int k = colA[p];
    cnt[p] = rowptrB[k + 1] - rowptrB[k];
  }
}

// row id of every CSR position (rows are short: one lane per row writes its span)
__global__ __launch_bounds__(GLA_BLOCK) void csr_rows_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                             int32_t* __restrict__ rows) {
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK)
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) rows[p] = (int32_t)r;
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void spgemm_expand_kernel(
    const int32_t* __restrict__ rowsA, const int32_t* __restrict__ colA, const T* __restrict__ valA, int64_t nnzA,
    const int32_t* __restrict__ rowptrB, const int32_t* __restrict__ colB, const T* __restrict__ valB,
    int64_t n_colsB, const int64_t* __restrict__ offs, uint64_t* __restrict__ keys, T* __restrict__ vals) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnzA; p += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t i = rowsA[p];
    const int k = colA[p];
    const T a = valA[p];
    int64_t o = offs[p];
    for (int q = rowptrB[k]; q < rowptrB[k + 1]; ++q, ++o) {
      keys[o] = (uint64_t)(i * n_colsB + colB[q]);
      vals[o] = a * valB[q];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void coo_keys_kernel(const int64_t* __restrict__ row,
                                                             const int64_t* __restrict__ col,
                                                             const T* __restrict__ val, int64_t stride, int64_t nnz,
                                                             int64_t n_cols, uint64_t* __restrict__ keys,
                                                             T* __restrict__ vals) {
  for (int64_t e = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * GLA_BLOCK) {
    keys[e] = (uint64_t)(row[e] * n_cols + col[e]);
    vals[e] = val[e * stride];
  }
}

__global__ __launch_bounds__(GLA_BLOCK) void head_flags_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                               int64_t* __restrict__ flags) {
  for (int64_t t = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * GLA_BLOCK)
    flags[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}

// one lane per run head: sequential sum of the run, written at its compacted position
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void compress_kernel(const uint64_t* __restrict__ keys,
                                                             const T* __restrict__ vals, int64_t n,
                                                             const int64_t* __restrict__ pos, int64_t n_cols,
                                                             int64_t* __restrict__ row_out,
                                                             int64_t* __restrict__ col_out, T* __restrict__ val_out) {
  for (int64_t t = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * GLA_BLOCK) {
    const uint64_t k = keys[t];
    if (t != 0 && keys[t - 1] == k) continue;
    T acc = vals[t];
    for (int64_t u = t + 1; u < n && keys[u] == k; ++u) acc = acc + vals[u];
    const int64_t o = pos[t];
    row_out[o] = (int64_t)(k / (uint64_t)n_cols);
    col_out[o] = (int64_t)(k % (uint64_t)n_cols);
    val_out[o] = acc;
  }
}

// Setup scratch comes from one grow-only device pool (gla_release_scratch frees it): a hipMalloc / hipFree pair of
// several GB per call costs up to seconds of page mapping, far more than the kernels.  Every setup entry point sizes
// its scratch up front, reserves the pool once and bump-allocates from it; the entry points synchronise before they
// return, so the pool is free again at every call boundary.  They are NOT re-entrant (one setup call at a time).
struct Pool {
  char* base = nullptr;
  size_t cap = 0, used = 0;
} g_pool;

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

int pool_reserve(size_t bytes) {
  g_pool.used = 0;
  if (bytes <= g_pool.cap) return GLA_OK;
  if (g_pool.base) (void)hipFree(g_pool.base);
  g_pool.base = nullptr;
  g_pool.cap = 0;
  const size_t want = bytes + bytes / 8 + (1 << 20);
  GLA_HIP_CHECK(hipMalloc((void**)&g_pool.base, want));
  g_pool.cap = want;
  return GLA_OK;
}

template <typename U>
U* pool_get(size_t count) {
  const size_t bytes = al256((count ? count : 1) * sizeof(U));
  if (g_pool.used + bytes > g_pool.cap) return nullptr;          // sizing bug: callers check
  U* p = reinterpret_cast<U*>(g_pool.base + g_pool.used);
  g_pool.used += bytes;
  return p;
}

unsigned key_bits(int64_t n_rows, int64_t n_cols) {
  unsigned end_bit = 1;
  const double span = (double)n_rows * (double)n_cols;
  while (end_bit < 64 && (double)((uint64_t)1 << end_bit) < span) ++end_bit;
  return end_bit;
}

template <typename T>
size_t sort_tmp_bytes(int64_t n, unsigned end_bit) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (T*)nullptr, (T*)nullptr,
                                  (size_t)n, 0u, end_bit, (hipStream_t)0);
  return b;
}

size_t scan_tmp_bytes(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n,
                                rocprim::plus<int64_t>(), (hipStream_t)0);
  return b;
}

template <typename T>
size_t compress_need(int64_t n, unsigned end_bit) {      // scratch of sort_and_compress
  return al256(n * 8) + al256(n * sizeof(T)) + 2 * al256(n * 8) + al256(sort_tmp_bytes<T>(n, end_bit)) +
         al256(scan_tmp_bytes(n)) + 4096;
}

// sort (keys, vals) and compress runs into (row_out, col_out, val_out); returns the number of unique keys
template <typename T>
int sort_and_compress(uint64_t* keys, T* vals, int64_t n, int64_t n_rows, int64_t n_cols, int64_t* row_out,
                      int64_t* col_out, T* val_out, int64_t* nnz_out, hipStream_t st) {
  *nnz_out = 0;
  if (n == 0) return GLA_OK;
  const unsigned end_bit = key_bits(n_rows, n_cols);
  size_t tmp_bytes = sort_tmp_bytes<T>(n, end_bit), tmp2_bytes = scan_tmp_bytes(n);
  uint64_t* keys2 = pool_get<uint64_t>(n);
  T* vals2 = pool_get<T>(n);
  int64_t* flags = pool_get<int64_t>(n);
  int64_t* pos = pool_get<int64_t>(n);
  char* tmp = pool_get<char>(tmp_bytes);
  char* tmp2 = pool_get<char>(tmp2_bytes);
  if (!keys2 || !vals2 || !flags || !pos || !tmp || !tmp2) return GLA_ERR_INVALID_ARG;
  GLA_HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (size_t)n, 0u, end_bit, st));
  hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n)), dim3(GLA_BLOCK), 0, st, keys2, n, flags);
  GLA_LAUNCH_CHECK();
  GLA_HIP_CHECK(rocprim::exclusive_scan(tmp2, tmp2_bytes, flags, pos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
  int64_t last[2] = {0, 0};
  GLA_HIP_CHECK(hipMemcpyAsync(&last[0], pos + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GLA_HIP_CHECK(hipMemcpyAsync(&last[1], flags + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
  hipLaunchKernelGGL((compress_kernel<T>), dim3(grid_for(n)), dim3(GLA_BLOCK), 0, st, keys2, vals2, n, pos, n_cols,
                     row_out, col_out, val_out);
  GLA_LAUNCH_CHECK();
  GLA_HIP_CHECK(hipStreamSynchronize(st));
  *nnz_out = last[0] + last[1];
  return GLA_OK;
}

template <typename T>
int spgemm_impl(gla_csr* A, gla_csr* B, int64_t n_products, int64_t* row_out, int64_t* col_out, T* val_out,
                int64_t* nnz_out, hipStream_t st) {
  *nnz_out = 0;
  if (n_products == 0 || A->nnz == 0) return GLA_OK;
  const unsigned end_bit = key_bits(A->n_rows, B->n_cols);
  size_t tmp_bytes = scan_tmp_bytes(A->nnz);
  int rc = pool_reserve(2 * al256(A->nnz * 8) + al256(A->nnz * 4) + al256(tmp_bytes) + al256(n_products * 8) +
                        al256(n_products * sizeof(T)) + compress_need<T>(n_products, end_bit));
  if (rc) return rc;
  int64_t* cnt = pool_get<int64_t>(A->nnz);
  int64_t* offs = pool_get<int64_t>(A->nnz);
  int32_t* rowsA = pool_get<int32_t>(A->nnz);
  char* tmp = pool_get<char>(tmp_bytes);
  uint64_t* keys = pool_get<uint64_t>(n_products);
  T* vals = pool_get<T>(n_products);
  if (!cnt || !offs || !rowsA || !tmp || !keys || !vals) return GLA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(spgemm_count_kernel, dim3(grid_for(A->nnz)), dim3(GLA_BLOCK), 0, st, A->col, A->nnz, B->rowptr, cnt);
  GLA_LAUNCH_CHECK();
  GLA_HIP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, cnt, offs, (int64_t)0, (size_t)A->nnz, rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(csr_rows_kernel, dim3(grid_for(A->n_rows)), dim3(GLA_BLOCK), 0, st, A->rowptr, A->n_rows, rowsA);
  GLA_LAUNCH_CHECK();
  hipLaunchKernelGGL((spgemm_expand_kernel<T>), dim3(grid_for(A->nnz)), dim3(GLA_BLOCK), 0, st, rowsA, A->col,
                     (const T*)A->val, A->nnz, B->rowptr, B->col, (const T*)B->val, B->n_cols, offs, keys, vals);
  GLA_LAUNCH_CHECK();
  return sort_and_compress<T>(keys, vals, n_products, A->n_rows, B->n_cols, row_out, col_out, val_out, nnz_out, st);
}

template <typename T>
int coalesce_impl(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col, const T* val,
                  int64_t stride, int64_t* row_out, int64_t* col_out, T* val_out, int64_t* nnz_out, hipStream_t st) {
  *nnz_out = 0;
  if (nnz == 0) return GLA_OK;
  int rc = pool_reserve(al256(nnz * 8) + al256(nnz * sizeof(T)) + compress_need<T>(nnz, key_bits(n_rows, n_cols)));
  if (rc) return rc;
  uint64_t* keys = pool_get<uint64_t>(nnz);
  T* vals = pool_get<T>(nnz);
  if (!keys || !vals) return GLA_ERR_INVALID_ARG;
  hipLaunchKernelGGL((coo_keys_kernel<T>), dim3(grid_for(nnz)), dim3(GLA_BLOCK), 0, st, row, col, val, stride, nnz,
                     n_cols, keys, vals);
  GLA_LAUNCH_CHECK();
  return sort_and_compress<T>(keys, vals, nnz, n_rows, n_cols, row_out, col_out, val_out, nnz_out, st);
}

}  // namespace

extern "C" {

int gla_release_scratch(void) {
  if (g_pool.base) (void)hipFree(g_pool.base);
  g_pool.base = nullptr;
  g_pool.cap = g_pool.used = 0;
  return GLA_OK;
}

int gla_spgemm_count(gla_csr_t A, gla_csr_t B, int64_t* n_products_host, gla_stream_t stream) {
  if (!A || !B || !n_products_host || A->n_cols != B->n_rows) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  *n_products_host = 0;
  if (A->nnz == 0) return GLA_OK;
  size_t tmp_bytes = 0;
  GLA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tmp_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (size_t)A->nnz,
                                        rocprim::plus<int64_t>(), st));
  int rc = pool_reserve(2 * al256(A->nnz * 8) + al256(tmp_bytes) + 4096);
  if (rc) return rc;
  int64_t* cnt = pool_get<int64_t>(A->nnz);
  int64_t* offs = pool_get<int64_t>(A->nnz);
  char* tmp = pool_get<char>(tmp_bytes);
  if (!cnt || !offs || !tmp) return GLA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(spgemm_count_kernel, dim3(grid_for(A->nnz)), dim3(GLA_BLOCK), 0, st, A->col, A->nnz, B->rowptr, cnt);
  GLA_LAUNCH_CHECK();
  GLA_HIP_CHECK(rocprim::inclusive_scan(tmp, tmp_bytes, cnt, offs, (size_t)A->nnz, rocprim::plus<int64_t>(), st));
  GLA_HIP_CHECK(hipMemcpyAsync(n_products_host, offs + (A->nnz - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GLA_HIP_CHECK(hipStreamSynchronize(st));
  return GLA_OK;
}

int gla_spgemm(gla_csr_t A, gla_csr_t B, int dtype, int64_t n_products, int64_t* row_out, int64_t* col_out,
               void* val_out, int64_t* nnz_out_host, gla_stream_t stream) {
  if (!A || !B || !nnz_out_host || A->n_cols != B->n_rows || n_products < 0 ||
      (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if ((A->nnz > 0 && A->val_dtype != dtype) || (B->nnz > 0 && B->val_dtype != dtype)) return GLA_ERR_NO_VALUES;
  if (n_products > 0 && (!row_out || !col_out || !val_out)) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return spgemm_impl<double>(A, B, n_products, row_out, col_out, (double*)val_out, nnz_out_host, (hipStream_t)stream);
  return spgemm_impl<float>(A, B, n_products, row_out, col_out, (float*)val_out, nnz_out_host, (hipStream_t)stream);
}

int gla_coo_coalesce(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col,
                     const void* val, int64_t val_stride, int dtype, int64_t* row_out, int64_t* col_out, void* val_out,
                     int64_t* nnz_out_host, gla_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0 || !nnz_out_host || val_stride < 1 || (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (nnz > 0 && (!row || !col || !val || !row_out || !col_out || !val_out)) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return coalesce_impl<double>(n_rows, n_cols, nnz, row, col, (const double*)val, val_stride, row_out, col_out,
                                 (double*)val_out, nnz_out_host, (hipStream_t)stream);
  return coalesce_impl<float>(n_rows, n_cols, nnz, row, col, (const float*)val, val_stride, row_out, col_out,
                              (float*)val_out, nnz_out_host, (hipStream_t)stream);
}

}  // extern "C"
