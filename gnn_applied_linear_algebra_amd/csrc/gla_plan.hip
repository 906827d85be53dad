// libgla_hip.so -- plan construction (COO edge list -> CSR row tiles) and graph builders (gfx950 only).
//
// The reference hands every layer a COO edge list (edgeij_pair int64 [2,nnz]) and lets
// torch_scatter.scatter re-discover the row structure on every call.  Here that work is
// done once per graph: validate indices, stable-sort by row if needed (edge order is preserved inside a
// row so that row sums keep the reference's summation order), build rowptr, measure the longest row and
// cut the rows into tiles of ~tile_nnz nonzeros for the CSR-stream kernels.
#include <algorithm>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "gla_internal.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int FLAG_UNSORTED = 1, FLAG_RANGE = 2;

inline unsigned grid_for(int64_t work, int64_t per_block = GLA_BLOCK, int64_t cap = 256 * 16) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

__global__ __launch_bounds__(GLA_BLOCK) void validate_kernel(const int64_t* __restrict__ row,
                                                             const int64_t* __restrict__ col, int64_t nnz,
                                                             int64_t n_rows, int64_t n_cols, int* flags) {
  int f = 0;
  for (int64_t e = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t r = row[e], c = col[e];
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) f |= FLAG_RANGE;
    if (e > 0 && row[e - 1] > r) f |= FLAG_UNSORTED;
  }
  if (f) atomicOr(flags, f);
}

// rowptr from a row-sorted key array: every edge that starts a new row fills the gap of (possibly empty)
// rows in front of it.
template <typename KeyT>
__global__ __launch_bounds__(GLA_BLOCK) void rowptr_kernel(const KeyT* __restrict__ rows_sorted, int64_t nnz,
                                                           int64_t n_rows, int32_t* __restrict__ rowptr) {
  for (int64_t e = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; e <= nnz; e += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t prev = (e == 0) ? -1 : (int64_t)rows_sorted[e - 1];
    const int64_t cur = (e == nnz) ? n_rows : (int64_t)rows_sorted[e];
    for (int64_t q = prev + 1; q <= cur; ++q) rowptr[q] = (int32_t)e;
  }
}

__global__ __launch_bounds__(GLA_BLOCK) void keys_kernel(const int64_t* __restrict__ row, int64_t nnz,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
  for (int64_t e = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * GLA_BLOCK) {
    keys[e] = (uint32_t)row[e];
    ids[e] = (int32_t)e;
  }
}

__global__ __launch_bounds__(GLA_BLOCK) void gather_col_kernel(const int64_t* __restrict__ col,
                                                               const int32_t* __restrict__ perm, int64_t nnz,
                                                               int32_t* __restrict__ col32) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnz + GLA_PAD;
       p += (int64_t)gridDim.x * GLA_BLOCK)
    col32[p] = (p < nnz) ? (int32_t)col[perm ? perm[p] : p] : 0;
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void gather_val_kernel(const T* __restrict__ edge_val, int64_t stride,
                                                               const int32_t* __restrict__ perm, int64_t nnz,
                                                               T* __restrict__ val) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnz + GLA_PAD;
       p += (int64_t)gridDim.x * GLA_BLOCK)
    val[p] = (p < nnz) ? edge_val[(perm ? (int64_t)perm[p] : p) * stride] : T(0);
}

__global__ __launch_bounds__(GLA_BLOCK) void max_row_len_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                                int* out) {
  int m = 0;
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK)
    m = max(m, rowptr[r + 1] - rowptr[r]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

// rows longer than `limit`: ids appended to list (any order, sorted on the host), out[0] = their number,
// out[1] = the longest of the OTHER rows (what the LDS tile of the stream bin has to hold)
__global__ __launch_bounds__(GLA_BLOCK) void long_rows_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                              int limit, int* out, int32_t* __restrict__ list, int cap) {
  int m = 0;
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    const int len = rowptr[r + 1] - rowptr[r];
    if (len > limit) {
      const int i = atomicAdd(out, 1);
      if (i < cap) list[i] = (int32_t)r;
    } else {
      m = max(m, len);
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out + 1, m);
}

// tile_row[k] = first row r with rowptr[r] >= k * tile_nnz (k < n_tiles); tile_row[n_tiles] = n_rows
__global__ __launch_bounds__(GLA_BLOCK) void tile_rows_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                              int n_tiles, int tile_nnz, int32_t* __restrict__ tile_row) {
  const int k = blockIdx.x * GLA_BLOCK + threadIdx.x;
  if (k > n_tiles) return;
  if (k == n_tiles) { tile_row[k] = (int32_t)n_rows; return; }
  if (k == 0) { tile_row[0] = 0; return; }
  const int64_t target = (int64_t)k * tile_nnz;
  int64_t lo = 0, hi = n_rows;                    // rowptr is non-decreasing on [0, n_rows]
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  tile_row[k] = (int32_t)lo;
}

__global__ __launch_bounds__(GLA_BLOCK) void tile_info_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ tile_row, int n_tiles,
                                                              int4* __restrict__ info) {
  const int t = blockIdx.x * GLA_BLOCK + threadIdx.x;
  if (t >= n_tiles) return;
  const int r0 = tile_row[t], r1 = tile_row[t + 1];
  info[t] = make_int4(r0, r1, rowptr[r0], rowptr[r1]);
}

// a tile is "boundary" when any of its columns is >= col_limit (a halo column of a row-block partition)
__global__ __launch_bounds__(GLA_BLOCK) void tile_flag_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col,
                                                              const int32_t* __restrict__ tile_row, int n_tiles,
                                                              int32_t col_limit, int32_t* __restrict__ flag) {
  const int t = blockIdx.x;
  if (t >= n_tiles) return;
  const int s = rowptr[tile_row[t]], e = rowptr[tile_row[t + 1]];
  int f = 0;
  for (int i = s + threadIdx.x; i < e; i += GLA_BLOCK) f |= (col[i] >= col_limit);
  if (__syncthreads_or(f) && threadIdx.x == 0) flag[t] = 1;
}

// ---- builders --------------------------------------------------------------------------------
// nnz in rows [0, k) of the Ny x Nx 5-point grid (vertex k = j*Nx + i)
__host__ __device__ inline int64_t lap_offset(int64_t k, int64_t Nx, int64_t Ny) {
  const int64_t j = k / Nx, i = k - j * Nx;
  int64_t off = 5 * k;
  off -= j + (i > 0 ? 1 : 0);             // rows < k with i == 0       (no left neighbour)
  off -= j;                               // rows < k with i == Nx - 1  (no right neighbour)
  off -= (j == 0 ? i : Nx);               // rows < k with j == 0       (no lower neighbour)
  off -= (j == Ny - 1 ? i : 0);           // rows < k with j == Ny - 1  (no upper neighbour)
  if (j >= Ny) off = 5 * Nx * Ny - 2 * Nx - 2 * Ny;   // k == Nx*Ny (one past the end)
  return off;
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void laplacian2d_kernel(int64_t Nx, int64_t Ny, int64_t row_lo,
                                                                int64_t row_hi, int64_t base,
                                                                int64_t* __restrict__ row, int64_t* __restrict__ col,
                                                                T* __restrict__ val) {
  for (int64_t k = row_lo + (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; k < row_hi;
       k += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t j = k / Nx, i = k - j * Nx;
    int64_t o = lap_offset(k, Nx, Ny) - base;
    if (j > 0) { row[o] = k; col[o] = k - Nx; val[o] = T(1); ++o; }
    if (i > 0) { row[o] = k; col[o] = k - 1; val[o] = T(1); ++o; }
    row[o] = k; col[o] = k; val[o] = T(-4); ++o;
    if (i < Nx - 1) { row[o] = k; col[o] = k + 1; val[o] = T(1); ++o; }
    if (j < Ny - 1) { row[o] = k; col[o] = k + Nx; val[o] = T(1); ++o; }
  }
}

// ---- order-preserving compaction (remove self loops) ---------------------------------------------
constexpr int CMP_ITEMS = 4;
constexpr int CMP_TILE = GLA_BLOCK * CMP_ITEMS;

__global__ __launch_bounds__(GLA_BLOCK) void compact_count_kernel(const int64_t* __restrict__ row,
                                                                   const int64_t* __restrict__ col, int64_t nnz,
                                                                   int64_t* __restrict__ block_count) {
  __shared__ int wsum[GLA_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * CMP_TILE;
  int c = 0;
  // blocked layout: thread t owns elements [base + t*ITEMS, base + (t+1)*ITEMS) so ranks are ordered
#pragma unroll
  for (int u = 0; u < CMP_ITEMS; ++u) {
    const int64_t e = base + (int64_t)threadIdx.x * CMP_ITEMS + u;
    if (e < nnz && row[e] != col[e]) ++c;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < GLA_BLOCK / 64; ++w) tot += wsum[w];
    block_count[blockIdx.x] = tot;
  }
}

// exclusive scan of block counts in place (single workgroup, sequential over chunks)
__global__ __launch_bounds__(1024) void compact_scan_kernel(int64_t* __restrict__ block_count, int64_t nblocks,
                                                            int64_t* __restrict__ total) {
  __shared__ int64_t buf[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblocks; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = (i < nblocks) ? block_count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int64_t add = (threadIdx.x >= (unsigned)off) ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) block_count[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void compact_apply_kernel(const int64_t* __restrict__ row,
                                                                  const int64_t* __restrict__ col,
                                                                  const T* __restrict__ val, int64_t val_stride,
                                                                  int64_t nnz, const int64_t* __restrict__ block_off,
                                                                  int64_t* __restrict__ row_out,
                                                                  int64_t* __restrict__ col_out, T* __restrict__ val_out) {
  __shared__ int tsum[GLA_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * CMP_TILE + (int64_t)threadIdx.x * CMP_ITEMS;
  int keep[CMP_ITEMS];
  int c = 0;
#pragma unroll
  for (int u = 0; u < CMP_ITEMS; ++u) {
    const int64_t e = base + u;
    keep[u] = (e < nnz && row[e] != col[e]) ? 1 : 0;
    c += keep[u];
  }
  tsum[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < GLA_BLOCK; off <<= 1) {       // inclusive Hillis-Steele scan over the block
    int add = (threadIdx.x >= (unsigned)off) ? tsum[threadIdx.x - off] : 0;
    __syncthreads();
    tsum[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t o = block_off[blockIdx.x] + tsum[threadIdx.x] - c;
#pragma unroll
  for (int u = 0; u < CMP_ITEMS; ++u) {
    if (keep[u]) {
      const int64_t e = base + u;
      row_out[o] = row[e];
      col_out[o] = col[e];
      if (val) val_out[o] = val[e * val_stride];
      ++o;
    }
  }
}


// ---- packed stream (option "compress") ---------------------------------------------------------------------
// col16[p] = col[p] - (first row of p's tile); flag set when a delta does not fit int16
__global__ __launch_bounds__(GLA_BLOCK) void col16_kernel(const int32_t* __restrict__ col, const int4* __restrict__ tile_info,
                                                          int n_tiles, int16_t* __restrict__ col16,
                                                          uint32_t* __restrict__ wide, int* fail) {
  const int t = blockIdx.x;
  if (t >= n_tiles) return;
  const int4 ti = tile_info[t];
  int bad = 0;
  for (int p = ti.z + threadIdx.x; p < ti.w; p += GLA_BLOCK) {
    const int d = col[p] - ti.x;
    if (d < -32768 || d > 32767) bad = 1;
    col16[p] = (int16_t)d;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) {
    atomicOr(&wide[t >> 5], 1u << (t & 31));
    atomicAdd(fail, 1);
  }
}

template <typename T, typename U>
__global__ __launch_bounds__(GLA_BLOCK) void val_bits_kernel(const T* __restrict__ val, int64_t nnz, U* __restrict__ keys) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * GLA_BLOCK) {
    U u;
    memcpy(&u, &val[p], sizeof(U));
    keys[p] = u;
  }
}

template <typename U>
__global__ __launch_bounds__(GLA_BLOCK) void uniq_flags_kernel(const U* __restrict__ keys, int64_t n, int64_t* __restrict__ flags) {
  for (int64_t t = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * GLA_BLOCK)
    flags[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}

template <typename U>
__global__ __launch_bounds__(GLA_BLOCK) void uniq_gather_kernel(const U* __restrict__ keys, int64_t n,
                                                                const int64_t* __restrict__ pos, U* __restrict__ dict) {
  for (int64_t t = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; t < n; t += (int64_t)gridDim.x * GLA_BLOCK)
    if ((t == 0 || keys[t] != keys[t - 1]) && pos[t] < 256) dict[pos[t]] = keys[t];
}

template <typename T, typename U>
__global__ __launch_bounds__(GLA_BLOCK) void val_code_kernel(const T* __restrict__ val, int64_t nnz, const U* __restrict__ dict,
                                                             int n_dict, uint8_t* __restrict__ code) {
  for (int64_t p = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; p < nnz + 16; p += (int64_t)gridDim.x * GLA_BLOCK) {
    if (p >= nnz) { code[p] = 0; continue; }
    U u;
    memcpy(&u, &val[p], sizeof(U));
    int lo = 0, hi = n_dict - 1;                       // dict is sorted by bit pattern; the value is present
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (dict[mid] < u) lo = mid + 1; else hi = mid;
    }
    code[p] = (uint8_t)lo;
  }
}

// ---- diagonal (stencil) layout (option "dia") ------------------------------------------------------------------
struct DiaOffsets { int32_t off[GLA_DIA_MAX]; int32_t D; };

// first row whose length equals max_len at or after `from` (the candidate that should hold every diagonal; searched
// from the middle so that the halo rows at either end of a row block are not picked) and first such row anywhere
__global__ __launch_bounds__(GLA_BLOCK) void dia_first_full_row_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                                                       int max_len, int64_t from,
                                                                       unsigned long long* first) {
  // per-thread minima, then ONE atomic pair per workgroup (an atomic per matching row serialises on two addresses:
  // 190 ms at 16.8 M rows)
  unsigned long long lo[2] = {~0ull, ~0ull};
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK)
    if (rowptr[r + 1] - rowptr[r] == max_len) {
      const int w = r >= from ? 0 : 1;
      if ((unsigned long long)r < lo[w]) lo[w] = (unsigned long long)r;
    }
  __shared__ unsigned long long smin[2][GLA_BLOCK];
  smin[0][threadIdx.x] = lo[0];
  smin[1][threadIdx.x] = lo[1];
  __syncthreads();
  for (int st = GLA_BLOCK / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      smin[0][threadIdx.x] = min(smin[0][threadIdx.x], smin[0][threadIdx.x + st]);
      smin[1][threadIdx.x] = min(smin[1][threadIdx.x], smin[1][threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (smin[0][0] != ~0ull) atomicMin(first, smin[0][0]);
    if (smin[1][0] != ~0ull) atomicMin(first + 1, smin[1][0]);
  }
}

__device__ __forceinline__ int dia_slot(const DiaOffsets& o, int off) {
  int d = -1;
#pragma unroll
  for (int i = 0; i < GLA_DIA_MAX; ++i) d = (i < o.D && o.off[i] == off) ? i : d;
  return d;
}

// every row: its offsets col - row must be diagonals of the candidate set, in strictly ascending order.  Rows that
// do not fit (the halo rows of a row block) get mask 0xFFFF and shrink the valid range [range[0], range[1]).
__global__ __launch_bounds__(GLA_BLOCK) void dia_verify_kernel(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col, int64_t n_rows,
                                                               DiaOffsets o, uint16_t* __restrict__ mask,
                                                               unsigned long long* range) {
  unsigned long long lo = 0, hi = ~0ull;          // this thread's contribution to [range[0], range[1])
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    unsigned m = 0;
    int last = -1, bad = 0;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const int d = dia_slot(o, (int)((int64_t)col[p] - r));
      if (d <= last) bad = 1;              // not a diagonal of the set (-1) or out of order / duplicate
      last = d;
      m |= d >= 0 ? 1u << d : 0u;
    }
    mask[r] = bad ? (uint16_t)0xFFFF : (uint16_t)m;
    if (bad) {
      if (r < n_rows / 2) lo = max(lo, (unsigned long long)(r + 1));
      else hi = min(hi, (unsigned long long)r);
    }
  }
  // one atomic pair per workgroup (a non-stencil matrix fails on almost every row)
  __shared__ unsigned long long slo[GLA_BLOCK], shi[GLA_BLOCK];
  slo[threadIdx.x] = lo;
  shi[threadIdx.x] = hi;
  __syncthreads();
  for (int st = GLA_BLOCK / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      slo[threadIdx.x] = max(slo[threadIdx.x], slo[threadIdx.x + st]);
      shi[threadIdx.x] = min(shi[threadIdx.x], shi[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (slo[0] != 0) atomicMax(range, slo[0]);
    if (shi[0] != ~0ull) atomicMin(range + 1, shi[0]);
  }
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void dia_scatter_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                const T* __restrict__ val, int64_t n_rows, DiaOffsets o,
                                                                T* __restrict__ dia_val) {
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK)
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const int d = dia_slot(o, (int)((int64_t)col[p] - r));
      if (d >= 0) dia_val[(int64_t)d * n_rows + r] = val[p];        // rows outside the valid range are never read
    }
}

DiaOffsets dia_offsets_of(const gla_csr* p) {
  DiaOffsets o;
  for (int i = 0; i < GLA_DIA_MAX; ++i) o.off[i] = i < p->dia_D ? p->dia_off[i] : 0;
  o.D = p->dia_D;
  return o;
}

// does any present entry (row r, diagonal d) leave its grid line under the reading off_d = a_d * S + b_d, i.e. is
// (r % S) + b_d outside [0, S)?  (True 2-D grids never do: a vertex in the first / last column has no left / right
// neighbour.  A reading with the wrong S -- the offsets -33, -32, -1, 0, 1, 32, 33 read with S = 32 as well as with the
// grid's 33 -- does.)
struct DiaB { int8_t b[GLA_DIA_MAX]; int32_t D; };
__global__ __launch_bounds__(GLA_BLOCK) void dia_wrap_kernel(const uint16_t* __restrict__ mask, int64_t n_rows, int32_t S,
                                                             DiaB db, int* __restrict__ wraps) {
  int bad = 0;
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    const unsigned m = mask[r];
    if (m == 0xFFFFu) continue;                      // row outside the stencil range (halo rows of a row block)
    const int gi = (int)(r % S);
#pragma unroll
    for (int d = 0; d < GLA_DIA_MAX; ++d)
      if (d < db.D && ((m >> d) & 1u) && (gi + db.b[d] < 0 || gi + db.b[d] >= S)) bad = 1;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(wraps, 1);
}

// Canonical 5-point reading, grid lines of S points: does EVERY row hold exactly the entries its grid position implies (north
// unless on the first line, west unless first in its line, the centre, east unless last in its line, south unless the row S
// further does not exist)?  The mask jacobi_march5_kernel expects of a row that takes its unmasked path; structure only.
__global__ __launch_bounds__(GLA_BLOCK) void dia_full5_kernel(const uint16_t* __restrict__ mask, int64_t n_rows, int64_t S,
                                                              int* __restrict__ broken) {
  int bad = 0;
  const int64_t nfull = n_rows / S, tail = n_rows - nfull * S;
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t gj = r / S, gi = r - gj * S;
    const bool has_down = gj + 1 < nfull || (gj + 1 == nfull && gi < tail);
    const unsigned em = 31u & ~(gi == 0 ? 2u : 0u) & ~(gi == S - 1 ? 8u : 0u) & ~(gj == 0 ? 1u : 0u) & ~(has_down ? 0u : 16u);
    if ((unsigned)mask[r] != em) bad = 1;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(broken, 1);
}

// Structure side, at plan creation: does every nonzero lie on one of <= 16 diagonals, rows in ascending order?
int build_dia(gla_csr* p, hipStream_t st) {
  p->dia_state = -1;
  if (!p->sorted_input || p->long_rows || p->n_rows == 0 || p->max_row_len < 1 || p->max_row_len >= GLA_DIA_MAX)
    return GLA_OK;
  hipError_t e;
#define DK(expr) do { e = (expr); if (e != hipSuccess) return -(1000 + (int)e); } while (0)
  unsigned long long* d_u = nullptr;          // [0..1] candidate rows, [2..3] valid row range
  DK(hipMalloc(&d_u, 4 * sizeof(unsigned long long)));
  unsigned long long h_u[4] = {~0ull, ~0ull, 0ull, (unsigned long long)p->n_rows};
  DK(hipMemcpyAsync(d_u, h_u, sizeof(h_u), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(dia_first_full_row_kernel, dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, p->rowptr, p->n_rows,
                     p->max_row_len, p->n_rows / 2, d_u);
  DK(hipMemcpyAsync(h_u, d_u, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  DK(hipStreamSynchronize(st));
  const unsigned long long h_first = h_u[0] != ~0ull ? h_u[0] : h_u[1];
  if (h_first >= (unsigned long long)p->n_rows) { (void)hipFree(d_u); return GLA_OK; }
  int32_t h_ptr[2], h_cols[GLA_DIA_MAX];
  DK(hipMemcpy(h_ptr, p->rowptr + h_first, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
  const int D = h_ptr[1] - h_ptr[0];
  DK(hipMemcpy(h_cols, p->col + h_ptr[0], D * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < D; ++i) {
    p->dia_off[i] = (int32_t)((int64_t)h_cols[i] - (int64_t)h_first);
    if (i > 0 && p->dia_off[i] <= p->dia_off[i - 1]) { (void)hipFree(d_u); return GLA_OK; }   // columns not ascending
  }
  p->dia_D = D;
  DK(hipMalloc(&p->dia_mask, (size_t)p->n_rows * sizeof(uint16_t)));
  hipLaunchKernelGGL(dia_verify_kernel, dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, p->rowptr, p->col, p->n_rows,
                     dia_offsets_of(p), p->dia_mask, d_u + 2);
  DK(hipMemcpyAsync(h_u + 2, d_u + 2, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  DK(hipStreamSynchronize(st));
  (void)hipFree(d_u);
  p->dia_row_lo = (int64_t)h_u[2];
  p->dia_row_hi = (int64_t)h_u[3];
  if (p->dia_row_hi <= p->dia_row_lo || 2 * (p->dia_row_hi - p->dia_row_lo) < p->n_rows) {   // mostly not a stencil
    (void)hipFree(p->dia_mask);
    p->dia_mask = nullptr;
    return GLA_OK;
  }
  p->dia_state = 1;
  // 2-D reading of the offsets: off = a * S + b with a line stride S among the far offsets, |a| <= 2, |b| <= 3.  Preferred:
  // a reading in which no entry leaves its grid line (the temporally blocked Jacobi kernel needs that; the blocked
  // multi-rhs kernel works with any reading), then the smallest max |b|.
  p->dia_S = 0;
  p->dia_nowrap = 0;
  {
    int* d_wrap = nullptr;
    DK(hipMalloc(&d_wrap, sizeof(int)));
    int best_b = 1 << 30, best_nowrap = -1;
    for (int c = 0; c < D; ++c) {
      const int64_t S = p->dia_off[c];
      if (S <= 8) continue;
      int maxa = 0, maxb = 0, ok = 1;
      int8_t ta[16], tb[16];
      for (int i = 0; i < D && ok; ++i) {
        const int64_t off = p->dia_off[i];
        int64_t a = (off >= 0 ? off + S / 2 : off - S / 2) / S;          // nearest multiple
        const int64_t b = off - a * S;
        if (a < -2 || a > 2 || b < -3 || b > 3) ok = 0;
        ta[i] = (int8_t)a;
        tb[i] = (int8_t)b;
        maxa = std::max(maxa, (int)(a < 0 ? -a : a));
        maxb = std::max(maxb, (int)(b < 0 ? -b : b));
      }
      if (!ok) continue;
      DiaB db;
      db.D = D;
      for (int i = 0; i < GLA_DIA_MAX; ++i) db.b[i] = i < D ? tb[i] : 0;
      int h_wrap = 0;
      DK(hipMemsetAsync(d_wrap, 0, sizeof(int), st));
      hipLaunchKernelGGL(dia_wrap_kernel, dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, p->dia_mask, p->n_rows, (int32_t)S,
                         db, d_wrap);
      DK(hipMemcpyAsync(&h_wrap, d_wrap, sizeof(int), hipMemcpyDeviceToHost, st));
      DK(hipStreamSynchronize(st));
      const int nowrap = h_wrap ? 0 : 1;
      if (nowrap > best_nowrap || (nowrap == best_nowrap && maxb < best_b)) {
        best_nowrap = nowrap;
        best_b = maxb;
        p->dia_S = (int32_t)S;
        p->dia_A = maxa;
        p->dia_B = maxb;
        p->dia_nowrap = nowrap;
        for (int i = 0; i < D; ++i) { p->dia_a[i] = ta[i]; p->dia_b[i] = tb[i]; }
      }
    }
    (void)hipFree(d_wrap);
  }
  p->dia_full = 0;
  if (gla_dia_canon5(p) && p->dia_nowrap && p->dia_row_lo == 0 && p->dia_row_hi == p->n_rows && p->n_rows == p->n_cols) {
    int* d_broken = nullptr;
    int h_broken = 1;
    DK(hipMalloc(&d_broken, sizeof(int)));
    DK(hipMemsetAsync(d_broken, 0, sizeof(int), st));
    hipLaunchKernelGGL(dia_full5_kernel, dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, p->dia_mask, p->n_rows,
                       (int64_t)p->dia_S, d_broken);
    DK(hipMemcpyAsync(&h_broken, d_broken, sizeof(int), hipMemcpyDeviceToHost, st));
    DK(hipStreamSynchronize(st));
    (void)hipFree(d_broken);
    p->dia_full = h_broken ? 0 : 1;
  }
  return GLA_OK;
#undef DK
}

// Option "compress" on a stencil plan: is every diagonal constant (bit pattern) over the rows that have it?
template <typename T, typename U>
__global__ __launch_bounds__(GLA_BLOCK) void dia_minmax_kernel(const T* __restrict__ dia_val,
                                                               const uint16_t* __restrict__ mask, int64_t n_rows, int D,
                                                               unsigned long long* __restrict__ mm) {   // [D][2]
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    const unsigned m = mask[r];
    for (int d = 0; d < D; ++d) {
      if (!((m >> d) & 1u)) continue;
      const unsigned long long b = (unsigned long long)__builtin_bit_cast(U, dia_val[(int64_t)d * n_rows + r]);
      if (b < mm[2 * d]) atomicMin(mm + 2 * d, b);
      if (b > mm[2 * d + 1]) atomicMax(mm + 2 * d + 1, b);
    }
  }
}

template <typename T, typename U>
int detect_dia_const(gla_csr* p, hipStream_t st) {
  p->dia_const = 0;
  if (p->dia_state != 1 || !p->dia_val || p->dia_row_lo != 0 || p->dia_row_hi != p->n_rows) return GLA_OK;
  unsigned long long h_mm[2 * GLA_DIA_MAX];
  for (int d = 0; d < GLA_DIA_MAX; ++d) { h_mm[2 * d] = ~0ull; h_mm[2 * d + 1] = 0ull; }
  unsigned long long* d_mm = nullptr;
  hipError_t e = hipMalloc(&d_mm, sizeof(h_mm));
  if (e != hipSuccess) return -(1000 + (int)e);
  e = hipMemcpyAsync(d_mm, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL((dia_minmax_kernel<T, U>), dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st,
                       (const T*)p->dia_val, p->dia_mask, p->n_rows, p->dia_D, d_mm);
    e = hipMemcpyAsync(h_mm, d_mm, sizeof(h_mm), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_mm);
  if (e != hipSuccess) return -(1000 + (int)e);
  for (int d = 0; d < p->dia_D; ++d) {
    if (h_mm[2 * d] != h_mm[2 * d + 1]) return GLA_OK;            // varies along the diagonal (or diagonal empty)
    p->dia_cval[d] = (double)__builtin_bit_cast(T, (U)h_mm[2 * d]);
  }
  p->dia_const = 1;
  return GLA_OK;
}

// Symmetric value sets (option "sym"): diagonal d of offset off < 0 with a partner q of offset -off is REDUNDANT when
// val[d][r] == val[q][r + off] bit for bit for every row r with r + off >= 0, and == +0 above (A_ij == A_ji along the pair,
// absent entries being stored zeros on both sides).  The K = 1 march kernels then read the partner instead: the negative
// Laplacian of UtilsGNN.py:53-67, the Q1 stiffness matrix of heateqnfem2dfun.m and every Galerkin operator P^T A P of theirs
// qualify; a convection term, a one-sided hole or a -0.0 against a +0.0 does not, and such a plan keeps all its diagonals.
template <typename T, typename U>
__global__ __launch_bounds__(GLA_BLOCK) void dia_sym_kernel(const T* __restrict__ dia_val, int64_t n_rows, int D, DiaOffsets off,
                                                            int* __restrict__ broken) {      // bit d: pair (d, partner) differs
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * GLA_BLOCK) {
    int bad = 0;
    for (int d = 0; d < D; ++d) {
      const int64_t o = off.off[d];
      if (o >= 0) break;                                      // (offsets ascend)
      int q = -1;
      for (int e = d + 1; e < D; ++e) if ((int64_t)off.off[e] == -o) q = e;
      if (q < 0) { bad |= 1 << d; continue; }
      const U mine = __builtin_bit_cast(U, dia_val[(int64_t)d * n_rows + r]);
      const U want = r + o >= 0 ? __builtin_bit_cast(U, dia_val[(int64_t)q * n_rows + r + o]) : (U)0;
      if (mine != want) bad |= 1 << d;
    }
    if (bad) atomicOr(broken, bad);
  }
}

template <typename T, typename U>
int detect_dia_sym(gla_csr* p, hipStream_t st) {
  if (p->sym_pending) (void)gla_dia_sym_bits(p);     // (a check still in flight writes the words this one reuses)
  p->dia_sym = 0;
  if (p->dia_state != 1 || !p->dia_val || p->dia_row_lo != 0 || p->dia_row_hi != p->n_rows || p->n_rows != p->n_cols ||
      p->dia_S <= 0 || !p->dia_nowrap)
    return GLA_OK;
  int n_neg = 0;
  for (int d = 0; d < p->dia_D; ++d) n_neg += p->dia_off[d] < 0;
  if (n_neg == 0) return GLA_OK;
  hipError_t e = hipSuccess;
  if (!p->sym_dev) {                                 // once per plan
    e = hipMalloc(&p->sym_dev, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(&p->sym_host, sizeof(int));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->sym_event, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
  }
  e = hipMemsetAsync(p->sym_dev, 0, sizeof(int), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL((dia_sym_kernel<T, U>), dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, (const T*)p->dia_val, p->n_rows,
                       p->dia_D, dia_offsets_of(p), p->sym_dev);
    e = hipMemcpyAsync(p->sym_host, p->sym_dev, sizeof(int), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipEventRecord(p->sym_event, st);
  if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
  p->sym_mask = (1 << n_neg) - 1;
  p->sym_pending = 1;                                // (resolved by the first reader: gla_dia_sym_bits)
  return GLA_OK;
}

// Value side, at gla_csr_set_values: scatter the CSR values onto their diagonals (absent entries stay 0)
template <typename T>
int fill_dia(gla_csr* p, hipStream_t st) {
  if (p->dia_state != 1) return GLA_OK;
  if (!p->dia_val) {
    const size_t bytes = (size_t)p->dia_D * (size_t)p->n_rows * sizeof(double);
    hipError_t e = hipMalloc(&p->dia_val, bytes);
    if (e != hipSuccess) return -(1000 + (int)e);
    e = hipMemsetAsync(p->dia_val, 0, bytes, st);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  hipLaunchKernelGGL((dia_scatter_kernel<T>), dim3(grid_for(p->n_rows)), dim3(GLA_BLOCK), 0, st, p->rowptr, p->col,
                     (const T*)p->val, p->n_rows, dia_offsets_of(p), (T*)p->dia_val);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

// rowoff16[r + t] = rowptr[r] - (first nonzero of tile t) for the rows r0 .. r1 (inclusive) of tile t
__global__ __launch_bounds__(GLA_BLOCK) void rowoff16_kernel(const int32_t* __restrict__ rowptr,
                                                             const int4* __restrict__ tile_info, int n_tiles,
                                                             uint16_t* __restrict__ rowoff16) {
  const int t = blockIdx.x;
  if (t >= n_tiles) return;
  const int4 ti = tile_info[t];
  for (int i = threadIdx.x; i <= ti.y - ti.x; i += GLA_BLOCK) rowoff16[ti.x + t + i] = (uint16_t)(rowptr[ti.x + i] - ti.z);
}

int build_col16(gla_csr* p, hipStream_t st) {
  if (p->col16_state != 0 || p->nnz == 0) return GLA_OK;
  hipError_t e;
#define PK(expr) do { e = (expr); if (e != hipSuccess) return -(1000 + (int)e); } while (0)
  int* d_fail = nullptr;
  {
    PK(hipMalloc(&p->col16, (p->nnz + 16) * sizeof(int16_t)));
    PK(hipMemsetAsync(p->col16, 0, (p->nnz + 16) * sizeof(int16_t), st));
    const size_t mask_bytes = ((size_t)p->n_tiles / 32 + 1) * sizeof(uint32_t);
    PK(hipMalloc(&p->col16_wide, mask_bytes));
    PK(hipMemsetAsync(p->col16_wide, 0, mask_bytes, st));
    PK(hipMalloc(&d_fail, sizeof(int)));
    PK(hipMemsetAsync(d_fail, 0, sizeof(int), st));
    hipLaunchKernelGGL(col16_kernel, dim3(p->n_tiles), dim3(GLA_BLOCK), 0, st, p->col,
                       reinterpret_cast<const int4*>(p->tile_info), p->n_tiles, p->col16, p->col16_wide, d_fail);
    PK(hipMalloc(&p->rowoff16, ((size_t)p->n_rows + p->n_tiles + 2) * sizeof(uint16_t)));
    hipLaunchKernelGGL(rowoff16_kernel, dim3(p->n_tiles), dim3(GLA_BLOCK), 0, st, p->rowptr,
                       reinterpret_cast<const int4*>(p->tile_info), p->n_tiles, p->rowoff16);
    int h_fail = 0;
    PK(hipMemcpyAsync(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    PK(hipStreamSynchronize(st));
    (void)hipFree(d_fail);
    p->col16_wide_tiles = h_fail;
    p->col16_state = (2 * (int64_t)h_fail > p->n_tiles) ? -1 : 1;   // mostly wide: not worth the second stream
    if (p->col16_state < 0) {
      (void)hipFree(p->col16);
      (void)hipFree(p->col16_wide);
      (void)hipFree(p->rowoff16);
      p->col16 = nullptr;
      p->col16_wide = nullptr;
      p->rowoff16 = nullptr;
    }
  }
  return GLA_OK;
#undef PK
}

template <typename T, typename U>
int build_packed(gla_csr* p, hipStream_t st) {
  p->packed = 0;
  if (p->nnz == 0 || p->long_rows) return GLA_OK;
  hipError_t e;
  {
    const int rc0 = build_col16(p, st);
    if (rc0) return rc0;
  }
  if (p->col16_state != 1 || p->col16_wide_tiles != 0) return GLA_OK;   // the packed kernel has no int32 fallback
  // value dictionary: sort the bit patterns, count the distinct ones
  U *keys = nullptr, *keys2 = nullptr;
  int64_t *flags = nullptr, *pos = nullptr;
  void *tmp = nullptr, *tmp2 = nullptr;
  int rc = GLA_OK;
  size_t tb = 0, tb2 = 0;
  int64_t last[2] = {0, 0};
  const int64_t n = p->nnz;
  unsigned gridn = (unsigned)((n + GLA_BLOCK - 1) / GLA_BLOCK > 4096 ? 4096 : (n + GLA_BLOCK - 1) / GLA_BLOCK);
#define PKG(expr) do { e = (expr); if (e != hipSuccess) { rc = -(1000 + (int)e); goto done; } } while (0)
  PKG(hipMalloc(&keys, n * sizeof(U)));
  PKG(hipMalloc(&keys2, n * sizeof(U)));
  hipLaunchKernelGGL((val_bits_kernel<T, U>), dim3(gridn), dim3(GLA_BLOCK), 0, st, (const T*)p->val, n, keys);
  PKG(rocprim::radix_sort_keys(nullptr, tb, keys, keys2, (size_t)n, 0u, (unsigned)(8 * sizeof(U)), st));
  PKG(hipMalloc(&tmp, tb ? tb : 16));
  PKG(rocprim::radix_sort_keys(tmp, tb, keys, keys2, (size_t)n, 0u, (unsigned)(8 * sizeof(U)), st));
  PKG(hipMalloc(&flags, n * sizeof(int64_t)));
  PKG(hipMalloc(&pos, n * sizeof(int64_t)));
  hipLaunchKernelGGL((uniq_flags_kernel<U>), dim3(gridn), dim3(GLA_BLOCK), 0, st, keys2, n, flags);
  PKG(rocprim::exclusive_scan(nullptr, tb2, flags, pos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
  PKG(hipMalloc(&tmp2, tb2 ? tb2 : 16));
  PKG(rocprim::exclusive_scan(tmp2, tb2, flags, pos, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), st));
  PKG(hipMemcpyAsync(&last[0], pos + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PKG(hipMemcpyAsync(&last[1], flags + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
  PKG(hipStreamSynchronize(st));
  if (last[0] + last[1] <= 256) {
    const int n_dict = (int)(last[0] + last[1]);
    if (!p->vdict) PKG(hipMalloc(&p->vdict, 256 * sizeof(double)));
    if (!p->vcode) PKG(hipMalloc(&p->vcode, (n + 16) * sizeof(uint8_t)));
    PKG(hipMemsetAsync(p->vdict, 0, 256 * sizeof(double), st));
    hipLaunchKernelGGL((uniq_gather_kernel<U>), dim3(gridn), dim3(GLA_BLOCK), 0, st, keys2, n, pos, (U*)p->vdict);
    hipLaunchKernelGGL((val_code_kernel<T, U>), dim3(gridn), dim3(GLA_BLOCK), 0, st, (const T*)p->val, n,
                       (const U*)p->vdict, n_dict, p->vcode);
    PKG(hipGetLastError());
    PKG(hipStreamSynchronize(st));
    p->packed = 1;
  }
done:
  if (keys) (void)hipFree(keys);
  if (keys2) (void)hipFree(keys2);
  if (flags) (void)hipFree(flags);
  if (pos) (void)hipFree(pos);
  if (tmp) (void)hipFree(tmp);
  if (tmp2) (void)hipFree(tmp2);
  return rc;
#undef PKG
}

}  // namespace

int gla_sort_rows_stable(int64_t nnz, int64_t n_rows, const int64_t* row, int32_t* perm_out, uint32_t* keys_sorted,
                         hipStream_t st) {
  uint32_t* keys_in = nullptr;
  int32_t* ids_in = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = GLA_OK;
  hipError_t e;
  unsigned end_bit = 1;
  while (end_bit < 32 && ((int64_t)1 << end_bit) < n_rows) ++end_bit;
  if ((e = hipMalloc(&keys_in, nnz * sizeof(uint32_t))) != hipSuccess) { rc = -(1000 + (int)e); goto done; }
  if ((e = hipMalloc(&ids_in, nnz * sizeof(int32_t))) != hipSuccess) { rc = -(1000 + (int)e); goto done; }
  hipLaunchKernelGGL(keys_kernel, dim3(grid_for(nnz)), dim3(GLA_BLOCK), 0, st, row, nnz, keys_in, ids_in);
  e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_sorted, ids_in, perm_out, (size_t)nnz, 0u,
                                end_bit, st);
  if (e != hipSuccess) { rc = -(1000 + (int)e); goto done; }
  if ((e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16)) != hipSuccess) { rc = -(1000 + (int)e); goto done; }
  e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_sorted, ids_in, perm_out, (size_t)nnz, 0u, end_bit,
                                st);
  if (e != hipSuccess) { rc = -(1000 + (int)e); goto done; }
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) rc = -(1000 + (int)e);
done:
  if (keys_in) (void)hipFree(keys_in);
  if (ids_in) (void)hipFree(ids_in);
  if (tmp) (void)hipFree(tmp);
  return rc;
}

gla_tuning g_tuning;

extern "C" {

int gla_version(void) { return 140; }   // 100: round-1 core; 110: + edge_attr, segment_stats, edge_dot, mlp_relu3; 140: + gla_comm_probe, options sym / full_mask, dia_sym, dia_full

int gla_set_option(const char* name, int64_t value) {
  if (!name) return GLA_ERR_INVALID_ARG;
  if (!strcmp(name, "spmv_variant")) { g_tuning.spmv_variant = (int)value; return GLA_OK; }
  if (!strcmp(name, "xcd_chunk")) { if (value < 0) return GLA_ERR_INVALID_ARG; g_tuning.xcd_chunk = (int)value; return GLA_OK; }
  if (!strcmp(name, "col16")) { g_tuning.col16 = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "dia")) { g_tuning.dia = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "sym")) { g_tuning.sym = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "jacobi_diag_flag")) { g_tuning.jacobi_diag_flag = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "m5_variant")) { g_tuning.m5_variant = (int)value; return GLA_OK; }
  if (!strcmp(name, "edge_prio")) { g_tuning.edge_prio = value != 0; return GLA_OK; }
  if (!strcmp(name, "dist_stagger")) { if (value < 0 || value > 2) return GLA_ERR_INVALID_ARG; g_tuning.dist_stagger = (int)value; return GLA_OK; }   // (2: staggered, plain event records -- A/B)
  if (!strcmp(name, "full_mask")) { g_tuning.full_mask = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "row_sum")) { g_tuning.row_sum = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "jacobi_fuse")) { if (value < 0 || value > 2) return GLA_ERR_INVALID_ARG; g_tuning.jacobi_fuse = (int)value; return GLA_OK; }
  if (!strcmp(name, "jacobi_min_rows")) { if (value < 0) return GLA_ERR_INVALID_ARG; g_tuning.jacobi_min_rows = value; return GLA_OK; }
  if (!strcmp(name, "jacobi_ts_max")) { if (value < 2 || value > 8) return GLA_ERR_INVALID_ARG; g_tuning.jacobi_ts_max = (int)value; return GLA_OK; }
  if (!strcmp(name, "jacobi_ts_max9")) { if (value < 2 || value > 6) return GLA_ERR_INVALID_ARG; g_tuning.jacobi_ts_max9 = (int)value; return GLA_OK; }
  if (!strcmp(name, "jacobi_march")) { g_tuning.jacobi_march = (int)value; return GLA_OK; }
  if (!strcmp(name, "jacobi_lines")) { if (value < 0 || value == 1 || value > 4096) return GLA_ERR_INVALID_ARG; g_tuning.jacobi_lines = (int)value; return GLA_OK; }
  if (!strcmp(name, "compress")) { g_tuning.compress = value ? 1 : 0; return GLA_OK; }
  if (!strcmp(name, "spmm_variant")) { g_tuning.spmm_variant = (int)value; return GLA_OK; }
  if (!strcmp(name, "tile_nnz")) {
    if (value != 0 && (value < 256 || value > GLA_LDS_ELEMS_MAX - 64)) return GLA_ERR_INVALID_ARG;
    g_tuning.tile_nnz = (int)value; return GLA_OK;
  }
  return GLA_ERR_INVALID_ARG;
}

}  // extern "C"

namespace {
// GLA_OPTIONS="dia=0,col16=0,compress=1": initial option values from the environment (read once at load time), so a
// whole application or test run can be put on another layout without code changes
struct EnvOptions {
  EnvOptions() {
    const char* env = getenv("GLA_OPTIONS");
    if (!env) return;
    char buf[256];
    strncpy(buf, env, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char* tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
      char* eq = strchr(tok, '=');
      if (!eq) continue;
      *eq = 0;
      (void)gla_set_option(tok, atoll(eq + 1));        // unknown names are ignored
    }
  }
};
const EnvOptions g_env_options;
}  // namespace

extern "C" {

const char* gla_error_string(int code) {
  switch (code) {
    case GLA_OK: return "ok";
    case GLA_ERR_INVALID_ARG: return "invalid argument";
    case GLA_ERR_INDEX_RANGE: return "edge index out of range";
    case GLA_ERR_TOO_LARGE: return "problem exceeds int32 CSR index range";
    case GLA_ERR_NO_VALUES: return "plan holds no values of the requested dtype";
    case GLA_ERR_UNSUPPORTED: return "unsupported configuration";
    case GLA_ERR_NO_DEVICE: return "no HIP device";
    default:
      if (code <= -3000) return "RCCL error (ncclResult_t = -code - 3000)";
      if (code <= -1000) return hipGetErrorString((hipError_t)(-code - 1000));
      return "unknown error";
  }
}

int gla_csr_destroy(gla_csr_t p) {
  if (!p) return GLA_OK;
  if (p->rowptr) (void)hipFree(p->rowptr);
  if (p->col) (void)hipFree(p->col);
  if (p->perm) (void)hipFree(p->perm);
  if (p->val) (void)hipFree(p->val);
  if (p->tile_row) (void)hipFree(p->tile_row);
  if (p->tile_info) (void)hipFree(p->tile_info);
  if (p->tile_lists) (void)hipFree(p->tile_lists);
  if (p->long_row_ids) (void)hipFree(p->long_row_ids);
  if (p->col16) (void)hipFree(p->col16);
  if (p->col16_wide) (void)hipFree(p->col16_wide);
  if (p->rowoff16) (void)hipFree(p->rowoff16);
  if (p->dia_mask) (void)hipFree(p->dia_mask);
  if (p->dia_val) (void)hipFree(p->dia_val);
  if (p->dg_flags) (void)hipFree(p->dg_flags);
  if (p->sym_pending && p->sym_event) (void)hipEventSynchronize(p->sym_event);
  if (p->sym_dev) (void)hipFree(p->sym_dev);
  if (p->sym_host) (void)hipHostFree(p->sym_host);
  if (p->sym_event) (void)hipEventDestroy(p->sym_event);
  if (p->vcode) (void)hipFree(p->vcode);
  if (p->vdict) (void)hipFree(p->vdict);
  delete p;
  return GLA_OK;
}

int gla_csr_create_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col,
                            gla_stream_t stream, gla_csr_t* out) {
  if (!out || n_rows < 0 || n_cols < 0 || nnz < 0 || (nnz > 0 && (!row || !col))) return GLA_ERR_INVALID_ARG;
  if (nnz > 0 && (n_rows == 0 || n_cols == 0)) return GLA_ERR_INDEX_RANGE;
  if (nnz >= (int64_t)INT32_MAX - GLA_PAD || n_rows >= INT32_MAX || n_cols >= INT32_MAX) return GLA_ERR_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  *out = nullptr;
  gla_csr* p = new gla_csr();
  p->n_rows = n_rows; p->n_cols = n_cols; p->nnz = nnz;
  int* d_flags = nullptr;                 // [0] validation flags, [1] max row length
  int h_flags[2] = {0, 0};
  uint32_t* keys_sorted = nullptr;
  int rc = GLA_OK;
  hipError_t e;
#define PLAN_CHECK(expr) do { e = (expr); if (e != hipSuccess) { rc = -(1000 + (int)e); goto fail; } } while (0)
  PLAN_CHECK(hipMalloc(&d_flags, 2 * sizeof(int)));
  PLAN_CHECK(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), st));
  PLAN_CHECK(hipMalloc(&p->rowptr, (n_rows + 1) * sizeof(int32_t)));
  PLAN_CHECK(hipMalloc(&p->col, (nnz + GLA_PAD) * sizeof(int32_t)));
  PLAN_CHECK(hipMalloc(&p->val, (nnz + GLA_PAD) * sizeof(double)));
  if (nnz > 0) {
    hipLaunchKernelGGL(validate_kernel, dim3(grid_for(nnz)), dim3(GLA_BLOCK), 0, st, row, col, nnz, n_rows, n_cols,
                       d_flags);
    PLAN_CHECK(hipGetLastError());
  }
  PLAN_CHECK(hipMemcpyAsync(h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
  PLAN_CHECK(hipStreamSynchronize(st));
  if (h_flags[0] & FLAG_RANGE) { rc = GLA_ERR_INDEX_RANGE; goto fail; }
  p->sorted_input = (h_flags[0] & FLAG_UNSORTED) ? 0 : 1;
  if (!p->sorted_input) {
    PLAN_CHECK(hipMalloc(&p->perm, nnz * sizeof(int32_t)));
    PLAN_CHECK(hipMalloc(&keys_sorted, nnz * sizeof(uint32_t)));
    rc = gla_sort_rows_stable(nnz, n_rows, row, p->perm, keys_sorted, st);
    if (rc) goto fail;
    hipLaunchKernelGGL((rowptr_kernel<uint32_t>), dim3(grid_for(nnz + 1)), dim3(GLA_BLOCK), 0, st, keys_sorted, nnz,
                       n_rows, p->rowptr);
  } else {
    hipLaunchKernelGGL((rowptr_kernel<int64_t>), dim3(grid_for(nnz + 1)), dim3(GLA_BLOCK), 0, st, row, nnz, n_rows,
                       p->rowptr);
  }
  PLAN_CHECK(hipGetLastError());
  hipLaunchKernelGGL(gather_col_kernel, dim3(grid_for(nnz + GLA_PAD)), dim3(GLA_BLOCK), 0, st, col, p->perm, nnz,
                     p->col);
  PLAN_CHECK(hipGetLastError());
  if (n_rows > 0) {
    hipLaunchKernelGGL(max_row_len_kernel, dim3(grid_for(n_rows)), dim3(GLA_BLOCK), 0, st, p->rowptr, n_rows,
                       d_flags + 1);
    PLAN_CHECK(hipGetLastError());
  }
  PLAN_CHECK(hipMemcpyAsync(h_flags + 1, d_flags + 1, sizeof(int), hipMemcpyDeviceToHost, st));
  PLAN_CHECK(hipStreamSynchronize(st));
  p->max_row_len = h_flags[1];
  {
    // CSR-adaptive binning.  Stream bin: ~256 rows per tile for the average row, bounded by the LDS tile.  Long bin:
    // every row with more than GLA_LONG_ROW nonzeros becomes a tile of its own (tile boundaries are inserted before and
    // after it), so one hub row neither blows the LDS tile of its neighbours nor changes how THEIR sums are formed.
    int short_max = p->max_row_len;
    std::vector<int32_t> long_ids;
    if (p->max_row_len > GLA_LONG_ROW) {
      const int cap = (int)(nnz / (GLA_LONG_ROW + 1)) + 1;
      int32_t* d_list = nullptr;
      int h_out[2] = {0, 0};
      PLAN_CHECK(hipMalloc(&d_list, (size_t)cap * sizeof(int32_t)));
      PLAN_CHECK(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), st));
      hipLaunchKernelGGL(long_rows_kernel, dim3(grid_for(n_rows)), dim3(GLA_BLOCK), 0, st, p->rowptr, n_rows,
                         GLA_LONG_ROW, d_flags, d_list, cap);
      PLAN_CHECK(hipGetLastError());
      PLAN_CHECK(hipMemcpyAsync(h_out, d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
      PLAN_CHECK(hipStreamSynchronize(st));
      long_ids.resize((size_t)h_out[0]);
      PLAN_CHECK(hipMemcpy(long_ids.data(), d_list, long_ids.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      (void)hipFree(d_list);
      std::sort(long_ids.begin(), long_ids.end());
      short_max = h_out[1];
      p->long_rows = (int32_t)long_ids.size();
      PLAN_CHECK(hipMalloc(&p->long_row_ids, long_ids.size() * sizeof(int32_t)));
      PLAN_CHECK(hipMemcpy(p->long_row_ids, long_ids.data(), long_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const double avg = n_rows > 0 ? (double)nnz / (double)n_rows : 0.0;
    int64_t target = (int64_t)(GLA_BLOCK * (avg < 1.0 ? 1.0 : avg) + 0.5);
    if (target < GLA_TILE_NNZ_MIN) target = GLA_TILE_NNZ_MIN;
    // packed stream: one batch of 8 nonzeros per lane covers 2048 entries (incl. alignment slack and the last row)
    if (g_tuning.compress && p->max_row_len <= 1000) target = 2040 - p->max_row_len;
    if (g_tuning.tile_nnz > 0) target = g_tuning.tile_nnz;
    if (target > GLA_LDS_ELEMS_MAX - short_max) target = GLA_LDS_ELEMS_MAX - short_max;
    p->tile_nnz = (int32_t)target;
    p->lds_elems = (int32_t)(target + short_max);
    int64_t nt = (nnz + target - 1) / target;
    if (nt < 1) nt = 1;
    p->n_tiles = (int32_t)nt;
    PLAN_CHECK(hipMalloc(&p->tile_row, (nt + 1 + 2 * long_ids.size()) * sizeof(int32_t)));
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((nt + 1 + GLA_BLOCK - 1) / GLA_BLOCK)), dim3(GLA_BLOCK), 0,
                       st, p->rowptr, n_rows, p->n_tiles, p->tile_nnz, p->tile_row);
    PLAN_CHECK(hipGetLastError());
    if (!long_ids.empty()) {
      // merge the regular boundaries with {r, r + 1} of every long row; equal boundaries (the empty tiles a long row
      // leaves behind in the regular grid) collapse
      std::vector<int32_t> bnd((size_t)nt + 1);
      PLAN_CHECK(hipMemcpyAsync(bnd.data(), p->tile_row, bnd.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      PLAN_CHECK(hipStreamSynchronize(st));
      for (int32_t r : long_ids) { bnd.push_back(r); bnd.push_back(r + 1); }
      std::sort(bnd.begin(), bnd.end());
      bnd.erase(std::unique(bnd.begin(), bnd.end()), bnd.end());
      nt = (int64_t)bnd.size() - 1;
      if (nt < 1) { nt = 1; bnd.assign({0, (int32_t)n_rows}); }
      p->n_tiles = (int32_t)nt;
      PLAN_CHECK(hipMemcpy(p->tile_row, bnd.data(), bnd.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    PLAN_CHECK(hipMalloc(&p->tile_info, nt * 4 * sizeof(int32_t)));
    hipLaunchKernelGGL(tile_info_kernel, dim3((unsigned)((nt + GLA_BLOCK - 1) / GLA_BLOCK)), dim3(GLA_BLOCK), 0, st,
                       p->rowptr, p->tile_row, p->n_tiles, reinterpret_cast<int4*>(p->tile_info));
    PLAN_CHECK(hipGetLastError());
  }
  PLAN_CHECK(hipStreamSynchronize(st));
  if (g_tuning.col16 || g_tuning.compress) {
    rc = build_col16(p, st);              // int16 tile-relative columns for the K = 1 kernels (when the band fits)
    if (rc) goto fail;
  }
  if (g_tuning.dia) {
    rc = build_dia(p, st);                // stencil matrices: diagonal layout for the K = 1 kernels
    if (rc) goto fail;
  }
  (void)hipFree(d_flags);
  if (keys_sorted) (void)hipFree(keys_sorted);
  *out = p;
  return GLA_OK;
fail:
  if (d_flags) (void)hipFree(d_flags);
  if (keys_sorted) (void)hipFree(keys_sorted);
  gla_csr_destroy(p);
  return rc;
#undef PLAN_CHECK
}

int gla_csr_set_values(gla_csr_t p, const void* edge_val, int64_t stride, int dtype, gla_stream_t stream) {
  if (!p || (dtype != GLA_F32 && dtype != GLA_F64) || stride < 1 || (p->nnz > 0 && !edge_val))
    return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLA_F64)
    hipLaunchKernelGGL((gather_val_kernel<double>), dim3(grid_for(p->nnz + GLA_PAD)), dim3(GLA_BLOCK), 0, st,
                       (const double*)edge_val, stride, p->perm, p->nnz, (double*)p->val);
  else
    hipLaunchKernelGGL((gather_val_kernel<float>), dim3(grid_for(p->nnz + GLA_PAD)), dim3(GLA_BLOCK), 0, st,
                       (const float*)edge_val, stride, p->perm, p->nnz, (float*)p->val);
  GLA_LAUNCH_CHECK();
  p->val_dtype = dtype;
  p->packed = 0;
  {
    const int rc = dtype == GLA_F64 ? fill_dia<double>(p, st) : fill_dia<float>(p, st);
    if (rc) return rc;
  }
  p->dia_const = 0;
  if (p->sym_pending) (void)gla_dia_sym_bits(p);
  p->dia_sym = 0;
  if (g_tuning.sym) {                                 // (one pass over the diagonals + a host wait, once per value set)
    const int rc = dtype == GLA_F64 ? detect_dia_sym<double, uint64_t>(p, st) : detect_dia_sym<float, uint32_t>(p, st);
    if (rc) return rc;
  }
  if (g_tuning.compress) {
    const int rc = dtype == GLA_F64 ? detect_dia_const<double, uint64_t>(p, st) : detect_dia_const<float, uint32_t>(p, st);
    if (rc) return rc;
    if (p->dia_const) return GLA_OK;                  // matrix-free stencil: nothing else to build
    return dtype == GLA_F64 ? build_packed<double, uint64_t>(p, st) : build_packed<float, uint32_t>(p, st);
  }
  return GLA_OK;
}

int gla_csr_info(gla_csr_t p, gla_csr_info_t* info) {
  if (!p || !info) return GLA_ERR_INVALID_ARG;
  info->n_rows = p->n_rows; info->n_cols = p->n_cols; info->nnz = p->nnz;
  info->sorted_input = p->sorted_input; info->max_row_len = p->max_row_len;
  info->n_tiles = p->n_tiles; info->tile_nnz = p->tile_nnz; info->lds_elems = p->lds_elems;
  info->long_rows = p->long_rows; info->val_dtype = p->val_dtype; info->packed = (p->dia_const && g_tuning.compress) ? 2 : p->packed;
  info->col16 = (p->col16_state == 1 && g_tuning.col16) ? 1 : 0; info->col16_wide_tiles = p->col16_wide_tiles;
  info->dia = (p->dia_state == 1 && g_tuning.dia) ? p->dia_D : 0;
  info->dia_stride = info->dia ? p->dia_S : 0;
  info->dia_row_lo = info->dia ? p->dia_row_lo : 0;
  info->dia_row_hi = info->dia ? p->dia_row_hi : 0;
  info->dia_A = info->dia ? p->dia_A : 0;
  info->dia_B = info->dia ? p->dia_B : 0;
  info->dia_canon5 = info->dia && gla_dia_canon5(p) ? 1 : (info->dia && gla_dia_canon9(p) ? 9 : 0);
  info->dia_sym = info->dia && g_tuning.sym ? gla_dia_sym_bits(p) : 0;
  info->dia_full = info->dia && gla_dia_full5(p) ? 1 : 0;
  return GLA_OK;
}

__global__ __launch_bounds__(GLA_BLOCK) static void iota_kernel(int32_t* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GLA_BLOCK)
    out[i] = (int32_t)i;
}

int gla_csr_export(gla_csr_t p, int32_t* rowptr_out, int32_t* col_out, int32_t* perm_out, gla_stream_t stream) {
  if (!p) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (rowptr_out)
    GLA_HIP_CHECK(hipMemcpyAsync(rowptr_out, p->rowptr, (p->n_rows + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  if (col_out && p->nnz)
    GLA_HIP_CHECK(hipMemcpyAsync(col_out, p->col, p->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  if (perm_out && p->nnz) {
    if (p->perm)
      GLA_HIP_CHECK(hipMemcpyAsync(perm_out, p->perm, p->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    else {
      hipLaunchKernelGGL(iota_kernel, dim3(grid_for(p->nnz)), dim3(GLA_BLOCK), 0, st, perm_out, p->nnz);
      GLA_LAUNCH_CHECK();
    }
  }
  return GLA_OK;
}

int gla_csr_classify_tiles(gla_csr_t p, int64_t col_limit, gla_stream_t stream, int32_t* n_interior,
                           int32_t* n_boundary) {
  if (!p || col_limit < 0) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nt = p->n_tiles;
  if (!p->tile_lists) GLA_HIP_CHECK(hipMalloc(&p->tile_lists, 2 * (size_t)nt * sizeof(int32_t)));
  int32_t* d_flag = nullptr;
  GLA_HIP_CHECK(hipMalloc(&d_flag, nt * sizeof(int32_t)));
  GLA_HIP_CHECK(hipMemsetAsync(d_flag, 0, nt * sizeof(int32_t), st));
  if (p->n_rows > 0) {
    const int32_t lim = (int32_t)(col_limit > INT32_MAX ? INT32_MAX : col_limit);
    hipLaunchKernelGGL(tile_flag_kernel, dim3(nt), dim3(GLA_BLOCK), 0, st, p->rowptr, p->col, p->tile_row, nt, lim,
                       d_flag);
  }
  int32_t* h_flag = new int32_t[nt];
  hipError_t e = hipMemcpyAsync(h_flag, d_flag, nt * sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  int rc = GLA_OK;
  if (e != hipSuccess) rc = -(1000 + (int)e);
  else {
    int32_t* h_list = new int32_t[2 * (size_t)nt];
    int ni = 0, nb = 0;
    for (int t = 0; t < nt; ++t) {
      if (h_flag[t]) h_list[nt + nb++] = t; else h_list[ni++] = t;
    }
    e = hipMemcpyAsync(p->tile_lists, h_list, 2 * (size_t)nt * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = -(1000 + (int)e);
    p->n_interior = ni; p->n_boundary = nb;
    p->int_row_lo = p->int_row_hi = -1;             // rows of the interior tiles when they are one contiguous run
    if (ni > 0 && h_list[ni - 1] - h_list[0] + 1 == ni && rc == GLA_OK && p->tile_row && p->dia_state == 1) {
      int32_t r_lo = 0, r_hi = 0;
      if (hipMemcpy(&r_lo, p->tile_row + h_list[0], sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess &&
          hipMemcpy(&r_hi, p->tile_row + h_list[ni - 1] + 1, sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess) {
        p->int_row_lo = r_lo;
        p->int_row_hi = r_hi;
      } else {
        (void)hipGetLastError();                    // leave no pending error behind: the CSR kernels take the tiles
      }
    }
    delete[] h_list;
    if (n_interior) *n_interior = ni;
    if (n_boundary) *n_boundary = nb;
  }
  delete[] h_flag;
  (void)hipFree(d_flag);
  return rc;
}

// ---- builders --------------------------------------------------------------------------------
int64_t gla_laplacian2d_nnz(int64_t Nx, int64_t Ny, int64_t row_lo, int64_t row_hi) {
  if (Nx < 1 || Ny < 1 || row_lo < 0 || row_hi > Nx * Ny || row_lo > row_hi) return -1;
  return lap_offset(row_hi, Nx, Ny) - lap_offset(row_lo, Nx, Ny);
}

int gla_laplacian2d_coo(int64_t Nx, int64_t Ny, int64_t row_lo, int64_t row_hi, int64_t* row, int64_t* col,
                        void* val, int dtype, gla_stream_t stream) {
  if (Nx < 1 || Ny < 1 || row_lo < 0 || row_hi > Nx * Ny || row_lo > row_hi || !row || !col || !val ||
      (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (row_lo == row_hi) return GLA_OK;
  const int64_t base = lap_offset(row_lo, Nx, Ny);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLA_F64)
    hipLaunchKernelGGL((laplacian2d_kernel<double>), dim3(grid_for(row_hi - row_lo)), dim3(GLA_BLOCK), 0, st, Nx, Ny,
                       row_lo, row_hi, base, row, col, (double*)val);
  else
    hipLaunchKernelGGL((laplacian2d_kernel<float>), dim3(grid_for(row_hi - row_lo)), dim3(GLA_BLOCK), 0, st, Nx, Ny,
                       row_lo, row_hi, base, row, col, (float*)val);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

int64_t gla_compact_work_bytes(int64_t nnz) {
  const int64_t nb = (nnz + CMP_TILE - 1) / CMP_TILE;
  return (nb + 2) * (int64_t)sizeof(int64_t);
}

int gla_remove_diag_count(int64_t nnz, const int64_t* row, const int64_t* col, void* work, int64_t* kept_host,
                          gla_stream_t stream) {
  if (nnz < 0 || !work || !kept_host || (nnz > 0 && (!row || !col))) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (nnz + CMP_TILE - 1) / CMP_TILE;
  int64_t* w = (int64_t*)work;            // [0..nb) block offsets, [nb] total
  if (nb > 0) {
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(GLA_BLOCK), 0, st, row, col, nnz, w);
    GLA_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, w, nb, w + nb);
  GLA_LAUNCH_CHECK();
  GLA_HIP_CHECK(hipMemcpyAsync(kept_host, w + nb, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GLA_HIP_CHECK(hipStreamSynchronize(st));
  return GLA_OK;
}

int gla_remove_diag_apply(int64_t nnz, const int64_t* row, const int64_t* col, const void* val, int64_t val_stride,
                          int dtype, const void* work, int64_t* row_out, int64_t* col_out, void* val_out,
                          gla_stream_t stream) {
  if (nnz < 0 || !work || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (nnz == 0) return GLA_OK;
  if (!row || !col || !row_out || !col_out || (val && (!val_out || val_stride < 1))) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = (nnz + CMP_TILE - 1) / CMP_TILE;
  if (dtype == GLA_F64)
    hipLaunchKernelGGL((compact_apply_kernel<double>), dim3((unsigned)nb), dim3(GLA_BLOCK), 0, st, row, col,
                       (const double*)val, val_stride, nnz, (const int64_t*)work, row_out, col_out, (double*)val_out);
  else
    hipLaunchKernelGGL((compact_apply_kernel<float>), dim3((unsigned)nb), dim3(GLA_BLOCK), 0, st, row, col,
                       (const float*)val, val_stride, nnz, (const int64_t*)work, row_out, col_out, (float*)val_out);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

}  // extern "C"
