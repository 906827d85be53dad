// A sliding LDS window of x for BANDED irregular matrices (round 5 experiment).
// profiles/r04/pmc_irregular_summary.txt: with columns within +-4096 of the row the CSR-adaptive stream kernel is bound by the
// L1 / L2 REQUEST rate of its 8-byte gathers (one tag lookup per nonzero), not by bytes.  Here: the same tiles, the same in-order /
// wave-tree row sums, two ways --
//   A  one workgroup per tile, x[col] gathered from global memory (the library's spmv_stream_vec2_kernel in outline);
//   B  persistent workgroups walking consecutive tiles; the x window of a tile ([first row + min offset, last row + max offset]) is
//      held in LDS (loaded with coalesced 16-byte loads, re-based only when a tile's range leaves it), the gathers are LDS reads,
//      and the next tile's nonzeros are fetched into registers while this tile's rows are summed.
// Both produce bit-identical y.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/band_spmv.hip -o tools/micro/band_spmv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

constexpr int WAVE_ROW = 64;

struct Tile { int r0, r1, s, e; };

__device__ __forceinline__ double wave_tree(double part) {
  for (int off = 1; off < 64; off <<= 1) part = part + __shfl_xor(part, off, 64);
  return part;
}

// row sums of one tile from the staged products (prod[j - s]): short rows one lane in edge order, rows of >= 64 nonzeros by their
// wave (64 chunks + xor tree), as gla_rows.hip does
template <int NT, bool RR>
__device__ __forceinline__ void tile_row_sums(const Tile& ti, const int* __restrict__ rowptr, const double* prod, double* __restrict__ y) {
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int NW = NT / 64;
  const int slot = RR ? (tid & 63) * NW + (tid >> 6) : tid;       // RR: the tile's rows go round-robin over the waves (a tile of a few
  for (int rbase = ti.r0; rbase < ti.r1; rbase += NT) {            // long rows keeps every wave busy, not wave 0 alone)
    const int rr = rbase + slot;
    const bool have = rr < ti.r1;
    int a = 0, b = 0;
    if (have) { a = rowptr[rr]; b = rowptr[rr + 1]; }
    const int len = b - a;
    const bool heavy = have && len >= WAVE_ROW;
    double acc = 0.0;
    if (have && !heavy) {
      for (int j = a - ti.s; j < b - ti.s; ++j) acc = acc + prod[j];
    }
    unsigned long long hm = __ballot(heavy);
    while (hm) {
      const int hl = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      const int ha = __shfl(a, hl, 64) - ti.s, hb = __shfl(b, hl, 64) - ti.s;
      const int chunk = (hb - ha + 63) >> 6;
      int j = ha + lane * chunk;
      const int jend = min(j + chunk, hb);
      double part = 0.0;
      for (; j < jend; ++j) part = part + prod[j];
      part = wave_tree(part);
      if (lane == hl) acc = part;
    }
    if (have) y[rr] = acc;
  }
}

// A: one workgroup per tile, global gathers
template <bool RR>
__global__ __launch_bounds__(256) void spmv_gather(const Tile* __restrict__ tiles, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
  extern __shared__ double prod[];
  const Tile ti = tiles[blockIdx.x];
  const int len = ti.e - ti.s;
  for (int base = 0; base < len; base += 256 * 4) {
    double v[4]; int c[4]; int q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { q[u] = min(base + u * 256 + (int)threadIdx.x, len - 1); v[u] = val[ti.s + q[u]]; c[u] = col[ti.s + q[u]]; }
    double xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) prod[q[u]] = v[u] * xv[u];
  }
  __syncthreads();
  tile_row_sums<256, RR>(ti, rowptr, prod, y);
}

// B: persistent workgroups, LDS window of x, the nonzeros of the next PD tiles prefetched into registers
template <int NT, int PPT, bool RR, int PD>      // PPT nonzeros per thread and tile: a tile holds at most NT * PPT nonzeros
__global__ __launch_bounds__(NT) void spmv_window(const Tile* __restrict__ tiles, int n_tiles, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, int n_cols,
                                                  int band_lo, int band_hi, int wsz) {
  extern __shared__ __align__(16) double smem[];
  double* win = smem;                      // [wsz]   x[base .. base + wsz)
  double* prod = smem + wsz;               // [NT * PPT]
  const int tid = threadIdx.x;
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(n_tiles, t0 + per);
  if (t0 >= t1) return;
  int base = 0, top = 0;                   // window holds x[base, top)
  Tile tq[PD];                             // the tiles whose nonzeros are in registers: tq[k] = tile t + k
  double v[PD][PPT]; int c[PD][PPT];
  auto fetch = [&](int k, int t) {
    tq[k] = tiles[min(t, t1 - 1)];
    const int len = tq[k].e - tq[k].s;
#pragma unroll
    for (int u = 0; u < PPT; ++u) { const int q = min(u * NT + tid, len - 1); v[k][u] = val[tq[k].s + q]; c[k][u] = col[tq[k].s + q]; }
  };
#pragma unroll
  for (int k = 0; k < PD; ++k) fetch(k, t0 + k);
  for (int tb = t0; tb < t1; tb += PD) {
#pragma unroll
    for (int k = 0; k < PD; ++k) {         // (static register names: slot k holds tile tb + k)
      const int t = tb + k;
      if (t >= t1) break;
      const Tile ti = tq[k];
      const int need_lo = max(0, ti.r0 + band_lo), need_hi = min(n_cols, ti.r1 + band_hi);      // [need_lo, need_hi)
      if (need_lo < base || need_hi > top) {                                                      // re-base (workgroup-uniform)
        __syncthreads();
        base = need_lo & ~1;
        top = min(n_cols, base + wsz);
        for (int i = 2 * tid; i < top - base; i += 2 * NT) {
          if (base + i + 1 < n_cols) *(double2*)(win + i) = *(const double2*)(x + base + i);
          else win[i] = x[base + i];
        }
        __syncthreads();
      }
      const int len = ti.e - ti.s;
#pragma unroll
      for (int u = 0; u < PPT; ++u) { const int q = min(u * NT + tid, len - 1); prod[q] = v[k][u] * win[c[k][u] - base]; }
      if (t + PD < t1) fetch(k, t + PD);                                                          // in flight during PD tiles' work
      __syncthreads();
      tile_row_sums<NT, RR>(ti, rowptr, prod, y);
      __syncthreads();
    }
  }
}

int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 0;     // 0: 200 k rows of 64 .. 1024 nonzeros; 1: 1 M rows, power-law 3 .. 3000
  const int band = argc > 2 ? atoi(argv[2]) : 4096;
  std::mt19937_64 gen(11);
  int n;
  std::vector<int> lens;
  if (which == 0) { n = 200000; lens.resize(n); for (auto& l : lens) l = 64 + (int)(gen() % 961); }
  else {
    n = 1000000; lens.resize(n);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (auto& l : lens) { double u = U(gen); double v = 3.0 * pow(1.0 - u, -1.0 / 1.2); l = (int)std::min(3000.0, v); }
  }
  std::vector<int> rowptr(n + 1, 0);
  for (int i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + lens[i];
  const int64_t nnz = rowptr[n];
  std::vector<int> col(nnz);
  std::vector<double> val(nnz), x(n);
  for (int i = 0; i < n; ++i)
    for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      int c = i + (int)(gen() % (2 * band + 1)) - band;
      col[j] = std::min(std::max(c, 0), n - 1);
      val[j] = (double)(gen() % 2000) / 2000.0 - 0.5;
    }
  for (auto& xv : x) xv = (double)(gen() % 1000) / 1000.0;
  // tiles: consecutive rows up to TILE nonzeros / 256 rows
  const int TILE = 3072;
  std::vector<Tile> tiles;
  for (int r = 0; r < n;) {
    int r1 = r, s = rowptr[r];
    while (r1 < n && r1 - r < 256 && rowptr[r1 + 1] - s <= TILE) ++r1;
    if (r1 == r) ++r1;
    tiles.push_back({r, r1, s, rowptr[r1]});
    r = r1;
  }
  int max_nnz = 0, max_rows = 0;
  for (auto& t : tiles) { max_nnz = std::max(max_nnz, t.e - t.s); max_rows = std::max(max_rows, t.r1 - t.r0); }
  printf("matrix %d: %d rows, %ld nonzeros, band +-%d, %zu tiles (max %d nonzeros, %d rows)\n", which, n, (long)nnz, band, tiles.size(), max_nnz, max_rows);
  if (max_nnz > TILE) { printf("a row exceeds the tile capacity; not handled here\n"); return 0; }
  int *d_rowptr, *d_col; double *d_val, *d_x, *d_y, *d_y2; Tile* d_tiles;
  hipMalloc(&d_rowptr, (n + 1) * 4); hipMalloc(&d_col, nnz * 4 + 64); hipMalloc(&d_val, nnz * 8 + 64); hipMalloc(&d_x, n * 8 + 64);
  hipMalloc(&d_y, n * 8); hipMalloc(&d_y2, n * 8); hipMalloc(&d_tiles, tiles.size() * sizeof(Tile));
  hipMemcpy(d_rowptr, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice); hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < 11; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2] * 1e3, B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("%-70s %8.1f us   %6.0f GB/s on SURVEY's bytes (%.3f of 8 TB/s)\n", name, us, B / us / 1e3, B / us / 1e3 / 8000.0);
    fflush(stdout);
  };
  const int nt = (int)tiles.size();
  std::vector<double> yb_(n);
  timeit("A  one workgroup per tile, x gathered from global memory", [&] { hipLaunchKernelGGL(spmv_gather<false>, dim3(nt), dim3(256), TILE * 8, 0, d_tiles, d_rowptr, d_col, d_val, d_x, d_y); });
  hipMemcpy(yb_.data(), d_y, n * 8, hipMemcpyDeviceToHost);
  timeit("A' the same, rows round-robin over the waves", [&] { hipLaunchKernelGGL(spmv_gather<true>, dim3(nt), dim3(256), TILE * 8, 0, d_tiles, d_rowptr, d_col, d_val, d_x, d_y); });
  std::vector<double> ya(n), yb(n);
  hipMemcpy(ya.data(), d_y, n * 8, hipMemcpyDeviceToHost);
  { size_t bad = 0; for (int i = 0; i < n; ++i) bad += ya[i] != yb_[i]; printf("      rows of A' that differ from A: %zu\n", bad); }
  const int span = 2 * band + 256 + 2;
  const int wsz = span + 2048;
  auto runB = [&](const char* what, auto kern, int nt_, int ppt) {
    const size_t lds = (size_t)(wsz + nt_ * ppt) * 8;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    char name[160];
    snprintf(name, sizeof name, "B  256 persistent workgroups x %d threads, %s (%zu KB LDS)", nt_, what, lds / 1024);
    hipMemset(d_y2, 0, n * 8);
    timeit(name, [&] { hipLaunchKernelGGL(kern, dim3(256), dim3(nt_), lds, 0, d_tiles, nt, d_rowptr, d_col, d_val, d_x, d_y2, n, -band, band + 1, wsz); });
    hipMemcpy(yb.data(), d_y2, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < n; ++i) bad += ya[i] != yb[i];
    printf("      rows that differ from A: %zu\n", bad);
  };
  runB("1 tile ahead", spmv_window<512, 6, true, 1>, 512, 6);
  runB("2 tiles ahead", spmv_window<512, 6, true, 2>, 512, 6);
  runB("3 tiles ahead", spmv_window<512, 6, true, 3>, 512, 6);
  runB("4 tiles ahead", spmv_window<512, 6, true, 4>, 512, 6);
  runB("2 tiles ahead, 1024 threads", spmv_window<1024, 3, true, 2>, 1024, 3);
  runB("4 tiles ahead, 1024 threads", spmv_window<1024, 3, true, 4>, 1024, 3);
  return 0;
}
