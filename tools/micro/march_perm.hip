// The MEMORY SIDE of the headline Jacobi pass, two ways (round 5).  tools/ab_m5_variants.py showed that the 7-sweep pass with its
// levels compiled out still takes 165 of its 185 us: the pass is bound by how fast its load pattern -- 510 workgroups marching
// down grid lines, five 2 KiB segments per line step, one line step of register prefetch -- pulls bytes (5.3 TB/s), and the
// registers (214 of 256) leave no room for a deeper register prefetch.  This tool streams the same pattern
//   A  with 8-byte register loads one line step ahead (what jacobi_march5_kernel does), and
//   B  with LDS-DMA (global_load_lds_dwordx4: no destination registers) NST - 1 line steps ahead into an LDS ring, counted
//      s_waitcnt vmcnt(N) + raw s_barrier,
// consumes every value (a sum) and writes one value per owned point, exactly the pass's traffic (3 diagonals, b, x in; x out).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/march_glds.hip -o tools/micro/march_glds ; run: tools/micro/march_glds [N=4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int HALO = 7;
#ifndef BWV
#define BWV 256
#endif
constexpr int BW = BWV, BI = BW - 2 * HALO;

struct Args {
  const double* arr[5];     // east, centre, south, b, x
  double* out;
  int64_t S, nlines;
  int lines, nstrips;
  int ramp, xcd_map, nblk;  // kernel A: extra lines loaded above a block; 1: consecutive tiles share an XCD; workgroups in all
  int perm, nlb;            // 1: line j of block k is STORED at line slot (j - J0) * nlb + k: what all the blocks read in one step is contiguous
};

__device__ __forceinline__ void geom(const Args& a, int& strip, int64_t& J0, int& L, int64_t& P0, bool& owner) {
  int blk = blockIdx.x;
  if (a.xcd_map) {                                   // workgroup b runs on XCD b % 8: give XCD e the tiles [e q, (e + 1) q)
    const int q = (a.nblk + 7) / 8;
    blk = (blockIdx.x % 8) * q + blockIdx.x / 8;
    if (blk >= a.nblk) { L = 0; J0 = 0; P0 = 0; strip = 0; owner = false; return; }
  }
  const int lb = blk / a.nstrips;
  strip = blk - lb * a.nstrips;
  J0 = (int64_t)lb * a.lines;
  L = (int)min((int64_t)a.lines, a.nlines - J0);
  P0 = max((int64_t)0, min((int64_t)strip * BI, a.S - BW));
  const int64_t gi = P0 + threadIdx.x;
  const int64_t lo = strip == 0 ? 0 : (int64_t)strip * BI + HALO, hi = strip == a.nstrips - 1 ? a.S : (int64_t)strip * BI + BW - HALO;
  owner = gi >= lo && gi < hi;
}

// A: register loads, one step ahead
template <int MINB>
__global__ __launch_bounds__(BW, MINB) void march_reg(Args a) {
  const int RAMP = a.ramp;
  __shared__ double line[2][BW + 2];
  int strip; int64_t J0, P0; int L; bool owner;
  geom(a, strip, J0, L, P0, owner);
  const int tid = threadIdx.x;
  const int lbq = (int)(J0 / a.lines);
  auto rowc = [&](int jl) -> int64_t {
    if (a.perm) {
      int64_t slot;
      if (jl >= 0) slot = (int64_t)jl * a.nlb + lbq;                           // (own lines)
      else slot = lbq > 0 ? (int64_t)(a.lines + jl) * a.nlb + (lbq - 1) : 0;   // (ramp: the tail of the block above)
      return slot * a.S + P0 + tid;
    }
    return min(max((J0 + jl) * a.S + P0 + tid, (int64_t)0), a.S * a.nlines - 1);
  };
  double v[5], w[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = a.arr[k][rowc(-RAMP)];
  for (int jl = -RAMP; jl < L; ++jl) {
    const int64_t rn = rowc(jl + 1);
#pragma unroll
    for (int k = 0; k < 5; ++k) w[k] = a.arr[k][rn];
    __syncthreads();
    line[jl & 1][1 + tid] = v[4];
    const double nb = line[(jl + 1) & 1][tid] + line[(jl + 1) & 1][2 + tid];
    const double s = (((v[0] + v[1]) + (v[2] + v[3])) + v[4]) + nb;
    if (jl >= 0 && owner) a.out[a.perm ? ((int64_t)jl * a.nlb + lbq) * a.S + P0 + tid : (J0 + jl) * a.S + P0 + tid] = s;
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = w[k];
  }
}

// B: LDS-DMA ring of NST stages; each wave issues 3 of the 12 DMA instructions of a line (10 real + 2 repeats)
template <int NST, int RAMP>
__global__ __launch_bounds__(BW, 2) void march_glds(Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* ring = reinterpret_cast<double*>(smem);                    // [NST][5][BW]
  double* line = ring + NST * 5 * BW;                                // [2][BW + 2]
  int strip; int64_t J0, P0; int L; bool owner;
  geom(a, strip, J0, L, P0, owner);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2 * (BW + 2); i += BW) line[i] = 0.0;
  const int64_t nrows = a.S * a.nlines;
  auto issue = [&](int jl) {                                         // DMA line jl into ring slot (jl + 64 NST) % NST
    const int slot = (jl + 64 * NST) % NST;
    const int64_t row0 = min(max((J0 + jl) * a.S + P0, (int64_t)0), nrows - BW);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = (wave + 4 * q) % 10;
      const int arr = i >> 1, half = i & 1;
      const double* src = a.arr[arr] + row0 + half * 128 + 2 * lane;
      double* dst = ring + ((size_t)slot * 5 + arr) * BW + half * 128;          // wave-uniform; the DMA adds lane x 16 bytes
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };
#pragma unroll
  for (int p = 0; p < NST - 1; ++p) issue(-RAMP + p);
  __syncthreads();
  int stores = 0;
  for (int jl = -RAMP; jl < L; ++jl) {
    // stage jl must have landed (this wave's 3 DMAs of it): the ops issued after them are the DMAs of stages jl+1 .. jl+NST-2
    // (3 each) and the stores of the steps since.  Stores are counted exactly only once NST - 1 of them are behind us.
    if (stores >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * (NST - 2) + (NST - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * (NST - 2)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(jl + NST - 1);                                             // into the slot every wave finished reading before the barrier
    const double* st = ring + (size_t)((jl + 64 * NST) % NST) * 5 * BW;
    double v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = st[k * BW + tid];
    line[(jl & 1) * (BW + 2) + 1 + tid] = v[4];
    const double nb = line[((jl + 1) & 1) * (BW + 2) + tid] + line[((jl + 1) & 1) * (BW + 2) + 2 + tid];
    const double s = (((v[0] + v[1]) + (v[2] + v[3])) + v[4]) + nb;
    if (jl >= 0) {                                                   // (every wave of a strip holds owner lanes: one store per wave and step)
      if (owner) a.out[(J0 + jl) * a.S + P0 + tid] = s;
      ++stores;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atol(argv[1]) : 4096;
  const int64_t n = N * N;
  Args a;
  double* bufs[6];
  for (int k = 0; k < 6; ++k) { hipMalloc(&bufs[k], n * 10 + 4096); hipMemset(bufs[k], 0, n * 10 + 4096); }
  std::vector<double> h(n);
  for (int k = 0; k < 5; ++k) {
    for (int64_t i = 0; i < n; ++i) h[i] = (double)((i * 7 + k * 13) % 1000) * 0.001;
    hipMemcpy(bufs[k], h.data(), n * 8, hipMemcpyHostToDevice);
    a.arr[k] = bufs[k];
  }
  a.out = bufs[5];
  a.S = N; a.nlines = N;
  a.nstrips = (int)((N + BI - 1) / BI);
  const int slots = 512;
  int nlb = slots / a.nstrips;
  a.lines = (int)((N + nlb - 1) / nlb);
  nlb = (int)((N + a.lines - 1) / a.lines);
  const int grid = nlb * a.nstrips;
  printf("N = %ld: %d strips x %d line blocks of %d lines = %d workgroups; bytes per pass (5 reads + 1 write of 8 B per row) = %.1f MB\n",
         (long)N, a.nstrips, nlb, a.lines, grid, 48.0 * n / 1e6);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> ref(n), got(n);
  auto timeit = [&](const char* name, auto launch, bool check) {
    for (int i = 0; i < 20; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < 15; ++i) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    hipMemcpy(got.data(), a.out, n * 8, hipMemcpyDeviceToHost);
    bool same = true;
    if (check) { for (int64_t i = 0; i < n && same; ++i) same = got[i] == ref[i]; }
    else ref = got;
    const double us = ts[ts.size() / 2] * 1e3;
    printf("%-52s %7.1f us  %6.0f GB/s on 48 B per row   %s\n", name, us, 48.0 * n / us / 1e3, check ? (same ? "same values" : "VALUES DIFFER") : "");
    hipMemset(a.out, 0, n * 8);
  };
  constexpr int RAMP = 12;
  a.ramp = RAMP; a.xcd_map = 0; a.nblk = grid; a.perm = 0; a.nlb = nlb;
  auto geometry = [&](int slots_, int ramp, int xcd) {     // kernel A with another decomposition
    int nlb_ = std::max(1, slots_ / a.nstrips);
    a.lines = (int)((N + nlb_ - 1) / nlb_);
    nlb_ = (int)((N + a.lines - 1) / a.lines);
    a.ramp = ramp; a.xcd_map = xcd; a.nblk = nlb_ * a.nstrips;
    return xcd ? ((a.nblk + 7) / 8) * 8 : a.nblk;
  };
  for (int rnd = 0; rnd < 2; ++rnd) {
    char name[128];
    for (int xcd : {0, 1})
      for (int slots_ : {512, 1024})
       for (int pm : {0, 1}) {
        for (int ramp : {12, 0}) {
          const int g_ = geometry(slots_, ramp, xcd);
          a.perm = pm; a.nlb = a.nblk / a.nstrips;
          snprintf(name, sizeof name, "A %6d wgs of %4d lines, ramp %2d, xcd map %d, line-interleaved storage %d", a.nblk, a.lines, ramp, xcd, pm);
          if (slots_ <= 512) timeit(name, [&] { hipLaunchKernelGGL((march_reg<BW == 256 ? 2 : 1>), dim3(g_), dim3(BW), 0, 0, a); }, false);
          else timeit(name, [&] { hipLaunchKernelGGL((march_reg<BW == 256 ? 4 : 2>), dim3(g_), dim3(BW), 0, 0, a); }, false);
        }
      }
    geometry(512, RAMP, 0); a.perm = 0;
    const int grid2 = a.nblk;
    if (BW == 256) timeit("B LDS-DMA ring, 2 stages (1 step ahead)", [&] { hipLaunchKernelGGL((march_glds<2, RAMP>), dim3(grid2), dim3(BW), (2 * 5 * BW + 2 * (BW + 2)) * 8, 0, a); }, false);
    if (BW == 256) timeit("B LDS-DMA ring, 4 stages (3 steps ahead)", [&] { hipLaunchKernelGGL((march_glds<4, RAMP>), dim3(grid2), dim3(BW), (4 * 5 * BW + 2 * (BW + 2)) * 8, 0, a); }, false);
  }
  return 0;
}
