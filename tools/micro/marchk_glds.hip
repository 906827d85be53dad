// The MEMORY SIDE of the K-wide Jacobi pass (jacobi_marchk_kernel<double, 8, 5, 5, 512>: 4096^2 x 16 right-hand sides), two ways.
// One 512-thread workgroup per CU (255 registers), lanes = (point, 16-byte rhs vector): per line step a workgroup loads 8 KiB of
// x and 8 KiB of b ONE step ahead (17 KB in flight per CU) and stores 8 KiB.  The pass moves 7.4 GB in 2.2 ms = 3.3 TB/s.
//   A  register loads one step ahead (what the kernel does)
//   B  LDS-DMA (global_load_lds_dwordx4) NST - 1 steps ahead into an LDS ring, counted s_waitcnt vmcnt(N), raw s_barrier
// Both consume every value and write one 16-byte vector per owned (point, vector).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/marchk_glds.hip -o tools/micro/marchk_glds ; run: tools/micro/marchk_glds [N=4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NT = 512, KV = 8, BP = NT / KV, HALO = 5, BI = BP - 2 * HALO, K = 16;
typedef double D2 __attribute__((ext_vector_type(2)));

struct Args {
  const double *x, *b;
  double* out;
  int64_t S, nlines;
  int lines, nstrips, nblk, ramp;
};

__device__ __forceinline__ bool geom(const Args& a, int blk, int64_t& J0, int& L, int64_t& P0, bool& owner) {
  if (blk >= a.nblk) return false;
  const int lb = blk / a.nstrips, strip = blk - lb * a.nstrips;
  J0 = (int64_t)lb * a.lines;
  L = (int)min((int64_t)a.lines, a.nlines - J0);
  P0 = max((int64_t)0, min((int64_t)strip * BI, a.S - BP));
  const int64_t gi = P0 + threadIdx.x / KV;
  const int64_t lo = strip == 0 ? 0 : (int64_t)strip * BI + HALO, hi = strip == a.nstrips - 1 ? a.S : (int64_t)strip * BI + BP - HALO;
  owner = gi >= lo && gi < hi;
  return true;
}

__global__ __launch_bounds__(NT, 1) void marchk_reg(Args a) {
  __shared__ D2 line[2][(BP + 2) * KV];
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  for (int blk = blockIdx.x; blk < a.nblk; blk += gridDim.x) {
    int64_t J0, P0; int L; bool owner;
    geom(a, blk, J0, L, P0, owner);
    const int64_t nrows = a.S * a.nlines;
    auto rowc = [&](int jl) -> int64_t { return min(max((J0 + jl) * a.S + P0 + ii, (int64_t)0), nrows - 1); };
    D2 xv = *(const D2*)(a.x + rowc(-a.ramp) * K + q * 2), bv = *(const D2*)(a.b + rowc(-a.ramp) * K + q * 2);
    for (int jl = -a.ramp; jl < L; ++jl) {
      const int64_t rn = rowc(jl + 1);
      const D2 xn = *(const D2*)(a.x + rn * K + q * 2), bn = *(const D2*)(a.b + rn * K + q * 2);
      __syncthreads();
      line[jl & 1][(1 + ii) * KV + q] = xv;
      const D2 nb = line[(jl + 1) & 1][ii * KV + q] + line[(jl + 1) & 1][(2 + ii) * KV + q];
      const D2 s = (xv + bv) + nb;
      if (jl >= 0 && owner) *(D2*)(a.out + ((J0 + jl) * a.S + P0 + ii) * K + q * 2) = s;
      xv = xn; bv = bn;
    }
    __syncthreads();
  }
}

template <int NST>
__global__ __launch_bounds__(NT, 1) void marchk_glds(Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  D2* ring = reinterpret_cast<D2*>(smem);                             // [NST][2][NT]  (x line, b line: lane-linear)
  D2* line = ring + NST * 2 * NT;                                     // [2][(BP + 2) KV]
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2 * (BP + 2) * KV; i += NT) line[i] = D2{0.0, 0.0};
  for (int blk = blockIdx.x; blk < a.nblk; blk += gridDim.x) {
    int64_t J0, P0; int L; bool owner;
    geom(a, blk, J0, L, P0, owner);
    const int64_t nrows = a.S * a.nlines;
    auto issue = [&](int jl) {                                        // this wave's 1 KiB of the x line and of the b line
      const int slot = (jl + 64 * NST) % NST;
      const int64_t row = min(max((J0 + jl) * a.S + P0 + ii, (int64_t)0), nrows - 1);
      D2* dx = ring + ((size_t)slot * 2 + 0) * NT + wave * 64;        // wave-uniform; the DMA adds lane x 16 bytes
      D2* db = ring + ((size_t)slot * 2 + 1) * NT + wave * 64;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.x + row * K + q * 2),
                                       (void __attribute__((address_space(3)))*)dx, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.b + row * K + q * 2),
                                       (void __attribute__((address_space(3)))*)db, 16, 0, 0);
      (void)lane;
    };
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) issue(-a.ramp + p);
    int stores = 0;
    for (int jl = -a.ramp; jl < L; ++jl) {
      if (stores >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (NST - 2) + (NST - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (NST - 2)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(jl + NST - 1);
      const D2* st = ring + (size_t)((jl + 64 * NST) % NST) * 2 * NT;
      const D2 xv = st[tid], bv = st[NT + tid];
      line[(jl & 1) * (BP + 2) * KV + (1 + ii) * KV + q] = xv;
      const D2 nb = line[((jl + 1) & 1) * (BP + 2) * KV + ii * KV + q] + line[((jl + 1) & 1) * (BP + 2) * KV + (2 + ii) * KV + q];
      const D2 s = (xv + bv) + nb;
      if (jl >= 0) {
        if (owner) *(D2*)(a.out + ((J0 + jl) * a.S + P0 + ii) * K + q * 2) = s;
        ++stores;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atol(argv[1]) : 4096;
  const int64_t n = N * N;
  Args a;
  double *x, *b, *out;
  hipMalloc(&x, n * K * 8 + 4096); hipMalloc(&b, n * K * 8 + 4096); hipMalloc(&out, n * K * 8 + 4096);
  hipMemset(x, 0, n * K * 8); hipMemset(b, 0, n * K * 8); hipMemset(out, 0, n * K * 8);
  a.x = x; a.b = b; a.out = out; a.S = N; a.nlines = N;
  a.nstrips = (int)((N + BI - 1) / BI);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < 9; ++i) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2] * 1e3;
    printf("%-64s %8.1f us  %6.0f GB/s on 3 x 128 B per row\n", name, us, 384.0 * n / us / 1e3);
    fflush(stdout);
  };
  printf("N = %ld, K = 16: %d strips of %d points (halo %d); bytes per pass (x, b in; x out) = %.2f GB\n", (long)N, a.nstrips, BP, HALO, 384.0 * n / 1e9);
  for (int rounds : {3, 1}) {
    for (int ramp : {8, 0}) {
      int nlb = std::max(1, 256 * rounds / a.nstrips);
      a.lines = (int)((N + nlb - 1) / nlb);
      nlb = (int)((N + a.lines - 1) / a.lines);
      a.nblk = nlb * a.nstrips; a.ramp = ramp;
      const int grid = std::min(a.nblk, 256);
      char name[160];
      snprintf(name, sizeof name, "A register loads 1 step ahead   (%d blocks of %d lines, ramp %d, %d wgs)", a.nblk, a.lines, ramp, grid);
      timeit(name, [&] { hipLaunchKernelGGL(marchk_reg, dim3(grid), dim3(NT), 0, 0, a); });
      const size_t lds_line = 2 * (BP + 2) * KV * 16;
      snprintf(name, sizeof name, "B LDS-DMA ring 2 stages (1 ahead)");
      timeit(name, [&] { hipLaunchKernelGGL((marchk_glds<2>), dim3(grid), dim3(NT), 2 * 2 * NT * 16 + lds_line, 0, a); });
      snprintf(name, sizeof name, "B LDS-DMA ring 3 stages (2 ahead)");
      timeit(name, [&] { hipLaunchKernelGGL((marchk_glds<3>), dim3(grid), dim3(NT), 3 * 2 * NT * 16 + lds_line, 0, a); });
      snprintf(name, sizeof name, "B LDS-DMA ring 4 stages (3 ahead)");
      timeit(name, [&] { hipLaunchKernelGGL((marchk_glds<4>), dim3(grid), dim3(NT), 4 * 2 * NT * 16 + lds_line, 0, a); });
      snprintf(name, sizeof name, "B LDS-DMA ring 6 stages (5 ahead)");
      timeit(name, [&] { hipLaunchKernelGGL((marchk_glds<6>), dim3(grid), dim3(NT), 6 * 2 * NT * 16 + lds_line, 0, a); });
    }
  }
  return 0;
}
