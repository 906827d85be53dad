// Streaming-bandwidth ceilings of one MI355X, second edition (round 5): the round-2 tool (membw.hip) topped out at
// 4.9-5.06 TB/s for a 1:1 copy while /opt/skills/guides/MI355X_MICROARCH.md records 6.29 TB/s (float4 copy).  This one sweeps
// what the first did not: grid size (one-shot grids of n / (256 U) workgroups vs persistent grids of 256 k), the order a
// workgroup walks memory in (grid-stride vs one contiguous chunk per workgroup), unroll depth, non-temporal loads AND stores,
// stream sizes from 256 MiB to 2 GiB, and the read : write mixes of the hot-path passes (5 : 1 = the 48-byte Jacobi pass).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/membw2.hip -o tools/micro/membw2 ; run: tools/micro/membw2 [MiB=1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float F4 __attribute__((ext_vector_type(4)));

struct Ptrs { const F4* r[8]; F4* w[4]; };

// ORDER 0: grid-stride (workgroup b takes 256-element pieces b, b + G, ...; U pieces in flight per lane)
// ORDER 1: one contiguous chunk of n / G elements per workgroup, walked in pieces of 256 * U
template <int NR, int NW, int NTL, int NTS, int U, int ORDER>
__global__ __launch_bounds__(256) void stream_kernel(Ptrs p, size_t n, float* sink) {
  F4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t G = gridDim.x;
  size_t base, end, step;
  if (ORDER == 0) { base = (size_t)blockIdx.x * 256 + threadIdx.x; end = n; step = G * 256; }
  else {
    const size_t per = ((n + G - 1) / G + 255) / 256 * 256;
    base = (size_t)blockIdx.x * per + threadIdx.x; end = min(n, ((size_t)blockIdx.x + 1) * per); step = 256;
  }
  for (; base < end; base += step * U) {
    F4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * step < end ? base + u * step : end - 1;
      F4 s = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
      for (int k = 0; k < NR; ++k) s += NTL ? __builtin_nontemporal_load(&p.r[k][i]) : p.r[k][i];
      v[u] = s;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * step < end ? base + u * step : end - 1;
      if (NW == 0) acc += v[u];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        if (NTS) __builtin_nontemporal_store(v[u], &p.w[k][i]);
        else p.w[k][i] = v[u];
      }
    }
  }
  if (NW == 0 && acc.x == 123.456f) *sink = acc.y;
}

template <int NR, int NW, int NTL, int NTS, int U, int ORDER>
double run(const Ptrs& p, size_t n, float* sink, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 9; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<NR, NW, NTL, NTS, U, ORDER>), dim3(blocks), dim3(256), 0, 0, p, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 3) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const size_t mib_max = argc > 1 ? atol(argv[1]) : 1024;
  const size_t bytes_max = mib_max << 20;
  F4* buf[12];
  // (a page's worth of skew between the streams: 12 arrays an exact power of two apart would share their channel phase)
  for (int i = 0; i < 12; ++i) { hipMalloc(&buf[i], bytes_max + (1 << 20)); hipMemset(buf[i], 0, bytes_max + (1 << 20)); }
  float* sink;
  hipMalloc(&sink, 8);
  for (int warm = 0; warm < 1; ++warm) {           // clocks up: ~0.3 s of copies
    for (int it = 0; it < 200; ++it) hipMemcpyAsync(buf[1], buf[0], bytes_max, hipMemcpyDeviceToDevice, 0);
    hipDeviceSynchronize();
  }
  auto ptrs = [&](int skew) {
    Ptrs p;
    for (int i = 0; i < 8; ++i) p.r[i] = buf[i] + (skew ? (size_t)i * 4096 / 16 * 3 : 0);
    for (int i = 0; i < 4; ++i) p.w[i] = buf[8 + i] + (skew ? (size_t)(8 + i) * 4096 / 16 * 3 : 0);
    return p;
  };
#define RUN(NR, NW, NTL, NTS, U, ORDER, B, MIB, SKEW)                                                            \
  {                                                                                                              \
    const size_t bytes = (size_t)(MIB) << 20, n = bytes / 16;                                                    \
    const int blocks = (B) > 0 ? (B) : (int)((n + 256 * (U) - 1) / (256 * (U)));                                 \
    const Ptrs p = ptrs(SKEW);                                                                                   \
    double ms = run<NR, NW, NTL, NTS, U, ORDER>(p, n, sink, blocks);                                             \
    printf("rd %d wr %d ntl=%d nts=%d U=%d order=%d blocks=%7d size=%5zu MiB skew=%d : %8.3f ms %7.1f GB/s\n", NR, NW, NTL, NTS, U,  \
           ORDER, blocks, (size_t)(MIB), SKEW, ms, (NR + NW) * (double)bytes / ms / 1e6);                        \
    fflush(stdout);                                                                                              \
  }
  const size_t M = mib_max;
  printf("== 1:1 copy, grid sweep (grid-stride), U = 4\n");
  for (int B : {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) RUN(1, 1, 0, 0, 4, 0, B, M, 0)
  printf("== 1:1 copy, one-shot grid, unroll sweep\n");
  RUN(1, 1, 0, 0, 1, 0, 0, M, 0) RUN(1, 1, 0, 0, 2, 0, 0, M, 0) RUN(1, 1, 0, 0, 8, 0, 0, M, 0)
  printf("== 1:1 copy, non-temporal variants (one-shot U = 4, and 2048 workgroups)\n");
  RUN(1, 1, 1, 0, 4, 0, 0, M, 0) RUN(1, 1, 0, 1, 4, 0, 0, M, 0) RUN(1, 1, 1, 1, 4, 0, 0, M, 0)
  RUN(1, 1, 1, 0, 4, 0, 2048, M, 0) RUN(1, 1, 0, 1, 4, 0, 2048, M, 0) RUN(1, 1, 1, 1, 4, 0, 2048, M, 0)
  printf("== 1:1 copy, contiguous chunk per workgroup\n");
  for (int B : {512, 1024, 2048, 8192}) RUN(1, 1, 0, 0, 4, 1, B, M, 0)
  RUN(1, 1, 1, 1, 4, 1, 2048, M, 0)
  printf("== 1:1 copy, stream size (one-shot U = 4 and 2048 workgroups)\n");
  for (size_t mib : {(size_t)256, (size_t)512, M}) { RUN(1, 1, 0, 0, 4, 0, 0, mib, 0) RUN(1, 1, 0, 0, 4, 0, 2048, mib, 0) }
  printf("== 1:1 copy, streams skewed by 3 pages each\n");
  RUN(1, 1, 0, 0, 4, 0, 0, M, 1) RUN(1, 1, 0, 0, 4, 0, 2048, M, 1) RUN(1, 1, 1, 1, 4, 0, 2048, M, 1)
  printf("== read-only / write-only\n");
  RUN(1, 0, 0, 0, 4, 0, 0, M, 0) RUN(1, 0, 0, 0, 4, 0, 2048, M, 0) RUN(1, 0, 1, 0, 4, 0, 2048, M, 0) RUN(1, 0, 0, 0, 8, 0, 2048, M, 0)
  RUN(0, 1, 0, 0, 4, 0, 0, M, 0) RUN(0, 1, 0, 0, 4, 0, 2048, M, 0) RUN(0, 1, 0, 1, 4, 0, 2048, M, 0)
  printf("== mixes of the hot-path passes (2048 workgroups and one-shot; plain and nt stores)\n");
  RUN(2, 1, 0, 0, 4, 0, 0, M, 0) RUN(2, 1, 0, 0, 4, 0, 2048, M, 0) RUN(2, 1, 0, 1, 4, 0, 2048, M, 0)
  RUN(3, 1, 0, 0, 2, 0, 0, M, 0) RUN(3, 1, 0, 0, 2, 0, 2048, M, 0) RUN(3, 1, 0, 1, 2, 0, 2048, M, 0)
  RUN(5, 1, 0, 0, 2, 0, 0, M, 0) RUN(5, 1, 0, 0, 2, 0, 2048, M, 0) RUN(5, 1, 0, 1, 2, 0, 2048, M, 0) RUN(5, 1, 0, 0, 2, 0, 512, M, 0)
  RUN(5, 1, 1, 1, 2, 0, 2048, M, 1)
  RUN(7, 1, 0, 0, 1, 0, 0, M, 0) RUN(7, 1, 0, 0, 1, 0, 2048, M, 0)
  RUN(3, 3, 0, 0, 2, 0, 0, M, 0) RUN(3, 3, 0, 0, 2, 0, 2048, M, 0)
  RUN(2, 3, 0, 0, 2, 0, 2048, M, 0) RUN(2, 3, 0, 1, 2, 0, 2048, M, 0)
  return 0;
}
