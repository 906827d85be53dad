// Is the march pattern's cost just its 8-byte-per-lane loads?  One-shot streaming of 5 read arrays + 1 write array (the pass's mix),
// with 8 bytes and with 16 bytes per lane and request, plain grids.  hipcc --offload-arch=gfx950 -O3 tools/micro/stream8.hip -o tools/micro/stream8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <typename V, int UNR>
__global__ __launch_bounds__(256) void mix(const V* a0, const V* a1, const V* a2, const V* a3, const V* a4, V* o, size_t n) {
  size_t i = ((size_t)blockIdx.x * UNR) * 256 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < UNR; ++u, i += 256)
    if (i < n) o[i] = ((a0[i] + a1[i]) + (a2[i] + a3[i])) + a4[i];
}
int main() {
  const size_t N = 4096, n = N * N;
  double* b[6];
  for (int k = 0; k < 6; ++k) { hipMalloc(&b[k], n * 8); hipMemset(b[k], 0, n * 8); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < 15; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double us = ts[7] * 1e3;
    printf("%-44s %7.1f us  %6.0f GB/s on 48 B per row\n", name, us, 48.0 * n / us / 1e3);
  };
  for (int rnd = 0; rnd < 2; ++rnd) {
    timeit("8 B per lane, 1 element per thread", [&] { hipLaunchKernelGGL((mix<double, 1>), dim3(n / 256), dim3(256), 0, 0, b[0], b[1], b[2], b[3], b[4], b[5], n); });
    timeit("8 B per lane, 4 elements per thread", [&] { hipLaunchKernelGGL((mix<double, 4>), dim3(n / 1024), dim3(256), 0, 0, b[0], b[1], b[2], b[3], b[4], b[5], n); });
    timeit("8 B per lane, 16 elements per thread", [&] { hipLaunchKernelGGL((mix<double, 16>), dim3(n / 4096), dim3(256), 0, 0, b[0], b[1], b[2], b[3], b[4], b[5], n); });
    typedef double d2 __attribute__((ext_vector_type(2)));
    timeit("16 B per lane, 1 vector per thread", [&] { hipLaunchKernelGGL((mix<d2, 1>), dim3(n / 512), dim3(256), 0, 0, (const d2*)b[0], (const d2*)b[1], (const d2*)b[2], (const d2*)b[3], (const d2*)b[4], (d2*)b[5], n / 2); });
    timeit("16 B per lane, 4 vectors per thread", [&] { hipLaunchKernelGGL((mix<d2, 4>), dim3(n / 2048), dim3(256), 0, 0, (const d2*)b[0], (const d2*)b[1], (const d2*)b[2], (const d2*)b[3], (const d2*)b[4], (d2*)b[5], n / 2); });
  }
  return 0;
}
