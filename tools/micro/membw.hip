// Streaming-bandwidth ceilings of one MI355X for the read/write mixes of the hot-path kernels (calibration for the
// roofline fractions in DESIGN.md): NR read streams + NW write streams of `bytes` each, 16 B per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/membw.hip -o tools/micro/membw ; run: tools/micro/membw [MiB=2048]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double D2 __attribute__((ext_vector_type(2)));

template <int NR, int NW, bool NT, int U>
__global__ __launch_bounds__(256) void stream_kernel(const D2* const* __restrict__ rd, D2* const* __restrict__ wr,
                                                     size_t n, double* sink) {
  const D2* r[NR > 0 ? NR : 1];
  D2* w[NW > 0 ? NW : 1];
#pragma unroll
  for (int i = 0; i < NR; ++i) r[i] = rd[i];
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = wr[i];
  D2 acc = {0.0, 0.0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t base = (size_t)blockIdx.x * 256 + threadIdx.x; base < n; base += stride * U) {
    D2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * stride < n ? base + u * stride : n - 1;
      D2 s = {1.0, 2.0};
#pragma unroll
      for (int k = 0; k < NR; ++k) s += r[k][i];
      v[u] = s;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * stride < n ? base + u * stride : n - 1;
      if (NW == 0) acc += v[u];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        if (NT) __builtin_nontemporal_store(v[u], &w[k][i]);
        else w[k][i] = v[u];
      }
    }
  }
  if (NW == 0 && acc.x == 123.456) *sink = acc.y;
}

template <int NR, int NW, bool NT, int U>
double run(const D2* const* rd, D2* const* wr, size_t n, double* sink, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 8; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<NR, NW, NT, U>), dim3(blocks), dim3(256), 0, 0, rd, wr, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? atol(argv[1]) : 2048;
  const size_t bytes = mib << 20, n = bytes / 16;
  D2* buf[6];
  for (auto& b : buf) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
  const D2** d_rd; D2** d_wr; double* sink;
  hipMalloc(&d_rd, 3 * sizeof(D2*));
  hipMalloc(&d_wr, 3 * sizeof(D2*));
  hipMalloc(&sink, 8);
  hipMemcpy(d_rd, buf, 3 * sizeof(D2*), hipMemcpyHostToDevice);
  hipMemcpy(d_wr, buf + 3, 3 * sizeof(D2*), hipMemcpyHostToDevice);
  printf("stream size %zu MiB each; GB/s = (NR + NW) * size / time\n", mib);
#define RUN(NR, NW, NT, U, B)                                                                      \
  {                                                                                                \
    double ms = run<NR, NW, NT, U>(d_rd, d_wr, n, sink, B);                                        \
    printf("read %d write %d  nt=%d unroll=%d blocks=%6d : %8.3f ms  %7.1f GB/s\n", NR, NW, (int)NT, U, B, ms, \
           (NR + NW) * (double)bytes / ms / 1e6);                                                  \
  }
  for (int B : {2048, 8192, 65536}) {
    RUN(1, 0, false, 4, B)
    RUN(0, 1, false, 4, B)
    RUN(0, 1, true, 4, B)
    RUN(1, 1, false, 4, B)
    RUN(1, 1, true, 4, B)
    RUN(2, 1, false, 4, B)
    RUN(2, 1, true, 4, B)
    RUN(3, 3, false, 2, B)
    RUN(3, 3, true, 2, B)
    RUN(1, 1, false, 1, B)
    RUN(1, 1, false, 8, B)
  }
  return 0;
}
