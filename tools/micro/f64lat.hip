// Issue / dependent-latency figures of the fp64 VALU operations the Jacobi march spends its time on (gfx950), per wave and
// with 1, 2, 4 waves per SIMD: a chain of N dependent operations vs N independent ones.  cycles = s_memtime ticks (100 MHz
// constant clock on this part: converted with the wall clock of the launch instead) -- reported as ns per operation per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/f64lat.hip -o tools/micro/f64lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (MODE == 0) { x0 = __builtin_fma(x0, b, a); }                                       // dependent fma chain
      if constexpr (MODE == 1) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
      if constexpr (MODE == 2) { x0 = x0 * b; x0 = x0 + a; }                                           // dependent mul, add
      if constexpr (MODE == 3) { x0 = a / x0; }                                                        // dependent IEEE division
      if constexpr (MODE == 4) { x0 = a / x0; x1 = a / x1; x2 = a / x2; x3 = a / x3; }
      if constexpr (MODE == 5) { x0 = __builtin_amdgcn_rcp(x0); }
      if constexpr (MODE == 6) { x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int MODE>
void run(const char* name, int ops_per_u, double* out) {
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0000001, 0.9999999);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double ops = (double)iters * 16 * ops_per_u;        // per wave
    printf("%-34s waves/SIMD=%d : %7.2f ns per op per wave  (%6.2f ns per op per SIMD)\n", name, wps, best * 1e6 / ops, best * 1e6 / ops / wps);
  }
}

int main() {
  double* out;
  CK(hipMalloc(&out, 256 * 4 * 256 * 8));
  run<0>("fma f64, dependent chain", 1, out);
  run<1>("fma f64, 4 independent chains", 4, out);
  run<2>("mul+add f64, dependent", 2, out);
  run<3>("IEEE div f64, dependent", 1, out);
  run<4>("IEEE div f64, 4 independent", 4, out);
  run<5>("v_rcp_f64, dependent", 1, out);
  run<6>("v_rcp_f64, 4 independent", 4, out);
  return 0;
}
