// Issue rates of the vector instructions the march kernels are made of (gfx950): waves x instructions / time, per SIMD, against
// v_fma_f64 -- are v_mul_f64 / v_add_f64 full rate?  what do 64-bit integer adds, compares and selects cost?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/valu_rates ; run: tools/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int ITER = 32768, UNR = 16;

#define KERNEL(NAME, DECL, BODY, SINK)                                                   \
  __global__ __launch_bounds__(256) void NAME(double* out, double a, double b, int n) { \
    DECL;                                                                                \
    const long long t0 = clock64();                                                      \
    for (int i = 0; i < n; ++i) {                                                        \
      _Pragma("unroll") for (int u = 0; u < UNR; ++u) { BODY; }                          \
    }                                                                                    \
    const long long t1 = clock64();                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[256] = (double)(t1 - t0);               \
    SINK;                                                                                \
  }

#define DECL_D double r[8]; for (int k = 0; k < 8; ++k) r[k] = a + threadIdx.x * 1e-9 + k
#define SINK_D double s = 0; for (int k = 0; k < 8; ++k) s += r[k]; if (s == 12345.678) out[threadIdx.x] = s

KERNEL(k_fma, DECL_D, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[u & 7]) : "v"(a), "v"(b)), SINK_D)
KERNEL(k_mul, DECL_D, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a)), SINK_D)
KERNEL(k_add, DECL_D, asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[u & 7]) : "v"(b)), SINK_D)
KERNEL(k_fma_dep, DECL_D, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b)), SINK_D)
KERNEL(k_fma2, DECL_D, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[u & 1]) : "v"(a), "v"(b)), SINK_D)
KERNEL(k_cmp, DECL_D, asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(r[u & 7]), "v"(b) : "vcc"), SINK_D)
#define DECL_I unsigned r[8]; for (int k = 0; k < 8; ++k) r[k] = threadIdx.x + k
#define SINK_I unsigned s = 0; for (int k = 0; k < 8; ++k) s += r[k]; if (s == 12345678u) out[threadIdx.x] = s
KERNEL(k_iadd, DECL_I, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(n)), SINK_I)
KERNEL(k_cnd, DECL_I, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[u & 7]) : "v"(n) : "vcc"), SINK_I)
KERNEL(k_mullo, DECL_I, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(n)), SINK_I)
#define DECL_L unsigned long long r[8]; for (int k = 0; k < 8; ++k) r[k] = threadIdx.x + k
#define SINK_L unsigned long long s = 0; for (int k = 0; k < 8; ++k) s += r[k]; if (s == 12345678ull) out[threadIdx.x] = (double)s
KERNEL(k_add64, DECL_L, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[u & 7]) : "v"((unsigned long long)n)), SINK_L)
KERNEL(k_mov64, DECL_L, asm volatile("v_mov_b64 %0, %1" : "+v"(r[u & 7]) : "v"((unsigned long long)n)), SINK_L)

int main() {
  double* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct K { const char* name; void (*fn)(double*, double, double, int); };
  K ks[] = {{"v_fma_f64 (8 independent)", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_fma_f64 (1 dependent chain)", k_fma_dep},
            {"v_fma_f64 (2 chains)", k_fma2}, {"v_cmp_gt_f64", k_cmp}, {"v_add_u32", k_iadd}, {"v_cndmask_b32", k_cnd},
            {"v_mul_lo_u32", k_mullo}, {"v_lshl_add_u64", k_add64}, {"v_mov_b64", k_mov64}};
  for (int wps : {1, 2}) {                     // waves per SIMD: 256 CUs x 4 SIMDs x wps waves = blocks of 256 threads: 256 x wps
    printf("---- %d wave(s) per SIMD (%d blocks of 4 waves) ----\n", wps, 256 * wps);
    for (auto& k : ks) {
      std::vector<float> ts;
      for (int rep = 0; rep < 7; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(256 * wps), dim3(256), 0, 0, out, 1.0000001, 1e-9, ITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep >= 2) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      const double us = ts[ts.size() / 2] * 1e3;
      const double per_wave = (double)ITER * UNR;                         // instructions of one wave
      // cycles per instruction and SIMD at an assumed 2.4 GHz: time x 2.4e3 / (instructions a SIMD issued)
      double cyc = 0; hipMemcpy(&cyc, out + 256, 8, hipMemcpyDeviceToHost);
      printf("%-32s %9.1f us   s_memtime ticks %.3g = %.1f MHz;  %.2f ns per instruction and SIMD = %.2f cycles at 2.4 GHz, %.2f at 2.0\n", k.name, us,
             cyc, cyc / us, us * 1e3 / (per_wave * wps), us * 2.4e3 / (per_wave * wps), us * 2.0e3 / (per_wave * wps));
    }
  }
  return 0;
}
