// Learned-Jacobi vertex model, inference: out[i] = W3 relu(W2 relu(W1 x_i + b1) + b2) + b3 for every vertex i.
// The reference runs torch.nn.Sequential(Linear(5,50), ReLU, Linear(50,20), ReLU, Linear(20,1)) over all vertices
// (TrainableJacobiDiag/TrainableJacobiGNN.py:86-91, :110-111), i.e. three GEMMs with [n,50] and [n,20]
// intermediates in memory.  Here one lane owns one vertex and the hidden layers never leave the register file:
// the first two layers are fused (h1_j is consumed by the second layer's accumulators as soon as it is formed) and
// the weights are SGPR operands fetched through the scalar cache.  The kernel streams 4*(IN+1) bytes per vertex, so
// it is bound by the VALU rate (2*(IN*H1 + H1*H2 + H2) flops per vertex), not by HBM.
#include "gla_internal.h"

namespace {

constexpr int MLP_IN_MAX = 8;       // inputs per vertex (zero padded)
constexpr int MLP_H2_MAX = 32;      // second hidden width (zero padded; accumulators are registers)
constexpr int MLP_H1_MAX = 512;     // first hidden width

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Packed weights (in units of T): W1p [H1][IN_MAX] (zero padded) | b1 [H1] | W2t [H1][H2_MAX] (transposed, zero
// padded) | b2 [H2_MAX] | W3 [H2_MAX] | b3 [1].  Every index into this block is wave-uniform in the main kernel, so
// the weights arrive through the scalar cache (s_load) and feed the FMAs as SGPR operands: no LDS, no vector loads.
// (A first version staged them in LDS and read them as 16-byte broadcasts: 1.1 ms at n = 4.2 M, LDS-pipe bound.)
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void mlp_pack_kernel(int in_dim, int h1, int h2, const T* __restrict__ W1,
                                                            const T* __restrict__ b1, const T* __restrict__ W2,
                                                            const T* __restrict__ b2, const T* __restrict__ W3,
                                                            const T* __restrict__ b3, T* __restrict__ wk) {
  T* sW1 = wk;
  T* sb1 = sW1 + (size_t)h1 * MLP_IN_MAX;
  T* sW2 = sb1 + h1;
  T* sb2 = sW2 + (size_t)h1 * MLP_H2_MAX;
  T* sW3 = sb2 + MLP_H2_MAX;
  T* sb3 = sW3 + MLP_H2_MAX;
  const int tid = blockIdx.x * GLA_BLOCK + threadIdx.x, nt = gridDim.x * GLA_BLOCK;
  for (int i = tid; i < h1 * MLP_IN_MAX; i += nt) {
    const int j = i / MLP_IN_MAX, c = i % MLP_IN_MAX;
    sW1[i] = c < in_dim ? W1[(size_t)j * in_dim + c] : T(0);
  }
  for (int i = tid; i < h1; i += nt) sb1[i] = b1[i];
  for (int i = tid; i < h1 * MLP_H2_MAX; i += nt) {
    const int j = i / MLP_H2_MAX, k = i % MLP_H2_MAX;
    sW2[i] = k < h2 ? W2[(size_t)k * h1 + j] : T(0);                 // transposed: [j][k]
  }
  for (int i = tid; i < MLP_H2_MAX; i += nt) {
    sb2[i] = i < h2 ? b2[i] : T(0);
    sW3[i] = i < h2 ? W3[i] : T(0);
  }
  if (tid == 0) sb3[0] = b3[0];
}

template <typename T, int H2P>        // H2P: second hidden width rounded up to a multiple of 8 (register accumulators)
__global__ __launch_bounds__(GLA_BLOCK) void mlp_relu3_kernel(int64_t n, int in_dim, int h1, const T* __restrict__ X,
                                                             int64_t ldx, const T* __restrict__ wk,
                                                             T* __restrict__ out) {
  const T* __restrict__ sW1 = wk;
  const T* __restrict__ sb1 = sW1 + (size_t)h1 * MLP_IN_MAX;
  const T* __restrict__ sW2 = sb1 + h1;
  const T* __restrict__ sb2 = sW2 + (size_t)h1 * MLP_H2_MAX;
  const T* __restrict__ sW3 = sb2 + MLP_H2_MAX;
  const T* __restrict__ sb3 = sW3 + MLP_H2_MAX;
  for (int64_t r = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * GLA_BLOCK) {
    T x[MLP_IN_MAX];
#pragma unroll
    for (int c = 0; c < MLP_IN_MAX; ++c) {
      const T v = X[r * ldx + min(c, in_dim - 1)];           // every lane issues every load (no exec-mask branches)
      x[c] = c < in_dim ? v : T(0);
    }
    T acc[H2P];
#pragma unroll
    for (int k = 0; k < H2P; ++k) acc[k] = sb2[k];
    for (int j = 0; j < h1; ++j) {
      T s = sb1[j];
#pragma unroll
      for (int c = 0; c < MLP_IN_MAX; ++c) s = fma_t(sW1[j * MLP_IN_MAX + c], x[c], s);
      s = s > T(0) ? s : T(0);
#pragma unroll
      for (int k = 0; k < H2P; ++k) acc[k] = fma_t(sW2[j * MLP_H2_MAX + k], s, acc[k]);
    }
    T o = sb3[0];
#pragma unroll
    for (int k = 0; k < H2P; ++k) o = fma_t(sW3[k], acc[k] > T(0) ? acc[k] : T(0), o);
    out[r] = o;
  }
}

template <typename T>
int launch_mlp(int64_t n, int in_dim, int h1, int h2, const void* X, int64_t ldx, const void* W1, const void* b1,
               const void* W2, const void* b2, const void* W3, const void* b3, void* out, void* work, hipStream_t st) {
  hipLaunchKernelGGL((mlp_pack_kernel<T>), dim3(8), dim3(GLA_BLOCK), 0, st, in_dim, h1, h2, (const T*)W1,
                     (const T*)b1, (const T*)W2, (const T*)b2, (const T*)W3, (const T*)b3, (T*)work);
  GLA_LAUNCH_CHECK();
  int64_t blocks = (n + GLA_BLOCK - 1) / GLA_BLOCK;
  if (blocks > 256 * 8) blocks = 256 * 8;
#define GLA_MLP(H2P_)                                                                                         \
  hipLaunchKernelGGL((mlp_relu3_kernel<T, H2P_>), dim3((unsigned)blocks), dim3(GLA_BLOCK), 0, st, n, in_dim, h1, \
                     (const T*)X, ldx, (const T*)work, (T*)out)
  if (h2 <= 8) GLA_MLP(8);
  else if (h2 <= 16) GLA_MLP(16);
  else if (h2 <= 24) GLA_MLP(24);
  else GLA_MLP(32);
#undef GLA_MLP
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

}  // namespace

extern "C" int64_t gla_mlp_work_bytes(int h1) {
  if (h1 < 1 || h1 > MLP_H1_MAX) return 0;
  return (int64_t)((size_t)h1 * (MLP_IN_MAX + 1 + MLP_H2_MAX) + 2 * MLP_H2_MAX + 1) * (int64_t)sizeof(double);
}

extern "C" int gla_mlp_relu3(int dtype, int64_t n, int in_dim, int h1, int h2, const void* X, int64_t ldx,
                             const void* W1, const void* b1, const void* W2, const void* b2, const void* W3,
                             const void* b3, void* out, void* work, gla_stream_t stream) {
  if ((dtype != GLA_F32 && dtype != GLA_F64) || n < 0 || in_dim < 1 || in_dim > MLP_IN_MAX || h1 < 1 ||
      h1 > MLP_H1_MAX || h2 < 1 || h2 > MLP_H2_MAX || ldx < in_dim)
    return GLA_ERR_INVALID_ARG;
  if (n == 0) return GLA_OK;
  if (!X || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out || !work) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return launch_mlp<double>(n, in_dim, h1, h2, X, ldx, W1, b1, W2, b2, W3, b3, out, work, (hipStream_t)stream);
  return launch_mlp<float>(n, in_dim, h1, h2, X, ldx, W1, b1, W2, b2, W3, b3, out, work, (hipStream_t)stream);
}
