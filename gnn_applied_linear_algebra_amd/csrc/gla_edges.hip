// libgla_hip.so -- edge-wise and aggregation kernels (outputs in EDGE order or per vertex), the strength-of-
// connection / interpolation kernels of the AMG setup, column reductions (gfx950 / MI355X only; wave = 64 lanes).
//
// Same tile structure as gla_rows.hip where a row tile is staged through LDS (SOC, DI, SA-SOC); plain
// lane-per-(row, column) or lane-group kernels otherwise.  Every kernel that sums does so in a fixed order.
#include "gla_device.h"

namespace {

// ------------------------------------------------------------------------------------------------
// edge-wise kernels (outputs in EDGE order)
// ------------------------------------------------------------------------------------------------
template <typename T, int KC>          // KC > 0: compile-time K (division becomes a shift), 0: runtime K
__global__ __launch_bounds__(GLA_BLOCK) void edge_products_kernel(
    const int32_t* __restrict__ col, const T* __restrict__ val, const int32_t* __restrict__ perm,
    const T* __restrict__ X, int64_t ldx, T* __restrict__ C, int64_t ldc, int64_t nnz, int Krt) {
  const int K = KC > 0 ? KC : Krt;
  const int64_t total = nnz * K;
  // 4 independent (col -> X) chains per lane and iteration; indices clamped so the batch is branch-free
  for (int64_t base = (int64_t)blockIdx.x * (GLA_BLOCK * 4) + threadIdx.x; base < total;
       base += (int64_t)gridDim.x * (GLA_BLOCK * 4)) {
    int64_t e[4];
    int k[4];
    T v[4];
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t idx = min(base + (int64_t)u * GLA_BLOCK, total - 1);
      const int64_t p = idx / K;
      k[u] = (int)(idx - p * K);
      e[u] = perm ? (int64_t)perm[p] : p;
      v[u] = val[p];
      c[u] = col[p];
    }
    T xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = X[(int64_t)c[u] * ldx + k[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) C[e[u] * ldc + k[u]] = v[u] * xv[u];
  }
}

template <typename T>
__device__ __forceinline__ T nan_max(T m, T a) { return (a > m || a != a) ? a : m; }

// v_i = max_j(-A_ij); S = relu(-A/v - theta).  SOCClassicGNN.py:50-129
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void soc_classic_kernel(
    const int32_t* __restrict__ rowptr, const T* __restrict__ val, const int32_t* __restrict__ perm,
    TileMap tm, T theta, T* __restrict__ v_out, T* __restrict__ S_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sv = reinterpret_cast<T*>(smem_raw);
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, len = ti.w - ti.z;
  if (r0 >= r1) return;
  const int tid = threadIdx.x;
  if (len > tm.lds_cap) {                      // CSR-adaptive long bin: one row, whole workgroup (max is order-free)
    const ChunkRange cr = chunk_of(s, s + len, tid, GLA_BLOCK);
    T m = T(-1) * val[s];
    for (int j = cr.a; j < cr.b; ++j) m = nan_max(m, T(-1) * val[j]);
    m = block_tree(m, sv, 1, [](T a, T b) { return nan_max(a, b); });
    if (tid == 0) v_out[r0] = m;
    for (int j = s + tid; j < s + len; j += GLA_BLOCK) {
      const T q = (T(-1) * val[j]) / m - theta;
      S_out[perm ? (int64_t)perm[j] : (int64_t)j] = (q != q) ? q : (q > T(0) ? q : T(0));
    }
    return;
  }
  {
    T v_[4];
#define SOC_LD(u, i) v_[u] = val[s + (i)]
#define SOC_ST(u, i) sv[i] = v_[u]
    GLA_TILE_LOOP(len, tid, SOC_LD, SOC_ST)
#undef SOC_LD
#undef SOC_ST
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += GLA_BLOCK) {
    const int a = rowptr[r] - s, b = rowptr[r + 1] - s;
    T m = T(0);                                   // torch_scatter: empty segment -> 0
    if (a < b) {
      m = T(-1) * sv[a];
      for (int j = a + 1; j < b; ++j) m = nan_max(m, T(-1) * sv[j]);
    }
    v_out[r] = m;
    for (int j = a; j < b; ++j) {
      const T q = (T(-1) * sv[j]) / m - theta;
      sv[j] = (q != q) ? q : (q > T(0) ? q : T(0));   // relu keeps NaN, like torch.nn.ReLU
    }
  }
  __syncthreads();
  {
    int64_t e_[4];
#define SOC_LD(u, i) e_[u] = perm ? (int64_t)perm[s + (i)] : (int64_t)(s + (i))
#define SOC_ST(u, i) S_out[e_[u]] = sv[i]
    GLA_TILE_LOOP(len, tid, SOC_LD, SOC_ST)
#undef SOC_LD
#undef SOC_ST
  }
}

// DirectInterpGNN.py:50-152
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void direct_interp_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const int32_t* __restrict__ perm, TileMap tm, int lds_elems, const T* __restrict__ diag,
    const T* __restrict__ cf, const T* __restrict__ S, T* __restrict__ w_out, T* __restrict__ alpha_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sa = reinterpret_cast<T*>(smem_raw);         // A_ik, later w_ik
  T* sd = sa + lds_elems;                         // A_ik * S_ik * C_k
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, len = ti.w - ti.z;
  if (r0 >= r1) return;
  const int tid = threadIdx.x;
  if (len > tm.lds_cap) {                      // CSR-adaptive long bin: one row, fixed-chunk sums by the whole workgroup
    const ChunkRange cr = chunk_of(s, s + len, tid, GLA_BLOCK);
    T num = T(0), den = T(0);
    for (int j = cr.a; j < cr.b; ++j) {
      const T a = val[j];
      num = num + a;
      den = den + (a * S[perm ? (int64_t)perm[j] : (int64_t)j]) * cf[col[j]];
    }
    num = block_tree(num, sa, 1, [](T a, T b) { return a + b; });
    den = block_tree(den, sa, 1, [](T a, T b) { return a + b; });
    const T gamma = num / den;
    const T alpha = (T(1) / diag[r0]) * gamma;
    if (alpha_out && tid == 0) alpha_out[r0] = alpha;
    const T omc = T(1) - cf[r0];
    for (int j = s + tid; j < s + len; j += GLA_BLOCK)
      w_out[perm ? (int64_t)perm[j] : (int64_t)j] = omc * ((T(-1) * val[j]) * alpha);
    return;
  }
  {
    T a_[4], s_[4], c_[4];
#define DI_LD(u, i)                                                                  \
  {                                                                                  \
    const int64_t e = perm ? (int64_t)perm[s + (i)] : (int64_t)(s + (i));            \
    a_[u] = val[s + (i)];                                                            \
    s_[u] = S[e];                                                                    \
    c_[u] = cf[col[s + (i)]];                                                        \
  }
#define DI_ST(u, i) { sa[i] = a_[u]; sd[i] = (a_[u] * s_[u]) * c_[u]; }
    GLA_TILE_LOOP(len, tid, DI_LD, DI_ST)
#undef DI_LD
#undef DI_ST
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += GLA_BLOCK) {
    const int a = rowptr[r] - s, b = rowptr[r + 1] - s;
    T num = T(0), den = T(0);
    for (int j = a; j < b; ++j) { num = num + sa[j]; den = den + sd[j]; }
    const T gamma = num / den;
    const T alpha = (T(1) / diag[r]) * gamma;
    if (alpha_out) alpha_out[r] = alpha;
    const T omc = T(1) - cf[r];
    for (int j = a; j < b; ++j) sa[j] = omc * ((T(-1) * sa[j]) * alpha);
  }
  __syncthreads();
  {
    int64_t e_[4];
#define DI_LD(u, i) e_[u] = perm ? (int64_t)perm[s + (i)] : (int64_t)(s + (i))
#define DI_ST(u, i) w_out[e_[u]] = sa[i]
    GLA_TILE_LOOP(len, tid, DI_LD, DI_ST)
#undef DI_LD
#undef DI_ST
  }
}

// SOCSAGNN.py:49-71: S = (A*A)/(A_ii*A_jj); the row endpoint is known per row-lane, the column endpoint
// is gathered while streaming
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void soc_sa_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const int32_t* __restrict__ perm, TileMap tm, int lds_elems, const T* __restrict__ diag,
    T* __restrict__ S_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sa = reinterpret_cast<T*>(smem_raw);         // A_ij^2, later S_ij
  T* sd = sa + lds_elems;                         // A_jj
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, len = ti.w - ti.z;
  if (r0 >= r1) return;
  const int tid = threadIdx.x;
  if (len > tm.lds_cap) {                      // CSR-adaptive long bin: one row, purely edge-wise
    const T di = diag[r0];
    for (int j = s + tid; j < s + len; j += GLA_BLOCK) {
      const T a = val[j];
      S_out[perm ? (int64_t)perm[j] : (int64_t)j] = (a * a) / (di * diag[col[j]]);
    }
    return;
  }
  {
    T a_[4], d_[4];
#define SA_LD(u, i) { a_[u] = val[s + (i)]; d_[u] = diag[col[s + (i)]]; }
#define SA_ST(u, i) { sa[i] = a_[u] * a_[u]; sd[i] = d_[u]; }
    GLA_TILE_LOOP(len, tid, SA_LD, SA_ST)
#undef SA_LD
#undef SA_ST
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += GLA_BLOCK) {
    const int a = rowptr[r] - s, b = rowptr[r + 1] - s;
    const T di = diag[r];
    for (int j = a; j < b; ++j) sa[j] = sa[j] / (di * sd[j]);
  }
  __syncthreads();
  {
    int64_t e_[4];
#define SA_LD(u, i) e_[u] = perm ? (int64_t)perm[s + (i)] : (int64_t)(s + (i))
#define SA_ST(u, i) S_out[e_[u]] = sa[i]
    GLA_TILE_LOOP(len, tid, SA_LD, SA_ST)
#undef SA_LD
#undef SA_ST
  }
}

// ------------------------------------------------------------------------------------------------
// generic edge -> vertex aggregation: the torch_scatter.scatter(src, edgeij_pair[0], dim=0, dim_size=n, reduce=...)
// primitive itself, for callers that keep their own EdgeUpdate / VertexUpdate modules.  Lanes are
// (row, column) pairs; a row's entries are visited in EDGE order, so `sum` equals the reference's sequential
// scatter-add bit for bit.  Empty rows give 0 for every op (torch_scatter semantics); max / min propagate NaN.
// ------------------------------------------------------------------------------------------------
template <typename T, int KP>
__global__ __launch_bounds__(GLA_BLOCK) void segment_reduce_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, const T* __restrict__ src, int64_t lds_,
    int K, int op, int64_t n_rows, T* __restrict__ out, int64_t ldo, const int32_t* __restrict__ long_ids,
    int n_long) {
  constexpr int RPP = GLA_BLOCK / KP;
  const int k0 = threadIdx.x % KP, rsub = threadIdx.x / KP;
  const int nb = (int)gridDim.x - n_long;        // the last n_long workgroups take one long row each
  if ((int)blockIdx.x >= nb) {
    __shared__ T sh[GLA_BLOCK];
    const int64_t r = long_ids[(int)blockIdx.x - nb];
    const int a = rowptr[r], b = rowptr[r + 1];
    const ChunkRange cr = chunk_of(a, b, threadIdx.x / KP, RPP);
    for (int kb = 0; kb < K; kb += KP) {
      const bool live = kb + k0 < K;
      const int k = live ? kb + k0 : K - 1;
      T acc = (op == 0 || op == 3) ? T(0) : src[(perm ? (int64_t)perm[a] : (int64_t)a) * lds_ + k];
      for (int p = cr.a; p < cr.b; ++p) {
        const T v = src[(perm ? (int64_t)perm[p] : (int64_t)p) * lds_ + k];
        if (op == 0 || op == 3) acc = acc + v;
        else if (op == 1) acc = (v > acc || v != v) ? v : acc;
        else acc = (v < acc || v != v) ? v : acc;
      }
      if (op == 0 || op == 3) acc = block_tree(acc, sh, KP, [](T x, T y) { return x + y; });
      else if (op == 1) acc = block_tree(acc, sh, KP, [](T x, T y) { return (y > x || y != y) ? y : x; });
      else acc = block_tree(acc, sh, KP, [](T x, T y) { return (y < x || y != y) ? y : x; });
      if (op == 3) acc = acc / T(b - a);
      if (threadIdx.x < KP && live) out[r * ldo + k] = acc;
    }
    return;
  }
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_rows; r += (int64_t)nb * RPP) {
    const int a = rowptr[r], b = rowptr[r + 1];
    if (n_long > 0 && b - a > GLA_LONG_ROW) continue;      // summed by its own workgroup above
    for (int k = k0; k < K; k += KP) {
      T acc = T(0);
      for (int p = a; p < b; ++p) {
        const int64_t e = perm ? (int64_t)perm[p] : (int64_t)p;
        const T v = src[e * lds_ + k];
        if (op == 0 || op == 3) acc = acc + v;
        else if (p == a) acc = v;
        else if (op == 1) acc = (v > acc || v != v) ? v : acc;
        else acc = (v < acc || v != v) ? v : acc;
      }
      if (op == 3 && b > a) acc = acc / T(b - a);
      out[r * ldo + k] = acc;
    }
  }
}

// The four row statistics of TrainableJacobiGNN.py:53-70 in one pass over src: out[r, :] = cat(min, mean, sum, max)
// (each K wide).  Same visiting order and empty-row / NaN rules as segment_reduce_kernel.
template <typename T, int KP>
__global__ __launch_bounds__(GLA_BLOCK) void segment_stats_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, const T* __restrict__ src, int64_t lds_,
    int K, int64_t n_rows, T* __restrict__ out, int64_t ldo) {
  constexpr int RPP = GLA_BLOCK / KP;
  const int k0 = threadIdx.x % KP, rsub = threadIdx.x / KP;
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_rows; r += (int64_t)gridDim.x * RPP) {
    const int a = rowptr[r], b = rowptr[r + 1];
    for (int k = k0; k < K; k += KP) {
      T sum = T(0), mn = T(0), mx = T(0);
      for (int p = a; p < b; ++p) {
        const int64_t e = perm ? (int64_t)perm[p] : (int64_t)p;
        const T v = src[e * lds_ + k];
        sum = sum + v;
        if (p == a) { mn = v; mx = v; }
        else {
          mx = (v > mx || v != v) ? v : mx;
          mn = (v < mn || v != v) ? v : mn;
        }
      }
      T* o = out + r * ldo + k;
      o[0] = mn;
      o[K] = b > a ? sum / T(b - a) : sum;
      o[2 * (int64_t)K] = sum;
      o[3 * (int64_t)K] = mx;
    }
  }
}

// Edge-wise dot products (SDDMM on the plan's pattern): out[e] = sum_k G[row(e), k] * X[col(e), k] -- the gradient of
// Y = A X with respect to the stored values (loss.py:92-128 differentiates through `A @ Y`).  KP lanes share a row;
// lane q owns the contiguous slice [q*kc, (q+1)*kc) of the K columns (kc <= GLA_EDOT_KC) and keeps its slice of
// G[row, :] in registers.  The row's edges are taken four at a time: the four columns are loaded first, then all
// 4*kc gathers are issued back to back (clamped indices, no exec-mask branches), then the KP partial sums are
// combined by a fixed xor-shuffle tree (deterministic; not the sequential k order -- this is a gradient).
#define GLA_EDOT_KC 8
template <typename T, int KP>
__global__ __launch_bounds__(GLA_BLOCK) void edge_dot_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx, int K, int64_t n_rows, int nnz,
    T* __restrict__ out) {
  constexpr int RPP = GLA_BLOCK / KP;
  const int q = threadIdx.x % KP, rsub = threadIdx.x / KP;
  const int kc = (K + KP - 1) / KP;
  const int k_lo = q * kc;
  const int64_t n_round = ((n_rows + RPP - 1) / RPP) * RPP;        // whole groups stay together for the shuffles
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_round; r += (int64_t)gridDim.x * RPP) {
    const bool live = r < n_rows;
    const int64_t rr = live ? r : n_rows - 1;
    const int a = live ? rowptr[rr] : 0, b = live ? rowptr[rr + 1] : 0;
    const int len = b - a;
    T g[GLA_EDOT_KC];
    int kk[GLA_EDOT_KC];
#pragma unroll
    for (int i = 0; i < GLA_EDOT_KC; ++i) {
      const bool in = i < kc && k_lo + i < K;
      kk[i] = in ? k_lo + i : 0;                                   // out-of-slice slots read column 0, times g = 0
      const T gv = G[rr * ldg + kk[i]];
      g[i] = in ? gv : T(0);
    }
    for (int j0 = 0; __any(j0 < len); j0 += 4) {                   // wave-uniform trip count (all lanes stay active)
      int64_t c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = col[min(a + min(j0 + u, max(len - 1, 0)), nnz - 1)];
      T acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        T xv[GLA_EDOT_KC];
#pragma unroll
        for (int i = 0; i < GLA_EDOT_KC; ++i) xv[i] = X[c[u] * ldx + kk[i]];
        T s = T(0);
#pragma unroll
        for (int i = 0; i < GLA_EDOT_KC; ++i) s = s + g[i] * xv[i];
        acc[u] = s;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = group_sum<KP>(acc[u]);
        const int p = a + j0 + u;
        if (j0 + u < len && q == 0) out[perm ? (int64_t)perm[p] : (int64_t)p] = acc[u];
      }
    }
  }
}

// Vector variant (default when K, the leading dimensions and the base pointers allow 16-byte accesses): the KP lanes
// of a row take CONSECUTIVE 16-byte vectors of the K-wide rows (lane q: vectors q, q + KP, ...), so one wave-level
// load reads whole rows of G / X -- every fetched line is consumed by the instruction that fetched it.  Measured on
// the 2048^2 pattern, K = 20 (profiles/r01/edge_dot_variants.txt): scalar slices spend their time in the texture
// address path (5e8 L1 accesses), one-lane-per-row 16-byte loads thrash L1 (1.2e8 L2 read requests, 3x write
// amplification); this mapping has neither problem.
template <typename T, int KP, int VC, bool PERM, int U>
__global__ __launch_bounds__(GLA_BLOCK) void edge_dot_vec_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx, int nvec, int64_t n_rows, int nnz,
    T* __restrict__ out) {
  constexpr int VW = 16 / sizeof(T);
  typedef T TV __attribute__((ext_vector_type(VW)));
  constexpr int RPP = GLA_BLOCK / KP;
  const int q = threadIdx.x % KP, rsub = threadIdx.x / KP;
  int vo[VC];
  bool in[VC];
#pragma unroll
  for (int i = 0; i < VC; ++i) {
    in[i] = q + i * KP < nvec;
    vo[i] = in[i] ? q + i * KP : 0;
  }
  const int64_t n_round = ((n_rows + RPP - 1) / RPP) * RPP;        // whole groups stay together (uniform trip counts)
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_round; r += (int64_t)gridDim.x * RPP) {
    const bool live = r < n_rows;
    const int64_t rr = live ? r : n_rows - 1;                      // dead lanes redo the last row, store nothing
    const int a = rowptr[rr], b = rowptr[rr + 1];
    const int len = live ? b - a : 0;
    TV g[VC];
    const TV* gp = reinterpret_cast<const TV*>(G + rr * ldg);
#pragma unroll
    for (int i = 0; i < VC; ++i) {
      const TV t = gp[vo[i]];
      g[i] = in[i] ? t : TV(T(0));
    }
    for (int j0 = 0; __any(j0 < len); j0 += U) {                   // wave-uniform trip count (all lanes stay active)
      const TV* xp[U];
      int64_t e[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pc = min(a + min(j0 + u, max(len - 1, 0)), nnz - 1);       // clamped: every lane issues every load
        const int64_t c = col[pc];
        if constexpr (PERM) e[u] = (int64_t)perm[pc];
        else e[u] = (int64_t)pc;
        xp[u] = reinterpret_cast<const TV*>(X + c * ldx);
      }
      TV xv[U][VC];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < VC; ++i) xv[u][i] = xp[u][vo[i]];
      T acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T s_ = T(0);
#pragma unroll
        for (int i = 0; i < VC; ++i)
#pragma unroll
          for (int k = 0; k < VW; ++k) s_ = s_ + g[i][k] * xv[u][i][k];
        acc[u] = group_sum<KP>(s_);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + u < len && q == 0) out[e[u]] = acc[u];
    }
  }
}

// K beyond KP * GLA_EDOT_KC columns: same lane layout, G re-read per edge (L1 hits), one edge at a time.
template <typename T, int KP>
__global__ __launch_bounds__(GLA_BLOCK) void edge_dot_wide_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const T* __restrict__ G, int64_t ldg, const T* __restrict__ X, int64_t ldx, int K, int64_t n_rows,
    T* __restrict__ out) {
  constexpr int RPP = GLA_BLOCK / KP;
  const int k0 = threadIdx.x % KP, rsub = threadIdx.x / KP;
  const int64_t n_round = ((n_rows + RPP - 1) / RPP) * RPP;
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_round; r += (int64_t)gridDim.x * RPP) {
    const bool live = r < n_rows;
    const int a = live ? rowptr[r] : 0, b = live ? rowptr[r + 1] : 0;
    const int len = b - a;
    for (int j = 0; __any(j < len); ++j) {
      const int p = a + j;
      T acc = T(0);
      if (j < len) {
        const int64_t c = col[p];
        for (int k = k0; k < K; k += KP) acc = acc + G[r * ldg + k] * X[c * ldx + k];
      }
      acc = group_sum<KP>(acc);
      if (j < len && k0 == 0) out[perm ? (int64_t)perm[p] : (int64_t)p] = acc;
    }
  }
}

// AoS -> SoA split of the reference's attribute tensors (vertex_attr [n, a] row-major -> a contiguous vectors): one
// pass over the AoS rows instead of `a` strided copies.  The a (<= 8) column pointers are passed by value.
struct SplitPtrs { void* p[8]; };

struct CopyBatch {                      // up to 8 strided vector copies in one launch (blockIdx.y = entry)
  const void* src[8];
  void* dst[8];
  int64_t count[8], sstride[8];
};

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void copy_batch_kernel(CopyBatch cb) {
  const int e = blockIdx.y;
  const T* __restrict__ src = static_cast<const T*>(cb.src[e]);
  T* __restrict__ dst = static_cast<T*>(cb.dst[e]);
  const int64_t n = cb.count[e], ss = cb.sstride[e];
  for (int64_t i = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * GLA_BLOCK)
    dst[i] = src[i * ss];
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void split_columns_kernel(const T* __restrict__ src, int64_t n, int a, int64_t ld,
                                                                  SplitPtrs out) {
  // lanes walk the flat [n, a] array (coalesced reads); each element goes to out[col][row]
  const int64_t total = n * a;
  for (int64_t idx = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t r = idx / a;
    const int c = (int)(idx - r * a);
    static_cast<T*>(out.p[c])[r] = src[r * ld + c];
  }
}

template <typename T, int KC>
__global__ __launch_bounds__(GLA_BLOCK) void col_scale_rsqrt_kernel(const T* __restrict__ X, int64_t ldx,
                                                                    const T* __restrict__ sumsq,
                                                                    T* __restrict__ Y, int64_t ldy, int64_t n,
                                                                    int Krt) {
  const int K = KC > 0 ? KC : Krt;
  const int64_t total = n * K;
  for (int64_t idx = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t r = idx / K;
    const int k = (int)(idx - r * K);
    Y[r * ldy + k] = X[r * ldx + k] / sqrt(sumsq[k]);
  }
}

// The returned edge attributes of every EdgeUpdate in one pass: out[e, :] = [A_e, A_e * X[col_e, 0..K)] -- the
// reference's torch.cat([A_ij, c_ij], 1) (e.g. JacobiGNN.py:88-89) written as whole rows of 1 + K values instead of a
// product pass plus a concatenation pass.  KC1 = 1 + K at compile time (0: runtime).
template <typename T, int KC1>
__global__ __launch_bounds__(GLA_BLOCK) void edge_attr_kernel(
    const int32_t* __restrict__ col, const T* __restrict__ val, const int32_t* __restrict__ perm,
    const T* __restrict__ X, int64_t ldx, T* __restrict__ out, int64_t ldo, int64_t nnz, int K1rt) {
  const int K1 = KC1 > 0 ? KC1 : K1rt;
  const int64_t total = nnz * K1;
  for (int64_t base = (int64_t)blockIdx.x * (GLA_BLOCK * 4) + threadIdx.x; base < total;
       base += (int64_t)gridDim.x * (GLA_BLOCK * 4)) {
    int64_t e[4];
    int j[4];
    T v[4];
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t idx = min(base + (int64_t)u * GLA_BLOCK, total - 1);
      const int64_t p = idx / K1;
      j[u] = (int)(idx - p * K1);
      e[u] = perm ? (int64_t)perm[p] : p;
      v[u] = val[p];
      c[u] = col[p];
    }
    T xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = X[(int64_t)c[u] * ldx + max(j[u] - 1, 0)];
#pragma unroll
    for (int u = 0; u < 4; ++u) out[e[u] * ldo + j[u]] = j[u] == 0 ? v[u] : v[u] * xv[u];
  }
}

template <typename T>
int launch_edge_attr(gla_csr* A, int K, const T* X, int64_t ldx, T* out, int64_t ldo, hipStream_t st) {
  const dim3 grid(grid_for(A->nnz * (K + 1), GLA_BLOCK * 4, 256 * 16));
  GLA_K_SWITCH(K, hipLaunchKernelGGL((edge_attr_kernel<T, (KC > 0 ? KC + 1 : 0)>), grid, dim3(GLA_BLOCK), 0, st,
                                     A->col, (const T*)A->val, A->perm, X, ldx, out, ldo, A->nnz, K + 1))
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_edge_products(gla_csr* A, int K, const T* X, int64_t ldx, T* C, int64_t ldc, hipStream_t st) {
  const dim3 grid(grid_for(A->nnz * K, GLA_BLOCK * 4, 256 * 16));
  GLA_K_SWITCH(K, hipLaunchKernelGGL((edge_products_kernel<T, KC>), grid, dim3(GLA_BLOCK), 0, st, A->col,
                                     (const T*)A->val, A->perm, X, ldx, C, ldc, A->nnz, K))
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_col_scale(int64_t n, int K, const T* X, int64_t ldx, const T* sumsq, T* Y, int64_t ldy, hipStream_t st) {
  const dim3 grid(grid_for(n * K, GLA_BLOCK * 4));
  GLA_K_SWITCH(K, hipLaunchKernelGGL((col_scale_rsqrt_kernel<T, KC>), grid, dim3(GLA_BLOCK), 0, st, X, ldx, sumsq,
                                     Y, ldy, n, K))
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_segment_reduce(gla_csr* A, int K, const T* src, int64_t ld_src, int op, T* out, int64_t ld_out,
                          hipStream_t st) {
  int kp = 1;
  while (kp < K && kp < 64) kp <<= 1;
  const unsigned grid = grid_for(A->n_rows, GLA_BLOCK / kp, 256 * 16) + (unsigned)A->long_rows;
#define GLA_SEG(KP_)                                                                                            \
  hipLaunchKernelGGL((segment_reduce_kernel<T, KP_>), dim3(grid), dim3(GLA_BLOCK), 0, st, A->rowptr, A->perm,  \
                     src, ld_src, K, op, A->n_rows, out, ld_out, A->long_row_ids, (int)A->long_rows)
  switch (kp) {
    case 1: GLA_SEG(1); break;
    case 2: GLA_SEG(2); break;
    case 4: GLA_SEG(4); break;
    case 8: GLA_SEG(8); break;
    case 16: GLA_SEG(16); break;
    case 32: GLA_SEG(32); break;
    default: GLA_SEG(64); break;
  }
#undef GLA_SEG
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_segment_stats(gla_csr* A, int K, const T* src, int64_t ld_src, T* out, int64_t ld_out, hipStream_t st) {
  int kp = 1;
  while (kp < K && kp < 64) kp <<= 1;
  const unsigned grid = grid_for(A->n_rows, GLA_BLOCK / kp, 256 * 16);
#define GLA_SEG(KP_)                                                                                            \
  hipLaunchKernelGGL((segment_stats_kernel<T, KP_>), dim3(grid), dim3(GLA_BLOCK), 0, st, A->rowptr, A->perm,   \
                     src, ld_src, K, A->n_rows, out, ld_out)
  switch (kp) {
    case 1: GLA_SEG(1); break;
    case 2: GLA_SEG(2); break;
    case 4: GLA_SEG(4); break;
    case 8: GLA_SEG(8); break;
    case 16: GLA_SEG(16); break;
    case 32: GLA_SEG(32); break;
    default: GLA_SEG(64); break;
  }
#undef GLA_SEG
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_edge_dot(gla_csr* A, int K, const T* G, int64_t ldg, const T* X, int64_t ldx, T* out, hipStream_t st) {
  int kp = 1;
  while (kp * GLA_EDOT_KC < K && kp < 16) kp <<= 1;   // fewest lanes per row whose register slice covers K
  const unsigned grid = grid_for(A->n_rows, GLA_BLOCK / kp, 256 * 16);
  {
    constexpr int VW = 16 / (int)sizeof(T);
    const int nvec = K / VW;
    const bool vec_ok = K % VW == 0 && ldg % VW == 0 && ldx % VW == 0 && nvec <= 64 &&
                        ((uintptr_t)G % 16) == 0 && ((uintptr_t)X % 16) == 0;
    if (vec_ok && g_tuning.spmm_variant != 7) {      // spmm_variant 7: scalar-slice kernel (A/B)
      int vkp = 1;
      while (vkp < nvec && vkp < 16) vkp <<= 1;
      const int vc = (nvec + vkp - 1) / vkp;
      const unsigned vgrid = grid_for(A->n_rows, GLA_BLOCK / vkp, 256 * 16);
      // edges are taken U at a time: U = 5 when the longest row (5-point, 9/10-point stencils) wastes fewer slots
      const int L = A->max_row_len > 0 ? A->max_row_len : 1;
      const bool u5 = ((L + 4) / 5) * 5 - L < ((L + 3) / 4) * 4 - L;
#define GLA_EDOTV2(KP_, VC_, PERM_)                                                                              \
  do {                                                                                                           \
    if (u5)                                                                                                      \
      hipLaunchKernelGGL((edge_dot_vec_kernel<T, KP_, VC_, PERM_, 5>), dim3(vgrid), dim3(GLA_BLOCK), 0, st,      \
                         A->rowptr, A->col, A->perm, G, ldg, X, ldx, nvec, A->n_rows, (int)A->nnz, out);         \
    else                                                                                                         \
      hipLaunchKernelGGL((edge_dot_vec_kernel<T, KP_, VC_, PERM_, 4>), dim3(vgrid), dim3(GLA_BLOCK), 0, st,      \
                         A->rowptr, A->col, A->perm, G, ldg, X, ldx, nvec, A->n_rows, (int)A->nnz, out);         \
  } while (0)
#define GLA_EDOTV(KP_, VC_)                                                                                      \
  do {                                                                                                           \
    if (A->perm) GLA_EDOTV2(KP_, VC_, true);                                                                     \
    else GLA_EDOTV2(KP_, VC_, false);                                                                            \
  } while (0)
      if (vkp == 1) GLA_EDOTV(1, 1);
      else if (vkp == 2) GLA_EDOTV(2, 1);
      else if (vkp == 4) GLA_EDOTV(4, 1);
      else if (vkp == 8) GLA_EDOTV(8, 1);
      else if (vc == 1) GLA_EDOTV(16, 1);
      else if (vc == 2) GLA_EDOTV(16, 2);
      else if (vc == 3) GLA_EDOTV(16, 3);
      else GLA_EDOTV(16, 4);
#undef GLA_EDOTV
#undef GLA_EDOTV2
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
  if ((int64_t)kp * GLA_EDOT_KC < K) {
    hipLaunchKernelGGL((edge_dot_wide_kernel<T, 16>), dim3(grid), dim3(GLA_BLOCK), 0, st, A->rowptr, A->col, A->perm,
                       G, ldg, X, ldx, K, A->n_rows, out);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  }
#define GLA_EDOT(KP_)                                                                                           \
  hipLaunchKernelGGL((edge_dot_kernel<T, KP_>), dim3(grid), dim3(GLA_BLOCK), 0, st, A->rowptr, A->col, A->perm, \
                     G, ldg, X, ldx, K, A->n_rows, (int)A->nnz, out)
  switch (kp) {
    case 1: GLA_EDOT(1); break;
    case 2: GLA_EDOT(2); break;
    case 4: GLA_EDOT(4); break;
    case 8: GLA_EDOT(8); break;
    default: GLA_EDOT(16); break;
  }
#undef GLA_EDOT
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int64_t gla_dot_work_bytes(int64_t n, int K, int dtype) {
  (void)n;
  return (int64_t)DOT_BLOCKS_MAX * (K < 1 ? 1 : K) * (dtype == GLA_F64 ? 8 : 4);
}

int gla_col_dot(int dtype, int64_t n, int K, const void* X, int64_t ldx, const void* Y, int64_t ldy, void* out,
                void* work, gla_stream_t stream) {
  if (n < 0 || K < 1 || !out || !work || (n > 0 && (!X || !Y)) || ldx < K || ldy < K ||
      (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return launch_col_dot<double>(n, K, (const double*)X, ldx, (const double*)Y, ldy, (double*)out, (double*)work,
                                  (hipStream_t)stream);
  return launch_col_dot<float>(n, K, (const float*)X, ldx, (const float*)Y, ldy, (float*)out, (float*)work,
                               (hipStream_t)stream);
}

int gla_col_scale_rsqrt(int dtype, int64_t n, int K, const void* X, int64_t ldx, const void* sumsq, void* Y,
                        int64_t ldy, gla_stream_t stream) {
  if (n < 0 || K < 1 || !sumsq || (n > 0 && (!X || !Y)) || ldx < K || ldy < K ||
      (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (n == 0) return GLA_OK;
  GLA_DISPATCH(dtype, {
    return launch_col_scale<T>(n, K, (const T*)X, ldx, (const T*)sumsq, (T*)Y, ldy, (hipStream_t)stream);
  })
}

int gla_segment_reduce(gla_csr_t A, int dtype, int K, const void* src, int64_t ld_src, int op, void* out,
                       int64_t ld_out, gla_stream_t stream) {
  if (!A || K < 1 || op < 0 || op > 3 || (dtype != GLA_F32 && dtype != GLA_F64) || ld_src < K || ld_out < K)
    return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0) return GLA_OK;
  if (!out || (A->nnz > 0 && !src)) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return launch_segment_reduce<double>(A, K, (const double*)src, ld_src, op, (double*)out, ld_out, (hipStream_t)stream);
  return launch_segment_reduce<float>(A, K, (const float*)src, ld_src, op, (float*)out, ld_out, (hipStream_t)stream);
}

int gla_segment_stats(gla_csr_t A, int dtype, int K, const void* src, int64_t ld_src, void* out, int64_t ld_out,
                      gla_stream_t stream) {
  if (!A || K < 1 || (dtype != GLA_F32 && dtype != GLA_F64) || ld_src < K || ld_out < 4 * (int64_t)K)
    return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0) return GLA_OK;
  if (!out || (A->nnz > 0 && !src)) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return launch_segment_stats<double>(A, K, (const double*)src, ld_src, (double*)out, ld_out, (hipStream_t)stream);
  return launch_segment_stats<float>(A, K, (const float*)src, ld_src, (float*)out, ld_out, (hipStream_t)stream);
}

int gla_edge_dot(gla_csr_t A, int dtype, int K, const void* G, int64_t ldg, const void* X, int64_t ldx,
                 void* out_edge, gla_stream_t stream) {
  if (!A || K < 1 || (dtype != GLA_F32 && dtype != GLA_F64) || ldg < K || ldx < K) return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0 || A->nnz == 0) return GLA_OK;
  if (!G || !X || !out_edge) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return launch_edge_dot<double>(A, K, (const double*)G, ldg, (const double*)X, ldx, (double*)out_edge,
                                   (hipStream_t)stream);
  return launch_edge_dot<float>(A, K, (const float*)G, ldg, (const float*)X, ldx, (float*)out_edge,
                                (hipStream_t)stream);
}

int gla_split_columns(int dtype, int64_t n, int a, const void* src, int64_t ld, void* const* cols_host,
                      gla_stream_t stream) {
  if (n < 0 || a < 1 || a > 8 || ld < a || !cols_host || (n > 0 && !src) || (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (n == 0) return GLA_OK;
  SplitPtrs sp;
  for (int c = 0; c < 8; ++c) sp.p[c] = c < a ? cols_host[c] : nullptr;
  for (int c = 0; c < a; ++c) if (!sp.p[c]) return GLA_ERR_INVALID_ARG;
  const unsigned grid = grid_for(n * a, GLA_BLOCK * 4, 256 * 16);
  if (dtype == GLA_F64)
    hipLaunchKernelGGL((split_columns_kernel<double>), dim3(grid), dim3(GLA_BLOCK), 0, (hipStream_t)stream,
                       (const double*)src, n, a, ld, sp);
  else
    hipLaunchKernelGGL((split_columns_kernel<float>), dim3(grid), dim3(GLA_BLOCK), 0, (hipStream_t)stream,
                       (const float*)src, n, a, ld, sp);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

int gla_copy_batch(int dtype, int n_copies, const void* const* src_host, void* const* dst_host, const int64_t* count_host,
                   const int64_t* src_stride_host, gla_stream_t stream) {
  if (n_copies < 0 || n_copies > 8 || (dtype != GLA_F32 && dtype != GLA_F64) ||
      (n_copies > 0 && (!src_host || !dst_host || !count_host)))
    return GLA_ERR_INVALID_ARG;
  CopyBatch cb;
  int64_t most = 0;
  int k = 0;
  for (int i = 0; i < n_copies; ++i) {
    if (count_host[i] < 0 || (count_host[i] > 0 && (!src_host[i] || !dst_host[i]))) return GLA_ERR_INVALID_ARG;
    if (count_host[i] == 0) continue;
    cb.src[k] = src_host[i]; cb.dst[k] = dst_host[i]; cb.count[k] = count_host[i];
    cb.sstride[k] = src_stride_host ? src_stride_host[i] : 1;
    if (cb.sstride[k] < 1) return GLA_ERR_INVALID_ARG;
    most = count_host[i] > most ? count_host[i] : most;
    ++k;
  }
  if (k == 0) return GLA_OK;
  for (int i = k; i < 8; ++i) { cb.src[i] = nullptr; cb.dst[i] = nullptr; cb.count[i] = 0; cb.sstride[i] = 1; }
  const dim3 grid(grid_for(most, GLA_BLOCK * 4, 256 * 8), (unsigned)k);
  if (dtype == GLA_F64)
    hipLaunchKernelGGL((copy_batch_kernel<double>), grid, dim3(GLA_BLOCK), 0, (hipStream_t)stream, cb);
  else
    hipLaunchKernelGGL((copy_batch_kernel<float>), grid, dim3(GLA_BLOCK), 0, (hipStream_t)stream, cb);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

int gla_edge_products(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* C, int64_t ldc,
                      gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!X || (A->nnz > 0 && !C) || ldx < K || ldc < K) return GLA_ERR_INVALID_ARG;
  if (A->nnz == 0) return GLA_OK;
  GLA_DISPATCH(dtype, {
    return launch_edge_products<T>(A, K, (const T*)X, ldx, (T*)C, ldc, (hipStream_t)stream);
  })
}

int gla_edge_attr(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* out, int64_t ldo,
                  gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0 || A->nnz == 0) return GLA_OK;
  if (!X || !out || ldx < K || ldo < K + 1) return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    return launch_edge_attr<T>(A, K, (const T*)X, ldx, (T*)out, ldo, (hipStream_t)stream);
  })
}

int gla_soc_classic(gla_csr_t A, int dtype, double theta, void* v_out, void* S_out, gla_stream_t stream) {
  int rc = check_plan(A, dtype, 1);
  if (rc) return rc;
  if (!v_out && A->n_rows > 0) return GLA_ERR_INVALID_ARG;
  if (!S_out && A->nnz > 0) return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0) return GLA_OK;
  TileMap tm = all_tiles(A);
  GLA_DISPATCH(dtype, {
    hipLaunchKernelGGL((soc_classic_kernel<T>), dim3(tm.tiles_per_xcd * GLA_XCDS), dim3(GLA_BLOCK),
                       (size_t)A->lds_elems * sizeof(T), (hipStream_t)stream, A->rowptr, (const T*)A->val,
                       A->perm, tm, (T)theta, (T*)v_out, (T*)S_out);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  })
}

int gla_direct_interp(gla_csr_t A, int dtype, const void* diag, const void* cf, const void* S, void* w_out,
                      void* alpha_out, gla_stream_t stream) {
  int rc = check_plan(A, dtype, 1);
  if (rc) return rc;
  if (A->n_cols < A->n_rows) return GLA_ERR_INVALID_ARG;      // cf is indexed by column AND by row: [n_cols] with the
  if (A->n_rows > 0 && (!diag || !cf)) return GLA_ERR_INVALID_ARG;   // plan's rows first (a row block + its halo columns)
  if (A->nnz > 0 && (!S || !w_out)) return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0) return GLA_OK;
  TileMap tm = all_tiles(A);
  GLA_DISPATCH(dtype, {
    hipLaunchKernelGGL((direct_interp_kernel<T>), dim3(tm.tiles_per_xcd * GLA_XCDS), dim3(GLA_BLOCK),
                       (size_t)A->lds_elems * sizeof(T) * 2, (hipStream_t)stream, A->rowptr, A->col,
                       (const T*)A->val, A->perm, tm, A->lds_elems, (const T*)diag, (const T*)cf, (const T*)S,
                       (T*)w_out, (T*)alpha_out);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  })
}

int gla_soc_sa(gla_csr_t A, int dtype, const void* diag, void* S_out, gla_stream_t stream) {
  int rc = check_plan(A, dtype, 1);
  if (rc) return rc;
  if (A->n_rows != A->n_cols) return GLA_ERR_INVALID_ARG;
  if ((A->n_rows > 0 && !diag) || (A->nnz > 0 && !S_out)) return GLA_ERR_INVALID_ARG;
  if (A->n_rows == 0) return GLA_OK;
  TileMap tm = all_tiles(A);
  GLA_DISPATCH(dtype, {
    hipLaunchKernelGGL((soc_sa_kernel<T>), dim3(tm.tiles_per_xcd * GLA_XCDS), dim3(GLA_BLOCK),
                       (size_t)A->lds_elems * sizeof(T) * 2, (hipStream_t)stream, A->rowptr, A->col,
                       (const T*)A->val, A->perm, tm, A->lds_elems, (const T*)diag, (T*)S_out);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  })
}

}  // extern "C"
