// libgla_hip.so -- row kernels: SpMV / SpMM with fused vertex updates (gfx950 / MI355X only; wave = 64 lanes).
//
// One CSR "row-tile" kernel family replaces the reference's per-layer
//   gather x[col] -> multiply -> torch.cat -> torch_scatter.scatter(sum) -> vertex update -> torch.cat
// (e.g. pytorch/JacobiGNN.py:52-123) with a single pass over HBM:
//
//   phase 1  the workgroup streams its tile's nnz (val f64/f32 + col int32, CSR order) with fully
//            coalesced loads, gathers x[col] (served by the XCD's L2: a tile's column window is a few
//            grid lines) and parks the products in LDS;
//   phase 2  one lane per row adds its products IN EDGE ORDER (sequential, no FMA contraction, so the
//            sum is bit-identical to the reference's CPU scatter-add) and applies the fused vertex
//            update (residual / Jacobi / Chebyshev ...), reading and writing the dense vectors once.
//
// Tiles are contiguous row ranges holding ~tile_nnz nonzeros (CSR-adaptive "stream" binning computed at
// plan creation); blockIdx is remapped so that the XCDs take turns of 64 consecutive tiles (gla_device.h,
// map_tile): all of them stream one common region of HBM and a tile's x window stays in its XCD's L2.  A row longer than
// GLA_LONG_ROW is a tile of its own (the "long" bin): every kernel sends such a tile to long_row_spmm (gla_device.h),
// where the whole workgroup sums it in fixed chunks; all other rows keep their strictly in-order sums.
//
// Multi-RHS (K > 1, row-major [n, K]): val/col are staged in LDS instead of products, and the lanes
// are (row, rhs) pairs so that every gather reads K contiguous values of one X row.
#include "gla_device.h"

#include <mutex>

namespace {

// ------------------------------------------------------------------------------------------------
// fused vertex updates ("epilogues"): called once per (row, rhs) with the finished row sum
// ------------------------------------------------------------------------------------------------
// Each epilogue is split into load -- the row's dense operands, issued EARLY so their latency overlaps the
// nnz stream -- and apply with the finished row sum.  Both are generic in the value type V: V = T for one
// right-hand side per lane, V = T2 (16-byte accesses) for two adjacent right-hand sides per lane.
// operator() = load + apply for kernels that do not prefetch.
#define GLA_EPI_SCALAR_API                                                                            \
  static constexpr int NRED = 0; /* no block-level reduction */                                       \
  template <typename V> struct GxV {};                                                                \
  template <typename V> __device__ __forceinline__ GxV<V> gx_load(int) const { return GxV<V>{}; }     \
  template <typename V> __device__ __forceinline__ V gx(V xv, const GxV<V>&) const { return xv; }     \
  template <typename V> __device__ __forceinline__ V redv(V, const PreV<V>&) const { return V(); }    \
  __device__ __forceinline__ T* partials_ptr() const { return nullptr; }                                                         \
  GLA_EPI_SCALAR_CORE

#define GLA_EPI_SCALAR_CORE                                                                           \
  typedef PreV<T> Pre;                                                                                \
  __device__ __forceinline__ Pre load(int64_t r, int k) const { return loadv<T>(r, k); }              \
  __device__ __forceinline__ void apply(int64_t r, int k, T acc, const Pre& p) const {                \
    applyv<T>(r, k, acc, p);                                                                          \
  }                                                                                                   \
  __device__ __forceinline__ void operator()(int64_t r, int k, T acc) const {                         \
    applyv<T>(r, k, acc, loadv<T>(r, k));                                                             \
  }

template <typename T>
struct EpiStore {                       // y = A x                       MatVecGNN.py:95-114
  T* y; int64_t ldy;
  template <typename V> struct PreV {};
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t, int) const { return PreV<V>{}; }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>&) const {
    stv<V>(y + r * ldy + k, acc);
  }
  bool vec_ok(int w) const { return al16(y) && ldy % w == 0; }
  GLA_EPI_SCALAR_API
};

template <typename T>
struct EpiAddTo {                       // y = z + A x                   the coarse-grid correction x + P xc, VCycle.py:228
  const T* z; int64_t ldz; T* y; int64_t ldy;
  template <typename V> struct PreV { V z; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{ldv<V>(z + r * ldz + k)};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>& p) const {
    stv<V>(y + r * ldy + k, p.z + acc);
  }
  bool vec_ok(int w) const { return al16(z) && al16(y) && ldz % w == 0 && ldy % w == 0; }
  GLA_EPI_SCALAR_API
};

template <typename T>
struct EpiResidual {                    // r = b - A x                   GNNResidual.py:97-118
  const T* b; int64_t ldb; T* res; int64_t ldr;
  template <typename V> struct PreV { V b; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{ldv<V>(b + r * ldb + k)};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>& p) const {
    stv<V>(res + r * ldr + k, p.b - acc);
  }
  bool vec_ok(int w) const { return al16(b) && al16(res) && ldb % w == 0 && ldr % w == 0; }
  GLA_EPI_SCALAR_API
};

template <typename T>
struct EpiJacobi {                      // x + (w*(b - A x))/A_ii        JacobiGNN.py:100-123 (:119)
  const T* diag; const T* b; int64_t ldb; const T* x; int64_t ldx; T* xo; int64_t ldxo; T w;
  template <typename V> struct PreV { T d; V b, x; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{diag[r], ldv<V>(b + r * ldb + k), ldv<V>(x + r * ldx + k)};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>& p) const {
    stv<V>(xo + r * ldxo + k, p.x + (w * (p.b - acc)) / p.d);
  }
  bool vec_ok(int w) const {
    return al16(b) && al16(x) && al16(xo) && ldb % w == 0 && ldx % w == 0 && ldxo % w == 0;
  }
  GLA_EPI_SCALAR_API
};

template <typename T>
struct EpiChebyFirst {                  // r = b - A x; p = r; x += alpha p    ChebyGNN.py:91-163
  const T* b; int64_t ldb; const T* x; int64_t ldx; T* res; int64_t ldr; T* p; int64_t ldp;
  T* xo; int64_t ldxo; T alpha;
  template <typename V> struct PreV { V b, x; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{ldv<V>(b + r * ldb + k), ldv<V>(x + r * ldx + k)};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>& q) const {
    const V ri = q.b - acc;
    stv<V>(res + r * ldr + k, ri);
    stv<V>(p + r * ldp + k, ri);
    stv<V>(xo + r * ldxo + k, q.x + alpha * ri);
  }
  bool vec_ok(int w) const {
    return al16(b) && al16(x) && al16(res) && al16(p) && al16(xo) && ldb % w == 0 && ldx % w == 0 &&
           ldr % w == 0 && ldp % w == 0 && ldxo % w == 0;
  }
  GLA_EPI_SCALAR_API
};

template <typename T>
struct EpiChebyNext {                   // r -= a_old*(A p); p = r + beta p; x += a_new p   ChebyGNN.py:186-243
  const T* p; int64_t ldp; T* po; int64_t ldpo; T* res; int64_t ldr; T* x; int64_t ldx;
  T a_old, a_new, beta;
  template <typename V> struct PreV { V r, p, x; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{ldv<V>(res + r * ldr + k), ldv<V>(p + r * ldp + k), ldv<V>(x + r * ldx + k)};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>& q) const {
    const V ri = q.r - a_old * acc;
    stv<V>(res + r * ldr + k, ri);
    const V pn = ri + beta * q.p;
    stv<V>(po + r * ldpo + k, pn);
    stv<V>(x + r * ldx + k, q.x + a_new * pn);
  }
  bool vec_ok(int w) const {
    return al16(p) && al16(po) && al16(res) && al16(x) && ldp % w == 0 && ldpo % w == 0 && ldr % w == 0 &&
           ldx % w == 0;
  }
  GLA_EPI_SCALAR_API
};

// Fused power-method step (PowerMethodGNN.py:86-207, three MetaLayers): the gathered operand is normalised on
// the fly (b_j = y_j / n, rounded like the reference's Layer 3), y_i = sum_j A_ij b_j is stored, and the workgroup
// leaves sum_i y_i^2 of its tile in partials[tile * K + k] for the deterministic second-stage reduction.
template <typename T>
__device__ __forceinline__ T div_by(T a, T b, T rb) {
  // a / b from the correctly rounded reciprocal rb = RN(1/b) with one Markstein correction step: q = RN(a*rb),
  // r = a - q*b (exact in an FMA), q' = RN(q + r*rb) = RN(a / b) (Markstein 1990) for every b whose significand is
  // not all ones; those divisors take the true division (all_ones, below).  3 instructions instead of the ~30 of an
  // IEEE fp64 division per gathered value; 671 M random quotients compared bit for bit (tools/exp_power_division.py).
  const T q = a * rb;
  const T r = fma_t(-q, b, a);
  return fma_t(r, rb, q);
}

template <typename T>
struct EpiPower {
  T* y; int64_t ldy; const T* nrm; const T* rnrm; T* partials;
  template <typename V> struct PreV {};
  static constexpr int NRED = 1;
  template <typename V> struct GxV { V n, rn; bool slow; };      // this lane's divisor(s), loaded once per kernel
  static __device__ __forceinline__ bool all_ones(T v) {         // significand all ones: the one case Markstein excludes
    if constexpr (sizeof(T) == 8) return (__builtin_bit_cast(unsigned long long, v) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
    else return (__builtin_bit_cast(unsigned, v) & 0x7FFFFFu) == 0x7FFFFFu;
  }
  template <typename V> __device__ __forceinline__ GxV<V> gx_load(int k) const {
    GxV<V> g;
    g.slow = false;
    if constexpr (sizeof(V) == sizeof(T)) {
      g.n = nrm ? nrm[k] : T(1);
      g.rn = nrm ? rnrm[k] : T(1);
      g.slow = all_ones(g.n);
    } else {
#pragma unroll
      for (int i = 0; i < (int)(sizeof(V) / sizeof(T)); ++i) {
        g.n[i] = nrm ? nrm[k + i] : T(1);
        g.rn[i] = nrm ? rnrm[k + i] : T(1);
        g.slow = g.slow || all_ones(g.n[i]);
      }
    }
    return g;
  }
  template <typename V> __device__ __forceinline__ V gx(V xv, const GxV<V>& g) const {
    if (!nrm) return xv;
    if (g.slow) return xv / g.n;                                 // IEEE division (never taken in practice)
    if constexpr (sizeof(V) == sizeof(T)) return div_by(xv, g.n, g.rn);
    else {
      V o;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(V) / sizeof(T)); ++i) o[i] = div_by(xv[i], g.n[i], g.rn[i]);
      return o;
    }
  }
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t, int) const { return PreV<V>{}; }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V acc, const PreV<V>&) const {
    stv<V>(y + r * ldy + k, acc);
  }
  template <typename V> __device__ __forceinline__ V redv(V acc, const PreV<V>&) const { return acc * acc; }
  __device__ __forceinline__ T* partials_ptr() const { return partials; }
  bool vec_ok(int w) const { return al16(y) && ldy % w == 0; }
  GLA_EPI_SCALAR_CORE
};

// ------------------------------------------------------------------------------------------------
// K = 1: products staged in LDS
// ------------------------------------------------------------------------------------------------
template <bool NT, typename U>
__device__ __forceinline__ U ld_stream(const U* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// Main K = 1 kernel: 16-byte val loads / 8-byte col loads -- each lane owns PAIRS of consecutive nnz; the tile's
// nnz window is aligned down to an even index (the extra element is staged but never summed).
// PF: the row lane's rowptr / dense operands are requested before the nnz stream so that the block runs through
// three dependent memory round trips (tile record -> streams -> x gather) instead of five.
template <typename T, class Epi, int PPT, bool NT, bool PF, bool COL16 = false, int MINW = 1>
__global__ __launch_bounds__(GLA_BLOCK, MINW) void spmv_stream_vec2_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const T* __restrict__ x, TileMap tm, Epi epi, const int16_t* __restrict__ col16 = nullptr,
    const uint32_t* __restrict__ wide_mask = nullptr, unsigned n_cols = 0,
    const uint16_t* __restrict__ rowoff16 = nullptr) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* prod = reinterpret_cast<T*>(smem_raw);
  typedef T T2 __attribute__((ext_vector_type(2)));
  typedef int I2 __attribute__((ext_vector_type(2)));
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, e = ti.w;
  if (r0 >= r1) return;
  if (e - s > tm.lds_cap) {                    // CSR-adaptive long bin: this tile is ONE row, summed by the whole workgroup
    long_row_spmm<T>(col, val, s, e, r0, t, x, (int64_t)1, 1, epi, prod);
    return;
  }
  const int tid = threadIdx.x;
  const int r = r0 + tid;
  int ra = 0, rb = 0;
  typename Epi::Pre pre;
  // COL16: row offsets are read as uint16 relative to the tile's first nonzero, entry r + t (2 B per row, not 4)
  auto row_range = [&](int row, int& a, int& b) {
    if constexpr (COL16) {
      a = s + (int)rowoff16[row + t];
      b = s + (int)rowoff16[row + t + 1];
    } else {
      a = rowptr[row];
      b = rowptr[row + 1];
    }
  };
  if (PF && r < r1) {
    row_range(r, ra, rb);
    pre = epi.load((int64_t)r, 0);
  }
  const auto gxs = epi.template gx_load<T>(0);
  const int sa = s & ~1;                       // even -> val+sa is 16-byte aligned, col+sa 8-byte aligned
  const int npairs = (e - sa + 1) >> 1;
  const T2* val2 = reinterpret_cast<const T2*>(val + sa);
  const I2* col2 = reinterpret_cast<const I2*>(col + sa);
  // COL16: columns are read as int16 offsets from the tile's first row (2 B instead of 4 B per nonzero); tiles
  // with a column out of that range (halo columns of a row-block partition) are flagged in wide_mask and read int32
  const unsigned* colp = reinterpret_cast<const unsigned*>(COL16 ? col16 + sa : nullptr);
  bool wide = false;
  if constexpr (COL16) wide = (wide_mask[t >> 5] >> (t & 31)) & 1u;
  T2* prod2 = reinterpret_cast<T2*>(prod);

  auto stream_tile = [&](auto c16_tag) {
  constexpr bool C16 = decltype(c16_tag)::value;
  for (int base = 0; base < npairs; base += GLA_BLOCK * PPT) {
    T2 v[PPT];
    I2 c[PPT];
    // Branch-free batch: indices are clamped into the tile, so every lane issues every load and every LDS
    // write (lanes past the end recompute the tile's last pair and store the identical value again).  A
    // predicated load or store makes hipcc sink the loads into exec-mask branches with a full s_waitcnt each,
    // which serialises the batch and halves the bytes in flight.
    int qc[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      qc[u] = min(base + u * GLA_BLOCK + tid, npairs - 1);
      if constexpr (C16) {
        const unsigned pk = ld_stream<NT>(colp + qc[u]);
        // the alignment-slack element belongs to a neighbouring tile and may decode out of range: clamp (never summed)
        c[u].x = (int)min((unsigned)(r0 + (int)(short)(pk & 0xFFFFu)), n_cols - 1);
        c[u].y = (int)min((unsigned)(r0 + (int)(short)(pk >> 16)), n_cols - 1);
      } else {
        c[u] = ld_stream<NT>(col2 + qc[u]);
      }
      v[u] = ld_stream<NT>(val2 + qc[u]);
    }
    T2 xv[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      xv[u].x = epi.gx(x[c[u].x], gxs);
      xv[u].y = epi.gx(x[c[u].y], gxs);
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      T2 p;
      p.x = v[u].x * xv[u].x;
      p.y = v[u].y * xv[u].y;
      prod2[qc[u]] = p;
    }
  }
  };
  if constexpr (COL16) {
    if (wide) stream_tile(std::false_type{}); else stream_tile(std::true_type{});
  } else {
    stream_tile(std::false_type{});
  }
  __syncthreads();

  // Row sums.  A row of fewer than GLA_WAVE_ROW (64) nonzeros is summed by ONE lane strictly in edge order (the association
  // order of the reference's sequential CPU scatter-add: bit-exact).  A longer row -- the "vector" bin of a CSR-adaptive
  // SpMV -- is summed by the 64 lanes of the wave its lane belongs to: lane l adds the l-th of 64 contiguous chunks of
  // ceil(len / 64) staged products in edge order and the 64 partial sums meet in a fixed xor tree of cross-lane shuffles
  // (option row_sum = 1, the default; row_sum = 0 keeps one lane per row for every row).  THE ORDER IS A FUNCTION OF THE
  // ROW'S OWN LENGTH, never of the tile the row happens to share: a row-block partition (whose tiles start at the rank's
  // first row) sums every row exactly as the single-rank plan does, so the multi-rank results stay bit-identical
  // (tests/test_gpu_dist.py::test_two_ranks_long_rows).  Round 3 picked the lane count from the tile's population.
  T red = T(0);
  const int lane = tid & 63;
  for (int rbase = r0; rbase < r1; rbase += GLA_BLOCK) {         // (one round unless the tile holds more than 256 short rows)
    const int rr = rbase + tid;
    const bool have = rr < r1;
    int a = 0, b = 0;
    typename Epi::Pre pr;
    if (PF && rbase == r0) { a = ra; b = rb; pr = pre; }
    else if (have) { row_range(rr, a, b); pr = epi.load((int64_t)rr, 0); }
    const int len = have ? b - a : 0;
    const bool heavy = tm.vec && len >= GLA_WAVE_ROW;
    T acc = T(0);
    if (have && !heavy) {
      int j = a - sa;
      const int jend = b - sa;
      for (; j + 4 <= jend; j += 4) {            // LDS reads issued together, added strictly in edge order
        const T p0 = prod[j], p1 = prod[j + 1], p2 = prod[j + 2], p3 = prod[j + 3];
        acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
      }
      for (; j < jend; ++j) acc = acc + prod[j];
    }
    unsigned long long hm = __ballot(heavy);
    while (hm) {                                  // (wave-uniform loop: every lane of the wave takes part)
      const int hl = __ffsll((long long)hm) - 1;
      hm &= hm - 1;
      const int ha = __shfl(a, hl, 64) - sa, hb = __shfl(b, hl, 64) - sa;
      const int chunk = (hb - ha + 63) >> 6;
      int j = ha + lane * chunk;
      const int jend = min(j + chunk, hb);
      T part = T(0);
      for (; j + 4 <= jend; j += 4) {
        const T p0 = prod[j], p1 = prod[j + 1], p2 = prod[j + 2], p3 = prod[j + 3];
        part = part + p0; part = part + p1; part = part + p2; part = part + p3;
      }
      for (; j < jend; ++j) part = part + prod[j];
      for (int off = 1; off < 64; off <<= 1) part = part + __shfl_xor(part, off, 64);
      if (lane == hl) acc = part;
    }
    if (have) {
      epi.apply((int64_t)rr, 0, acc, pr);
      if constexpr (Epi::NRED > 0) red = red + epi.redv(acc, pr);
    }
  }
  if constexpr (Epi::NRED > 0) {                                  // fixed-order workgroup sum -> partials[tile]
    __shared__ T wred[GLA_BLOCK / 64];
    red = wave_sum(red);
    if ((tid & 63) == 0) wred[tid >> 6] = red;
    __syncthreads();
    if (tid == 0) epi.partials_ptr()[t] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
  }
}

// Packed-stream K = 1 kernel (option "compress"): the matrix is read as 1-byte value codes + 2-byte tile-relative
// columns (3 B per nonzero instead of 12).  Each lane owns 8 consecutive nonzeros: one 8-byte code load and one
// 16-byte delta load, 8 gathers in flight; values come from the <= 256-entry dictionary staged in LDS.  Products,
// in-order row sums and the fused vertex update are exactly those of spmv_stream_vec2_kernel (bit-identical).
template <typename T, class Epi>
__global__ __launch_bounds__(GLA_BLOCK) void spmv_packed_kernel(
    const int32_t* __restrict__ rowptr, const int16_t* __restrict__ col16, const uint8_t* __restrict__ vcode,
    const T* __restrict__ vdict, const T* __restrict__ x, unsigned n_cols, TileMap tm, Epi epi,
    const uint16_t* __restrict__ rowoff16) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sdict = reinterpret_cast<T*>(smem_raw);               // [256]
  T* prod = sdict + 256;
  typedef T T2 __attribute__((ext_vector_type(2)));
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, e = ti.w;
  if (r0 >= r1) return;
  const int tid = threadIdx.x;
  // packed tiles hold ~2040 nonzeros = up to 2 rows per lane for 5-point rows: both rows' offsets and dense
  // operands are requested before the nonzero stream (rows past the tile are clamped to its last row, not stored)
  constexpr int RPL = 2;
  int rr[RPL], ra[RPL], rb[RPL];
  typename Epi::Pre pre[RPL];
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    rr[i] = r0 + i * GLA_BLOCK + tid;
    const int rc = min(rr[i], r1 - 1);
    ra[i] = s + (int)rowoff16[rc + t];           // uint16 offsets relative to the tile's first nonzero
    rb[i] = s + (int)rowoff16[rc + t + 1];
    pre[i] = epi.load((int64_t)rc, 0);
  }
  const auto gxs = epi.template gx_load<T>(0);
  const int sa = s & ~7;                       // multiple of 8 -> 8-byte aligned codes, 16-byte aligned deltas
  const int ngrp = (e - sa + 7) >> 3;
  const uint2* code8 = reinterpret_cast<const uint2*>(vcode + sa);
  const int4* delta8 = reinterpret_cast<const int4*>(col16 + sa);
  sdict[tid] = vdict[tid];
  bool first = true;
  for (int base = 0; base < ngrp; base += GLA_BLOCK) {
    const int q = min(base + tid, ngrp - 1);   // branch-free: surplus lanes redo the last group
    const uint2 cw = code8[q];
    const int4 dw = delta8[q];
    if (first) { __syncthreads(); first = false; }          // dictionary visible (loads above are in flight)
    unsigned cc[8];
    T v[8], xv[8];
    const unsigned dd[4] = {(unsigned)dw.x, (unsigned)dw.y, (unsigned)dw.z, (unsigned)dw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lo = (int)(short)(dd[i] & 0xFFFFu), hi = (int)(short)(dd[i] >> 16);
      cc[2 * i] = min((unsigned)(r0 + lo), n_cols - 1);      // elements of neighbouring tiles (alignment slack)
      cc[2 * i + 1] = min((unsigned)(r0 + hi), n_cols - 1);  // may decode out of range: clamp, never summed
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = epi.gx(x[cc[i]], gxs);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = sdict[((i < 4 ? cw.x : cw.y) >> (8 * (i & 3))) & 0xFFu];
    T2* dst = reinterpret_cast<T2*>(prod + 8 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T2 pp;
      pp.x = v[2 * i] * xv[2 * i];
      pp.y = v[2 * i + 1] * xv[2 * i + 1];
      dst[i] = pp;
    }
  }
  __syncthreads();

  T red = T(0);
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    if (rr[i] < r1) {
      T acc = T(0);
      int j = ra[i] - sa;
      const int jend = rb[i] - sa;
      for (; j + 4 <= jend; j += 4) {
        const T p0 = prod[j], p1 = prod[j + 1], p2 = prod[j + 2], p3 = prod[j + 3];
        acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
      }
      for (; j < jend; ++j) acc = acc + prod[j];
      epi.apply((int64_t)rr[i], 0, acc, pre[i]);
      if constexpr (Epi::NRED > 0) red = red + epi.redv(acc, pre[i]);
    }
  }
  for (int r2 = r0 + RPL * GLA_BLOCK + tid; r2 < r1; r2 += GLA_BLOCK) {
    const int a = s + (int)rowoff16[r2 + t] - sa, b = s + (int)rowoff16[r2 + t + 1] - sa;
    T acc = T(0);
    for (int j = a; j < b; ++j) acc = acc + prod[j];
    const auto pre2 = epi.load((int64_t)r2, 0);
    epi.apply((int64_t)r2, 0, acc, pre2);
    if constexpr (Epi::NRED > 0) red = red + epi.redv(acc, pre2);
  }
  if constexpr (Epi::NRED > 0) {
    __shared__ T wred[GLA_BLOCK / 64];
    red = wave_sum(red);
    if ((tid & 63) == 0) wred[tid >> 6] = red;
    __syncthreads();
    if (tid == 0) epi.partials_ptr()[t] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
  }
}

// ------------------------------------------------------------------------------------------------
// K = 1, stencil matrices (option "dia"): every nonzero lies on one of D <= 16 diagonals col - row = off[d], values are
// kept diagonal-major.  One lane per row; val[d][r] and x[r + off[d]] are coalesced streaming loads -- no index stream,
// no LDS, no dependent gather phase.  The row's entries are added in ascending diagonal order = ascending column =
// edge order of a row-sorted list with ascending columns (the only inputs that get this layout), product and sum
// rounded separately: bit-identical to the CSR kernels.  Bytes per row: 8 D + 2 (mask) + the vectors, against
// 10 nnz + 2 + the vectors of the compact CSR stream.
// ------------------------------------------------------------------------------------------------
struct DiaView {
  const void* val;            // [D][n_rows]
  const uint16_t* mask;       // [n_rows] which diagonals the row has
  int32_t off[GLA_DIA_MAX];
  int32_t D;
  int64_t n_rows, n_cols;
  int64_t row_lo, row_hi;     // rows handled by this launch
  double cval[GLA_DIA_MAX];   // CONSTV kernels: the one value of each diagonal (option "compress")
  int64_t part_off;           // reducing epilogues: this launch's row blocks write partials[part_off + block]
  int32_t partner[GLA_DIA_MAX];   // >= 0: diagonal d (offset < 0) is read from this diagonal at row r + off[d] (symmetric value
};                                // sets, option "sym": A_ij == A_ji bit for bit, checked at gla_csr_set_values); -1: from itself

template <typename T, class Epi, int DMAX, bool CONSTV = false>
__global__ __launch_bounds__(GLA_BLOCK) void spmv_dia_kernel(DiaView dv, const T* __restrict__ x, TileMap tm, Epi epi) {
  const int blk = map_tile(tm);                  // row blocks of 256 take the same XCD turns as the CSR tiles
  if (blk < 0) return;
  const int64_t r_raw = dv.row_lo + (int64_t)blk * GLA_BLOCK + threadIdx.x;
  const bool live = r_raw < dv.row_hi;
  if constexpr (Epi::NRED == 0) {
    if (!live) return;
  }
  const int64_t r = live ? r_raw : dv.row_hi - 1;       // reducing epilogues keep every lane for the block sum
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  const unsigned m = dv.mask[r];
  const auto pre = epi.load(r, 0);
  const auto gxs = epi.template gx_load<T>(0);
  T v[DMAX], xv[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {               // every lane issues every load (absent diagonals: clamped, unused)
    const int dd = min(d, dv.D - 1);
    const int64_t c_raw = r + (int64_t)dv.off[dd];
    const int64_t c = min(max(c_raw, (int64_t)0), dv.n_cols - 1);
    if constexpr (CONSTV) v[d] = (T)dv.cval[dd];            // constant diagonals: the stencil is matrix-free
    else {
      // (a lower diagonal of a symmetric value set: its value is the partner's at the row this entry points to -- bytes some
      // other lane streams anyway, so only the upper diagonals cross HBM)
      const int sp = dv.partner[dd];
      const T vv = val[sp >= 0 ? (int64_t)sp * dv.n_rows + c : (int64_t)dd * dv.n_rows + r];
      v[d] = sp >= 0 && c_raw < 0 ? T(0) : vv;
    }
    xv[d] = epi.gx(x[c], gxs);
  }
  T acc = T(0);
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    const T nxt = acc + v[d] * xv[d];
    acc = ((m >> d) & 1u) ? nxt : acc;
  }
  if (live) epi.apply(r, 0, acc, pre);
  if constexpr (Epi::NRED > 0) {                 // fixed-order workgroup sum -> partials[part_off + block]
    __shared__ T wred[GLA_BLOCK / 64];
    T red = live ? epi.redv(acc, pre) : T(0);
    red = wave_sum(red);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = red;
    __syncthreads();
    if (threadIdx.x == 0) epi.partials_ptr()[dv.part_off + blk] = ((wred[0] + wred[1]) + wred[2]) + wred[3];
  }
}

// ------------------------------------------------------------------------------------------------
// K > 1: val/col staged in LDS, lanes = (row, rhs)
// ------------------------------------------------------------------------------------------------
template <typename T, int KP, class Epi>
__global__ __launch_bounds__(GLA_BLOCK) void spmm_stream_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const T* __restrict__ X, int64_t ldx, int K, int lds_elems, TileMap tm, Epi epi) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sval = reinterpret_cast<T*>(smem_raw);
  int32_t* scol = reinterpret_cast<int32_t*>(sval + lds_elems);
  const int t = map_tile(tm);
  if (t < 0) return;
  const int r0 = tm.tile_row[t], r1 = tm.tile_row[t + 1];
  if (r0 >= r1) return;
  const int s = rowptr[r0];
  const int len = rowptr[r1] - s;
  if (len > tm.lds_cap) {                      // CSR-adaptive long bin: one row, whole workgroup
    long_row_spmm<T>(col, val, s, s + len, r0, t, X, ldx, K, epi, sval);
    return;
  }
  const int tid = threadIdx.x;
  {
    T v_[4];
    int c_[4];
#define MM_LD(u, i) { v_[u] = val[s + (i)]; c_[u] = col[s + (i)]; }
#define MM_ST(u, i) { sval[i] = v_[u]; scol[i] = c_[u]; }
    GLA_TILE_LOOP(len, tid, MM_LD, MM_ST)
#undef MM_LD
#undef MM_ST
  }
  __syncthreads();
  constexpr int RPP = GLA_BLOCK / KP;          // rows per pass
  const int k0 = tid % KP, rsub = tid / KP;
  for (int r = r0 + rsub; r < r1; r += RPP) {
    const int a = rowptr[r] - s, b = rowptr[r + 1] - s;
    for (int k = k0; k < K; k += KP) {
      T acc = T(0);
#pragma unroll 4
      for (int j = a; j < b; ++j) acc = acc + sval[j] * X[(int64_t)scol[j] * ldx + k];
      epi((int64_t)r, k, acc);
    }
  }
}

// K even: lanes = (row, PAIR of adjacent rhs) with 16-byte X / epilogue accesses.  val, col AND the tile's row
// offsets are staged in LDS, so phase 2 touches global memory only for the X gathers and the epilogue vectors.
// A row's gathers are issued four at a time (independent 16-byte loads), its sum stays in edge order.
// LPR = lanes per row (power of two >= K/2).  MODE 0: exact trip counts (4-batches + scalar tail);
// MODE 1: branch-free clamped batches (surplus lanes re-read the last entry and are masked out of the sum).
constexpr int SPMM_ROWCAP = 1024;               // row offsets staged per tile (more rows -> read rowptr directly)

// Sum over the N consecutive chunks c0 .. c0 + N - 1 of a heavy row in the association order of the K = 1 kernel's xor tree
// (level l adds the blocks of 2^l chunks that differ in bit l): a balanced binary tree over aligned blocks of chunks.
template <int N, typename V, class F>
__device__ __forceinline__ V chunk_tree(int c0, const F& chunk) {
  if constexpr (N == 1) {
    return chunk(c0);
  } else {
    const V lo = chunk_tree<N / 2, V>(c0, chunk);
    const V hi = chunk_tree<N / 2, V>(c0 + N / 2, chunk);
    return lo + hi;
  }
}

template <typename T, int LPR, int MODE, class Epi, int VW = 2, bool FLAT = false>
__global__ __launch_bounds__(GLA_BLOCK) void spmm_stream_vec2_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const T* __restrict__ X, int64_t ldx, int K, int lds_elems, TileMap tm, Epi epi) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef T T2 __attribute__((ext_vector_type(VW)));      // VW adjacent right-hand sides per lane (2, or 4 for fp32)
  T* sval = reinterpret_cast<T*>(smem_raw);
  int32_t* scol = reinterpret_cast<int32_t*>(sval + lds_elems);
  int32_t* srow = scol + lds_elems;             // [SPMM_ROWCAP + 1] offsets relative to the tile's first nnz
  const int t = map_tile(tm);
  if (t < 0) return;
  const int4 ti = tm.tile_info[t];
  const int r0 = ti.x, r1 = ti.y, s = ti.z, len = ti.w - ti.z;
  if (r0 >= r1) return;
  if (len > tm.lds_cap) {                      // CSR-adaptive long bin: one row, whole workgroup
    long_row_spmm<T>(col, val, s, s + len, r0, t, X, ldx, K, epi, sval);
    return;
  }
  const int tid = threadIdx.x;
  const int nrow_staged = min(r1 - r0, SPMM_ROWCAP);
  for (int i = tid; i <= nrow_staged; i += GLA_BLOCK) srow[i] = rowptr[r0 + i] - s;
  {
    T v_[4];
    int c_[4];
#define MM_LD(u, i) { v_[u] = val[s + (i)]; c_[u] = col[s + (i)]; }
#define MM_ST(u, i) { sval[i] = v_[u]; scol[i] = c_[u]; }
    GLA_TILE_LOOP(len, tid, MM_LD, MM_ST)
#undef MM_LD
#undef MM_ST
  }
  __syncthreads();
  T2 red = T2(T(0));
  // one (row, vector of VW right-hand sides) unit: gather the row's X vectors, in-order sum, fused vertex update
  auto row_vec = [&](int r, int k, int a, int b, const auto& gxs) {
    const auto pre = epi.template loadv<T2>((int64_t)r, k);
    T2 acc = T2(T(0));
    const T* Xk = X + k;
    if constexpr (MODE == 0) {
      int j = a;
      for (; j + 4 <= b; j += 4) {
        T2 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = epi.gx(ldv<T2>(Xk + (int64_t)scol[j + u] * ldx), gxs);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = acc + sval[j + u] * xv[u];
      }
      if (j + 2 <= b) {
        const T2 x0 = epi.gx(ldv<T2>(Xk + (int64_t)scol[j] * ldx), gxs);
        const T2 x1 = epi.gx(ldv<T2>(Xk + (int64_t)scol[j + 1] * ldx), gxs);
        acc = acc + sval[j] * x0;
        acc = acc + sval[j + 1] * x1;
        j += 2;
      }
      if (j < b) acc = acc + sval[j] * epi.gx(ldv<T2>(Xk + (int64_t)scol[j] * ldx), gxs);
    } else {
      for (int j = a; j < b; j += 4) {
        T v[4];
        T2 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = min(j + u, b - 1);
          v[u] = sval[jj];
          xv[u] = epi.gx(ldv<T2>(Xk + (int64_t)scol[jj] * ldx), gxs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const T2 nxt = acc + v[u] * xv[u];
          acc = (j + u < b) ? nxt : acc;
        }
      }
    }
    epi.template applyv<T2>((int64_t)r, k, acc, pre);
    if constexpr (Epi::NRED > 0) red = red + epi.template redv<T2>(acc, pre);
  };
  auto row_bounds = [&](int r, int& a, int& b) {
    if (r - r0 < SPMM_ROWCAP) { a = srow[r - r0]; b = srow[r - r0 + 1]; }
    else { a = rowptr[r] - s; b = rowptr[r + 1] - s; }
  };
  const int k0 = (tid % LPR) * VW;
  if constexpr (FLAT) {
    // K / VW is not a power of two (e.g. K = 20): lanes walk the flattened (row, vector) index space of the tile, so
    // every lane of a wave-level gather is busy (with LPR = 16 lane groups only 10 of 16 would be at K = 20)
    static_assert(Epi::NRED == 0, "the block reduction assumes the lane-group layout");
    const int nvec = K / VW;
    const int total = (r1 - r0) * nvec;
    const auto gxs = epi.template gx_load<T2>(0);
    for (int f = tid; f < total; f += GLA_BLOCK) {
      const int rr = f / nvec;
      int a, b;
      row_bounds(r0 + rr, a, b);
      row_vec(r0 + rr, (f - rr * nvec) * VW, a, b, gxs);
    }
  } else {
    constexpr int RPP = GLA_BLOCK / LPR;          // rows per pass
    const int rsub = tid / LPR;
    const auto gxs0 = epi.template gx_load<T2>(min(k0, max(K - VW, 0)));
    // The VECTOR BIN for K > 1 (round 5; option row_sum, as for K = 1).  A row of GLA_WAVE_ROW (64) nonzeros or more is summed
    // by ALL 64 lanes of the wave its lane group belongs to instead of by its LPR lanes alone (a tile of a few long rows kept
    // 8 x LPR of 256 lanes busy): the row is cut into the SAME 64 contiguous chunks of ceil(len / 64) nonzeros as in
    // spmv_stream_vec2_kernel, lane (g, kv) -- group g of 64 / LPR, right-hand-side vector kv -- sums chunks g LPR .. g LPR +
    // LPR - 1 for its vector in edge order and combines them in the xor tree's order (chunk_tree), the groups meet in
    // cross-lane shuffles at distances LPR, 2 LPR, ..: per right-hand side exactly the association order of the K = 1 kernel,
    // a function of the row's own length -- column k of an SpMM equals the SpMV on column k bit for bit, on any partition.
    constexpr bool VECBIN = MODE == 0 && LPR <= 32;
    const int lane = tid & 63;
    for (int rb = r0; rb < r1; rb += RPP) {       // (wave-uniform: the heavy rows of a wave are worked on by all its lanes)
      const int r = rb + rsub;
      const bool have = r < r1;
      int a = 0, b = 0;
      if (have) row_bounds(r, a, b);
      bool heavy = false;
      unsigned long long hm0 = 0;
      if constexpr (VECBIN) {
        heavy = tm.vec && have && (b - a) >= GLA_WAVE_ROW;
        hm0 = __ballot(heavy && (lane % LPR) == 0);
      }
      for (int kk = 0; kk < K; kk += VW * LPR) {
        const int k = kk + k0;
        const bool kact = k < K;
        T2 hacc = T2(T(0));
        if constexpr (VECBIN) {
          unsigned long long hm = hm0;
          while (hm) {                            // (wave-uniform loop)
            const int hl = __ffsll((long long)hm) - 1;
            hm &= hm - 1;
            const int ha = __shfl(a, hl, 64), hb = __shfl(b, hl, 64);
            const int chunk = (hb - ha + 63) >> 6;
            const auto gxs = epi.template gx_load<T2>(kact ? k : max(K - VW, 0));
            const T* Xk = X + (kact ? k : 0);
            auto chunk_sum = [&](int c) -> T2 {
              int j = ha + c * chunk;
              const int jend = min(j + chunk, hb);
              T2 acc = T2(T(0));
              for (; j + 4 <= jend; j += 4) {
                T2 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = epi.gx(ldv<T2>(Xk + (int64_t)scol[j + u] * ldx), gxs);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = acc + sval[j + u] * xv[u];
              }
              for (; j < jend; ++j) acc = acc + sval[j] * epi.gx(ldv<T2>(Xk + (int64_t)scol[j] * ldx), gxs);
              return acc;
            };
            T2 part = chunk_tree<LPR, T2>((lane / LPR) * LPR, chunk_sum);
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) {
              T2 oth;
#pragma unroll
              for (int i = 0; i < VW; ++i) oth[i] = __shfl_xor(part[i], off, 64);
              part = part + oth;
            }
            if (lane / LPR == hl / LPR) hacc = part;
          }
        }
        if (have && kact) {
          if (heavy) {
            const auto pre = epi.template loadv<T2>((int64_t)r, k);
            epi.template applyv<T2>((int64_t)r, k, hacc, pre);
            if constexpr (Epi::NRED > 0) red = red + epi.template redv<T2>(hacc, pre);
          } else if (kk == 0) {
            row_vec(r, k, a, b, gxs0);
          } else {
            row_vec(r, k, a, b, epi.template gx_load<T2>(k));
          }
        }
      }
    }
  }
  if constexpr (Epi::NRED > 0) {       // lanes with the same rhs pair are LPR apart: fixed-order sum over the rows
    const T2 tot = block_lane_group_sum<LPR>(red, reinterpret_cast<T2*>(smem_raw));
    if (tid < LPR) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
        if (k0 + i < K) epi.partials_ptr()[(int64_t)t * K + k0 + i] = tot[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K > 1, stencil matrices (option "dia"): 2-D blocked kernel with an LDS window of X.
// The diagonal offsets are read as a * S + b (S = grid-line stride, plan->dia_S): a workgroup owns a block of BJ grid
// lines x BI points (BI = 256 / KV, KV = 16-byte rhs vectors per row), stages the (BJ + 2A) x (BI + 2B) rows of X it
// touches in LDS ONCE -- every X element is fetched once per workgroup instead of once per nonzero through L1 -- and
// forms all products from LDS (conflict-free ds_read_b128: the lanes of a wave read 1 KiB of consecutive bytes).
// Lanes = (point, rhs vector) as in the CSR tile kernel; a row's products are added in ascending diagonal order =
// edge order, product and sum rounded separately: bit-identical to the CSR kernels.  Workgroups are numbered so that
// an XCD works through chunks of 8 x 8 blocks: the halo lines of a block are L2 hits (its neighbours' lines).
// ------------------------------------------------------------------------------------------------
struct Dia2D {
  const void* val;            // [D][n_rows]
  const uint16_t* mask;       // [n_rows]
  int32_t D, S, A, B;
  int32_t da[GLA_DIA_MAX], db[GLA_DIA_MAX];
  int64_t n_rows, n_cols;
  int32_t nbi, nbj;           // blocks per grid line, block rows
  int32_t nci, nchunks;       // chunks (8 x 8 blocks) per row of chunks, chunks in all
  uint32_t w_magic;           // ceil(2^32 / W): x / W == umulhi(x, w_magic) for x < 2^16
  int64_t part_off;           // reducing epilogues: block (jb, ib) writes partials[(part_off + jb * nbi + ib) * K + k]
  int64_t aos_off, aos_rows;  // marching Jacobi kernels, AoS first pass: plan row r is row r - aos_off of an AoS tensor of
};                            // aos_rows rows (multi-GPU: the plan covers the rank's rows plus ghost lines, the tensor its own rows)

template <typename T, class Epi, int KV, int DMAX, int BJ>
__global__ __launch_bounds__(GLA_BLOCK) void spmm_dia2d_kernel(Dia2D dv, const T* __restrict__ X, int64_t ldx, Epi epi) {
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int K = KV * VW;
  constexpr int BI = GLA_BLOCK / KV;
  constexpr int NPT = BJ * BI;                    // grid points of a block
  typedef T TV __attribute__((ext_vector_type(VW)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int W = BI + 2 * dv.B, H = BJ + 2 * dv.A;
  TV* sx = reinterpret_cast<TV*>(smem_raw);                                  // [H * W * KV]
  T* sv = reinterpret_cast<T*>(sx + H * W * KV);                             // [DMAX][NPT]
  int* sm = reinterpret_cast<int*>(sv + DMAX * NPT);                         // [NPT] diagonal mask | live bit 16
  // workgroup -> block: XCD e (= blockIdx % 8) takes chunk e of every round of 8 chunks; a chunk is 8 x 8 blocks
  const int bid = blockIdx.x, jx = bid / GLA_XCDS;
  const int chunk = (jx / 64) * GLA_XCDS + bid % GLA_XCDS, within = jx % 64;
  if (chunk >= dv.nchunks) return;
  const int jb = (chunk / dv.nci) * 8 + within / 8, ib = (chunk % dv.nci) * 8 + within % 8;
  if (jb >= dv.nbj || ib >= dv.nbi) return;
  const int tid = threadIdx.x, q = tid % KV;
  const int64_t J0 = (int64_t)jb * BJ, I0 = (int64_t)ib * BI;
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  // ---- stage the X window (branch-free batches of 4 16-byte loads per lane) and the block's values / masks
  {
    const int total = H * W * KV;
    const auto gxs = epi.template gx_load<TV>(q * VW);
    for (int base = 0; base < total; base += 4 * GLA_BLOCK) {
      TV v_[4];
      int f_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int f = base + u * GLA_BLOCK + tid;              // f % KV == q for every slot of this lane
        f = f < total ? f : total - KV + q;              // surplus lanes redo the last slot of their rhs vector
        f_[u] = f;
        const int xr = f / KV;
        const int hh = (int)__umulhi((unsigned)xr, dv.w_magic), ww = xr - hh * W;
        int64_t flat = (J0 + hh - dv.A) * dv.S + (I0 + ww - dv.B);
        flat = min(max(flat, (int64_t)0), dv.n_cols - 1);
        v_[u] = ldv<TV>(X + flat * ldx + q * VW);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sx[f_[u]] = epi.gx(v_[u], gxs);
    }
    for (int p = tid; p < NPT; p += GLA_BLOCK) {
      const int jj = p / BI, ii = p % BI;
      const int64_t r = (J0 + jj) * dv.S + I0 + ii;
      const bool live = I0 + ii < dv.S && r < dv.n_rows;
      const int64_t rc = live ? r : 0;
#pragma unroll
      for (int d = 0; d < DMAX; ++d) sv[d * NPT + p] = val[(int64_t)min(d, dv.D - 1) * dv.n_rows + rc];
      sm[p] = live ? ((int)dv.mask[rc] | 0x10000) : 0;
    }
  }
  int dl[DMAX];                                    // LDS offset of diagonal d relative to the row's own X slot
#pragma unroll
  for (int d = 0; d < DMAX; ++d) { const int dd = min(d, dv.D - 1); dl[d] = (dv.da[dd] * W + dv.db[dd]) * KV; }
  __syncthreads();
  // ---- products and in-order row sums from LDS, fused vertex update; BJ passes of BI rows, 4 passes in flight
  const int ii = tid / KV;
  TV red = TV(T(0));
  [[maybe_unused]] TV red2 = TV(T(0));
  constexpr int G = DMAX > 5 ? 2 : (BJ < 4 ? BJ : 4);      // 9+ diagonals: two passes in flight keep the kernel under 128 VGPRs
#pragma unroll 1
  for (int g = 0; g < BJ; g += G) {
    typename Epi::template PreV<TV> pre[G];
    int m[G];
    int64_t rr[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int p = (g + u) * BI + ii;
      m[u] = sm[p];
      rr[u] = (J0 + g + u) * dv.S + I0 + ii;
      pre[u] = epi.template loadv<TV>((m[u] & 0x10000) ? rr[u] : (int64_t)0, q * VW);
    }
    TV acc[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int p = (g + u) * BI + ii;
      const int xs = ((g + u + dv.A) * W + ii + dv.B) * KV + q;
      TV xv[DMAX];
      T v[DMAX];
#pragma unroll
      for (int d = 0; d < DMAX; ++d) { xv[d] = sx[xs + dl[d]]; v[d] = sv[d * NPT + p]; }
      TV a = TV(T(0));
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {
        const TV nxt = a + v[d] * xv[d];
        a = ((m[u] >> d) & 1) ? nxt : a;
      }
      acc[u] = a;
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (m[u] & 0x10000) {
        epi.template applyv<TV>(rr[u], q * VW, acc[u], pre[u]);
        if constexpr (Epi::NRED > 0) red = red + epi.template redv<TV>(acc[u], pre[u]);
        if constexpr (Epi::NRED > 1) red2 = red2 + epi.template redv2<TV>(acc[u], pre[u]);
      }
    }
  }
  if constexpr (Epi::NRED > 0) {       // lanes with the same rhs vector are KV apart: fixed-order sum over the rows
    const TV tot = block_lane_group_sum<KV>(red, reinterpret_cast<TV*>(smem_raw));
    const int64_t blk = dv.part_off + (int64_t)jb * dv.nbi + ib;
    if (tid < KV) {
#pragma unroll
      for (int i = 0; i < VW; ++i) epi.partials_ptr()[blk * K + q * VW + i] = tot[i];
    }
    if constexpr (Epi::NRED > 1) {
      __syncthreads();
      const TV tot2 = block_lane_group_sum<KV>(red2, reinterpret_cast<TV*>(smem_raw));
      if (tid < KV) {
#pragma unroll
        for (int i = 0; i < VW; ++i) epi.partials2_ptr()[blk * K + q * VW + i] = tot2[i];
      }
    }
  }
}

// EXPERIMENT (gla_set_option("spmm_variant", 12), never the default): the same 2-D blocked kernel with the products
// formed on the matrix cores -- north_star's "MFMA tile for the dense multi-RHS SpMM".  fp64, K = 16, 5-point stencils
// only.  A wave owns tiles of 16 consecutive grid points x 16 right-hand sides; the 16 x 16 diagonal blocks of the lines
// above / below (4 slices each) and the 16 x 18 band of the own line (5 slices) are multiplied as dense 16 x 4 slices by
// v_mfma_f64_16x16x4_f64: 13 MFMAs per tile, 80 of whose 13 * 64 = 832 coefficients are non-zero (9.6 %).  The MFMA's
// fused chain rounds differently from the reference's separate multiply and add (bit-identical only when the products
// are exact, as with the Laplacian's -4 / +1), and a zero coefficient times an Inf / NaN of X pollutes the row, which
// the masked VALU path cannot.  Measured A/B (profiles/r02/ab_mfma_f64_tile.txt): 1.1 - 1.55 x SLOWER than the VALU
// path of the same kernel -- the path is HBM-bound and the matrix pipe spends 90 % of its cycles on zeros.
typedef double D4 __attribute__((ext_vector_type(4)));

template <class Epi, int BJ>
__global__ __launch_bounds__(GLA_BLOCK) void spmm_dia2d_mfma_kernel(Dia2D dv, const double* __restrict__ X, int64_t ldx,
                                                                    Epi epi) {
  typedef double T;
  constexpr int KV = 8, VW = 2, K = 16, DMAX = 9;
  constexpr int BI = GLA_BLOCK / KV;              // 32 points per line
  constexpr int NPT = BJ * BI;
  typedef T TV __attribute__((ext_vector_type(VW)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int W = BI + 2 * dv.B, H = BJ + 2 * dv.A;
  TV* sx = reinterpret_cast<TV*>(smem_raw);
  T* sv = reinterpret_cast<T*>(sx + H * W * KV);
  int* sm = reinterpret_cast<int*>(sv + DMAX * NPT);
  const int bid = blockIdx.x, jx = bid / GLA_XCDS;
  const int chunk = (jx / 64) * GLA_XCDS + bid % GLA_XCDS, within = jx % 64;
  if (chunk >= dv.nchunks) return;
  const int jb = (chunk / dv.nci) * 8 + within / 8, ib = (chunk % dv.nci) * 8 + within % 8;
  if (jb >= dv.nbj || ib >= dv.nbi) return;
  const int tid = threadIdx.x, q = tid % KV;
  const int64_t J0 = (int64_t)jb * BJ, I0 = (int64_t)ib * BI;
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  {                                               // staging: identical to spmm_dia2d_kernel
    const int total = H * W * KV;
    for (int base = 0; base < total; base += 4 * GLA_BLOCK) {
      TV v_[4];
      int f_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int f = base + u * GLA_BLOCK + tid;
        f = f < total ? f : total - KV + q;
        f_[u] = f;
        const int xr = f / KV;
        const int hh = (int)__umulhi((unsigned)xr, dv.w_magic), ww = xr - hh * W;
        int64_t flat = (J0 + hh - dv.A) * dv.S + (I0 + ww - dv.B);
        flat = min(max(flat, (int64_t)0), dv.n_cols - 1);
        v_[u] = ldv<TV>(X + flat * ldx + q * VW);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sx[f_[u]] = v_[u];
    }
    for (int p = tid; p < NPT; p += GLA_BLOCK) {
      const int jj = p / BI, ii = p % BI;
      const int64_t r = (J0 + jj) * dv.S + I0 + ii;
      const bool live = I0 + ii < dv.S && r < dv.n_rows;
      const int64_t rc = live ? r : 0;
      for (int d = 0; d < dv.D; ++d) sv[d * NPT + p] = val[(int64_t)d * dv.n_rows + rc];
      sm[p] = live ? ((int)dv.mask[rc] | 0x10000) : 0;
    }
  }
  __syncthreads();
  const T* sxs = reinterpret_cast<const T*>(sx);  // X window as scalars: slot * K + rhs
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kk = lane >> 4;        // A operand: row m, column kk of the 16 x 4 slice; B: row kk, column m
  // 5-point structure (checked by the launcher): d = 0: a = -1; d = 1, 2, 3: a = 0, b = -1, 0, +1; d = 4: a = +1
#pragma unroll 1
  for (int tile = wave; tile < BJ * (BI / 16); tile += GLA_BLOCK / 64) {
    const int jj = tile / (BI / 16), i0 = (tile % (BI / 16)) * 16;
    const int pm = jj * BI + i0 + m;              // this lane's A row (grid point) inside the block
    const int msk = sm[pm];
    double vd[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) vd[d] = ((msk >> d) & 1) ? sv[d * NPT + pm] : 0.0;
    D4 acc = {0.0, 0.0, 0.0, 0.0};
    const T* xl[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) xl[g] = sxs + (int64_t)((jj + dv.A + g - 1) * W + dv.B + i0) * K + m;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {              // line above: 16 x 16 diagonal block, X points i0 + 4 s + kk
      const int xi = 4 * s4 + kk;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xi == m ? vd[0] : 0.0, xl[0][xi * K], acc, 0, 0, 0);
    }
#pragma unroll
    for (int s4 = 0; s4 < 5; ++s4) {              // own line: 16 x 18 band, X points i0 - 1 + 4 s + kk
      const int xi = 4 * s4 + kk - 1;
      const double av = xi - m == -1 ? vd[1] : (xi - m == 0 ? vd[2] : (xi - m == 1 ? vd[3] : 0.0));
      const int xc = min(xi, BI - i0 + dv.B - 1);              // the two padding rows of the last slice stay in the window
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xl[1][xc * K], acc, 0, 0, 0);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {              // line below
      const int xi = 4 * s4 + kk;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xi == m ? vd[4] : 0.0, xl[2][xi * K], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * i + kk;                 // D layout of v_mfma_f64_16x16x4_f64: register i, lane l -> row 4 i + l / 16, column l % 16
      const int pi = jj * BI + i0 + row;
      if (sm[pi] & 0x10000) {
        const int64_t r = (J0 + jj) * dv.S + I0 + i0 + row;
        epi((int64_t)r, m, acc[i]);
      }
    }
  }
}

// > 0: the blocked stencil kernel takes this launch, with that many 16-byte rhs vectors per row
template <typename T, class Epi>
int dia2d_kv(const gla_csr* A, int K, const T* X, int64_t ldx, const Epi& epi, int which) {
  constexpr int VW = 16 / (int)sizeof(T);
  if (!(K > 1 && which == 0 && A->dia_state == 1 && A->dia_val && A->dia_S > 0 && g_tuning.dia &&
        g_tuning.spmm_variant != 9 && A->dia_row_lo == 0 && A->dia_row_hi == A->n_rows && A->n_rows == A->n_cols))
    return 0;
  const int kv = K / VW;
  if (K % VW != 0 || !(kv == 1 || kv == 2 || kv == 4 || kv == 8) || ldx % VW != 0 || !al16(X) || !epi.vec_ok(VW)) return 0;
  const int bj = g_tuning.spmm_variant == 10 ? 8 : 4, bi = GLA_BLOCK / kv;
  const int dmax = A->dia_D <= 5 ? 5 : (A->dia_D <= 9 ? 9 : GLA_DIA_MAX);
  const size_t lds = (size_t)(bj + 2 * A->dia_A) * (bi + 2 * A->dia_B) * kv * 16 + (size_t)dmax * bj * bi * sizeof(T) +
                     (size_t)bj * bi * sizeof(int);
  return lds <= 64 * 1024 ? kv : 0;
}

inline int64_t dia2d_nblocks(const gla_csr* A, int kv) {
  const int bj = g_tuning.spmm_variant == 10 ? 8 : 4, bi = GLA_BLOCK / kv;
  const int64_t lines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  return ((lines + bj - 1) / bj) * ((A->dia_S + bi - 1) / bi);
}

template <typename T, class Epi, int KV, int BJ>
int launch_dia2d(gla_csr* A, const T* X, int64_t ldx, const Epi& epi, hipStream_t st) {
  constexpr int BI = GLA_BLOCK / KV;
  Dia2D dv;
  dv.val = A->dia_val;
  dv.mask = A->dia_mask;
  dv.D = A->dia_D; dv.S = A->dia_S; dv.A = A->dia_A; dv.B = A->dia_B;
  for (int i = 0; i < GLA_DIA_MAX; ++i) { dv.da[i] = i < A->dia_D ? A->dia_a[i] : 0; dv.db[i] = i < A->dia_D ? A->dia_b[i] : 0; }
  dv.n_rows = A->n_rows; dv.n_cols = A->n_cols;
  const int64_t lines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  dv.nbi = (int)((A->dia_S + BI - 1) / BI);
  dv.nbj = (int)((lines + BJ - 1) / BJ);
  dv.nci = (dv.nbi + 7) / 8;
  const int ncj = (dv.nbj + 7) / 8;
  dv.nchunks = dv.nci * ncj;
  const int W = BI + 2 * dv.B, H = BJ + 2 * dv.A;
  dv.w_magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)W + 1);
  dv.part_off = 0;                        // the only kernel of its launch: partial rows [0, blocks)
  const int dmax = A->dia_D <= 5 ? 5 : (A->dia_D <= 9 ? 9 : GLA_DIA_MAX);
  const size_t lds = (size_t)H * W * KV * 16 + (size_t)dmax * BJ * BI * sizeof(T) + (size_t)BJ * BI * sizeof(int);
  const unsigned grid = (unsigned)(((dv.nchunks + GLA_XCDS - 1) / GLA_XCDS) * GLA_XCDS * 64);
  if constexpr (sizeof(T) == 8 && KV == 8 && Epi::NRED == 0) {
    const bool five_point = A->dia_D == 5 && A->dia_A == 1 && A->dia_B == 1 && A->dia_a[0] == -1 && A->dia_b[0] == 0 &&
                            A->dia_a[1] == 0 && A->dia_b[1] == -1 && A->dia_a[2] == 0 && A->dia_b[2] == 0 &&
                            A->dia_a[3] == 0 && A->dia_b[3] == 1 && A->dia_a[4] == 1 && A->dia_b[4] == 0;
    if (g_tuning.spmm_variant == 12 && five_point) {        // MFMA tile experiment (see the kernel)
      const size_t lds_m = (size_t)H * W * KV * 16 + (size_t)9 * BJ * BI * sizeof(T) + (size_t)BJ * BI * sizeof(int);
      hipLaunchKernelGGL((spmm_dia2d_mfma_kernel<Epi, BJ>), dim3(grid), dim3(GLA_BLOCK), lds_m, st, dv,
                         (const double*)X, ldx, epi);
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
  if (dmax == 5)
    hipLaunchKernelGGL((spmm_dia2d_kernel<T, Epi, KV, 5, BJ>), dim3(grid), dim3(GLA_BLOCK), lds, st, dv, X, ldx, epi);
  else if (dmax == 9)
    hipLaunchKernelGGL((spmm_dia2d_kernel<T, Epi, KV, 9, BJ>), dim3(grid), dim3(GLA_BLOCK), lds, st, dv, X, ldx, epi);
  else
    hipLaunchKernelGGL((spmm_dia2d_kernel<T, Epi, KV, GLA_DIA_MAX, BJ>), dim3(grid), dim3(GLA_BLOCK), lds, st, dv, X,
                       ldx, epi);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

// The tail of PowerMethodGNN.forward in one pass of the blocked stencil kernel (PowerMethodGNN.py:187-294): b = y / n (the
// last iteration's layer 3, applied while X is staged exactly as in EpiPower), then the Rayleigh layers' A b, n_A = sum b (A b)
// and sum b^2 as two block partials, and the module's [b, b^2] result written straight out -- A b itself is never stored.
template <typename T>
struct EpiRayleigh : EpiPower<T> {
  const T* x; int64_t ldx; T* bout; int64_t ldb; T* b2out; int64_t ldb2; T* partials2;
  static constexpr int NRED = 2;
  template <typename V> struct PreV { V b; };
  template <typename V> __device__ __forceinline__ PreV<V> loadv(int64_t r, int k) const {
    return PreV<V>{this->template gx<V>(ldv<V>(x + r * ldx + k), this->template gx_load<V>(k))};
  }
  template <typename V> __device__ __forceinline__ void applyv(int64_t r, int k, V, const PreV<V>& p) const {
    stv<V>(bout + r * ldb + k, p.b);
    stv<V>(b2out + r * ldb2 + k, p.b * p.b);
  }
  template <typename V> __device__ __forceinline__ V redv(V acc, const PreV<V>& p) const { return p.b * acc; }
  template <typename V> __device__ __forceinline__ V redv2(V, const PreV<V>& p) const { return p.b * p.b; }
  __device__ __forceinline__ T* partials2_ptr() const { return partials2; }
  bool vec_ok(int w) const { return al16(x) && al16(bout) && al16(b2out) && ldx % w == 0 && ldb % w == 0 && ldb2 % w == 0; }
};

// number of 2-D blocks the blocked stencil kernel may use (upper bound over its block shapes): partials of reducing
// epilogues need one slot per block
inline int64_t dia2d_blocks_bound(const gla_csr* A) {
  if (A->dia_state != 1 || A->dia_S <= 0) return 0;
  const int64_t lines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  return ((lines + 3) / 4 + 1) * ((int64_t)(A->dia_S + 31) / 32 + 1);
}

// ------------------------------------------------------------------------------------------------
// K = 1, 2-D stencil plans: TS = 2, 4 or 5 Jacobi sweeps per pass over the data (temporal blocking, gla_jacobi_sweeps).
// A sweep is HBM-bound at 74 B per row (5-point: 40 B of values, 2 B mask, b, A_ii, x in, x out); x^(k+1) .. x^(k+TS-1)
// are only ever needed to form the next iterate, so a workgroup whose 256 lanes cover a strip of 256 points marches
// down `lines` grid lines keeping four lines of each intermediate iterate in LDS rings and the coefficients of the
// last TS lines in registers: every matrix value, b and A_ii is read ONCE per TS sweeps and the intermediate
// iterates never touch HBM -- 74 / TS bytes per row and sweep, plus the overlap: iterate t is valid on the strip minus
// t * B points either side (strips overlap by 2 * TS * B points), blocks overlap by 2 * (TS - 1) coefficient lines.
// Every load is a coalesced 2 KiB row segment.  Per row exactly the arithmetic of spmv_dia_kernel + EpiJacobi, TS times:
// in-order masked sum over the diagonals, then x + (w * (b - sum)) / A_ii (JacobiGNN.py:119) -- bit-identical to TS
// single sweeps.
//   step jl:  ring[0] <- x^(k) line jl + 1 (prefetched) | prefetch x^(k) line jl + 2, coefficients of line jl + 1 |
//             for t = 1 .. TS:  sync;  x^(k+t) on line jl - (t - 1) from ring[t - 1] lines -1 .. +1 -> ring[t] (t = TS: HBM)
// ------------------------------------------------------------------------------------------------
// Line block lb of a marching kernel -> first line and height.  edge = 0: uniform blocks of `lines` lines.  edge > 0 (the
// multi-GPU interior / boundary split): the first and the last block are `edge` lines high, the nlb - 2 blocks between them
// `lines` -- the two edge blocks are all that has to wait for the ghost lines, so they are kept as short as the split allows.
__device__ __forceinline__ void march_block(int lb, int lines, int edge, int nlb, int64_t nlines, int64_t& J0, int& L) {
  if (edge == 0) { J0 = (int64_t)lb * lines; L = (int)min((int64_t)lines, nlines - J0); }
  else if (lb == 0) { J0 = 0; L = edge; }
  else if (lb == nlb - 1) { J0 = nlines - edge; L = edge; }
  else { J0 = edge + (int64_t)(lb - 1) * lines; L = (int)min((int64_t)lines, nlines - edge - J0); }
}

template <typename T, int DMAX>
struct DiaCoef {                                  // one row's matrix data
  T v[DMAX], b, dg;
  int m, lv;                                      // diagonal mask exactly as loaded (no arithmetic on it before its first use: the
};                                                // load must stay in flight across the line step); lv = row exists

// AOS = true (first pass of gla_jacobi_sweeps_aos): A_ii, b and x^(k) are columns 0, 1, 2 of the reference's row-major
// vertex attribute tensor (`x` points at it, leading dimension ldva; JacobiGNN.py:104-106); the lane that owns a row also
// leaves contiguous copies of its A_ii and b in diag_out / b_out for the later passes -- the AoS split costs 16 B per row
// of extra writes inside this pass instead of a separate 48 B per row pass.
template <typename T, int DMAX, int TS, bool AOS = false>
__global__ __launch_bounds__(GLA_BLOCK) void jacobi_fused_dia2d_kernel(Dia2D dv, const T* __restrict__ diag,
                                                                       const T* __restrict__ bvec,
                                                                       const T* __restrict__ x, T* __restrict__ xo, T w,
                                                                       int lines, int nstrips, int64_t nlines, TileMap tm,
                                                                       int tile_base0, int tile_split, int tile_base1,
                                                                       int64_t ldva = 1, T* __restrict__ diag_out = nullptr,
                                                                       T* __restrict__ b_out = nullptr) {
  constexpr int BW = GLA_BLOCK;                   // window width = lanes
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* ring = reinterpret_cast<T*>(smem_raw);       // [TS][4][BW]: ring t holds four lines of x^(k+t)
  const int slot = map_tile(tm);
  if (slot < 0) return;
  // the launch covers tiles base0 + [0, split) and base1 + [0, n - split) (all / interior / first + last line block)
  const int blk = slot < tile_split ? tile_base0 + slot : tile_base1 + (slot - tile_split);
  const int B = dv.B, halo = TS * B, bi_eff = BW - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t P0 = (int64_t)strip * bi_eff - halo;   // lane tid <-> point P0 + tid
  int64_t J0;
  int L;
  march_block(lb, lines, dv.nbi, dv.nbj, nlines, J0, L);
  const int tid = threadIdx.x;
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  int da[DMAX], db[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) { const int dd = min(d, dv.D - 1); da[d] = dv.da[dd]; db[d] = dv.db[dd]; }

  auto x0_at = [&](int jl) -> T {                 // x^(k) at line J0 + jl, this lane's point (clamped: never used dead)
    int64_t flat = (J0 + jl) * dv.S + P0 + tid;
    flat = min(max(flat, (int64_t)0), dv.n_cols - 1);
    if constexpr (AOS) return x[min(max(flat - dv.aos_off, (int64_t)0), dv.aos_rows - 1) * ldva + 2];
    else return x[flat];
  };
  auto coef_at = [&](int jl) -> DiaCoef<T, DMAX> {   // row (line J0 + jl, this lane's point); dead rows: mask 0
    const int64_t gj = J0 + jl, gi = P0 + tid, r = gj * dv.S + gi;
    const bool live = gj >= 0 && gj < nlines && gi >= 0 && gi < dv.S && r < dv.n_rows;
    const int64_t rc = live ? r : 0;
    DiaCoef<T, DMAX> c;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) c.v[d] = val[(int64_t)min(d, dv.D - 1) * dv.n_rows + rc];
    if constexpr (AOS) {
      const int64_t ra = min(max(rc - dv.aos_off, (int64_t)0), dv.aos_rows - 1);
      c.dg = x[ra * ldva];
      c.b = x[ra * ldva + 1];
    }
    else { c.b = bvec[rc]; c.dg = diag[rc]; }
    c.m = (int)dv.mask[rc];
    c.lv = live;
    return c;
  };
  // one Jacobi update of this lane's row on line jl from ring r (lines jl - 1 .. jl + 1; slot of line j: (j + 8) & 3)
  auto update = [&](const DiaCoef<T, DMAX>& c, const T* r, int jl) -> T {
    T acc = T(0);
    const int m = c.lv ? c.m : 0;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
      const int pos = min(max(tid + db[d], 0), BW - 1);            // window edge lanes: their results are never used
      const T xv = r[((jl + da[d] + 8) & 3) * BW + pos];
      const T nxt = acc + c.v[d] * xv;
      acc = ((m >> d) & 1) ? nxt : acc;
    }
    const T xc = r[((jl + 8) & 3) * BW + tid];
    return xc + (w * (c.b - acc)) / c.dg;
  };

  // x^(k+1) is formed on lines -(TS-1) .. L-1+(TS-1): it needs x^(k) on lines -TS .. L-1+TS
  const int j_first = -(TS - 1), j_last = L - 1 + (TS - 1);
  ring[((j_first - 1 + 8) & 3) * BW + tid] = x0_at(j_first - 1);
  ring[((j_first + 8) & 3) * BW + tid] = x0_at(j_first);
  T xpre = x0_at(j_first + 1);
  DiaCoef<T, DMAX> c[TS];                         // c[t]: coefficients of line jl - t
  c[0] = coef_at(j_first);
#pragma unroll
  for (int t = 1; t < TS; ++t) c[t] = c[0];
  DiaCoef<T, DMAX> nxt = c[0];
#pragma unroll 1
  for (int jl = j_first; jl <= j_last; ++jl) {
    ring[((jl + 1 + 8) & 3) * BW + tid] = xpre;
    if (jl + 2 <= j_last + 1) xpre = x0_at(jl + 2);                // prefetch for the next step
    if (jl + 1 <= j_last) nxt = coef_at(jl + 1);
    if constexpr (AOS) {                           // the row's owner leaves contiguous copies of A_ii and b (c[0]: line jl)
      if (jl >= 0 && jl < L && c[0].lv && tid >= halo && tid < BW - halo) {
        const int64_t r = (J0 + jl) * dv.S + P0 + tid;
        diag_out[r] = c[0].dg;
        b_out[r] = c[0].b;
      }
    }
#pragma unroll
    for (int t = 1; t <= TS; ++t) {
      __syncthreads();
      const int line = jl - (t - 1);              // x^(k+t) is needed on lines -(TS-t) .. L-1+(TS-t)
      if (line >= -(TS - t) && line <= L - 1 + (TS - t)) {
        const T xn = update(c[t - 1], ring + (size_t)(t - 1) * 4 * BW, line);
        if (t < TS) ring[(size_t)t * 4 * BW + ((line + 8) & 3) * BW + tid] = xn;
        else if (c[t - 1].lv && tid >= halo && tid < BW - halo) xo[(J0 + line) * dv.S + P0 + tid] = xn;
      }
    }
#pragma unroll
    for (int t = TS - 1; t >= 1; --t) c[t] = c[t - 1];
    c[0] = nxt;
  }
}

// (the march for the canonical 5-point pattern, jacobi_march5_kernel, lives in gla_march5.hip: gla_march5_pass)

// ------------------------------------------------------------------------------------------------
// The march for the canonical 9-point pattern (diagonals (-1,-1) (-1,0) (-1,1) (0,-1) (0,0) (0,1) (1,-1) (1,0) (1,1): the Q1
// FEM operator of heateqnfem2dfun).  As jacobi_march5_kernel -- own column of every level in registers, static coefficient
// set names, wave-uniform full-pattern fast path -- but a row also needs the i -/+ 1 neighbours of the line BELOW, which the
// level underneath produces in the same line step: three lines per level in LDS (slot = line mod 3), one barrier per level
// and line step, three ds_read2 per update.  Same arithmetic as spmv_dia_kernel + EpiJacobi: bit-identical.
// ------------------------------------------------------------------------------------------------
// SYM (round 4): a symmetric value set (gla_csr_set_values found A_ij == A_ji bit for bit along all four diagonal pairs; option
// "sym") streams only the centre, E, SW, S and SE diagonals: W = E of the point before, NW / NE = SE / SW of the line above at
// the points before / after (cache hits: those lines were streamed a line step ago), N = S of the line above (the lane's own
// previous set).  40 instead of 72 bytes of values per row.
template <typename T, int TS, bool AOS = false, bool RES = false, bool SYM = false>
__global__ __launch_bounds__(GLA_BLOCK) void jacobi_march9_kernel(Dia2D dv, const T* __restrict__ diag,
                                                                  const T* __restrict__ bvec, const T* __restrict__ x,
                                                                  T* __restrict__ xo, T w, int lines, int nstrips,
                                                                  int64_t nlines, TileMap tm, int tile_base0, int tile_split,
                                                                  int tile_base1, int64_t ldva = 1,
                                                                  T* __restrict__ diag_out = nullptr,
                                                                  T* __restrict__ b_out = nullptr) {
  constexpr int BW = GLA_BLOCK, LW = BW + 2;      // a line in LDS: lane tid at [1 + tid], one pad either side
  constexpr int LV = TS + (RES ? 1 : 0);          // levels: TS sweeps; RES: + one level that forms r = b - A x^(k+TS) (-> b_out)
  constexpr int NC = LV + 1, U = NC;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* buf = reinterpret_cast<T*>(smem_raw);        // [TS][3][LW]: buf[t][s] = line of x^(k+t) with line mod 3 == s
  const int slot = map_tile(tm);
  if (slot < 0) return;
  const int blk = slot < tile_split ? tile_base0 + slot : tile_base1 + (slot - tile_split);
  constexpr int halo = LV, bi_eff = BW - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t P0 = (int64_t)strip * bi_eff - halo;
  int64_t J0;
  int L;
  march_block(lb, lines, dv.nbi, dv.nbj, nlines, J0, L);
  const int tid = threadIdx.x;
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  // A_ii: the caller's array, or (diag == NULL: "A_ii is the matrix diagonal", as in every use the reference makes of JacobiGNN)
  // the plan's own centre diagonal -- the row's v[4], an L1 hit: no A_ii stream
  const T* const dg_base = diag ? diag : val + 4 * dv.n_rows;
  const int64_t gi = P0 + tid;
  const bool in_i = gi >= 0 && gi < dv.S;
  const bool owner = tid >= halo && tid < BW - halo;

  auto x0_at = [&](int jl) -> T {
    int64_t flat = (J0 + jl) * dv.S + gi;
    flat = min(max(flat, (int64_t)0), dv.n_cols - 1);
    if constexpr (AOS) return x[min(max(flat - dv.aos_off, (int64_t)0), dv.aos_rows - 1) * ldva + 2];
    else return x[flat];
  };
  auto coef_at = [&](int jl) -> DiaCoef<T, 9> {
    const int64_t gj = J0 + jl, r = gj * dv.S + gi;
    const bool live = in_i && gj >= 0 && gj < nlines && r < dv.n_rows;
    const int64_t rc = live ? r : 0;
    DiaCoef<T, 9> c;
    if constexpr (SYM) {
#pragma unroll
      for (int d = 4; d < 9; ++d) c.v[d] = val[(int64_t)d * dv.n_rows + rc];
      const int64_t rw_ = rc - 1, rnw = rc - dv.S - 1, rne = rc - dv.S + 1;
      const T vw = val[5 * dv.n_rows + max(rw_, (int64_t)0)];
      const T vnw = val[8 * dv.n_rows + max(rnw, (int64_t)0)];
      const T vne = val[6 * dv.n_rows + max(rne, (int64_t)0)];
      c.v[3] = rw_ >= 0 ? vw : T(0);
      c.v[0] = rnw >= 0 ? vnw : T(0);
      c.v[2] = rne >= 0 ? vne : T(0);
      c.v[1] = T(0);                               // north: filled in at the set's first use
    } else {
#pragma unroll
      for (int d = 0; d < 9; ++d) c.v[d] = val[(int64_t)d * dv.n_rows + rc];
    }
    if constexpr (AOS) {
      const int64_t ra = min(max(rc - dv.aos_off, (int64_t)0), dv.aos_rows - 1);
      c.dg = x[ra * ldva];
      c.b = x[ra * ldva + 1];
    }
    else { c.b = bvec[rc]; c.dg = dg_base[rc]; }
    c.m = (int)dv.mask[rc];
    c.lv = live;
    return c;
  };
  // xs: the nine neighbours in diagonal order
  auto update = [&](const DiaCoef<T, 9>& c, bool full, const T (&xs)[9], bool residual) -> T {
    T acc = T(0);
    if (full) {
#pragma unroll
      for (int d = 0; d < 9; ++d) acc = acc + c.v[d] * xs[d];
    } else {
      const int m = c.lv ? c.m : 0;
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        const T nxt = acc + c.v[d] * xs[d];
        acc = ((m >> d) & 1) ? nxt : acc;
      }
    }
    if (residual) return c.b - acc;                 // (GNNResidual.py:97-118, as EpiResidual)
    return xs[4] + (w * (c.b - acc)) / c.dg;
  };

  constexpr int j_first = -(LV - 1);
  const int j_last = L - 1 + (LV - 1);
  T h[LV][3];                                      // h[t]: x^(k+t) of this lane on the last three lines level t produced
#pragma unroll
  for (int t = 0; t < LV; ++t) h[t][0] = h[t][1] = h[t][2] = T(0);
  h[0][1] = x0_at(j_first - 1);
  h[0][2] = x0_at(j_first);
  buf[((j_first - 1 + 63) % 3) * LW + 1 + tid] = h[0][1];
  buf[((j_first + 63) % 3) * LW + 1 + tid] = h[0][2];
  T xpre = x0_at(j_first + 1);
  DiaCoef<T, 9> c[NC];                             // set of line j_first + s lives in c[s % NC]
  bool cfull[NC];
  c[0] = coef_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
  for (int s = 0; s < NC; ++s) cfull[s] = false;
  if constexpr (SYM) {                             // (the set "before" the first one: its south coefficient)
    const int64_t rp = (J0 + j_first - 1) * dv.S + gi;
    c[NC - 1].v[7] = val[7 * dv.n_rows + min(max(rp, (int64_t)0), dv.n_rows - 1)];
  }

  auto do_step = [&](const int jl, const int u, const int nact) {
    if ((jl & 1) ^ (((int)blockIdx.x / GLA_XCDS) >> 5 & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (priority alternation between a CU's two workgroups: gla_march5.hip)
    // LDS rows of the lines jl + 1, jl, jl - 1 (mod 3): level t reads lines jl - t + 2, jl - t + 1, jl - t of level t - 1 --
    // the same three rows rotated by t - 1
    const int s1 = (jl + 1 + 63) % 3, s0 = (jl + 63) % 3, s2 = (jl - 1 + 63) % 3;
    T* const rw[3] = {buf + s1 * LW + tid, buf + s0 * LW + tid, buf + s2 * LW + tid};   // [0] left, [1] own, [2] right
    h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
    rw[0][1] = xpre;                               // level 0, line jl + 1
    xpre = x0_at(jl + 2);                          // clamped / dead beyond the block: harmless, never used
    c[(u + 1) % NC] = coef_at(jl + 1);
    if constexpr (SYM)                             // north(line jl) = south(line jl - 1); the grid's first line has none
      c[u % NC].v[1] = J0 + jl <= 0 ? T(0) : c[(u + NC - 1) % NC].v[7];
    cfull[u % NC] = __all(c[u % NC].lv && (c[u % NC].m & 0x1ff) == 0x1ff) != 0;   // (at first use: loaded a step ago)
    if constexpr (AOS) {
      const DiaCoef<T, 9>& c0 = c[u % NC];
      if (jl >= 0 && jl < L && c0.lv && owner) {
        const int64_t r = (J0 + jl) * dv.S + gi;
        diag_out[r] = c0.dg;
        b_out[r] = c0.b;
      }
    }
#pragma unroll
    for (int t = 1; t <= LV; ++t) {
      __syncthreads();
      if (t <= nact) {
        // lines line + 1, line, line - 1 of level t - 1 (line = jl - (t - 1)): rows rotated by t - 1
        T* const below = rw[(t - 1) % 3] + (t - 1) * 3 * LW;
        T* const same = rw[t % 3] + (t - 1) * 3 * LW;
        T* const above = rw[(t + 1) % 3] + (t - 1) * 3 * LW;
        const int ci = ((u - (t - 1)) % NC + NC) % NC;
        const T xs[9] = {above[0], h[t - 1][0], above[2], same[0], h[t - 1][1], same[2], below[0], h[t - 1][2], below[2]};
        const T xn = update(c[ci], cfull[ci], xs, RES && t == LV);
        if (t < LV) {
          h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
          same[3 * LW + 1] = xn;                   // level t, same line (same slot)
        }
        if (t == TS) {                             // the iterate after TS sweeps (RES: its level also runs one line either side)
          const int line = jl - (TS - 1);
          if (c[ci].lv && owner && (!RES || (line >= 0 && line < L))) xo[(J0 + line) * dv.S + gi] = xn;
        } else if (RES && t == LV) {               // ... and its residual
          if (c[ci].lv && owner) b_out[(J0 + jl - (LV - 1)) * dv.S + gi] = xn;
        }
      }
    }
  };
#pragma unroll
  for (int s = 0; s < 2 * (LV - 1); ++s) do_step(j_first + s, s % U, (s + 2) / 2);
  constexpr int u0 = (2 * (LV - 1)) % U;
#pragma unroll 1
  for (int jl0 = LV - 1;; jl0 += U) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      if (jl0 + q > j_last) return;
      do_step(jl0 + q, (u0 + q) % U, LV);
    }
  }
}

template <typename T>
bool jacobi_fused_ok(const gla_csr* A) {
  return g_tuning.jacobi_fuse && g_tuning.dia && g_tuning.spmv_variant == 0 && A->dia_state == 1 && A->dia_val &&
         A->dia_S > 0 && A->dia_nowrap && A->dia_A == 1 && A->dia_B >= 1 && A->dia_B <= 2 && A->dia_D <= 9 && A->dia_row_lo == 0 &&
         A->dia_row_hi == A->n_rows && A->n_rows == A->n_cols && !(g_tuning.compress && (A->dia_const || A->packed));
}

// the canonical 5-point pattern jacobi_march5_kernel is written for (option jacobi_march = 0: generic kernel instead)
inline bool jacobi_canon(const gla_csr* A) { return g_tuning.jacobi_march && gla_dia_canon5(A); }
// ... and the canonical 9-point pattern of jacobi_march9_kernel (2 .. 6 sweeps per pass)
inline bool jacobi_canon9(const gla_csr* A) { return g_tuning.jacobi_march && gla_dia_canon9(A); }
constexpr int JACOBI_TS_MAX9 = 6;

inline bool jacobi_sym9(const gla_csr* A) { return g_tuning.sym && gla_dia_canon9(A) && (gla_dia_sym_bits(A) & 15) == 15; }
// the SYM instances that keep two waves per SIMD (256 registers; the others are not built: those passes read all nine diagonals)
template <typename T, int TS, bool AOS, bool RES>
constexpr bool march9_sym_fits() { return sizeof(T) == 4 || (AOS || RES ? TS <= 4 : TS <= 5); }

template <typename T, int TS, bool AOS>
int jacobi_march9_blocks_per_cu() {
  static int cached = 0;
  if (!cached) {
    int nb = 0;
    if constexpr (TS <= JACOBI_TS_MAX9) {
      const size_t lds9 = (size_t)TS * 3 * (GLA_BLOCK + 2) * sizeof(T);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, jacobi_march9_kernel<T, TS, AOS>, GLA_BLOCK, lds9) != hipSuccess) nb = 0;
    }
    if (nb < 1) { (void)hipGetLastError(); nb = 2; }
    cached = nb;
  }
  return cached;
}

template <typename T, int TS>
int launch_jacobi_fused(gla_csr* A, const T* diag, const T* b, const T* x, T* xo, T w, hipStream_t st, int part = 0,
                        int64_t ldva = 0, T* diag_out = nullptr, T* b_out = nullptr, int ghost = TS, int64_t aos_off = 0,
                        int64_t aos_rows = -1, int64_t ldd = 1, int* dg_flag = nullptr, int dg_epoch = 0) {
  constexpr bool generic_ts = TS == 2 || TS == 4 || TS == 5;      // what jacobi_fused_dia2d_kernel is instantiated for
  const bool canon = jacobi_canon(A);
  const bool canon9 = TS <= JACOBI_TS_MAX9 && jacobi_canon9(A);
  if (canon) {                                      // register-history march for the canonical 5-point pattern (gla_march5.hip)
    gla_march5_args a;
    a.A = A; a.dtype = sizeof(T) == 8 ? GLA_F64 : GLA_F32; a.ts = TS; a.res = false;
    a.diag = diag; a.b = b; a.x = x; a.xo = xo; a.w = (double)w; a.st = st; a.part = part;
    // (ghost >> 16, internal callers only -- gla_dist.hip's staggered passes: an explicit height of the first / last line block)
    a.ldva = ldva; a.diag_out = diag_out; a.b_out = b_out; a.ghost = ghost & 0xffff; a.edge_h = ghost >> 16; a.aos_off = aos_off; a.aos_rows = aos_rows;
    a.ldd = ldd; a.dg_flag = dg_flag; a.dg_epoch = dg_epoch;
    if (part < 0 || part > 2 || a.ghost < TS || (a.edge_h && (part == 0 || a.edge_h < a.ghost + TS))) return GLA_ERR_INVALID_ARG;
    return gla_march5_pass(a);
  }
  if (ghost >> 16) return GLA_ERR_UNSUPPORTED;
  // (A_ii from the centre diagonal / a strided array / by a device flag: the 5-point march kernel only)
  if (ldd != 1 || dg_flag || (ldva <= 0 && !diag && !canon9)) return GLA_ERR_UNSUPPORTED;
  if (!canon && !canon9 && !generic_ts) return GLA_ERR_UNSUPPORTED;
  if (part < 0 || part > 2 || ghost < TS) return GLA_ERR_INVALID_ARG;
  Dia2D dv;
  dv.val = A->dia_val;
  dv.mask = A->dia_mask;
  dv.D = A->dia_D; dv.S = A->dia_S; dv.A = A->dia_A; dv.B = A->dia_B;
  for (int i = 0; i < GLA_DIA_MAX; ++i) { dv.da[i] = i < A->dia_D ? A->dia_a[i] : 0; dv.db[i] = i < A->dia_D ? A->dia_b[i] : 0; }
  dv.n_rows = A->n_rows; dv.n_cols = A->n_cols;
  dv.nbi = dv.nbj = dv.nci = dv.nchunks = 0; dv.w_magic = 0; dv.part_off = 0;
  dv.aos_off = aos_off;
  dv.aos_rows = aos_rows >= 0 ? aos_rows : A->n_rows;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  const int bi_eff = GLA_BLOCK - 2 * TS * dv.B;
  const int nstrips = (int)((A->dia_S + bi_eff - 1) / bi_eff);
  // Interior / boundary split (part 1 / 2, the multi-GPU route): the first and the last `ghost` grid lines of x arrive while
  // the interior blocks run, so no interior block may read them: the first and the last line block are ghost + TS lines
  // high (an interior block reads x from TS lines above its first line), everything between is split as for part 0.
  // Grids too short for that (fewer than 2 edge blocks + 1 line) count as boundary as a whole.
  const int edge_h = ghost + TS > 2 * ghost ? ghost + TS : 2 * ghost;     // (at least 2 x ghost: gla_march5.hip)
  const int edge = part != 0 && nlines > 2 * (int64_t)edge_h ? edge_h : 0;
  const int64_t inner_lines = nlines - 2 * edge;
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) {
    // block height: the fewest lines per workgroup (least serial work, but every block re-reads 2 (TS - 1) coefficient
    // lines) for which all workgroups are resident at once -- 256 CUs x the workgroups per CU the register budget allows
    int64_t per_cu;
    if (canon9) {
      per_cu = ldva > 0 ? jacobi_march9_blocks_per_cu<T, TS, true>() : jacobi_march9_blocks_per_cu<T, TS, false>();
      if (per_cu > 2) per_cu = 2;
    }
    else per_cu = sizeof(T) == 8 ? (A->dia_D <= 5 ? (TS >= 4 ? 4 : 7) : 4) : (A->dia_D <= 5 ? 8 : 5);
    int64_t slots = 256 * per_cu;                     // (interior / boundary split: room for the two edge line blocks PER XCD, gla_march5.hip)
    if (edge && slots > 4 * (int64_t)nstrips) {
      const int64_t edge_x = (2 * (int64_t)nstrips + GLA_XCDS - 1) / GLA_XCDS;
      const int64_t int_x = 32 * (int64_t)per_cu - (g_tuning.edge_prio ? 1 : 0) - edge_x;
      slots = g_tuning.edge_prio ? GLA_XCDS * int_x : slots - 2 * (int64_t)nstrips;
    }
    lines = gla_pick_block_lines(inner_lines, nstrips, slots, 2 * (TS - 1), 2 * TS);
    if (lines > 1024) lines = 1024;
  }
  if (lines < 2 * TS && canon9) lines = 2 * TS;
  const int nlb_all = (int)((inner_lines + lines - 1) / lines) + (edge ? 2 : 0);
  dv.nbi = edge;
  dv.nbj = nlb_all;
  int base0 = 0, split = nlb_all * nstrips, base1 = 0, count = nlb_all * nstrips;
  if (part == 1) {                                  // line blocks 1 .. nlb - 2
    if (!edge) return GLA_OK;
    base0 = nstrips; count = split = (nlb_all - 2) * nstrips;
  } else if (part == 2) {                           // the first and the last line block (all of them on short grids)
    if (edge) { base0 = 0; split = nstrips; base1 = (nlb_all - 1) * nstrips; count = 2 * nstrips; }
  }
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = count;
  set_tile_mapping(tm);
  if (part != 0 && edge && g_tuning.edge_prio) {    // the two launches of a split pass must fit PER XCD (map_tile, gla_device.h)
    tm.balanced = part == 1 ? 1 : 2;
    tm.tiles_per_xcd = (count + GLA_XCDS - 1) / GLA_XCDS;
  }
  const dim3 grid((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), block(GLA_BLOCK);
  if constexpr (TS <= JACOBI_TS_MAX9) {
    if (canon9) {                                   // the same march for the canonical 9-point pattern
      const size_t lds9 = (size_t)TS * 3 * (GLA_BLOCK + 2) * sizeof(T);
      const bool sym9 = jacobi_sym9(A);
      bool done = false;
      if constexpr (march9_sym_fits<T, TS, true, false>()) {
        if (ldva > 0 && sym9) {
          hipLaunchKernelGGL((jacobi_march9_kernel<T, TS, true, false, true>), grid, block, lds9, st, dv, diag, b, x, xo, w, lines,
                             nstrips, nlines, tm, base0, split, base1, ldva, diag_out, b_out);
          done = true;
        }
      }
      if constexpr (march9_sym_fits<T, TS, false, false>()) {
        if (!done && ldva <= 0 && sym9) {
          hipLaunchKernelGGL((jacobi_march9_kernel<T, TS, false, false, true>), grid, block, lds9, st, dv, diag, b, x, xo, w, lines,
                             nstrips, nlines, tm, base0, split, base1);
          done = true;
        }
      }
      if (done) {}
      else if (ldva > 0)
        hipLaunchKernelGGL((jacobi_march9_kernel<T, TS, true>), grid, block, lds9, st, dv, diag, b, x, xo, w, lines, nstrips,
                           nlines, tm, base0, split, base1, ldva, diag_out, b_out);
      else
        hipLaunchKernelGGL((jacobi_march9_kernel<T, TS>), grid, block, lds9, st, dv, diag, b, x, xo, w, lines, nstrips, nlines,
                           tm, base0, split, base1);
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
  if constexpr (generic_ts) {
    const size_t lds = (size_t)TS * 4 * GLA_BLOCK * sizeof(T);
    if (ldva > 0) {                                   // x = the AoS vertex attributes (first pass of gla_jacobi_sweeps_aos)
      if (A->dia_D <= 5)
        hipLaunchKernelGGL((jacobi_fused_dia2d_kernel<T, 5, TS, true>), grid, block, lds, st, dv, diag, b, x, xo, w, lines,
                           nstrips, nlines, tm, base0, split, base1, ldva, diag_out, b_out);
      else
        hipLaunchKernelGGL((jacobi_fused_dia2d_kernel<T, 9, TS, true>), grid, block, lds, st, dv, diag, b, x, xo, w, lines,
                           nstrips, nlines, tm, base0, split, base1, ldva, diag_out, b_out);
    } else if (A->dia_D <= 5)
      hipLaunchKernelGGL((jacobi_fused_dia2d_kernel<T, 5, TS>), grid, block, lds, st, dv, diag, b, x, xo, w, lines, nstrips,
                         nlines, tm, base0, split, base1);
    else
      hipLaunchKernelGGL((jacobi_fused_dia2d_kernel<T, 9, TS>), grid, block, lds, st, dv, diag, b, x, xo, w, lines, nstrips,
                         nlines, tm, base0, split, base1);
    GLA_LAUNCH_CHECK();
  }
  return GLA_OK;
}

// `TS` sweeps AND the residual of the result in one pass (march kernels with RES = true: one more level that forms
// r = b - A x^(k+TS); canonical 5-point plans TS = 1 .. 7, canonical 9-point plans TS = 1 .. 5)
template <typename T, int TS>
int launch_jacobi_fused_res(gla_csr* A, const T* diag, const T* b, const T* x, T* xo, T* ro, T w, hipStream_t st) {
  const bool canon = TS + 1 <= 8 && jacobi_canon(A);
  const bool canon9 = TS + 1 <= JACOBI_TS_MAX9 && jacobi_canon9(A);
  if (canon) {                                      // gla_march5.hip
    gla_march5_args a;
    a.A = A; a.dtype = sizeof(T) == 8 ? GLA_F64 : GLA_F32; a.ts = TS; a.res = true;
    a.diag = diag; a.b = b; a.x = x; a.xo = xo; a.w = (double)w; a.st = st; a.part = 0;
    a.ldva = 0; a.diag_out = nullptr; a.b_out = ro; a.ghost = TS + 1; a.aos_off = 0; a.aos_rows = -1;
    return gla_march5_pass(a);
  }
  if (!canon && !canon9) return GLA_ERR_UNSUPPORTED;
  constexpr int LV = TS + 1;
  Dia2D dv;
  dv.val = A->dia_val;
  dv.mask = A->dia_mask;
  dv.D = A->dia_D; dv.S = A->dia_S; dv.A = A->dia_A; dv.B = A->dia_B;
  for (int i = 0; i < GLA_DIA_MAX; ++i) { dv.da[i] = i < A->dia_D ? A->dia_a[i] : 0; dv.db[i] = i < A->dia_D ? A->dia_b[i] : 0; }
  dv.n_rows = A->n_rows; dv.n_cols = A->n_cols;
  dv.nbi = dv.nbj = dv.nci = dv.nchunks = 0; dv.w_magic = 0; dv.part_off = 0;
  dv.aos_off = 0; dv.aos_rows = A->n_rows;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  const int bi_eff = GLA_BLOCK - 2 * LV;
  const int nstrips = (int)((A->dia_S + bi_eff - 1) / bi_eff);
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) lines = gla_pick_block_lines(nlines, nstrips, 256 * 2, 2 * (LV - 1), 2 * LV);
  if (lines > 1024) lines = 1024;
  if (lines < 2 * LV) lines = 2 * LV;
  dv.nbj = (int)((nlines + lines - 1) / lines);
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = dv.nbj * nstrips;
  set_tile_mapping(tm);
  const dim3 grid((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), block(GLA_BLOCK);
  {
    if constexpr (TS + 1 <= JACOBI_TS_MAX9) {
      bool done = false;
      if constexpr (march9_sym_fits<T, TS, false, true>()) {
        if (jacobi_sym9(A)) {
          hipLaunchKernelGGL((jacobi_march9_kernel<T, TS, false, true, true>), grid, block, (size_t)LV * 3 * (GLA_BLOCK + 2) * sizeof(T),
                             st, dv, diag, b, x, xo, w, lines, nstrips, nlines, tm, 0, tm.n_tiles, 0, (int64_t)1, (T*)nullptr, ro);
          done = true;
        }
      }
      if (!done)
        hipLaunchKernelGGL((jacobi_march9_kernel<T, TS, false, true>), grid, block, (size_t)LV * 3 * (GLA_BLOCK + 2) * sizeof(T), st, dv,
                           diag, b, x, xo, w, lines, nstrips, nlines, tm, 0, tm.n_tiles, 0, (int64_t)1, (T*)nullptr, ro);
    }
  }
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T>
int launch_jacobi_fused_res_ts(int ts, gla_csr* A, const T* diag, const T* b, const T* x, T* xo, T* ro, T w, hipStream_t st) {
  switch (ts) {
    case 1: return launch_jacobi_fused_res<T, 1>(A, diag, b, x, xo, ro, w, st);
    case 2: return launch_jacobi_fused_res<T, 2>(A, diag, b, x, xo, ro, w, st);
    case 3: return launch_jacobi_fused_res<T, 3>(A, diag, b, x, xo, ro, w, st);
    case 4: return launch_jacobi_fused_res<T, 4>(A, diag, b, x, xo, ro, w, st);
    case 5: return launch_jacobi_fused_res<T, 5>(A, diag, b, x, xo, ro, w, st);
    case 6: return launch_jacobi_fused_res<T, 6>(A, diag, b, x, xo, ro, w, st);
    case 7: return launch_jacobi_fused_res<T, 7>(A, diag, b, x, xo, ro, w, st);
    default: return GLA_ERR_UNSUPPORTED;
  }
}

// run-time sweeps-per-pass -> template (2 .. 8; 3, 6, 7, 8 exist for the canonical 5-point pattern only)
template <typename T>
int launch_jacobi_fused_ts(int ts, gla_csr* A, const T* diag, const T* b, const T* x, T* xo, T w, hipStream_t st, int part = 0,
                           int64_t ldva = 0, T* diag_out = nullptr, T* b_out = nullptr, int ghost = -1, int64_t aos_off = 0,
                           int64_t aos_rows = -1, int64_t ldd = 1, int* dg_flag = nullptr, int dg_epoch = 0) {
  if (ghost < 0) ghost = ts;
  if (ldd != 1 || dg_flag) {                       // (rare argument combination: not threaded through the switch below)
    switch (ts) {
      case 2: return launch_jacobi_fused<T, 2>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 3: return launch_jacobi_fused<T, 3>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 4: return launch_jacobi_fused<T, 4>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 5: return launch_jacobi_fused<T, 5>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 6: return launch_jacobi_fused<T, 6>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 7: return launch_jacobi_fused<T, 7>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      case 8: return launch_jacobi_fused<T, 8>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows, ldd, dg_flag, dg_epoch);
      default: return GLA_ERR_UNSUPPORTED;
    }
  }
  switch (ts) {
    case 2: return launch_jacobi_fused<T, 2>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 3: return launch_jacobi_fused<T, 3>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 4: return launch_jacobi_fused<T, 4>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 5: return launch_jacobi_fused<T, 5>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 6: return launch_jacobi_fused<T, 6>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 7: return launch_jacobi_fused<T, 7>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    case 8: return launch_jacobi_fused<T, 8>(A, diag, b, x, xo, w, st, part, ldva, diag_out, b_out, ghost, aos_off, aos_rows);
    default: return GLA_ERR_UNSUPPORTED;
  }
}

// ------------------------------------------------------------------------------------------------
// launch helper
// ------------------------------------------------------------------------------------------------
template <typename T, class Epi>
int launch_rows(gla_csr* A, int K, const T* X, int64_t ldx, const Epi& epi, int which, hipStream_t st) {
  if (A->n_rows == 0) return GLA_OK;
  const T* val = static_cast<const T*>(A->val);
  if (Epi::NRED > 0 && K > 1 && !((K % 2 == 0) && (ldx % 2 == 0) && al16(X) && epi.vec_ok(2)))
    return GLA_ERR_UNSUPPORTED;           // reducing epilogues exist for the K = 1 and paired-rhs kernels only
  {
    int64_t d_lo = -1, d_hi = -1;                 // rows the diagonal kernel may take for this launch
    const bool constv = A->dia_const && g_tuning.compress;      // opted-in: constant diagonals, no value stream
    // (a (row, vector)-lane version of this kernel for K > 1 lost to the CSR tile kernel in a same-process A/B at
    // K = 16 -- 1268 vs 1091 us per SpMM: with K-wide rows the tile's LDS-staged matrix and 256-row blocks matter more
    // than the missing index stream -- so K > 1 stays on the CSR kernels)
    if (K == 1 && ldx == 1 && A->dia_state == 1 && A->dia_val && g_tuning.dia && g_tuning.spmv_variant == 0 &&
        (constv || !(A->packed && g_tuning.compress))) {        // an opted-in packed CSR stream takes precedence
      if (which == 0 && A->dia_row_lo == 0 && A->dia_row_hi == A->n_rows) { d_lo = 0; d_hi = A->n_rows; }
      if (which == 1 && A->int_row_lo >= A->dia_row_lo && A->int_row_hi <= A->dia_row_hi && A->int_row_lo >= 0) {
        d_lo = A->int_row_lo;                       // interior tiles of a row block: one contiguous run of stencil rows
        d_hi = A->int_row_hi;
      }
    }
    if (d_hi > d_lo) {
      DiaView dv;
      dv.val = A->dia_val;
      dv.mask = A->dia_mask;
      for (int i = 0; i < GLA_DIA_MAX; ++i) dv.off[i] = i < A->dia_D ? A->dia_off[i] : 0;
      dv.D = A->dia_D;
      dv.n_rows = A->n_rows;
      dv.n_cols = A->n_cols;
      dv.row_lo = d_lo;
      dv.row_hi = d_hi;
      dv.part_off = A->n_tiles;                     // CSR tiles own partials [0, n_tiles), row blocks the rest
      TileMap bm;
      bm.tile_row = nullptr;
      bm.tile_info = nullptr;
      bm.tile_list = nullptr;
      bm.n_tiles = (int)((d_hi - d_lo + GLA_BLOCK - 1) / GLA_BLOCK);
      set_tile_mapping(bm);
      const dim3 dgrid((unsigned)(bm.tiles_per_xcd * GLA_XCDS)), dblock(GLA_BLOCK);
      for (int i = 0; i < GLA_DIA_MAX; ++i) dv.cval[i] = A->dia_cval[i];
      for (int i = 0; i < GLA_DIA_MAX; ++i) dv.partner[i] = -1;
      if (g_tuning.sym && d_lo == 0 && d_hi == A->n_rows && A->n_rows == A->n_cols)
        for (int i = 0; i < A->dia_D; ++i)
          if (A->dia_off[i] < 0 && ((gla_dia_sym_bits(A) >> i) & 1))
            for (int q = i + 1; q < A->dia_D; ++q)
              if (A->dia_off[q] == -A->dia_off[i]) dv.partner[i] = q;
      if (constv && which == 0 && A->dia_D <= 5)
        hipLaunchKernelGGL((spmv_dia_kernel<T, Epi, 5, true>), dgrid, dblock, 0, st, dv, X, bm, epi);
      else if (constv && which == 0 && A->dia_D <= 9)
        hipLaunchKernelGGL((spmv_dia_kernel<T, Epi, 9, true>), dgrid, dblock, 0, st, dv, X, bm, epi);
      else if (A->dia_D <= 5)
        hipLaunchKernelGGL((spmv_dia_kernel<T, Epi, 5>), dgrid, dblock, 0, st, dv, X, bm, epi);
      else if (A->dia_D <= 9)
        hipLaunchKernelGGL((spmv_dia_kernel<T, Epi, 9>), dgrid, dblock, 0, st, dv, X, bm, epi);
      else
        hipLaunchKernelGGL((spmv_dia_kernel<T, Epi, GLA_DIA_MAX>), dgrid, dblock, 0, st, dv, X, bm, epi);
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
  if (const int kv = dia2d_kv<T>(A, K, X, ldx, epi, which)) {
    // stencil plans, K = 1, 2, 4 or 8 16-byte vectors of right-hand sides: 2-D blocks with an LDS window of X
    const bool bj8 = g_tuning.spmm_variant == 10;        // A/B: 8-line blocks (larger LDS window, less halo, fewer waves)
    switch (kv) {
      case 1: return bj8 ? launch_dia2d<T, Epi, 1, 8>(A, X, ldx, epi, st) : launch_dia2d<T, Epi, 1, 4>(A, X, ldx, epi, st);
      case 2: return bj8 ? launch_dia2d<T, Epi, 2, 8>(A, X, ldx, epi, st) : launch_dia2d<T, Epi, 2, 4>(A, X, ldx, epi, st);
      case 4: return bj8 ? launch_dia2d<T, Epi, 4, 8>(A, X, ldx, epi, st) : launch_dia2d<T, Epi, 4, 4>(A, X, ldx, epi, st);
      default: return bj8 ? launch_dia2d<T, Epi, 8, 8>(A, X, ldx, epi, st) : launch_dia2d<T, Epi, 8, 4>(A, X, ldx, epi, st);
    }
  }
  TileMap tm;
  tm.lds_cap = A->lds_elems;                 // tiles above it are single long rows (whole-workgroup path in every kernel)
  tm.tile_row = A->tile_row;
  tm.tile_info = reinterpret_cast<const int4*>(A->tile_info);
  if (which == 0) { tm.tile_list = nullptr; tm.n_tiles = A->n_tiles; }
  else if (which == 1) { tm.tile_list = A->tile_lists; tm.n_tiles = A->n_interior; }
  else if (which == 2) { tm.tile_list = A->tile_lists + A->n_tiles; tm.n_tiles = A->n_boundary; }
  else return GLA_ERR_INVALID_ARG;
  if (which != 0 && A->tile_lists == nullptr) return GLA_ERR_INVALID_ARG;
  if (tm.n_tiles == 0) return GLA_OK;
  set_tile_mapping(tm);
  const dim3 grid((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), block(GLA_BLOCK);
  if (K == 1 && ldx == 1 && A->packed && !A->long_rows && g_tuning.compress && A->n_cols * (int64_t)sizeof(T) < ((int64_t)1 << 32)) {
    const size_t ldsp = (size_t)(256 + A->lds_elems + 16) * sizeof(T);
    hipLaunchKernelGGL((spmv_packed_kernel<T, Epi>), grid, block, ldsp, st, A->rowptr, A->col16, A->vcode,
                       (const T*)A->vdict, X, (unsigned)A->n_cols, tm, epi, A->rowoff16);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  }
  // the K = 1 kernels gather x[col] from a CONTIGUOUS vector; a strided single column (ldx > 1, e.g. a column view of
  // an AoS attribute tensor) takes the general (row, rhs)-lane kernel below, which honours the leading dimension
  if (K == 1 && ldx != 1 && Epi::NRED > 0) return GLA_ERR_UNSUPPORTED;
  if (K == 1 && ldx == 1) {
    const size_t lds = (size_t)(A->lds_elems + 2) * sizeof(T);
#define GLA_SPMV2(N_, NT_, PF_) \
  hipLaunchKernelGGL((spmv_stream_vec2_kernel<T, Epi, N_, NT_, PF_>), grid, block, lds, st, A->rowptr, A->col, val, \
                     X, tm, epi)
    if (A->col16_state == 1 && g_tuning.col16 && (g_tuning.spmv_variant == 0 || g_tuning.spmv_variant == 5)) {
      // MINW = 6 caps the kernel at 80 VGPRs (it takes 83 unconstrained): 7 instead of 5 waves per SIMD, +1.5 %
      if (g_tuning.spmv_variant == 5)
        hipLaunchKernelGGL((spmv_stream_vec2_kernel<T, Epi, 3, false, true, true, 1>), grid, block, lds, st, A->rowptr,
                           A->col, val, X, tm, epi, A->col16, A->col16_wide, (unsigned)A->n_cols, A->rowoff16);
      else
        hipLaunchKernelGGL((spmv_stream_vec2_kernel<T, Epi, 3, false, true, true, 6>), grid, block, lds, st, A->rowptr,
                           A->col, val, X, tm, epi, A->col16, A->col16_wide, (unsigned)A->n_cols, A->rowoff16);
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
    switch (g_tuning.spmv_variant) {          // A/B knobs (tools/ab_spmv.py); 0 is the tuned default
      case 1: GLA_SPMV2(3, false, false); break;
      case 2: GLA_SPMV2(2, false, true); break;
      case 3: GLA_SPMV2(4, false, true); break;
      case 4: GLA_SPMV2(3, true, true); break;
      default:
        hipLaunchKernelGGL((spmv_stream_vec2_kernel<T, Epi, 3, false, true, false, 6>), grid, block, lds, st,
                           A->rowptr, A->col, val, X, tm, epi);
        break;
    }
#undef GLA_SPMV2
  } else {
    const size_t lds = (size_t)A->lds_elems * (sizeof(T) + sizeof(int32_t));
    const bool vec2 = (K % 2 == 0) && (ldx % 2 == 0) && al16(X) && epi.vec_ok(2) &&
                      (g_tuning.spmm_variant != 1 || Epi::NRED > 0);
#define GLA_SPMM_CASE(KP_)                                                                              \
  hipLaunchKernelGGL((spmm_stream_kernel<T, KP_, Epi>), grid, block, lds, st, A->rowptr, A->col, val, X, \
                     ldx, K, A->lds_elems, tm, epi)
#define GLA_SPMM2_CASE(LPR_)                                                                                   \
  do {                                                                                                         \
    const size_t lds2 = lds + (SPMM_ROWCAP + 1) * sizeof(int32_t);                                             \
    if (g_tuning.spmm_variant == 2)                                                                            \
      hipLaunchKernelGGL((spmm_stream_vec2_kernel<T, LPR_, 1, Epi>), grid, block, lds2, st, A->rowptr, A->col, \
                         val, X, ldx, K, A->lds_elems, tm, epi);                                               \
    else                                                                                                       \
      hipLaunchKernelGGL((spmm_stream_vec2_kernel<T, LPR_, 0, Epi>), grid, block, lds2, st, A->rowptr, A->col, \
                         val, X, ldx, K, A->lds_elems, tm, epi);                                               \
  } while (0)
    const bool vec4 = sizeof(T) == 4 && vec2 && (K % 4 == 0) && (ldx % 4 == 0) && epi.vec_ok(4) &&
                      g_tuning.spmm_variant != 3;      // fp32: four right-hand sides (16 bytes) per lane
    {
      const int vw = vec4 ? 4 : 2;
      const int nvec = K / vw;
      if constexpr (Epi::NRED == 0) {
        if ((vec4 || vec2) && (nvec & (nvec - 1)) != 0 && g_tuning.spmm_variant != 4) {   // 4: lane groups (A/B)
          const size_t ldsf = lds + (SPMM_ROWCAP + 1) * sizeof(int32_t);
          if (vec4)
            hipLaunchKernelGGL((spmm_stream_vec2_kernel<T, 1, 0, Epi, (sizeof(T) == 4 ? 4 : 2), true>), grid, block, ldsf,
                               st, A->rowptr, A->col, val, X, ldx, K, A->lds_elems, tm, epi);
          else
            hipLaunchKernelGGL((spmm_stream_vec2_kernel<T, 1, 0, Epi, 2, true>), grid, block, ldsf, st, A->rowptr, A->col,
                               val, X, ldx, K, A->lds_elems, tm, epi);
          GLA_LAUNCH_CHECK();
          return GLA_OK;
        }
      }
    }
    if (vec4) {
#define GLA_SPMM4_CASE(LPR_)                                                                                      \
  hipLaunchKernelGGL((spmm_stream_vec2_kernel<T, LPR_, 0, Epi, 4>), grid, block,                                  \
                     lds + (SPMM_ROWCAP + 1) * sizeof(int32_t), st, A->rowptr, A->col, val, X, ldx, K,            \
                     A->lds_elems, tm, epi)
      if (K <= 4) GLA_SPMM4_CASE(1);
      else if (K <= 8) GLA_SPMM4_CASE(2);
      else if (K <= 16) GLA_SPMM4_CASE(4);
      else if (K <= 32) GLA_SPMM4_CASE(8);
      else if (K <= 64) GLA_SPMM4_CASE(16);
      else GLA_SPMM4_CASE(32);
#undef GLA_SPMM4_CASE
    } else if (vec2) {
      if (K <= 2) GLA_SPMM2_CASE(1);
      else if (K <= 4) GLA_SPMM2_CASE(2);
      else if (K <= 8) GLA_SPMM2_CASE(4);
      else if (K <= 16) GLA_SPMM2_CASE(8);
      else if (K <= 32) GLA_SPMM2_CASE(16);
      else GLA_SPMM2_CASE(32);
    } else if (K <= 2) GLA_SPMM_CASE(2);
    else if (K <= 4) GLA_SPMM_CASE(4);
    else if (K <= 8) GLA_SPMM_CASE(8);
    else if (K <= 16) GLA_SPMM_CASE(16);
    else if (K <= 32) GLA_SPMM_CASE(32);
    else GLA_SPMM_CASE(64);
#undef GLA_SPMM_CASE
#undef GLA_SPMM2_CASE
  }
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

// the launches of a schedule in order: group g = count[g] launches of sweeps[g] sweeps each
struct JacobiPlan { int count[4], sweeps[4]; };
inline int jacobi_plan_launches(const JacobiPlan& p) { return p.count[0] + p.count[1] + p.count[2] + p.count[3]; }
inline int jacobi_plan_ts(const JacobiPlan& p, int i) {      // sweeps of launch i (0-based)
  for (int g = 0; g < 4; ++g) { if (i < p.count[g]) return p.sweeps[g]; i -= p.count[g]; }
  return 1;
}

std::mutex g_dg_mutex;                               // guards every plan's dg_flags / dg_calls (taken once per gla_jacobi_sweeps_aos call)

}  // namespace

// The "A_ii is not the matrix diagonal" flag of ONE call (gla_jacobi_sweeps_aos, gla_dist_jacobi_temporal).  The flags live in
// the plan (zero when allocated).  EAGER calls: call number k uses word k mod GLA_DG_SLOTS and stores k into it to raise it, so no
// word ever has to be reset -- a memset in front of every call was 15-19 us of launch gap per step.  A call made while `st` is
// being CAPTURED into a HIP graph would freeze its word and its number into the graph, and a replay could then meet an eager call
// k + 64 j on the same word: a captured call gets a word of its own for the plan's lifetime (GLA_DG_CAPTURE_SLOTS of
// them; GLA_ERR_UNSUPPORTED once they are used up), the number 1, and a memset NODE in front of its first pass -- inside a graph
// that is an edge, not a launch gap.  The words are allocated at the first eager call (hipMalloc is not capturable; a capture
// of a plan that has never run eagerly is refused: callers warm up first, as torch's graph capture requires anyway).
int gla_jacobi_dg_slot(gla_csr* A, hipStream_t st, int** flag, int* epoch) {
  std::lock_guard<std::mutex> lock(g_dg_mutex);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if (cap != hipStreamCaptureStatusNone) {
    if (!A->dg_flags || A->dg_captured >= GLA_DG_CAPTURE_SLOTS) return GLA_ERR_UNSUPPORTED;
    int* word = A->dg_flags + GLA_DG_SLOTS + A->dg_captured++;
    GLA_HIP_CHECK(hipMemsetAsync(word, 0, sizeof(int), st));
    *epoch = 1;
    *flag = word;
    return GLA_OK;
  }
  if (!A->dg_flags) {
    GLA_HIP_CHECK(hipMalloc(&A->dg_flags, (GLA_DG_SLOTS + GLA_DG_CAPTURE_SLOTS) * sizeof(int)));
    GLA_HIP_CHECK(hipMemset(A->dg_flags, 0, (GLA_DG_SLOTS + GLA_DG_CAPTURE_SLOTS) * sizeof(int)));
  }
  if (++A->dg_calls >= 0x7fffff00u) {                 // (wrap-around after 2^31 calls: start over with clean words)
    GLA_HIP_CHECK(hipDeviceSynchronize());
    GLA_HIP_CHECK(hipMemset(A->dg_flags, 0, GLA_DG_SLOTS * sizeof(int)));
    A->dg_calls = 1;
  }
  *epoch = (int)A->dg_calls;
  *flag = A->dg_flags + (A->dg_calls % GLA_DG_SLOTS);
  return GLA_OK;
}

namespace {

template <typename T>
int jacobi_sweeps_aos(gla_csr* A, const T* va, int64_t ld, T* dwork, T* bwork, T* Xout, T* Xtmp, T w,
                      const JacobiPlan& plan, hipStream_t st) {
  const int n_launch = jacobi_plan_launches(plan);
  const T* src = va;
  // Canonical 5-point plans whose passes are all march passes: no contiguous copy of A_ii is made.  The first pass compares
  // A_ii with the matrix diagonal row by row and raises a flag (the first word of diag_work) on a difference; the later passes
  // take A_ii from the centre coefficient they load anyway -- 8 bytes per row less to read, 8 less for the first pass to write --
  // or, flag raised, from column 0 of the AoS rows.  (diag_work is not touched in this flow.)  (The reference always passes the diagonal: JacobiGNN.py:104, VCycle.py:164.)
  bool flagged = g_tuning.jacobi_diag_flag && jacobi_canon(A) && n_launch >= 2;
  for (int i = 0; i < n_launch && flagged; ++i) flagged = jacobi_plan_ts(plan, i) >= 2;
  int* flag = nullptr;
  int epoch = 0;
  if (flagged) {
    const int rc = gla_jacobi_dg_slot(A, st, &flag, &epoch);
    if (rc) return rc;
  }
  (void)dwork;
  for (int i = 1; i <= n_launch; ++i) {
    T* dst = ((n_launch - i) % 2 == 0) ? Xout : Xtmp;
    const int ts = jacobi_plan_ts(plan, i - 1);
    int rc;
    if (flagged && i == 1)
      rc = launch_jacobi_fused_ts<T>(ts, A, nullptr, nullptr, va, dst, w, st, 0, ld, nullptr, bwork, -1, 0, -1, 1, flag, epoch);
    else if (flagged)
      rc = launch_jacobi_fused_ts<T>(ts, A, va, bwork, src, dst, w, st, 0, 0, nullptr, nullptr, -1, 0, -1, ld, flag, epoch);
    else if (i == 1)                                // reads the AoS rows, writes the contiguous A_ii / b copies
      rc = launch_jacobi_fused_ts<T>(ts, A, nullptr, nullptr, va, dst, w, st, 0, ld, dwork, bwork);
    else if (ts > 1)
      rc = launch_jacobi_fused_ts<T>(ts, A, dwork, bwork, src, dst, w, st, 0);
    else
      rc = gla_jacobi_sweep(A, sizeof(T) == 8 ? GLA_F64 : GLA_F32, 1, dwork, bwork, 1, src, 1, dst, 1, (double)w, 0, st);
    if (rc) return rc;
    src = dst;
  }
  return GLA_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int gla_spmm(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* Y, int64_t ldy, int which,
             gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!X || !Y || ldx < K || ldy < K) return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiStore<T> epi{(T*)Y, ldy};
    return launch_rows<T>(A, K, (const T*)X, ldx, epi, which, (hipStream_t)stream);
  })
}

int gla_residual(gla_csr_t A, int dtype, int K, const void* B, int64_t ldb, const void* X, int64_t ldx, void* R,
                 int64_t ldr, int which, gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!B || !X || !R || ldb < K || ldx < K || ldr < K) return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiResidual<T> epi{(const T*)B, ldb, (T*)R, ldr};
    return launch_rows<T>(A, K, (const T*)X, ldx, epi, which, (hipStream_t)stream);
  })
}

int gla_spmm_add(gla_csr_t A, int dtype, int K, const void* Z, int64_t ldz, const void* X, int64_t ldx, void* Y,
                 int64_t ldy, int which, gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!Z || !X || !Y || ldz < K || ldx < K || ldy < K) return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiAddTo<T> epi{(const T*)Z, ldz, (T*)Y, ldy};
    return launch_rows<T>(A, K, (const T*)X, ldx, epi, which, (hipStream_t)stream);
  })
}

int gla_jacobi_sweep(gla_csr_t A, int dtype, int K, const void* diag, const void* B, int64_t ldb, const void* X,
                     int64_t ldx, void* Xout, int64_t ldxo, double w, int which, gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!diag || !B || !X || !Xout || X == Xout || ldb < K || ldx < K || ldxo < K) return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiJacobi<T> epi{(const T*)diag, (const T*)B, ldb, (const T*)X, ldx, (T*)Xout, ldxo, (T)w};
    return launch_rows<T>(A, K, (const T*)X, ldx, epi, which, (hipStream_t)stream);
  })
}

// The launches gla_jacobi_sweeps uses for n_iters sweeps on contiguous single-rhs operands.  A pass costs about the same
// whatever it carries (it is bound by one trip of the matrix through HBM), so: the fewest passes.  Canonical 5-point plans
// (jacobi_march5_kernel, 2 .. jacobi_ts_max = 8 sweeps per pass): ceil(n / 8) passes of equal size +- 1 (20 = 7 + 7 + 6).
// Other 2-D stencil plans (jacobi_fused_dia2d_kernel: 5, 4 or 2): the fewest passes, and among those the fewest unfused
// single sweeps (6 = 4 + 2, not 5 + 1).
static void jacobi_schedule(gla_csr* A, int dtype, bool contiguous_k1, int n_iters, JacobiPlan& p) {
  for (int g = 0; g < 4; ++g) { p.count[g] = 0; p.sweeps[g] = 1; }
  p.count[3] = n_iters;
  if (!(contiguous_k1 && n_iters >= 2 && A->n_rows > 0 && A->n_rows >= g_tuning.jacobi_min_rows && A->val_dtype == dtype &&
        (dtype == GLA_F64 ? jacobi_fused_ok<double>(A) : jacobi_fused_ok<float>(A))))
    return;
  if (g_tuning.jacobi_fuse == 2) {
    p.count[2] = n_iters / 2; p.sweeps[2] = 2; p.count[3] = n_iters % 2;
    return;
  }
  if (jacobi_canon(A) || jacobi_canon9(A)) {
    const int tmax = jacobi_canon(A) ? g_tuning.jacobi_ts_max : std::min(g_tuning.jacobi_ts_max, g_tuning.jacobi_ts_max9);
    const int k = (n_iters + tmax - 1) / tmax, base = n_iters / k, r = n_iters % k;
    p.count[0] = r; p.sweeps[0] = base + 1;
    p.count[1] = k - r; p.sweeps[1] = base;
    p.count[3] = 0;
    return;
  }
  p.sweeps[0] = 5; p.sweeps[1] = 4; p.sweeps[2] = 2;
  int best = 1 << 30;
  for (int a = n_iters / 5; a >= 0 && a >= n_iters / 5 - 3; --a) {
    int rem = n_iters - 5 * a;
    const int b = rem / 4; rem %= 4;
    const int c = rem / 2; rem %= 2;
    const int cost = 2 * (a + b + c + rem) + rem;          // passes first, then single sweeps
    if (cost < best) { best = cost; p.count[0] = a; p.count[1] = b; p.count[2] = c; p.count[3] = rem; }
  }
}

int gla_jacobi_sweeps_schedule(gla_csr_t A, int dtype, int n_iters, int* counts4_host, int* sweeps4_host) {
  if (!A || n_iters < 0 || !counts4_host || !sweeps4_host || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  JacobiPlan p;
  jacobi_schedule(A, dtype, true, n_iters, p);
  for (int g = 0; g < 4; ++g) { counts4_host[g] = p.count[g]; sweeps4_host[g] = p.sweeps[g]; }
  return GLA_OK;
}

}  // extern "C"

int gla_jacobi_parts_status(const gla_csr* A, int dtype, int sweeps) {
  if (!A || (dtype != GLA_F32 && dtype != GLA_F64) || sweeps < 2 || sweeps > 8) return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  if (!(dtype == GLA_F64 ? jacobi_fused_ok<double>(A) : jacobi_fused_ok<float>(A)) || !jacobi_canon(A)) return GLA_ERR_UNSUPPORTED;
  return GLA_OK;
}

// gla_jacobi_fused_parts / gla_jacobi_fused_parts_aos with the A_ii flag of a call (gla_dist.hip; arguments checked by the
// caller, canonical 5-point plans only).  va != NULL: the AoS first pass -- copies of A_ii / b go to diag_out / b_out AND every
// owned row whose A_ii differs from its centre coefficient raises the flag; va == NULL: a plain pass that takes A_ii from the
// centre coefficient unless the flag is raised (then from `diag`, of stride ldva elements where ldva > 0: the AoS rows themselves).
int gla_jacobi_parts_flagged(gla_csr* A, int dtype, const void* va, int64_t ldva, int64_t row_offset, int64_t n_attr_rows,
                             const void* diag, const void* B, const void* X, void* diag_out, void* b_out, void* Xout, double w,
                             int sweeps, int part, int ghost_lines, int* dg_flag, int dg_epoch, hipStream_t st) {
  if (!jacobi_canon(A)) return GLA_ERR_UNSUPPORTED;
  if (dtype == GLA_F64) {
    if (va) return launch_jacobi_fused_ts<double>(sweeps, A, nullptr, nullptr, (const double*)va, (double*)Xout, (double)w, st, part, ldva,
                                                  (double*)diag_out, (double*)b_out, ghost_lines, row_offset, n_attr_rows, 1, dg_flag, dg_epoch);
    return launch_jacobi_fused_ts<double>(sweeps, A, (const double*)diag, (const double*)B, (const double*)X, (double*)Xout, (double)w, st,
                                          part, 0, nullptr, nullptr, ghost_lines, 0, -1, ldva > 0 ? ldva : 1, dg_flag, dg_epoch);
  }
  if (va) return launch_jacobi_fused_ts<float>(sweeps, A, nullptr, nullptr, (const float*)va, (float*)Xout, (float)w, st, part, ldva,
                                               (float*)diag_out, (float*)b_out, ghost_lines, row_offset, n_attr_rows, 1, dg_flag, dg_epoch);
  return launch_jacobi_fused_ts<float>(sweeps, A, (const float*)diag, (const float*)B, (const float*)X, (float*)Xout, (float)w, st, part,
                                       0, nullptr, nullptr, ghost_lines, 0, -1, ldva > 0 ? ldva : 1, dg_flag, dg_epoch);
}

extern "C" {

int gla_jacobi_fused_parts(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, double w,
                           int sweeps, int part, int ghost_lines, gla_stream_t stream) {
  if (!A || !B || !X || !Xout || X == Xout || sweeps < 2 || sweeps > 8 || part < 0 || part > 2 ||
      ghost_lines < sweeps || (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  if (A->n_rows == 0) return GLA_OK;
  if (!(dtype == GLA_F64 ? jacobi_fused_ok<double>(A) : jacobi_fused_ok<float>(A))) return GLA_ERR_UNSUPPORTED;
  // diag == NULL: A_ii is the plan's own centre diagonal (the 5- / 9-point march kernels)
  if (!diag && !jacobi_canon(A) && !(jacobi_canon9(A) && sweeps <= JACOBI_TS_MAX9)) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLA_F64)
    return launch_jacobi_fused_ts<double>(sweeps, A, (const double*)diag, (const double*)B, (const double*)X, (double*)Xout,
                                          (double)w, st, part, 0, nullptr, nullptr, ghost_lines);
  return launch_jacobi_fused_ts<float>(sweeps, A, (const float*)diag, (const float*)B, (const float*)X, (float*)Xout, (float)w,
                                       st, part, 0, nullptr, nullptr, ghost_lines);
}

int gla_jacobi_fused_residual(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, void* Rout,
                              double w, int sweeps, gla_stream_t stream) {
  if (!A || !B || !X || !Xout || !Rout || X == Xout || Xout == Rout || X == Rout || sweeps < 1 ||
      (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  if (A->n_rows == 0) return GLA_OK;
  if (A->n_rows < g_tuning.jacobi_min_rows || g_tuning.jacobi_fuse != 1 ||
      !(dtype == GLA_F64 ? jacobi_fused_ok<double>(A) : jacobi_fused_ok<float>(A)))
    return GLA_ERR_UNSUPPORTED;
  if (!diag && !jacobi_canon(A) && !jacobi_canon9(A)) return GLA_ERR_INVALID_ARG;      // (diag == NULL: the march kernels only)
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLA_F64)
    return launch_jacobi_fused_res_ts<double>(sweeps, A, (const double*)diag, (const double*)B, (const double*)X, (double*)Xout,
                                              (double*)Rout, (double)w, st);
  return launch_jacobi_fused_res_ts<float>(sweeps, A, (const float*)diag, (const float*)B, (const float*)X, (float*)Xout,
                                           (float*)Rout, (float)w, st);
}

int gla_jacobi_fused_parts_aos(gla_csr_t A, int dtype, const void* vertex_attr, int64_t ld, int64_t row_offset,
                               int64_t n_attr_rows, void* diag_work, void* b_work, void* Xout, double w, int sweeps, int part,
                               int ghost_lines, gla_stream_t stream) {
  if (!A || !vertex_attr || ld < 3 || row_offset < 0 || n_attr_rows < 1 || !diag_work || !b_work || !Xout || sweeps < 2 ||
      sweeps > 8 || part < 0 || part > 2 || ghost_lines < sweeps || (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  if (A->n_rows == 0) return GLA_OK;
  // parts 0 and 2 read every row of the plan: only a tensor that holds them all will do
  if (part != 1 && (row_offset != 0 || n_attr_rows < A->n_rows)) return GLA_ERR_INVALID_ARG;
  // part 1 reads the plan rows outside the first / last ghost_lines grid lines: the tensor must hold those
  if (part == 1 && A->dia_S > 0 &&
      (row_offset > (int64_t)ghost_lines * A->dia_S || row_offset + n_attr_rows < A->n_rows - (int64_t)ghost_lines * A->dia_S))
    return GLA_ERR_INVALID_ARG;
  if (!(dtype == GLA_F64 ? jacobi_fused_ok<double>(A) : jacobi_fused_ok<float>(A))) return GLA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GLA_F64)
    return launch_jacobi_fused_ts<double>(sweeps, A, nullptr, nullptr, (const double*)vertex_attr, (double*)Xout, (double)w, st,
                                          part, ld, (double*)diag_work, (double*)b_work, ghost_lines, row_offset, n_attr_rows);
  return launch_jacobi_fused_ts<float>(sweeps, A, nullptr, nullptr, (const float*)vertex_attr, (float*)Xout, (float)w, st, part,
                                       ld, (float*)diag_work, (float*)b_work, ghost_lines, row_offset, n_attr_rows);
}

int gla_jacobi_fused(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, double w,
                     int sweeps, int part, gla_stream_t stream) {
  return gla_jacobi_fused_parts(A, dtype, diag, B, X, Xout, w, sweeps, part, sweeps, stream);
}

int gla_jacobi_sweeps_aos(gla_csr_t A, int dtype, const void* vertex_attr, int64_t ld, void* diag_work, void* b_work,
                          void* Xout, void* Xtmp, double w, int n_iters, gla_stream_t stream) {
  if (!A || !vertex_attr || ld < 3 || !diag_work || !b_work || !Xout || n_iters < 1 || Xout == Xtmp ||
      (dtype != GLA_F32 && dtype != GLA_F64) || A->n_rows != A->n_cols)
    return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  if (A->n_rows == 0) return GLA_OK;
  JacobiPlan plan;
  jacobi_schedule(A, dtype, true, n_iters, plan);
  if (jacobi_plan_ts(plan, 0) < 2) return GLA_ERR_UNSUPPORTED;              // no fused first pass: split and call gla_jacobi_sweeps
  if (jacobi_plan_launches(plan) > 1 && !Xtmp) return GLA_ERR_INVALID_ARG;
  if (dtype == GLA_F64)
    return jacobi_sweeps_aos<double>(A, (const double*)vertex_attr, ld, (double*)diag_work, (double*)b_work, (double*)Xout,
                                     (double*)Xtmp, (double)w, plan, (hipStream_t)stream);
  return jacobi_sweeps_aos<float>(A, (const float*)vertex_attr, ld, (float*)diag_work, (float*)b_work, (float*)Xout,
                                  (float*)Xtmp, (float)w, plan, (hipStream_t)stream);
}

int gla_jacobi_sweeps(gla_csr_t A, int dtype, int K, const void* diag, const void* B, int64_t ldb, const void* X0,
                      int64_t ldx0, void* Xout, void* Xtmp, int64_t ldw, double w, int n_iters,
                      gla_stream_t stream) {
  if (!A || n_iters < 0) return GLA_ERR_INVALID_ARG;
  if (A->n_rows != A->n_cols) return GLA_ERR_INVALID_ARG;
  if (n_iters == 0) return GLA_OK;
  if (!Xout || (n_iters > 1 && !Xtmp) || Xout == Xtmp) return GLA_ERR_INVALID_ARG;
  // 2-D stencil plans, single right-hand side, contiguous vectors: the sweeps are taken several at a time by the temporally
  // blocked kernels (the intermediate iterates stay on chip; jacobi_schedule); what is left over runs as an ordinary sweep.
  // Same arithmetic per row: bit-identical.  Option "jacobi_fuse": 0 off, 1 (default), 2 at most two per pass.
  JacobiPlan plan;
  // diag == NULL ("A_ii is the matrix diagonal": the march kernels then take it from the centre coefficient, no A_ii stream):
  // single right-hand side on canonical 5- / 9-point plans whose schedule is march passes only
  const bool dg_null_ok = !diag && K == 1 && A->n_rows > 0 && (jacobi_canon(A) || jacobi_canon9(A));
  if (!diag && A->n_rows > 0 && !dg_null_ok) return GLA_ERR_INVALID_ARG;
  jacobi_schedule(A, dtype, K == 1 && ldb == 1 && ldx0 == 1 && ldw == 1 && (diag || dg_null_ok) && B && X0, n_iters, plan);
  if (!diag && A->n_rows > 0)
    for (int i = 0; i < jacobi_plan_launches(plan); ++i)
      if (jacobi_plan_ts(plan, i) < 2) return GLA_ERR_INVALID_ARG;
  // K right-hand sides on canonical 5- / 9-point plans: passes of up to GLA_JACOBIK_TS_MAX sweeps (jacobi_marchk_kernel,
  // gla_jacobik.hip), ceil(n / 4) (9-point: n / 3) passes of equal size +- 1; anything it declines keeps one sweep per launch
  if (K > 1 && n_iters >= 2) {
    // the operand checks of gla_jacobi_sweep, BEFORE any pass writes: the K-wide route below does not go through it
    const int rc0 = check_plan(A, dtype, K);
    if (rc0) return rc0;
    if (A->n_rows > 0 && (!diag || !B || !X0 || X0 == Xout || X0 == Xtmp || ldb < K || ldx0 < K || ldw < K)) return GLA_ERR_INVALID_ARG;
  }
  const int vw_k = dtype == GLA_F64 ? 2 : 4;        // 16-byte rhs vectors: both ping-pong buffers must take them, or no pass may start
  if (K > 1 && n_iters >= 2 && g_tuning.jacobi_fuse == 1 && diag && B && X0 && A->val_dtype == dtype && A->n_rows > 0 &&
      al16(Xout) && al16(Xtmp) && ldw % vw_k == 0 &&
      A->n_rows * (int64_t)K >= g_tuning.jacobi_min_rows && (gla_dia_canon5(A) || gla_dia_canon9(A))) {
    const int tmax = gla_dia_canon5(A) ? GLA_JACOBIK_TS_MAX : 3;      // (5-point: five sweeps per pass, 9-point: three)
    const int k = (n_iters + tmax - 1) / tmax, base = n_iters / k, r = n_iters % k;       // base >= 2 for n_iters >= 2
    const void* src = X0;
    int64_t lsrc = ldx0;
    int rc = GLA_OK;
    bool declined = false;
    for (int i = 1; i <= k; ++i) {
      void* dst = ((k - i) % 2 == 0) ? Xout : Xtmp;
      const int ts = i <= r ? base + 1 : base;
      if (ts >= 2) {
        rc = gla_jacobik_pass(A, dtype, K, diag, B, ldb, src, lsrc, dst, ldw, w, ts, (hipStream_t)stream);
        if (rc == GLA_ERR_UNSUPPORTED && i == 1) { declined = true; break; }      // (later passes cannot decline: same plan, buffers checked above)
      } else {
        rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, src, lsrc, dst, ldw, w, 0, stream);
      }
      if (rc) return rc;
      src = dst;
      lsrc = ldw;
    }
    if (!declined) return GLA_OK;
  }
  const int n_launch = jacobi_plan_launches(plan);
  const void* src = X0;
  int64_t lds_ = ldx0;
  for (int i = 1; i <= n_launch; ++i) {
    void* dst = ((n_launch - i) % 2 == 0) ? Xout : Xtmp;
    const int ts = jacobi_plan_ts(plan, i - 1);
    int rc;
    if (ts > 1)
      rc = gla_jacobi_fused(A, dtype, diag, B, src, dst, w, ts, 0, stream);
    else
      rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, src, lds_, dst, ldw, w, 0, stream);
    if (rc) return rc;
    src = dst;
    lds_ = ldw;
  }
  return GLA_OK;
}

int gla_cheby_first(gla_csr_t A, int dtype, int K, const void* B, int64_t ldb, const void* X, int64_t ldx, void* R,
                    int64_t ldr, void* P, int64_t ldp, void* Xout, int64_t ldxo, double alpha, int which,
                    gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!B || !X || !R || !P || !Xout || X == Xout || ldb < K || ldx < K || ldr < K || ldp < K || ldxo < K)
    return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiChebyFirst<T> epi{(const T*)B, ldb, (const T*)X, ldx, (T*)R, ldr, (T*)P, ldp, (T*)Xout, ldxo, (T)alpha};
    return launch_rows<T>(A, K, (const T*)X, ldx, epi, which, (hipStream_t)stream);
  })
}

int gla_cheby_next(gla_csr_t A, int dtype, int K, const void* P, int64_t ldp, void* Pout, int64_t ldpo, void* R,
                   int64_t ldr, void* X, int64_t ldx, double alpha_old, double alpha_new, double beta, int which,
                   gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!P || !Pout || !R || !X || P == Pout || ldp < K || ldpo < K || ldr < K || ldx < K)
    return GLA_ERR_INVALID_ARG;
  GLA_DISPATCH(dtype, {
    EpiChebyNext<T> epi{(const T*)P, ldp, (T*)Pout, ldpo, (T*)R, ldr, (T*)X, ldx,
                        (T)alpha_old, (T)alpha_new, (T)beta};
    return launch_rows<T>(A, K, (const T*)P, ldp, epi, which, (hipStream_t)stream);
  })
}

int gla_cheby_steps(gla_csr_t A, int dtype, int K, int n_steps, void* P0, int64_t ldp0, void* P1, int64_t ldp1, void* R,
                    int64_t ldr, void* X, int64_t ldx, const double* alpha_old_host, const double* alpha_new_host,
                    const double* beta_host, gla_stream_t stream) {
  if (n_steps < 0 || (n_steps > 0 && (!alpha_old_host || !alpha_new_host || !beta_host))) return GLA_ERR_INVALID_ARG;
  void* p = P0;
  void* pn = P1;
  int64_t lp = ldp0, lpn = ldp1;
  for (int i = 0; i < n_steps; ++i) {              // one launch per step, enqueued back to back: no host work between them
    const int rc = gla_cheby_next(A, dtype, K, p, lp, pn, lpn, R, ldr, X, ldx, alpha_old_host[i], alpha_new_host[i],
                                  beta_host[i], 0, stream);
    if (rc) return rc;
    void* t = p; p = pn; pn = t;
    const int64_t tl = lp; lp = lpn; lpn = tl;
  }
  return GLA_OK;
}

static inline int64_t power_partials(const gla_csr* A) {
  const int64_t row_blocks = (A->n_rows + GLA_BLOCK - 1) / GLA_BLOCK, b2 = dia2d_blocks_bound(A);
  return (int64_t)(A->n_tiles > 0 ? A->n_tiles : 1) + (row_blocks > b2 ? row_blocks : b2);
}

int64_t gla_power_work_bytes(gla_csr_t A, int K, int dtype) {
  if (!A || K < 1) return -1;      // partials of the CSR tiles [n_tiles, K] and of the diagonal kernel's row blocks
                                   // [ceil(n_rows / 256), K], then the second-stage partials [DOT_BLOCKS_MAX, K]
  return (power_partials(A) + DOT_BLOCKS_MAX) * K * (dtype == GLA_F64 ? 8 : 4);
}

int gla_power_step(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, const void* nrm_in,
                   const void* rnrm_in, void* Y, int64_t ldy, void* sumsq_out, void* work, int which,
                   gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;      // nothing to compute; empty operands may be null
  if (!X || !Y || !work || ldx < K || ldy < K || ((nrm_in == nullptr) != (rnrm_in == nullptr)))
    return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == GLA_F64 ? 8 : 4;
  GLA_DISPATCH(dtype, {
    EpiPower<T> epi{(T*)Y, ldy, (const T*)nrm_in, (const T*)rnrm_in, (T*)work};
    // partial rows in use: the blocked stencil kernel writes one per block (every block, no zero-fill needed);
    // otherwise CSR tiles [0, n_tiles) and the diagonal kernel's row blocks behind them
    const int kv = dia2d_kv<T>(A, K, (const T*)X, ldx, epi, which);
    const int64_t rows_used = kv ? dia2d_nblocks(A, kv) : power_partials(A);
    if (which != 2 && !kv)   // boundary launches add to the partials the interior launch already zeroed
      GLA_HIP_CHECK(hipMemsetAsync(work, 0, (size_t)rows_used * K * esz, st));
    rc = launch_rows<T>(A, K, (const T*)X, ldx, epi, which, st);
    if (rc) return rc;
    if (sumsq_out && which != 1) {     // column sums of the partials once every tile has contributed
      T* tile_part = (T*)work;
      return launch_col_dot<T>(rows_used, K, tile_part, K, (const T*)nullptr, K, (T*)sumsq_out,
                               tile_part + power_partials(A) * K, st);
    }
    return GLA_OK;
  })
}

int gla_power_rayleigh(gla_csr_t A, int dtype, int K, const void* Y, int64_t ldy, const void* nrm_in, const void* rnrm_in,
                       void* Bout, int64_t ldb, void* B2out, int64_t ldb2, void* nA_out, void* bb_out, void* work,
                       gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (A->n_rows == 0) return GLA_OK;
  if (!Y || !nrm_in || !rnrm_in || !Bout || !B2out || !nA_out || !bb_out || !work || ldy < K || ldb < K || ldb2 < K ||
      Y == Bout || Y == B2out || Bout == B2out)
    return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  GLA_DISPATCH(dtype, {
    // work: [partials of b (A b) | partials of b^2 | scratch of the column sums], each gla_power_work_bytes / 3 ... the
    // caller passes 2 x gla_power_work_bytes: first half as gla_power_step uses it, second half for the b^2 partials
    const int64_t half = power_partials(A) * K + (int64_t)DOT_BLOCKS_MAX * K;      // elements of one gla_power_step work area
    T* w1 = (T*)work;
    T* w2 = w1 + half;
    if (g_tuning.jacobi_fuse == 1) {                  // canonical stencil plans: one level of the march kernel (gla_powerk.hip)
      const int rt = gla_power_tail_pass(A, dtype, K, Y, ldy, nrm_in, Bout, ldb, B2out, ldb2, nA_out, bb_out, work,
                                         2 * half * (int64_t)sizeof(T), st);
      if (rt != GLA_ERR_UNSUPPORTED) return rt;
    }
    EpiRayleigh<T> epi{{nullptr, 0, (const T*)nrm_in, (const T*)rnrm_in, w1}, (const T*)Y, ldy, (T*)Bout, ldb, (T*)B2out, ldb2, w2};
    const int kv = dia2d_kv<T>(A, K, (const T*)Y, ldy, epi, 0);
    if (!kv) return GLA_ERR_UNSUPPORTED;              // (no blocked stencil kernel for this plan / K: unfused tail)
    const int bj = g_tuning.spmm_variant == 10 ? 8 : 4;
    if (bj != 4) return GLA_ERR_UNSUPPORTED;
    switch (kv) {
      case 1: rc = launch_dia2d<T, EpiRayleigh<T>, 1, 4>(A, (const T*)Y, ldy, epi, st); break;
      case 2: rc = launch_dia2d<T, EpiRayleigh<T>, 2, 4>(A, (const T*)Y, ldy, epi, st); break;
      case 4: rc = launch_dia2d<T, EpiRayleigh<T>, 4, 4>(A, (const T*)Y, ldy, epi, st); break;
      default: rc = launch_dia2d<T, EpiRayleigh<T>, 8, 4>(A, (const T*)Y, ldy, epi, st); break;
    }
    if (rc) return rc;
    const int64_t rows_used = dia2d_nblocks(A, kv);
    rc = launch_col_dot<T>(rows_used, K, w1, K, (const T*)nullptr, K, (T*)nA_out, w1 + power_partials(A) * K, st);
    if (rc) return rc;
    return launch_col_dot<T>(rows_used, K, w2, K, (const T*)nullptr, K, (T*)bb_out, w2 + power_partials(A) * K, st);
  })
}

}  // extern "C"
