// libgla_hip.so -- hot-path kernels (gfx950 / MI355X only; wave = 64 lanes).
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
// plan creation); blockIdx is remapped so that each XCD walks a contiguous span of tiles and the x window
// stays in that XCD's 4 MiB L2.  Matrices with a row longer than the LDS tile fall back to a
// wave-per-row (CSR-vector) kernel whose sums are tree-ordered (documented in DESIGN.md).
//
// Multi-RHS (K > 1, row-major [n, K]): val/col are staged in LDS instead of products, and the lanes
// are (row, rhs) pairs so that every gather reads K contiguous values of one X row.
#include "gla_internal.h"
#include <type_traits>
#include <mutex>

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------------
// device views
// ------------------------------------------------------------------------------------------------
struct TileMap {
  const int4* tile_info;      // per tile {first row, end row, first nnz, end nnz}
  const int32_t* tile_row;
  const int32_t* tile_list;   // nullptr = identity
  int n_tiles;                // tiles to run in this launch
  int tiles_per_xcd;
  int xcd_chunk;              // see map_tile
};

// grid = 8 * tiles_per_xcd workgroups; with chunking the per-XCD count is rounded up to whole chunks so that the
// slot formula of map_tile is a bijection onto [0, 8 * tiles_per_xcd) (slots >= n_tiles exit at once)
inline void set_tile_mapping(TileMap& tm) {
  tm.xcd_chunk = g_tuning.xcd_chunk;
  const int C = tm.xcd_chunk > 0 ? tm.xcd_chunk : 1;
  const int per_round = GLA_XCDS * C;
  tm.tiles_per_xcd = ((tm.n_tiles + per_round - 1) / per_round) * C;
}

__device__ __forceinline__ int map_tile(const TileMap& m) {
  const int bid = blockIdx.x;
  // Workgroups b and b+8 share an XCD (round-robin dispatch).  xcd_chunk = C > 0: XCD e takes C consecutive tiles,
  // then the next XCD the next C, ... (C = 1 is the plain identity mapping); xcd_chunk = 0: each XCD walks one
  // contiguous eighth of the tiles.  Measured (tools/ab_spmv.py): small chunks win -- all XCDs then stream one
  // common region of HBM, and a tile's x window (a few grid lines) is re-fetched by neighbouring XCDs from the
  // Infinity Cache rather than from DRAM.
  int slot;
  if (m.xcd_chunk > 0) {
    const int C = m.xcd_chunk, xcd = bid % GLA_XCDS, j = bid / GLA_XCDS;
    slot = (j / C) * (GLA_XCDS * C) + xcd * C + (j % C);
  } else {
    slot = (bid % GLA_XCDS) * m.tiles_per_xcd + bid / GLA_XCDS;
  }
  if (slot >= m.n_tiles) return -1;
  return m.tile_list ? m.tile_list[slot] : slot;
}

// Branch-free tile loop: every lane runs every iteration with its index clamped into [0, len) (lanes past the end
// redo the last element and rewrite the identical value), so the loads of a batch are issued back to back instead
// of one exec-masked load + wait at a time.  BODY_LOAD(u, i) fills per-slot registers, BODY_STORE(u, i) consumes.
#define GLA_TILE_LOOP(len, tid, BODY_LOAD, BODY_STORE)                                  \
  for (int base_ = 0; base_ < (len); base_ += 4 * GLA_BLOCK) {                          \
    int ii_[4];                                                                         \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                     \
      ii_[u] = min(base_ + u * GLA_BLOCK + (tid), (len)-1);                             \
      BODY_LOAD(u, ii_[u]);                                                             \
    }                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { BODY_STORE(u, ii_[u]); }            \
  }

// ------------------------------------------------------------------------------------------------
// fused vertex updates ("epilogues"): called once per (row, rhs) with the finished row sum
// ------------------------------------------------------------------------------------------------
// Each epilogue is split into load -- the row's dense operands, issued EARLY so their latency overlaps the
// nnz stream -- and apply with the finished row sum.  Both are generic in the value type V: V = T for one
// right-hand side per lane, V = T2 (16-byte accesses) for two adjacent right-hand sides per lane.
// operator() = load + apply for kernels that do not prefetch.
template <typename V, typename T>
__device__ __forceinline__ V ldv(const T* p) { return *reinterpret_cast<const V*>(p); }
template <typename V, typename T>
__device__ __forceinline__ void stv(T* p, V v) { *reinterpret_cast<V*>(p) = v; }

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ T div_by(T a, T b, T rb) {
  // a / b from the correctly rounded reciprocal rb = RN(1/b) with one Markstein correction step: q = RN(a*rb),
  // r = a - q*b (exact in an FMA), q' = RN(q + r*rb) -- the correctly rounded quotient except for rare
  // last-place cases, at 3 instructions instead of the ~30 of an IEEE fp64 division per gathered value.
  const T q = a * rb;
  const T r = fma_t(-q, b, a);
  return fma_t(r, rb, q);
}

template <typename T>
struct EpiPower {
  T* y; int64_t ldy; const T* nrm; const T* rnrm; T* partials;
  template <typename V> struct PreV {};
  static constexpr int NRED = 1;
  template <typename V> struct GxV { V n, rn; };                // this lane's divisor(s), loaded once per kernel
  template <typename V> __device__ __forceinline__ GxV<V> gx_load(int k) const {
    GxV<V> g;
    if constexpr (sizeof(V) == sizeof(T)) { g.n = nrm ? nrm[k] : T(1); g.rn = nrm ? rnrm[k] : T(1); }
    else {
#pragma unroll
      for (int i = 0; i < (int)(sizeof(V) / sizeof(T)); ++i) {
        g.n[i] = nrm ? nrm[k + i] : T(1);
        g.rn[i] = nrm ? rnrm[k + i] : T(1);
      }
    }
    return g;
  }
  template <typename V> __device__ __forceinline__ V gx(V xv, const GxV<V>& g) const {
    if (!nrm) return xv;
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

// Sum over aligned groups of KP (<= 16) lanes with DPP moves (full-rate VALU; ds_bpermute-based __shfl_xor runs
// through the LDS pipe and was the bottleneck of gla_edge_dot).  quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror: after log2(KP) steps every lane of the group holds the group's sum (fixed order).
template <int CTRL, typename T>
__device__ __forceinline__ T dpp_move(T v) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
  } else {
    typedef int I2v __attribute__((ext_vector_type(2)));
    I2v w = __builtin_bit_cast(I2v, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, CTRL, 0xF, 0xF, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(T, w);
  }
}

template <int KP, typename T>
__device__ __forceinline__ T group_sum(T v) {
  static_assert(KP == 1 || KP == 2 || KP == 4 || KP == 8 || KP == 16, "group of 1..16 lanes");
  if constexpr (KP >= 2) v = v + dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
  if constexpr (KP >= 4) v = v + dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
  if constexpr (KP >= 8) v = v + dpp_move<0x141>(v);     // row_half_mirror
  if constexpr (KP >= 16) v = v + dpp_move<0x140>(v);    // row_mirror
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
  return v;
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

  T red = T(0);
  if (r < r1) {
    if (!PF) {
      row_range(r, ra, rb);
      pre = epi.load((int64_t)r, 0);
    }
    T acc = T(0);
    int j = ra - sa;
    const int jend = rb - sa;
    for (; j + 4 <= jend; j += 4) {              // LDS reads issued together, added strictly in edge order
      const T p0 = prod[j], p1 = prod[j + 1], p2 = prod[j + 2], p3 = prod[j + 3];
      acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
    }
    for (; j < jend; ++j) acc = acc + prod[j];
    epi.apply((int64_t)r, 0, acc, pre);
    if constexpr (Epi::NRED > 0) red = epi.redv(acc, pre);
  }
  for (int r2 = r + GLA_BLOCK; r2 < r1; r2 += GLA_BLOCK) {       // tiles with more than 256 (short) rows
    int a, b;
    row_range(r2, a, b);
    a -= sa;
    b -= sa;
    T acc = T(0);
    for (int j = a; j < b; ++j) acc = acc + prod[j];
    const auto pre2 = epi.load((int64_t)r2, 0);
    epi.apply((int64_t)r2, 0, acc, pre2);
    if constexpr (Epi::NRED > 0) red = red + epi.redv(acc, pre2);
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
    const T* __restrict__ vdict, const T* __restrict__ x, unsigned n_cols, TileMap tm, Epi epi) {
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
    ra[i] = rowptr[rc];
    rb[i] = rowptr[rc + 1];
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
    const int a = rowptr[r2] - sa, b = rowptr[r2 + 1] - sa;
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
    for (int r = r0 + rsub; r < r1; r += RPP) {
      int a, b;
      row_bounds(r, a, b);
      for (int k = k0; k < K; k += VW * LPR) {
        if (k == k0) row_vec(r, k, a, b, gxs0);
        else row_vec(r, k, a, b, epi.template gx_load<T2>(k));
      }
    }
  }
  if constexpr (Epi::NRED > 0) {       // lanes with the same rhs pair are LPR apart: fixed-order tree over the rows
    __syncthreads();
    T2* sred = reinterpret_cast<T2*>(smem_raw);
    sred[tid] = red;
    __syncthreads();
    for (int st = GLA_BLOCK / 2; st >= LPR; st >>= 1) {
      if (tid < st) sred[tid] = sred[tid] + sred[tid + st];
      __syncthreads();
    }
    if (tid < LPR) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
        if (k0 + i < K) epi.partials_ptr()[(int64_t)t * K + k0 + i] = sred[tid][i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fallback for plans with a row longer than the LDS tile: one wave per row, tree-ordered sum
// ------------------------------------------------------------------------------------------------
template <typename T, class Epi>
__global__ __launch_bounds__(GLA_BLOCK) void spmm_vector_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const T* __restrict__ val,
    const T* __restrict__ X, int64_t ldx, int K, int64_t n_rows, Epi epi) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * GLA_BLOCK) >> 6;
  for (int64_t r = wave; r < n_rows; r += n_waves) {
    const int a = rowptr[r], b = rowptr[r + 1];
    for (int k = 0; k < K; ++k) {
      T acc = T(0);
      for (int j = a + lane; j < b; j += 64) acc = acc + val[j] * X[(int64_t)col[j] * ldx + k];
      acc = wave_sum(acc);
      if (lane == 0) epi(r, k, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launch helper
// ------------------------------------------------------------------------------------------------
template <typename T, class Epi>
int launch_rows(gla_csr* A, int K, const T* X, int64_t ldx, const Epi& epi, int which, hipStream_t st) {
  if (A->n_rows == 0) return GLA_OK;
  const T* val = static_cast<const T*>(A->val);
  if (Epi::NRED > 0 && (A->long_rows || (K > 1 && !((K % 2 == 0) && (ldx % 2 == 0) && al16(X) && epi.vec_ok(2)))))
    return GLA_ERR_UNSUPPORTED;           // reducing epilogues exist for the K = 1 and paired-rhs kernels only
  if (A->long_rows) {
    if (which == 1) return GLA_OK;        // nothing is "interior" for the fallback: all rows run after the halo
    const int64_t waves_needed = A->n_rows;
    int64_t blocks = (waves_needed * 64 + GLA_BLOCK - 1) / GLA_BLOCK;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((spmm_vector_kernel<T, Epi>), dim3((unsigned)blocks), dim3(GLA_BLOCK), 0, st,
                       A->rowptr, A->col, val, X, ldx, K, A->n_rows, epi);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  }
  TileMap tm;
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
  if (K == 1 && A->packed && g_tuning.compress && A->n_cols * (int64_t)sizeof(T) < ((int64_t)1 << 32)) {
    const size_t ldsp = (size_t)(256 + A->lds_elems + 16) * sizeof(T);
    hipLaunchKernelGGL((spmv_packed_kernel<T, Epi>), grid, block, ldsp, st, A->rowptr, A->col16, A->vcode,
                       (const T*)A->vdict, X, (unsigned)A->n_cols, tm, epi);
    GLA_LAUNCH_CHECK();
    return GLA_OK;
  }
  if (K == 1) {
    const size_t lds = (size_t)(A->lds_elems + 2) * sizeof(T);
#define GLA_SPMV2(N_, NT_, PF_) \
  hipLaunchKernelGGL((spmv_stream_vec2_kernel<T, Epi, N_, NT_, PF_>), grid, block, lds, st, A->rowptr, A->col, val, \
                     X, tm, epi)
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

inline int check_plan(gla_csr* A, int dtype, int K) {
  if (!A || K < 1 || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  return GLA_OK;
}

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
    int K, int op, int64_t n_rows, T* __restrict__ out, int64_t ldo) {
  constexpr int RPP = GLA_BLOCK / KP;
  const int k0 = threadIdx.x % KP, rsub = threadIdx.x / KP;
  for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n_rows; r += (int64_t)gridDim.x * RPP) {
    const int a = rowptr[r], b = rowptr[r + 1];
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

// ------------------------------------------------------------------------------------------------
// column dots / scaling (global aggregations of the power method)
// ------------------------------------------------------------------------------------------------
constexpr int DOT_BLOCKS_MAX = 1024;

template <typename T, int KP>
__global__ __launch_bounds__(GLA_BLOCK) void col_dot_stage1(const T* __restrict__ X, int64_t ldx,
                                                            const T* __restrict__ Y, int64_t ldy, int64_t n,
                                                            int K, T* __restrict__ partial) {
  __shared__ T red[GLA_BLOCK];
  constexpr int RPP = GLA_BLOCK / KP;
  const int tid = threadIdx.x, k0 = tid % KP, rsub = tid / KP;
  for (int kb = 0; kb < K; kb += KP) {
    const int k = kb + k0;
    T acc = T(0);
    if (k < K)
      for (int64_t r = (int64_t)blockIdx.x * RPP + rsub; r < n; r += (int64_t)gridDim.x * RPP)
        acc = acc + (Y ? X[r * ldx + k] * Y[r * ldy + k] : X[r * ldx + k]);   // Y == nullptr: plain column sums
    red[tid] = acc;
    __syncthreads();
    for (int st = GLA_BLOCK / 2; st >= KP; st >>= 1) {
      if (tid < st) red[tid] = red[tid] + red[tid + st];
      __syncthreads();
    }
    if (tid < KP && k < K) partial[(int64_t)blockIdx.x * K + k] = red[tid];
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void col_dot_stage2(const T* __restrict__ partial, int nblocks, int K,
                                                            T* __restrict__ out) {
  __shared__ T red[GLA_BLOCK];
  const int tid = threadIdx.x;
  for (int k = 0; k < K; ++k) {
    T acc = T(0);
    for (int b = tid; b < nblocks; b += GLA_BLOCK) acc = acc + partial[(int64_t)b * K + k];
    red[tid] = acc;
    __syncthreads();
    for (int st = GLA_BLOCK / 2; st > 0; st >>= 1) {
      if (tid < st) red[tid] = red[tid] + red[tid + st];
      __syncthreads();
    }
    if (tid == 0) out[k] = red[0];
    __syncthreads();
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

inline unsigned grid_for(int64_t work, int per_block = GLA_BLOCK, int cap = 256 * 8) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

inline TileMap all_tiles(gla_csr* A) {
  TileMap tm;
  tm.tile_row = A->tile_row;
  tm.tile_info = reinterpret_cast<const int4*>(A->tile_info);
  tm.tile_list = nullptr;
  tm.n_tiles = A->n_tiles;
  set_tile_mapping(tm);
  return tm;
}

template <typename T>
int launch_col_dot(int64_t n, int K, const T* x, int64_t ldx, const T* y, int64_t ldy, T* out, T* pw,
                   hipStream_t st) {
  int kp = 1;
  while (kp < K && kp < 64) kp <<= 1;
  const int rpp = GLA_BLOCK / kp;
  const unsigned nb = grid_for(n, rpp * 8, DOT_BLOCKS_MAX);
#define GLA_DOT_CASE(KP_) \
  hipLaunchKernelGGL((col_dot_stage1<T, KP_>), dim3(nb), dim3(GLA_BLOCK), 0, st, x, ldx, y, ldy, n, K, pw)
  switch (kp) {
    case 1: GLA_DOT_CASE(1); break;
    case 2: GLA_DOT_CASE(2); break;
    case 4: GLA_DOT_CASE(4); break;
    case 8: GLA_DOT_CASE(8); break;
    case 16: GLA_DOT_CASE(16); break;
    case 32: GLA_DOT_CASE(32); break;
    default: GLA_DOT_CASE(64); break;
  }
#undef GLA_DOT_CASE
  GLA_LAUNCH_CHECK();
  hipLaunchKernelGGL((col_dot_stage2<T>), dim3(1), dim3(GLA_BLOCK), 0, st, pw, (int)nb, K, out);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
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

#define GLA_K_SWITCH(K, ...)                      \
  switch (K) {                                    \
    case 1: { constexpr int KC = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int KC = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int KC = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int KC = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int KC = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int KC = 32; __VA_ARGS__; } break; \
    default: { constexpr int KC = 0; __VA_ARGS__; } break;  \
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
  const unsigned grid = grid_for(A->n_rows, GLA_BLOCK / kp, 256 * 16);
#define GLA_SEG(KP_)                                                                                            \
  hipLaunchKernelGGL((segment_reduce_kernel<T, KP_>), dim3(grid), dim3(GLA_BLOCK), 0, st, A->rowptr, A->perm,  \
                     src, ld_src, K, op, A->n_rows, out, ld_out)
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
#define GLA_DISPATCH(dtype, ...)                              \
  if ((dtype) == GLA_F64) { using T = double; __VA_ARGS__; }  \
  else { using T = float; __VA_ARGS__; }

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

int gla_jacobi_sweeps(gla_csr_t A, int dtype, int K, const void* diag, const void* B, int64_t ldb, const void* X0,
                      int64_t ldx0, void* Xout, void* Xtmp, int64_t ldw, double w, int n_iters,
                      gla_stream_t stream) {
  if (!A || n_iters < 0) return GLA_ERR_INVALID_ARG;
  if (A->n_rows != A->n_cols) return GLA_ERR_INVALID_ARG;
  if (n_iters == 0) return GLA_OK;
  if (!Xout || (n_iters > 1 && !Xtmp) || Xout == Xtmp) return GLA_ERR_INVALID_ARG;
  const void* src = X0;
  int64_t lds_ = ldx0;
  for (int i = 1; i <= n_iters; ++i) {
    void* dst = ((n_iters - i) % 2 == 0) ? Xout : Xtmp;
    int rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, src, lds_, dst, ldw, w, 0, stream);
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

int64_t gla_power_work_bytes(gla_csr_t A, int K, int dtype) {
  if (!A || K < 1) return -1;      // tile partials [n_tiles, K] + second-stage partials [DOT_BLOCKS_MAX, K]
  return ((int64_t)(A->n_tiles > 0 ? A->n_tiles : 1) + DOT_BLOCKS_MAX) * K * (dtype == GLA_F64 ? 8 : 4);
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
  if (which != 2)          // boundary launches add to the partials the interior launch already zeroed
    GLA_HIP_CHECK(hipMemsetAsync(work, 0, (size_t)(A->n_tiles > 0 ? A->n_tiles : 1) * K * esz, st));
  GLA_DISPATCH(dtype, {
    EpiPower<T> epi{(T*)Y, ldy, (const T*)nrm_in, (const T*)rnrm_in, (T*)work};
    rc = launch_rows<T>(A, K, (const T*)X, ldx, epi, which, st);
    if (rc) return rc;
    if (sumsq_out && which != 1) {     // column sums of the tile partials once every tile has contributed
      T* tile_part = (T*)work;
      return launch_col_dot<T>(A->n_tiles, K, tile_part, K, (const T*)nullptr, K, (T*)sumsq_out,
                               tile_part + (int64_t)(A->n_tiles > 0 ? A->n_tiles : 1) * K, st);
    }
    return GLA_OK;
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
  if (A->long_rows) return GLA_ERR_UNSUPPORTED;
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
  if (A->n_rows != A->n_cols) return GLA_ERR_INVALID_ARG;
  if (A->n_rows > 0 && (!diag || !cf)) return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && (!S || !w_out)) return GLA_ERR_INVALID_ARG;
  if (A->long_rows) return GLA_ERR_UNSUPPORTED;
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
  if (A->long_rows) return GLA_ERR_UNSUPPORTED;
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
