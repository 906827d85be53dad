// Shared device / launch helpers of libgla_hip.so (gfx950 only; wave = 64 lanes): tile map and workgroup -> tile
// mapping, branch-free tile loop, vector load / store, DPP lane-group sums, the two-stage column reduction, grid
// sizing and the dtype / K dispatch macros.  Included by gla_rows.hip and gla_edges.hip; everything lives in an
// anonymous namespace (one private copy per translation unit).
#pragma once
#include "gla_internal.h"

#include <type_traits>

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
  int balanced = 0;           // 1 / 2: XCD e takes floor(n / 8) consecutive tiles, the n % 8 extra ones go to the first (1) / last (2) XCDs
  int lds_cap = 0x7fffffff;   // tiles with more nonzeros than this are single long rows (CSR-adaptive "long" bin)
  int vec = 0;                // CSR-adaptive "vector" bin: rows of >= GLA_WAVE_ROW nonzeros are summed by their wave (option row_sum)
};

// grid = 8 * tiles_per_xcd workgroups; with chunking the per-XCD count is rounded up to whole chunks so that the
// slot formula of map_tile is a bijection onto [0, 8 * tiles_per_xcd) (slots >= n_tiles exit at once)
inline void set_tile_mapping(TileMap& tm) {
  tm.xcd_chunk = g_tuning.xcd_chunk;
  tm.vec = g_tuning.row_sum;
  int C = tm.xcd_chunk > 0 ? tm.xcd_chunk : 1;
  int rounds = (tm.n_tiles + GLA_XCDS * C - 1) / (GLA_XCDS * C);
  if (rounds < 1) rounds = 1;
  if (tm.xcd_chunk > 0) {
    // the tiles spread EVENLY over the XCDs: the chunk shrinks to an eighth of a round's tiles.  With a fixed chunk of 64 a launch
    // of 64 tiles (a 16 k-row multigrid level, the edge blocks of the multi-GPU split) ran on XCD 0 alone -- 32 of the 256 CUs
    C = (tm.n_tiles + GLA_XCDS * rounds - 1) / (GLA_XCDS * rounds);
    if (C < 1) C = 1;
    tm.xcd_chunk = C;
  }
  tm.tiles_per_xcd = rounds * C;
}

__device__ __forceinline__ int map_tile(const TileMap& m) {
  const int bid = blockIdx.x;
  // Workgroups b and b+8 share an XCD (round-robin dispatch).  xcd_chunk = C > 0: XCD e takes C consecutive tiles,
  // then the next XCD the next C, ... (C = 1 is the plain identity mapping); xcd_chunk = 0: each XCD walks one
  // contiguous eighth of the tiles.  Measured (tools/ab_spmv.py): small chunks win -- all XCDs then stream one
  // common region of HBM, and a tile's x window (a few grid lines) is re-fetched by neighbouring XCDs from the
  // Infinity Cache rather than from DRAM.
  int slot;
  if (m.balanced) {
    // Two launches that share the chip (the interior and the edge launch of a split multi-GPU pass: 442 + 68 workgroups on 512
    // slots, two per CU) must fit PER XCD -- 64 slots each.  The chunked numbering below gives seven XCDs 56 + 9 = 65 workgroups:
    // one edge workgroup per XCD waited ~95 us for an interior workgroup to finish, and with it the next pass's exchange
    // (tools/time_edge_blocks.py: the edge launch took 90 us beside the interior launch, 32 us alone).  Balanced: every XCD gets
    // floor(n / 8) tiles and the n % 8 extra ones go to the FIRST XCDs in one launch (1), to the LAST in the other (2).
    const int xcd = bid % GLA_XCDS, j = bid / GLA_XCDS, q = m.n_tiles / GLA_XCDS, r = m.n_tiles % GLA_XCDS;
    const int extra = m.balanced == 1 ? (xcd < r ? 1 : 0) : (xcd >= GLA_XCDS - r ? 1 : 0);
    if (j >= q + extra) return -1;
    const int before = m.balanced == 1 ? (xcd < r ? xcd : r) : (xcd > GLA_XCDS - r ? xcd - (GLA_XCDS - r) : 0);
    slot = xcd * q + before + j;
    return m.tile_list ? m.tile_list[slot] : slot;
  }
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
// small device helpers
// ------------------------------------------------------------------------------------------------
template <typename V, typename T>
__device__ __forceinline__ V ldv(const T* p) { return *reinterpret_cast<const V*>(p); }
template <typename V, typename T>
__device__ __forceinline__ void stv(T* p, V v) { *reinterpret_cast<V*>(p) = v; }

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

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

// ------------------------------------------------------------------------------------------------
// CSR-adaptive "long" bin: a row with more than GLA_LONG_ROW nonzeros is a row tile of its own.  The whole workgroup
// works on it: the nonzeros are cut into C = 256 / KL contiguous chunks (KL = lanes across the right-hand sides), every
// lane adds its chunk in edge order, and the C partial sums are combined by a fixed binary tree over the chunk index --
// deterministic and launch-independent, but (unlike the stream tiles' strictly sequential sums) not the association
// order of a CPU scatter-add.  `sh` is workgroup LDS scratch of GLA_BLOCK elements.
// ------------------------------------------------------------------------------------------------
template <typename T, class Op>
__device__ __forceinline__ T block_tree(T v, T* sh, int KL, Op op) {
  const int tid = threadIdx.x;
  __syncthreads();
  sh[tid] = v;
  __syncthreads();
  for (int st = GLA_BLOCK / 2; st >= KL; st >>= 1) {
    if (tid < st) sh[tid] = op(sh[tid], sh[tid + st]);
    __syncthreads();
  }
  return sh[tid % KL];
}

struct ChunkRange { int a, b; };
__device__ __forceinline__ ChunkRange chunk_of(int s, int e, int c, int C) {
  const int len = e - s, clen = (len + C - 1) / C;
  return ChunkRange{s + min(c * clen, len), s + min((c + 1) * clen, len)};
}

template <typename T, class Epi>
__device__ void long_row_spmm(const int32_t* __restrict__ col, const T* __restrict__ val, int s, int e, int r, int t,
                              const T* __restrict__ X, int64_t ldx, int K, const Epi& epi, T* sh) {
  int KL = 1;
  while (KL * 2 <= K && KL < 64) KL <<= 1;
  const int tid = threadIdx.x, kk = tid % KL;
  const ChunkRange cr = chunk_of(s, e, tid / KL, GLA_BLOCK / KL);
  for (int kb = 0; kb < K; kb += KL) {
    const bool live = kb + kk < K;
    const int k = live ? kb + kk : K - 1;
    const auto gxs = epi.template gx_load<T>(k);
    T acc = T(0);
    int j = cr.a;
    for (; j + 8 <= cr.b; j += 8) {                 // eight independent gathers in flight, added in edge order
      T v[8], xv[8];
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = val[j + u]; c[u] = col[j + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = epi.gx(X[(int64_t)c[u] * ldx + k], gxs);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = acc + v[u] * xv[u];
    }
    for (; j < cr.b; ++j) acc = acc + val[j] * epi.gx(X[(int64_t)col[j] * ldx + k], gxs);
    const T tot = block_tree(acc, sh, KL, [](T a, T b) { return a + b; });
    if (tid < KL && live) {
      const auto pre = epi.load((int64_t)r, k);
      epi.apply((int64_t)r, k, tot, pre);
      if constexpr (Epi::NRED > 0) epi.partials_ptr()[(int64_t)t * K + k] = epi.redv(tot, pre);
    }
  }
}

inline int check_plan(gla_csr* A, int dtype, int K) {
  if (!A || K < 1 || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (A->nnz > 0 && A->val_dtype != dtype) return GLA_ERR_NO_VALUES;
  return GLA_OK;
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
  __shared__ T red[GLA_BLOCK];                   // one workgroup per column k = blockIdx.x, fixed-order tree
  const int tid = threadIdx.x, k = blockIdx.x;
  T acc = T(0);
  for (int b = tid; b < nblocks; b += GLA_BLOCK) acc = acc + partial[(int64_t)b * K + k];
  red[tid] = acc;
  __syncthreads();
  for (int st = GLA_BLOCK / 2; st > 0; st >>= 1) {
    if (tid < st) red[tid] = red[tid] + red[tid + st];
    __syncthreads();
  }
  if (tid == 0) out[k] = red[0];
}

// Sum of per-lane vectors over the lanes of a workgroup that share tid % KV (KV a power of two <= 32): xor-shuffles
// inside each wave, then the four wave results through LDS in wave order.  Fixed order; the result is valid in lanes
// tid < KV.  `sh` holds NW * KV vectors (NW = waves of the workgroup).
template <int KV, typename V, int NW = 4>
__device__ __forceinline__ V block_lane_group_sum(V v, V* sh) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int off = 32; off >= KV; off >>= 1) {
    V o;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(V) / sizeof(v[0])); ++i) o[i] = __shfl_xor(v[i], off, 64);
    v = v + o;
  }
  __syncthreads();
  if ((tid & 63) < KV) sh[(tid >> 6) * KV + (tid & 63)] = v;
  __syncthreads();
  V tot = V();
  if (tid < KV) {
    tot = sh[tid] + sh[KV + tid];
#pragma unroll
    for (int w = 2; w < NW; ++w) tot = tot + sh[w * KV + tid];
  }
  return tot;
}

inline unsigned grid_for(int64_t work, int per_block = GLA_BLOCK, int cap = 256 * 8) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

inline TileMap all_tiles(gla_csr* A) {
  TileMap tm;
  tm.lds_cap = A->lds_elems;
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
  hipLaunchKernelGGL((col_dot_stage2<T>), dim3(K), dim3(GLA_BLOCK), 0, st, pw, (int)nb, K, out);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
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

}  // namespace

#define GLA_DISPATCH(dtype, ...)                              \
  if ((dtype) == GLA_F64) { using T = double; __VA_ARGS__; }  \
  else { using T = float; __VA_ARGS__; }
