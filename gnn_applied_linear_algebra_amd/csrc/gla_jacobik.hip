// libgla_hip.so -- temporally blocked Jacobi sweeps for K right-hand sides on canonical stencil plans (gfx950 / MI355X).
//
// JacobiGNN on K right-hand sides (vertex_attr [n, 3, K]: the multi-RHS extension of JacobiGNN.py:138-148; the smoother of a
// V-cycle run on several right-hand sides at once) is, sweep by sweep, the blocked SpMM kernel with the EpiJacobi epilogue:
// every sweep streams b, x in and x out (24 K bytes per row) plus the matrix.  Only the last iterate leaves the call, so --
// exactly like jacobi_march5 / jacobi_march9 (gla_rows.hip) and cheby_march5 (gla_cheby.hip) -- TS sweeps are taken in ONE
// pass: a workgroup owns a strip of 256 / KV grid points x K right-hand sides (lanes = (point, 16-byte rhs vector)) and
// marches down the grid lines; the own column of every level lives in registers (h[t][0..2]), LDS carries the i -/+ 1
// neighbours (three lines per level, slot = line mod 3, one barrier per level and line step), coefficients and b of the last
// TS lines stay in registers under static names, a wave whose rows all carry the full pattern skips the mask selects.
// Per (row, rhs) the arithmetic of the blocked SpMM kernel + EpiJacobi -- in-order masked sum over the diagonals, then
// x + (w * (b - sum)) / A_ii (JacobiGNN.py:119) -- so the result is bit-identical to TS single sweeps.
#include "gla_device.h"

namespace {

struct JacK {
  const void* val;                     // [D][n_rows] diagonal-major values
  const uint16_t* mask;                // [n_rows]
  int32_t S;
  int64_t n_rows, n_cols;
};

template <typename T, int D>
struct KCoef {                                     // one row's matrix data, A_ii, mask as loaded, row exists
  T v[(D + 3) & ~3];                               // (padded to whole 16-byte groups: the compiler keeps the set in registers)
  T dg;
  int m, lv;
};

// D = 5: diagonals (-1,0) (0,-1) (0,0) (0,1) (1,0); D = 9: (a, b) in {-1, 0, 1}^2 row-major
template <typename T, int KV, int TS, int D>
__global__ __launch_bounds__(GLA_BLOCK) void jacobi_marchk_kernel(JacK jv, const T* __restrict__ diag,
                                                                  const T* __restrict__ bvec, int64_t ldb,
                                                                  const T* __restrict__ x0, int64_t ldx0,
                                                                  T* __restrict__ xout, int64_t ldxo, T w, int lines,
                                                                  int nstrips, int64_t nlines, TileMap tm) {
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int BP = GLA_BLOCK / KV;               // grid points of a strip
  constexpr int LWV = (BP + 2) * KV;               // 16-byte vectors of a line in LDS: point ii at [(1 + ii) * KV + q]
  constexpr int NC = TS + 1, U = NC;
  constexpr unsigned FULL = (1u << D) - 1u;
  typedef T TV __attribute__((ext_vector_type(VW)));
  typedef KCoef<T, D> Coef;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* buf = reinterpret_cast<TV*>(smem_raw);       // 9-point: [TS][3][LWV], slot = line mod 3; 5-point: [TS][2][LWV], slot = parity
  const int blk = map_tile(tm);
  if (blk < 0) return;
  constexpr int halo = TS, bp_eff = BP - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t J0 = (int64_t)lb * lines, P0 = (int64_t)strip * bp_eff - halo;
  const int L = (int)min((int64_t)lines, nlines - J0);
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  const T* __restrict__ val = static_cast<const T*>(jv.val);
  const int64_t gi = P0 + ii;
  const bool in_i = gi >= 0 && gi < jv.S;
  const bool owner = ii >= halo && ii < BP - halo;

  auto x0_at = [&](int jl) -> TV {                 // clamped: values of dead rows are never used
    int64_t flat = (J0 + jl) * jv.S + gi;
    flat = min(max(flat, (int64_t)0), jv.n_cols - 1);
    return ldv<TV>(x0 + flat * ldx0 + q * VW);
  };
  auto coef_at = [&](int jl) -> Coef {
    const int64_t gj = J0 + jl, r = gj * jv.S + gi;
    const bool live = in_i && gj >= 0 && gj < nlines && r < jv.n_rows;
    const int64_t rc = live ? r : 0;
    Coef c;
#pragma unroll
    for (int d = 0; d < D; ++d) c.v[d] = val[(int64_t)d * jv.n_rows + rc];
    c.dg = diag[rc];
    c.m = (int)jv.mask[rc];
    c.lv = live;
    return c;
  };
  auto b_at = [&](int jl) -> TV {                  // b of the row (line J0 + jl, this lane's point), this lane's rhs vector
    const int64_t gj = J0 + jl, r = gj * jv.S + gi;
    const bool live = in_i && gj >= 0 && gj < nlines && r < jv.n_rows;
    return ldv<TV>(bvec + (live ? r : 0) * ldb + q * VW);
  };
  auto update = [&](const Coef& c, TV cb, bool full, const TV (&xs)[D], TV ctr) -> TV {
    TV acc = TV(T(0));
    if (full) {
#pragma unroll
      for (int d = 0; d < D; ++d) acc = acc + c.v[d] * xs[d];
    } else {
      const int m = c.lv ? c.m : 0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const TV nxt = acc + c.v[d] * xs[d];
        acc = ((m >> d) & 1) ? nxt : acc;
      }
    }
    return ctr + (w * (cb - acc)) / c.dg;
  };

  constexpr int j_first = -(TS - 1);
  const int j_last = L - 1 + (TS - 1);
  TV h[TS][3];                                     // h[t]: x^(k+t) of this lane on the last three lines level t produced
#pragma unroll
  for (int t = 0; t < TS; ++t) h[t][0] = h[t][1] = h[t][2] = TV(T(0));
  h[0][1] = x0_at(j_first - 1);
  h[0][2] = x0_at(j_first);
  if constexpr (D == 5) {
    buf[((j_first + 64) & 1) * LWV + (1 + ii) * KV + q] = h[0][2];
  } else {
    buf[((j_first - 1 + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][1];
    buf[((j_first + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][2];
  }
  TV xpre = x0_at(j_first + 1);
  Coef c[NC];                                      // set of line j_first + s lives in c[s % NC]
  TV cb[NC];                                       // ... and the row's b
  bool cfull[NC];
  c[0] = coef_at(j_first);
  cb[0] = b_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) { c[s] = c[0]; cb[s] = cb[0]; }
#pragma unroll
  for (int s = 0; s < NC; ++s) cfull[s] = false;

  auto do_step = [&](const int jl, const int u, const int nact) {
    if ((jl & 1) ^ (((int)blockIdx.x / GLA_XCDS) >> 5 & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (priority alternation between a CU's two workgroups: gla_march5.hip)
    if constexpr (D == 5) {
      // 5-point: a level only needs the i -/+ 1 neighbours of ITS OWN line, which the level below wrote one line step
      // earlier: two lines per level (slot = line parity) and ONE barrier per line step (as jacobi_march5 / cheby_march5)
      __syncthreads();
      const int pj = (jl + 64) & 1;
      TV* const row_e = buf + pj * LWV + ii * KV + q;          // lines of the parity of jl: [0] left, [KV] own, [2 KV] right
      TV* const row_o = buf + (1 - pj) * LWV + ii * KV + q;
      h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
      row_o[KV] = xpre;                            // level 0, line jl + 1
      xpre = x0_at(jl + 2);                        // clamped / dead beyond the block: harmless, never used
      c[(u + 1) % NC] = coef_at(jl + 1);
      cb[(u + 1) % NC] = b_at(jl + 1);
      cfull[u % NC] = __all(c[u % NC].lv && ((unsigned)c[u % NC].m & FULL) == FULL) != 0;   // (at first use)
#pragma unroll
      for (int t = 1; t <= TS; ++t) {
        if (t <= nact) {
          TV* const row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LWV;   // level t - 1, line jl - (t - 1)
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          const TV xs[5] = {h[t - 1][0], row[0], h[t - 1][1], row[2 * KV], h[t - 1][2]};
          const TV xn = update(c[ci], cb[ci], cfull[ci], xs, h[t - 1][1]);
          if (t < TS) {
            h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
            row[2 * LWV + KV] = xn;                // level t, same line (same parity)
          } else if (c[ci].lv && owner) {
            stv<TV>(xout + ((J0 + jl - (TS - 1)) * jv.S + gi) * ldxo + q * VW, xn);
          }
        }
      }
    } else {
      const int s1 = (jl + 1 + 63) % 3, s0 = (jl + 63) % 3, s2 = (jl - 1 + 63) % 3;
      // LDS rows of the lines jl + 1, jl, jl - 1 (level 0): [0] left, [KV] own, [2 KV] right of this lane's point
      TV* const rw[3] = {buf + s1 * LWV + ii * KV + q, buf + s0 * LWV + ii * KV + q, buf + s2 * LWV + ii * KV + q};
      h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
      rw[0][KV] = xpre;                            // level 0, line jl + 1
      xpre = x0_at(jl + 2);                        // clamped / dead beyond the block: harmless, never used
      c[(u + 1) % NC] = coef_at(jl + 1);
      cb[(u + 1) % NC] = b_at(jl + 1);
      cfull[u % NC] = __all(c[u % NC].lv && ((unsigned)c[u % NC].m & FULL) == FULL) != 0;   // (at first use)
#pragma unroll
      for (int t = 1; t <= TS; ++t) {
        __syncthreads();
        if (t <= nact) {
          // lines line + 1, line, line - 1 of level t - 1 (line = jl - (t - 1)): the three rows rotated by t - 1
          TV* const below = rw[(t - 1) % 3] + (t - 1) * 3 * LWV;
          TV* const same = rw[t % 3] + (t - 1) * 3 * LWV;
          TV* const above = rw[(t + 1) % 3] + (t - 1) * 3 * LWV;
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          const TV xs[9] = {above[0], h[t - 1][0], above[2 * KV], same[0], h[t - 1][1], same[2 * KV],
                            below[0], h[t - 1][2], below[2 * KV]};
          const TV xn = update(c[ci], cb[ci], cfull[ci], xs, h[t - 1][1]);
          if (t < TS) {
            h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
            same[3 * LWV + KV] = xn;               // level t, same line (same slot)
          } else if (c[ci].lv && owner) {
            stv<TV>(xout + ((J0 + jl - (TS - 1)) * jv.S + gi) * ldxo + q * VW, xn);
          }
        }
      }
    }
  };
  // ramp-up: step s = jl - j_first < 2 (TS - 1) has levels 1 .. (s + 2) / 2 active (x^(k+t) starts on line -(TS - t))
#pragma unroll
  for (int s = 0; s < 2 * (TS - 1); ++s) do_step(j_first + s, s % U, (s + 2) / 2);
  constexpr int u0 = (2 * (TS - 1)) % U;
#pragma unroll 1
  for (int jl0 = TS - 1;; jl0 += U) {
#pragma unroll
    for (int qq = 0; qq < U; ++qq) {
      if (jl0 + qq > j_last) return;
      do_step(jl0 + qq, (u0 + qq) % U, TS);
    }
  }
}

template <typename T, int KV, int TS, int D>
int launch_marchk(gla_csr* A, const T* diag, const T* b, int64_t ldb, const T* x, int64_t ldx, T* xo, int64_t ldxo, T w,
                  hipStream_t st) {
  constexpr int BP = GLA_BLOCK / KV;
  constexpr int bp_eff = BP - 2 * TS;
  static_assert(bp_eff >= 1, "strip too narrow");
  JacK jv;
  jv.val = A->dia_val;
  jv.mask = A->dia_mask;
  jv.S = A->dia_S;
  jv.n_rows = A->n_rows; jv.n_cols = A->n_cols;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  const int nstrips = (int)((A->dia_S + bp_eff - 1) / bp_eff);
  const size_t lds = (size_t)TS * (D == 5 ? 2 : 3) * (BP + 2) * KV * 16;
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) lines = gla_pick_block_lines(nlines, nstrips, 256 * 2, 2 * (TS - 1), 2 * TS);
  if (lines < 2 * TS) lines = 2 * TS;
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = (int)(((nlines + lines - 1) / lines) * nstrips);
  set_tile_mapping(tm);
  hipLaunchKernelGGL((jacobi_marchk_kernel<T, KV, TS, D>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(GLA_BLOCK), lds,
                     st, jv, diag, b, ldb, x, ldx, xo, ldxo, w, lines, nstrips, nlines, tm);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T, int KV, int D>
int dispatch_ts(int ts, gla_csr* A, const T* diag, const T* b, int64_t ldb, const T* x, int64_t ldx, T* xo, int64_t ldxo, T w,
                hipStream_t st) {
  switch (ts) {
    case 2: return launch_marchk<T, KV, 2, D>(A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    case 3: return launch_marchk<T, KV, 3, D>(A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    case 4:
      if constexpr (D == 5) return launch_marchk<T, KV, 4, D>(A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
      else return GLA_ERR_UNSUPPORTED;               // (9-point, four sweeps: 255 VGPRs, one wave per SIMD)
    default: return GLA_ERR_UNSUPPORTED;
  }
}

template <typename T, int D>
int dispatch_kv(int kv, int ts, gla_csr* A, const T* diag, const T* b, int64_t ldb, const T* x, int64_t ldx, T* xo,
                int64_t ldxo, T w, hipStream_t st) {
  switch (kv) {
    case 1: return dispatch_ts<T, 1, D>(ts, A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    case 2: return dispatch_ts<T, 2, D>(ts, A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    case 4: return dispatch_ts<T, 4, D>(ts, A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    case 8: return dispatch_ts<T, 8, D>(ts, A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
    default: return GLA_ERR_UNSUPPORTED;
  }
}

}  // namespace

// `sweeps` (2 .. GLA_JACOBIK_TS_MAX) Jacobi sweeps of K right-hand sides in one pass; GLA_ERR_UNSUPPORTED when the plan is not
// a canonical 5- / 9-point stencil, K is not 1, 2, 4 or 8 16-byte vectors, or an operand is not 16-byte aligned.
int gla_jacobik_pass(gla_csr* A, int dtype, int K, const void* diag, const void* B, int64_t ldb, const void* X, int64_t ldx,
                     void* Xout, int64_t ldxo, double w, int sweeps, hipStream_t st) {
  const int vw = dtype == GLA_F64 ? 2 : 4;
  const bool c5 = gla_dia_canon5(A), c9 = gla_dia_canon9(A);
  if (!g_tuning.jacobi_march || !g_tuning.dia || !(c5 || c9) || !A->dia_val || !A->dia_nowrap || A->dia_row_lo != 0 ||
      A->dia_row_hi != A->n_rows || A->n_rows != A->n_cols || K % vw != 0 || ldb % vw != 0 || ldx % vw != 0 || ldxo % vw != 0 ||
      !al16(B) || !al16(X) || !al16(Xout) || sweeps < 2 || sweeps > GLA_JACOBIK_TS_MAX)
    return GLA_ERR_UNSUPPORTED;
  const int kv = K / vw;
  if (dtype == GLA_F64) {
    if (c5) return dispatch_kv<double, 5>(kv, sweeps, A, (const double*)diag, (const double*)B, ldb, (const double*)X, ldx, (double*)Xout, ldxo, (double)w, st);
    return dispatch_kv<double, 9>(kv, sweeps, A, (const double*)diag, (const double*)B, ldb, (const double*)X, ldx, (double*)Xout, ldxo, (double)w, st);
  }
  if (c5) return dispatch_kv<float, 5>(kv, sweeps, A, (const float*)diag, (const float*)B, ldb, (const float*)X, ldx, (float*)Xout, ldxo, (float)w, st);
  return dispatch_kv<float, 9>(kv, sweeps, A, (const float*)diag, (const float*)B, ldb, (const float*)X, ldx, (float*)Xout, ldxo, (float)w, st);
}
