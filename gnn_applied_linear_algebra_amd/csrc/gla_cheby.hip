// libgla_hip.so -- temporally blocked Chebyshev relaxation on 2-D stencil plans, K right-hand sides (gfx950 / MI355X).
//
// ChebyRelaxGNN(deg) (pytorch/ChebyGNN.py:287-353) is deg dependent SpMM steps,
//     step 1     r = b - A x;  p = r;  x += alpha_1 r                         (ChebyGNN.py:49-163)
//     step k>=2  r -= alpha_{k-1} (A p);  p = r + beta_k p;  x += alpha_k p     (ChebyGNN.py:166-243)
// each of which streams b / x / r / p (K values per vertex each) through HBM when run as one kernel per step
// (gla_cheby_first / gla_cheby_next: 11.4 + 13.6 (deg - 1) GB at 4096^2 x 16 fp64).  Only the LAST x, r, p leave the
// module, so -- exactly like the temporally blocked Jacobi sweeps (gla_rows.hip) -- the steps are fused into ONE pass:
// a workgroup owns a strip of BP = 256 / KV grid points x K right-hand sides (lanes = (point, 16-byte rhs vector)) and
// marches down the grid lines; the operand each step gathers (x for step 1, p_{k-1} for step k) lives in a 3-line LDS
// ring per step, r and x travel from one step to the next in registers (a row meets step k one line step after step
// k - 1), the coefficients of the last DEG lines stay in registers.  HBM sees b and x once in, x, r, p once out.
// Step t is valid on the strip minus t * B points either side (strips overlap by 2 DEG B points).
// Per row the arithmetic is that of the EpiChebyFirst / EpiChebyNext epilogues (in-order masked sum over the diagonals,
// then the same vector updates, product and sum rounded separately): bit-identical to the step-by-step kernels.
#include "gla_device.h"

namespace {

constexpr int CH_DMAX = 5;             // 5-point-type stencils (the 9-point FEM operator keeps one kernel per step)
constexpr int CH_DEG_MAX = 4;

struct Cheb2D {
  const void* val;                     // [D][n_rows] diagonal-major values
  const uint16_t* mask;                // [n_rows]
  int32_t D, S, B;
  int32_t da[CH_DMAX], db[CH_DMAX];
  int64_t n_rows, n_cols;
  double alpha[CH_DEG_MAX], beta[CH_DEG_MAX];      // alpha[k - 1] = alpha_k, beta[k - 1] = beta_k (beta_1 unused)
  // continuation pass (march kernels only): the pass starts from (r, x, p) of an earlier step instead of (b, x0) -- its
  // first step is an EpiChebyNext step too: `bvec` holds r, `x0` the operand p, `bcopy` (an INPUT then) x, alpha_prev the
  // alpha of the step before, alpha[0] / beta[0] those of the pass's first step
  int32_t cont;
  double alpha_prev;
};

template <typename T>
struct ChCoef9 { T v[12]; int m, lv, pad_[2]; };   // the 9-point pattern's row (values padded to whole 16-byte groups)
template <typename T>
struct ChCoef { T v[CH_DMAX]; int m, lv, pad_; };        // one row's diagonal values, mask as loaded (untouched until its first
                                                   // use, so that the load stays in flight across the line step), row exists

template <typename T, int KV, int DEG>
__global__ __launch_bounds__(GLA_BLOCK) void cheby_fused_dia2d_kernel(
    Cheb2D cv, const T* __restrict__ bvec, int64_t ldb, const T* __restrict__ x0, int64_t ldx0, T* __restrict__ xout,
    int64_t ldxo, T* __restrict__ rout, int64_t ldr, T* __restrict__ pout, int64_t ldp, T* __restrict__ pprev,
    int64_t ldpp, T* __restrict__ bcopy, int64_t ldbc, int lines, int nstrips, int64_t nlines, TileMap tm) {
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int BP = GLA_BLOCK / KV;               // grid points of a strip
  typedef T TV __attribute__((ext_vector_type(VW)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* ring = reinterpret_cast<TV*>(smem_raw);      // [DEG][3][BP * KV]: ring t = three lines of the operand step t + 1 gathers
  constexpr int RING = 3 * BP * KV;
  const int blk = map_tile(tm);
  if (blk < 0) return;
  const int B = cv.B, halo = DEG * B, bp_eff = BP - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t J0 = (int64_t)lb * lines, P0 = (int64_t)strip * bp_eff - halo;
  const int L = (int)min((int64_t)lines, nlines - J0);
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  const T* __restrict__ val = static_cast<const T*>(cv.val);
  int da[CH_DMAX], db[CH_DMAX];
#pragma unroll
  for (int d = 0; d < CH_DMAX; ++d) { const int dd = min(d, cv.D - 1); da[d] = cv.da[dd]; db[d] = cv.db[dd]; }

  auto slot = [](int line) { return (line + 9) % 3; };
  auto live_row = [&](int jl, int64_t& r) -> bool {
    const int64_t gj = J0 + jl, gi = P0 + ii;
    r = gj * cv.S + gi;
    return gj >= 0 && gj < nlines && gi >= 0 && gi < cv.S && r < cv.n_rows;
  };
  auto x0_at = [&](int jl) -> TV {                 // clamped: values of dead rows are never used
    int64_t flat = (J0 + jl) * cv.S + P0 + ii;
    flat = min(max(flat, (int64_t)0), cv.n_cols - 1);
    return ldv<TV>(x0 + flat * ldx0 + q * VW);
  };
  auto b_at = [&](int jl) -> TV {
    int64_t r;
    const bool live = live_row(jl, r);
    return ldv<TV>(bvec + (live ? r : 0) * ldb + q * VW);
  };
  auto coef_at = [&](int jl) -> ChCoef<T> {
    int64_t r;
    const bool live = live_row(jl, r);
    const int64_t rc = live ? r : 0;
    ChCoef<T> c;
#pragma unroll
    for (int d = 0; d < CH_DMAX; ++d) c.v[d] = val[(int64_t)min(d, cv.D - 1) * cv.n_rows + rc];
    c.m = (int)cv.mask[rc];
    c.lv = live;
    return c;
  };
  // in-order masked row sum of A g for this lane's (row, rhs vector) from ring t (lines line - 1 .. line + 1)
  auto stencil = [&](const ChCoef<T>& c, int t, int line) -> TV {
    const TV* rg = ring + (size_t)t * RING;
    TV acc = TV(T(0));
    const int m = c.lv ? c.m : 0;
#pragma unroll
    for (int d = 0; d < CH_DMAX; ++d) {
      const int pos = min(max(ii + db[d], 0), BP - 1);             // strip-edge lanes: their results are never used
      const TV xv = rg[slot(line + da[d]) * BP * KV + pos * KV + q];
      const TV nxt = acc + c.v[d] * xv;
      acc = ((m >> d) & 1) ? nxt : acc;
    }
    return acc;
  };
  auto centre = [&](int t, int line) -> TV { return ring[(size_t)t * RING + slot(line) * BP * KV + ii * KV + q]; };

  // step 1 is formed on lines -(DEG-1) .. L-1+(DEG-1): it needs x on lines -DEG .. L-1+DEG
  const int j_first = -(DEG - 1), j_last = L - 1 + (DEG - 1);
  ring[slot(j_first - 1) * BP * KV + ii * KV + q] = x0_at(j_first - 1);
  ring[slot(j_first) * BP * KV + ii * KV + q] = x0_at(j_first);
  TV xpre = x0_at(j_first + 1), bcur = b_at(j_first), bnxt = bcur;
  ChCoef<T> c[DEG];                                // c[t]: coefficients of line jl - t
  c[0] = coef_at(j_first);
#pragma unroll
  for (int t = 1; t < DEG; ++t) c[t] = c[0];
  ChCoef<T> cnxt = c[0];
  TV cr[DEG], cx[DEG];                             // r and x of the row step t + 1 meets at the NEXT line step (cr[t]: after step t)
#pragma unroll
  for (int t = 0; t < DEG; ++t) cr[t] = cx[t] = TV(T(0));
  const bool interior = ii >= halo && ii < BP - halo;
#pragma unroll 1
  for (int jl = j_first; jl <= j_last; ++jl) {
    ring[slot(jl + 1) * BP * KV + ii * KV + q] = xpre;               // x line jl + 1 joins ring 0
    if (jl + 2 <= j_last + 1) xpre = x0_at(jl + 2);                  // prefetch for the next line step
    if (jl + 1 <= j_last) { cnxt = coef_at(jl + 1); bnxt = b_at(jl + 1); }
    TV in_r = TV(T(0)), in_x = TV(T(0));
#pragma unroll
    for (int t = 1; t <= DEG; ++t) {
      __syncthreads();
      const int line = jl - (t - 1);
      const TV use_r = in_r, use_x = in_x;          // what step t - 1 left for THIS row one line step ago
      if (t < DEG) { in_r = cr[t]; in_x = cx[t]; }  // keep step t's previous results for step t + 1 before overwriting them
      const TV acc = stencil(c[t - 1], t - 1, line);
      TV ri, pn, xn;
      if (t == 1) {                                 // EpiChebyFirst: r = b - A x; p = r; x = x + alpha_1 r
        if (bcopy) {                                // the caller's [b, x, r, p] result wants b too: it is in a register anyway
          int64_t rb;
          if (live_row(line, rb) && interior && line >= 0 && line < L) stv<TV>(bcopy + rb * ldbc + q * VW, bcur);
        }
        ri = bcur - acc;
        pn = ri;
        xn = centre(0, line) + (T)cv.alpha[0] * ri;
      } else {                                      // EpiChebyNext: r -= a_old (A p); p = r + beta p; x += a_new p
        ri = use_r - (T)cv.alpha[t - 2] * acc;
        pn = ri + (T)cv.beta[t - 1] * centre(t - 1, line);
        xn = use_x + (T)cv.alpha[t - 1] * pn;
      }
      int64_t r;
      const bool store_ok = live_row(line, r) && interior && line >= 0 && line < L && c[t - 1].lv;
      if (t < DEG) {
        ring[(size_t)t * RING + slot(line) * BP * KV + ii * KV + q] = pn;
        cr[t] = ri;
        cx[t] = xn;
        if (t == DEG - 1 && pprev && store_ok) stv<TV>(pprev + r * ldpp + q * VW, pn);   // the operand the last step gathers
      } else if (store_ok) {
        stv<TV>(xout + r * ldxo + q * VW, xn);
        stv<TV>(rout + r * ldr + q * VW, ri);
        stv<TV>(pout + r * ldp + q * VW, pn);
      }
    }
#pragma unroll
    for (int t = DEG - 1; t >= 1; --t) c[t] = c[t - 1];
    c[0] = cnxt;
    bcur = bnxt;
  }
}

// ------------------------------------------------------------------------------------------------
// The same pass for the canonical 5-point pattern (diagonals (-1,0) (0,-1) (0,0) (0,1) (1,0), B = 1), organised like
// jacobi_march5_kernel (gla_rows.hip): the j -/+ 1 neighbours and the centre of the operand a step gathers are the lane's
// OWN values of the last three line steps (registers, h[t][0..2]); LDS only carries the i -/+ 1 neighbours -- one padded
// line per step and line parity, one 16-byte write + two 16-byte reads per update -- and since a step reads what the step
// below wrote one line step EARLIER there is ONE barrier per line step; the line loop is unrolled over the DEG + 1
// coefficient sets (static names, no register rotation); a wave whose rows all carry the full pattern skips the mask
// selects.  Same arithmetic per (row, rhs) as the kernel above and as the step-by-step epilogues: bit-identical.
// Rows above a step's first valid line carry garbage through the first line steps; it never reaches a stored value
// (step t is valid from line -(DEG - t) on, and a valid line only reads valid lines of the step below).
// ------------------------------------------------------------------------------------------------
template <typename T, int KV, int DEG, int VW = 16 / (int)sizeof(T), int NT = GLA_BLOCK>      // VW = 1: one right-hand side (lanes = points)
__global__ __launch_bounds__(NT) void cheby_march5_kernel(
    Cheb2D cv, const T* __restrict__ bvec, int64_t ldb, const T* __restrict__ x0, int64_t ldx0, T* __restrict__ xout,
    int64_t ldxo, T* __restrict__ rout, int64_t ldr, T* __restrict__ pout, int64_t ldp, T* __restrict__ pprev,
    int64_t ldpp, T* __restrict__ bcopy, int64_t ldbc, int lines, int nstrips, int64_t nlines, TileMap tm) {
  constexpr int BP = NT / KV;               // grid points of a strip
  constexpr int LWV = (BP + 2) * KV;               // 16-byte vectors of a line in LDS: point ii at [(1 + ii) * KV + q]
  constexpr int NC = DEG + 1, U = NC;
  typedef T TV __attribute__((ext_vector_type(VW)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* buf = reinterpret_cast<TV*>(smem_raw);       // [DEG][2][LWV]: buf[t][p] = latest line of parity p of the operand of step t + 1
  const int blk = map_tile(tm);
  if (blk < 0) return;
  constexpr int halo = DEG, bp_eff = BP - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t J0 = (int64_t)lb * lines, P0 = (int64_t)strip * bp_eff - halo;
  const int L = (int)min((int64_t)lines, nlines - J0);
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  const T* __restrict__ val = static_cast<const T*>(cv.val);
  const int64_t gi = P0 + ii;
  const bool in_i = gi >= 0 && gi < cv.S;
  const bool interior = ii >= halo && ii < BP - halo;

  auto live_row = [&](int jl, int64_t& r) -> bool {
    const int64_t gj = J0 + jl;
    r = gj * cv.S + gi;
    return in_i && gj >= 0 && gj < nlines && r < cv.n_rows;
  };
  auto x0_at = [&](int jl) -> TV {                 // clamped: values of dead rows are never used
    int64_t flat = (J0 + jl) * cv.S + gi;
    flat = min(max(flat, (int64_t)0), cv.n_cols - 1);
    return ldv<TV>(x0 + flat * ldx0 + q * VW);
  };
  auto b_at = [&](int jl) -> TV {
    int64_t r;
    const bool live = live_row(jl, r);
    return ldv<TV>(bvec + (live ? r : 0) * ldb + q * VW);
  };
  auto coef_at = [&](int jl) -> ChCoef<T> {
    int64_t r;
    const bool live = live_row(jl, r);
    const int64_t rc = live ? r : 0;
    ChCoef<T> c;
#pragma unroll
    for (int d = 0; d < CH_DMAX; ++d) c.v[d] = val[(int64_t)d * cv.n_rows + rc];
    c.m = (int)cv.mask[rc];
    c.lv = live;
    return c;
  };
  auto stencil = [&](const ChCoef<T>& c, bool full, TV up, TV left, TV ctr, TV right, TV down) -> TV {
    TV acc = TV(T(0));
    if (full) {
      acc = acc + c.v[0] * up;
      acc = acc + c.v[1] * left;
      acc = acc + c.v[2] * ctr;
      acc = acc + c.v[3] * right;
      acc = acc + c.v[4] * down;
    } else {
      const TV xs[5] = {up, left, ctr, right, down};
      const int m = c.lv ? c.m : 0;
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const TV nxt = acc + c.v[d] * xs[d];
        acc = ((m >> d) & 1) ? nxt : acc;
      }
    }
    return acc;
  };

  constexpr int j_first = -(DEG - 1);
  const int j_last = L - 1 + (DEG - 1);
  TV h[DEG][3];                                    // h[t]: the operand of step t + 1 (x for t = 0, p_t after) of this lane on
#pragma unroll                                     // the last three lines it was produced on
  for (int t = 0; t < DEG; ++t) h[t][0] = h[t][1] = h[t][2] = TV(T(0));
  h[0][1] = x0_at(j_first - 1);
  h[0][2] = x0_at(j_first);
  buf[((j_first + 64) & 1) * LWV + (1 + ii) * KV + q] = h[0][2];
  auto xin_at = [&](int jl) -> TV {                // continuation pass: x of the row (line J0 + jl), an input
    int64_t r;
    const bool live = live_row(jl, r);
    return ldv<TV>(bcopy + (live ? r : 0) * ldbc + q * VW);
  };
  TV xpre = x0_at(j_first + 1), bq = b_at(j_first);
  TV xinq = TV(T(0));
  if (cv.cont) xinq = xin_at(j_first);
  ChCoef<T> c[NC];                                 // set of line j_first + s lives in c[s % NC]
  bool cfull[NC];
  c[0] = coef_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
  for (int s = 0; s < NC; ++s) cfull[s] = false;
  TV cr[DEG], cx[DEG];                             // r and x step t left for this lane's row one line step ago
#pragma unroll
  for (int t = 0; t < DEG; ++t) cr[t] = cx[t] = TV(T(0));

  auto do_step = [&](const int jl, const int u) {
    if ((jl & 1) ^ (NT >= 512 ? ((int)threadIdx.x >> 8) & 1 : ((int)blockIdx.x / GLA_XCDS) >> 5 & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (priority alternation between a CU's two workgroups: gla_march5.hip)
    __syncthreads();
    const int pj = (jl + 64) & 1;
    TV* const row_e = buf + pj * LWV + ii * KV + q;          // lines of the parity of jl: [0] left, [KV] own, [2 KV] right
    TV* const row_o = buf + (1 - pj) * LWV + ii * KV + q;    // lines of the other parity
    h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
    row_o[KV] = xpre;                              // x, line jl + 1
    const TV bnow = bq;                            // b of line jl (loaded a step ago)
    const TV xinnow = xinq;
    if (cv.cont) xinq = xin_at(jl + 1);
    xpre = x0_at(jl + 2);                          // clamped / dead beyond the block: harmless, never used
    bq = b_at(jl + 1);
    c[(u + 1) % NC] = coef_at(jl + 1);
    cfull[u % NC] = __all(c[u % NC].lv && (c[u % NC].m & 0x1f) == 0x1f) != 0;   // (at first use: loaded a step ago)
    TV in_r = TV(T(0)), in_x = TV(T(0));
#pragma unroll
    for (int t = 1; t <= DEG; ++t) {
      const int line = jl - (t - 1);
      TV* const row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LWV;   // operand of step t, line `line`
      const int ci = ((u - (t - 1)) % NC + NC) % NC;
      const TV use_r = in_r, use_x = in_x;         // what step t - 1 left for THIS row one line step ago
      if (t < DEG) { in_r = cr[t]; in_x = cx[t]; } // keep step t's previous results for step t + 1 before overwriting them
      const TV acc = stencil(c[ci], cfull[ci], h[t - 1][0], row[0], h[t - 1][1], row[2 * KV], h[t - 1][2]);
      int64_t r;
      const bool store_ok = live_row(line, r) && interior && line >= 0 && line < L;
      TV ri, pn, xn;
      if (t == 1) {                                // EpiChebyFirst: r = b - A x; p = r; x = x + alpha_1 r
        if (!cv.cont) {
          if (bcopy && store_ok) stv<TV>(bcopy + r * ldbc + q * VW, bnow);
          ri = bnow - acc;
          pn = ri;
          xn = h[0][1] + (T)cv.alpha[0] * ri;
        } else {                                   // continuation: the pass's first step is a "next" step on loaded r, x
          ri = bnow - (T)cv.alpha_prev * acc;
          pn = ri + (T)cv.beta[0] * h[0][1];
          xn = xinnow + (T)cv.alpha[0] * pn;
        }
      } else {                                     // EpiChebyNext: r -= a_old (A p); p = r + beta p; x += a_new p
        ri = use_r - (T)cv.alpha[t - 2] * acc;
        pn = ri + (T)cv.beta[t - 1] * h[t - 1][1];
        xn = use_x + (T)cv.alpha[t - 1] * pn;
      }
      if (t < DEG) {
        h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = pn;
        row[2 * LWV + KV] = pn;                    // operand of step t + 1, same line (same parity)
        cr[t] = ri;
        cx[t] = xn;
        if (t == DEG - 1 && pprev && store_ok) stv<TV>(pprev + r * ldpp + q * VW, pn);   // the operand the last step gathers
      } else if (store_ok) {
        stv<TV>(xout + r * ldxo + q * VW, xn);
        stv<TV>(rout + r * ldr + q * VW, ri);
        stv<TV>(pout + r * ldp + q * VW, pn);
      }
    }
  };
#pragma unroll 1
  for (int jl0 = j_first;; jl0 += U) {
#pragma unroll
    for (int qq = 0; qq < U; ++qq) {
      if (jl0 + qq > j_last) return;
      do_step(jl0 + qq, qq);
    }
  }
}

// The same pass for the canonical 9-point pattern ((a, b) in {-1, 0, 1}^2: the Q1 FEM operator): a row also needs the i -/+ 1
// neighbours of the operand's line BELOW, which the step underneath produces in the same line step -- three lines per step in
// LDS (slot = line mod 3), one barrier per step and line step (as jacobi_march9_kernel).
template <typename T, int KV, int DEG, int VW = 16 / (int)sizeof(T), int NT = GLA_BLOCK>
__global__ __launch_bounds__(NT) void cheby_march9_kernel(
    Cheb2D cv, const T* __restrict__ bvec, int64_t ldb, const T* __restrict__ x0, int64_t ldx0, T* __restrict__ xout,
    int64_t ldxo, T* __restrict__ rout, int64_t ldr, T* __restrict__ pout, int64_t ldp, T* __restrict__ pprev,
    int64_t ldpp, T* __restrict__ bcopy, int64_t ldbc, int lines, int nstrips, int64_t nlines, TileMap tm) {
  constexpr int BP = NT / KV;               // grid points of a strip
  constexpr int LWV = (BP + 2) * KV;               // 16-byte vectors of a line in LDS: point ii at [(1 + ii) * KV + q]
  constexpr int NC = DEG + 1, U = NC;
  typedef T TV __attribute__((ext_vector_type(VW)));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* buf = reinterpret_cast<TV*>(smem_raw);       // [DEG][3][LWV]: buf[t][s] = line (mod 3 == s) of the operand of step t + 1
  const int blk = map_tile(tm);
  if (blk < 0) return;
  constexpr int halo = DEG, bp_eff = BP - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t J0 = (int64_t)lb * lines, P0 = (int64_t)strip * bp_eff - halo;
  const int L = (int)min((int64_t)lines, nlines - J0);
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  const T* __restrict__ val = static_cast<const T*>(cv.val);
  const int64_t gi = P0 + ii;
  const bool in_i = gi >= 0 && gi < cv.S;
  const bool interior = ii >= halo && ii < BP - halo;

  auto live_row = [&](int jl, int64_t& r) -> bool {
    const int64_t gj = J0 + jl;
    r = gj * cv.S + gi;
    return in_i && gj >= 0 && gj < nlines && r < cv.n_rows;
  };
  auto x0_at = [&](int jl) -> TV {                 // clamped: values of dead rows are never used
    int64_t flat = (J0 + jl) * cv.S + gi;
    flat = min(max(flat, (int64_t)0), cv.n_cols - 1);
    return ldv<TV>(x0 + flat * ldx0 + q * VW);
  };
  auto b_at = [&](int jl) -> TV {
    int64_t r;
    const bool live = live_row(jl, r);
    return ldv<TV>(bvec + (live ? r : 0) * ldb + q * VW);
  };
  auto coef_at = [&](int jl) -> ChCoef9<T> {
    int64_t r;
    const bool live = live_row(jl, r);
    const int64_t rc = live ? r : 0;
    ChCoef9<T> c;
#pragma unroll
    for (int d = 0; d < 9; ++d) c.v[d] = val[(int64_t)d * cv.n_rows + rc];
    c.m = (int)cv.mask[rc];
    c.lv = live;
    return c;
  };
  auto stencil = [&](const ChCoef9<T>& c, bool full, const TV (&xs)[9]) -> TV {
    TV acc = TV(T(0));
    if (full) {
#pragma unroll
      for (int d = 0; d < 9; ++d) acc = acc + c.v[d] * xs[d];
    } else {
      const int m = c.lv ? c.m : 0;
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        const TV nxt = acc + c.v[d] * xs[d];
        acc = ((m >> d) & 1) ? nxt : acc;
      }
    }
    return acc;
  };

  constexpr int j_first = -(DEG - 1);
  const int j_last = L - 1 + (DEG - 1);
  TV h[DEG][3];                                    // h[t]: the operand of step t + 1 (x for t = 0, p_t after) of this lane on
#pragma unroll                                     // the last three lines it was produced on
  for (int t = 0; t < DEG; ++t) h[t][0] = h[t][1] = h[t][2] = TV(T(0));
  h[0][1] = x0_at(j_first - 1);
  h[0][2] = x0_at(j_first);
  buf[((j_first - 1 + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][1];
  buf[((j_first + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][2];
  auto xin_at = [&](int jl) -> TV {                // continuation pass: x of the row (line J0 + jl), an input
    int64_t r;
    const bool live = live_row(jl, r);
    return ldv<TV>(bcopy + (live ? r : 0) * ldbc + q * VW);
  };
  TV xpre = x0_at(j_first + 1), bq = b_at(j_first);
  TV xinq = TV(T(0));
  if (cv.cont) xinq = xin_at(j_first);
  ChCoef9<T> c[NC];                                // set of line j_first + s lives in c[s % NC]
  bool cfull[NC];
  c[0] = coef_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
  for (int s = 0; s < NC; ++s) cfull[s] = false;
  TV cr[DEG], cx[DEG];                             // r and x step t left for this lane's row one line step ago
#pragma unroll
  for (int t = 0; t < DEG; ++t) cr[t] = cx[t] = TV(T(0));

  auto do_step = [&](const int jl, const int u) {
    if ((jl & 1) ^ (NT >= 512 ? ((int)threadIdx.x >> 8) & 1 : ((int)blockIdx.x / GLA_XCDS) >> 5 & 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (priority alternation between a CU's two workgroups: gla_march5.hip)
    const int s1 = (jl + 1 + 63) % 3, s0 = (jl + 63) % 3, s2 = (jl - 1 + 63) % 3;
    // LDS rows of the lines jl + 1, jl, jl - 1 (operand of step 1): [0] left, [KV] own, [2 KV] right of this lane's point
    TV* const rw[3] = {buf + s1 * LWV + ii * KV + q, buf + s0 * LWV + ii * KV + q, buf + s2 * LWV + ii * KV + q};
    h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
    rw[0][KV] = xpre;                              // x, line jl + 1
    const TV bnow = bq;                            // b of line jl (loaded a step ago)
    const TV xinnow = xinq;
    if (cv.cont) xinq = xin_at(jl + 1);
    xpre = x0_at(jl + 2);                          // clamped / dead beyond the block: harmless, never used
    bq = b_at(jl + 1);
    c[(u + 1) % NC] = coef_at(jl + 1);
    cfull[u % NC] = __all(c[u % NC].lv && (c[u % NC].m & 0x1ff) == 0x1ff) != 0;   // (at first use: loaded a step ago)
    TV in_r = TV(T(0)), in_x = TV(T(0));
#pragma unroll
    for (int t = 1; t <= DEG; ++t) {
      __syncthreads();
      const int line = jl - (t - 1);
      // lines line + 1, line, line - 1 of the operand of step t: the three rows rotated by t - 1
      TV* const below = rw[(t - 1) % 3] + (t - 1) * 3 * LWV;
      TV* const same = rw[t % 3] + (t - 1) * 3 * LWV;
      TV* const above = rw[(t + 1) % 3] + (t - 1) * 3 * LWV;
      const int ci = ((u - (t - 1)) % NC + NC) % NC;
      const TV use_r = in_r, use_x = in_x;         // what step t - 1 left for THIS row one line step ago
      if (t < DEG) { in_r = cr[t]; in_x = cx[t]; } // keep step t's previous results for step t + 1 before overwriting them
      const TV xs[9] = {above[0], h[t - 1][0], above[2 * KV], same[0], h[t - 1][1], same[2 * KV],
                        below[0], h[t - 1][2], below[2 * KV]};
      const TV acc = stencil(c[ci], cfull[ci], xs);
      int64_t r;
      const bool store_ok = live_row(line, r) && interior && line >= 0 && line < L;
      TV ri, pn, xn;
      if (t == 1) {                                // EpiChebyFirst: r = b - A x; p = r; x = x + alpha_1 r
        if (!cv.cont) {
          if (bcopy && store_ok) stv<TV>(bcopy + r * ldbc + q * VW, bnow);
          ri = bnow - acc;
          pn = ri;
          xn = h[0][1] + (T)cv.alpha[0] * ri;
        } else {                                   // continuation: the pass's first step is a "next" step on loaded r, x
          ri = bnow - (T)cv.alpha_prev * acc;
          pn = ri + (T)cv.beta[0] * h[0][1];
          xn = xinnow + (T)cv.alpha[0] * pn;
        }
      } else {                                     // EpiChebyNext: r -= a_old (A p); p = r + beta p; x += a_new p
        ri = use_r - (T)cv.alpha[t - 2] * acc;
        pn = ri + (T)cv.beta[t - 1] * h[t - 1][1];
        xn = use_x + (T)cv.alpha[t - 1] * pn;
      }
      if (t < DEG) {
        h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = pn;
        same[3 * LWV + KV] = pn;                   // operand of step t + 1, same line (same slot)
        cr[t] = ri;
        cx[t] = xn;
        if (t == DEG - 1 && pprev && store_ok) stv<TV>(pprev + r * ldpp + q * VW, pn);   // the operand the last step gathers
      } else if (store_ok) {
        stv<TV>(xout + r * ldxo + q * VW, xn);
        stv<TV>(rout + r * ldr + q * VW, ri);
        stv<TV>(pout + r * ldp + q * VW, pn);
      }
    }
  };
#pragma unroll 1
  for (int jl0 = j_first;; jl0 += U) {
#pragma unroll
    for (int qq = 0; qq < U; ++qq) {
      if (jl0 + qq > j_last) return;
      do_step(jl0 + qq, qq);
    }
  }
}

// one right-hand side: the march kernel with scalar lanes (256 points per strip); canonical 5-point plans only
template <typename T, int DEG>
int launch_cheby_fused_k1(gla_csr* A, const Cheb2D& cv, const T* b, int64_t ldb, const T* x0, int64_t ldx0, T* xout,
                          int64_t ldxo, T* r, int64_t ldr, T* p, int64_t ldp, T* pprev, int64_t ldpp, T* bcopy, int64_t ldbc,
                          hipStream_t st) {
  constexpr int BP = GLA_BLOCK;
  const int bp_eff = BP - 2 * DEG;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  const int nstrips = (int)((A->dia_S + bp_eff - 1) / bp_eff);
  const size_t lds = (size_t)DEG * 2 * (BP + 2) * sizeof(T);
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) lines = gla_pick_block_lines(nlines, nstrips, 256 * 2, 2 * (DEG - 1), 2 * DEG);
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = (int)(((nlines + lines - 1) / lines) * nstrips);
  set_tile_mapping(tm);
  if (gla_dia_canon9(A))
    hipLaunchKernelGGL((cheby_march9_kernel<T, 1, DEG, 1>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(GLA_BLOCK),
                       (size_t)DEG * 3 * (BP + 2) * sizeof(T), st, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp,
                       bcopy, ldbc, lines, nstrips, nlines, tm);
  else
    hipLaunchKernelGGL((cheby_march5_kernel<T, 1, DEG, 1>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(GLA_BLOCK), lds, st,
                       cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, lines, nstrips, nlines, tm);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T, int KV, int DEG>
int launch_cheby_fused(gla_csr* A, const Cheb2D& cv, const T* b, int64_t ldb, const T* x0, int64_t ldx0, T* xout,
                       int64_t ldxo, T* r, int64_t ldr, T* p, int64_t ldp, T* pprev, int64_t ldpp, T* bcopy, int64_t ldbc,
                       hipStream_t st) {
  const bool canon = g_tuning.jacobi_march && gla_dia_canon5(A);
  const bool canon9 = g_tuning.jacobi_march && gla_dia_canon9(A);
  // march kernels, rows of four or more 16-byte vectors: 512 threads per workgroup -- a strip of 64 / 128 points instead of
  // 32 / 64, so the 2 DEG halo points either side that every strip recomputes are a seventh of its work, not a third
  constexpr int NTM = KV >= 4 ? 512 : GLA_BLOCK;
  const int NTrun = (canon || canon9) ? NTM : GLA_BLOCK;
  const int BP = NTrun / KV;
  const int bp_eff = BP - 2 * DEG * cv.B;
  if (bp_eff < 1) return GLA_ERR_UNSUPPORTED;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  const int nstrips = (int)((A->dia_S + bp_eff - 1) / bp_eff);
  if (A->dia_D > CH_DMAX && !canon9) return GLA_ERR_UNSUPPORTED;
  if (canon9 && DEG > 3) return GLA_ERR_UNSUPPORTED;              // (9-point, four steps, K-wide: > 256 VGPRs; the caller fuses three)
  const size_t lds = canon ? (size_t)DEG * 2 * (BP + 2) * KV * 16
                           : (canon9 ? (size_t)DEG * 3 * (BP + 2) * KV * 16 : (size_t)DEG * 3 * BP * KV * 16);
  // block height: all workgroups resident at once (LDS: 12 KB per fused step and workgroup; registers: 3 waves per SIMD;
  // the march kernel: two workgroups per CU, as for the Jacobi march)
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) {
    const int64_t per_cu = (canon || canon9) ? 2 * GLA_BLOCK / NTrun : std::min<int64_t>(3, (160 * 1024) / (int64_t)lds);
    lines = gla_pick_block_lines(nlines, nstrips, 256 * (per_cu > 0 ? per_cu : 1), 2 * (DEG - 1), 2 * DEG);
  }
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = (int)(((nlines + lines - 1) / lines) * nstrips);
  set_tile_mapping(tm);
  constexpr int VWT = 16 / (int)sizeof(T);
  if (canon) {
    auto* kern = cheby_march5_kernel<T, KV, DEG, VWT, NTM>;
    if (lds > 64 * 1024) {                          // (once per instance: the default dynamic LDS limit is 64 KiB)
      static bool raised = false;
      if (!raised) { GLA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); raised = true; }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NTM), lds,
                       st, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, lines, nstrips, nlines, tm);
  } else if (canon9) {
    if constexpr (DEG <= 3) {
      auto* kern = cheby_march9_kernel<T, KV, DEG, VWT, NTM>;
      if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) { GLA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); raised = true; }
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NTM), lds,
                         st, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, lines, nstrips, nlines, tm);
    }
  } else
    hipLaunchKernelGGL((cheby_fused_dia2d_kernel<T, KV, DEG>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(GLA_BLOCK), lds,
                       st, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, lines, nstrips, nlines, tm);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T, int KV>
int dispatch_deg(int deg, gla_csr* A, const Cheb2D& cv, const T* b, int64_t ldb, const T* x0, int64_t ldx0, T* xout,
                 int64_t ldxo, T* r, int64_t ldr, T* p, int64_t ldp, T* pprev, int64_t ldpp, T* bcopy, int64_t ldbc,
                 hipStream_t st) {
  switch (deg) {
    case 2: return launch_cheby_fused<T, KV, 2>(A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    case 3: return launch_cheby_fused<T, KV, 3>(A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    case 4: return launch_cheby_fused<T, KV, 4>(A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    default: return GLA_ERR_UNSUPPORTED;
  }
}

template <typename T>
int cheby_fused_impl(gla_csr* A, int K, int deg, const T* b, int64_t ldb, const T* x0, int64_t ldx0, T* xout, int64_t ldxo,
                     T* r, int64_t ldr, T* p, int64_t ldp, T* pprev, int64_t ldpp, T* bcopy, int64_t ldbc, const double* alpha,
                     const double* beta, hipStream_t st, int cont = 0, double alpha_prev = 0.0) {
  if (cont && !(g_tuning.jacobi_march && (gla_dia_canon5(A) || gla_dia_canon9(A)))) return GLA_ERR_UNSUPPORTED;
  constexpr int VW = 16 / (int)sizeof(T);
  if (K == 1) {                                     // single right-hand side (the reference's own ChebyRelaxGNN): scalar lanes
    if (!(g_tuning.jacobi_march && (gla_dia_canon5(A) || gla_dia_canon9(A)))) return GLA_ERR_UNSUPPORTED;
    Cheb2D cv1;
    cv1.val = A->dia_val;
    cv1.mask = A->dia_mask;
    cv1.D = A->dia_D; cv1.S = A->dia_S; cv1.B = A->dia_B;
    for (int i = 0; i < CH_DMAX; ++i) { cv1.da[i] = A->dia_a[i]; cv1.db[i] = A->dia_b[i]; }      // (unused by the march kernels)
    cv1.n_rows = A->n_rows; cv1.n_cols = A->n_cols;
    for (int i = 0; i < CH_DEG_MAX; ++i) { cv1.alpha[i] = i < deg ? alpha[i] : 0.0; cv1.beta[i] = i < deg ? beta[i] : 0.0; }
    cv1.cont = cont; cv1.alpha_prev = alpha_prev;
    switch (deg) {
      case 2: return launch_cheby_fused_k1<T, 2>(A, cv1, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
      case 3: return launch_cheby_fused_k1<T, 3>(A, cv1, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
      case 4: return launch_cheby_fused_k1<T, 4>(A, cv1, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
      default: return GLA_ERR_UNSUPPORTED;
    }
  }
  const int kv = K / VW;
  const bool ptr_ok = al16(b) && al16(x0) && al16(xout) && al16(r) && al16(p) && (!pprev || al16(pprev)) &&
                      (!bcopy || (al16(bcopy) && ldbc % VW == 0)) &&
                      ldb % VW == 0 && ldx0 % VW == 0 && ldxo % VW == 0 && ldr % VW == 0 && ldp % VW == 0 &&
                      (!pprev || ldpp % VW == 0);
  if (K % VW != 0 || !(kv == 1 || kv == 2 || kv == 4 || kv == 8) || !ptr_ok) return GLA_ERR_UNSUPPORTED;
  Cheb2D cv;
  cv.val = A->dia_val;
  cv.mask = A->dia_mask;
  cv.D = A->dia_D; cv.S = A->dia_S; cv.B = A->dia_B;
  for (int i = 0; i < CH_DMAX; ++i) { cv.da[i] = i < A->dia_D ? A->dia_a[i] : 0; cv.db[i] = i < A->dia_D ? A->dia_b[i] : 0; }
  cv.n_rows = A->n_rows; cv.n_cols = A->n_cols;
  for (int i = 0; i < CH_DEG_MAX; ++i) { cv.alpha[i] = i < deg ? alpha[i] : 0.0; cv.beta[i] = i < deg ? beta[i] : 0.0; }
  cv.cont = cont; cv.alpha_prev = alpha_prev;
  switch (kv) {
    case 1: return dispatch_deg<T, 1>(deg, A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    case 2: return dispatch_deg<T, 2>(deg, A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    case 4: return dispatch_deg<T, 4>(deg, A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
    default: return dispatch_deg<T, 8>(deg, A, cv, b, ldb, x0, ldx0, xout, ldxo, r, ldr, p, ldp, pprev, ldpp, bcopy, ldbc, st);
  }
}

}  // namespace

extern "C" {

int gla_cheby_fused(gla_csr_t A, int dtype, int K, int deg, const void* B, int64_t ldb, const void* X0, int64_t ldx0,
                    void* Xout, int64_t ldxo, void* R, int64_t ldr, void* P, int64_t ldp, void* Pprev, int64_t ldpp,
                    void* Bcopy, int64_t ldbc, const double* alpha_host, const double* beta_host, gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (!B || !X0 || !Xout || !R || !P || X0 == Xout || !alpha_host || !beta_host || ldb < K || ldx0 < K || ldxo < K ||
      ldr < K || ldp < K || (Pprev && ldpp < K) || (Bcopy && ldbc < K))
    return GLA_ERR_INVALID_ARG;
  if (deg < 2 || deg > CH_DEG_MAX) return GLA_ERR_UNSUPPORTED;
  // 2-D stencil plans whose entries never leave their grid line, one line either side, at most five diagonals
  if (!(g_tuning.dia && g_tuning.jacobi_fuse && A->dia_state == 1 && A->dia_val && A->dia_S > 0 && A->dia_nowrap &&
        A->dia_A == 1 && A->dia_B >= 1 && A->dia_B <= 2 && (A->dia_D <= CH_DMAX || gla_dia_canon9(A)) && A->dia_row_lo == 0 &&
        A->dia_row_hi == A->n_rows && A->n_rows == A->n_cols && A->n_rows * (int64_t)K >= g_tuning.jacobi_min_rows))
    return GLA_ERR_UNSUPPORTED;
  if (A->n_rows == 0) return GLA_OK;
  if (dtype == GLA_F64)
    return cheby_fused_impl<double>(A, K, deg, (const double*)B, ldb, (const double*)X0, ldx0, (double*)Xout, ldxo,
                                    (double*)R, ldr, (double*)P, ldp, (double*)Pprev, ldpp, (double*)Bcopy, ldbc, alpha_host,
                                    beta_host, (hipStream_t)stream);
  return cheby_fused_impl<float>(A, K, deg, (const float*)B, ldb, (const float*)X0, ldx0, (float*)Xout, ldxo, (float*)R,
                                 ldr, (float*)P, ldp, (float*)Pprev, ldpp, (float*)Bcopy, ldbc, alpha_host, beta_host,
                                 (hipStream_t)stream);
}

int gla_cheby_fused_next(gla_csr_t A, int dtype, int K, int deg, const void* Rin, int64_t ldri, const void* Xin, int64_t ldxi,
                         const void* Pin, int64_t ldpi, void* Xout, int64_t ldxo, void* R, int64_t ldr, void* P, int64_t ldp,
                         void* Pprev, int64_t ldpp, double alpha_prev, const double* alpha_host, const double* beta_host,
                         gla_stream_t stream) {
  int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (!Rin || !Xin || !Pin || !Xout || !R || !P || Xin == Xout || Rin == R || Pin == P || !alpha_host || !beta_host ||
      ldri < K || ldxi < K || ldpi < K || ldxo < K || ldr < K || ldp < K || (Pprev && ldpp < K))
    return GLA_ERR_INVALID_ARG;
  if (deg < 2 || deg > CH_DEG_MAX) return GLA_ERR_UNSUPPORTED;
  if (!(g_tuning.dia && g_tuning.jacobi_fuse && A->dia_state == 1 && A->dia_val && A->dia_S > 0 && A->dia_nowrap &&
        A->dia_row_lo == 0 && A->dia_row_hi == A->n_rows && A->n_rows == A->n_cols &&
        A->n_rows * (int64_t)K >= g_tuning.jacobi_min_rows))
    return GLA_ERR_UNSUPPORTED;
  if (A->n_rows == 0) return GLA_OK;
  if (dtype == GLA_F64)
    return cheby_fused_impl<double>(A, K, deg, (const double*)Rin, ldri, (const double*)Pin, ldpi, (double*)Xout, ldxo,
                                    (double*)R, ldr, (double*)P, ldp, (double*)Pprev, ldpp, (double*)const_cast<void*>(Xin), ldxi,
                                    alpha_host, beta_host, (hipStream_t)stream, 1, alpha_prev);
  return cheby_fused_impl<float>(A, K, deg, (const float*)Rin, ldri, (const float*)Pin, ldpi, (float*)Xout, ldxo, (float*)R, ldr,
                                 (float*)P, ldp, (float*)Pprev, ldpp, (float*)const_cast<void*>(Xin), ldxi, alpha_host, beta_host,
                                 (hipStream_t)stream, 1, alpha_prev);
}

}  // extern "C"
