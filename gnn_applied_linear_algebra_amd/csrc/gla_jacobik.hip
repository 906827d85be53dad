// libgla_hip.so -- temporally blocked Jacobi sweeps for K right-hand sides on canonical stencil plans (gfx950 / MI355X).
//
// JacobiGNN on K right-hand sides (vertex_attr [n, 3, K]: the multi-RHS extension of JacobiGNN.py:138-148; the smoother of a
// V-cycle run on several right-hand sides at once) is, sweep by sweep, the blocked SpMM kernel with the EpiJacobi epilogue:
// every sweep streams b, x in and x out (24 K bytes per row) plus the matrix.  Only the last iterate leaves the call, so --
// exactly like jacobi_march5 (gla_march5.hip) and cheby_march5 (gla_cheby.hip) -- TS sweeps are taken in ONE pass: a
// workgroup owns a strip of NT / KV grid points x K right-hand sides (lanes = (point, 16-byte rhs vector)) and marches down
// the grid lines; the own column of every level lives in registers (h[t][0..2]), LDS carries the i -/+ 1 neighbours,
// coefficients and b of the last TS lines stay in registers under static names.
//
// Round 3, as in gla_march5.hip (the reasons and measurements are in its header):
//   * strips are aligned with the grid (no lane outside it); a row whose mask is "every neighbour that exists" takes the
//     unmasked sum -- missing neighbours read +0 (LDS pads, lines beyond the grid), their stored coefficients are zeros;
//   * the division by A_ii goes through RN(1 / A_ii) (one IEEE division per row and pass, shared by the K right-hand sides)
//     and one Markstein correction per level and right-hand side; lanes whose operands leave the range the correction is
//     proven for have the level (9-point) / the line step (5-point) redone with IEEE divisions;
//   * 512 threads per workgroup where a row is 4 or more 16-byte vectors wide: a strip is then 64 (128) points instead of
//     32 (64), and the 2 TS halo points either side that every strip recomputes drop from a third to a seventh of the work;
//   * the two waves that share a SIMD alternate s_setprio every line step in antiphase.
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
struct KCoef {                                     // one row's matrix data, A_ii, mask as loaded
  T v[(D + 3) & ~3];                               // (padded to whole 16-byte groups: the compiler keeps the set in registers)
  T dg;
  int m;
};

template <typename T> struct KLim;
template <> struct KLim<double> {
  static constexpr double dlo = 0x1p-500, dhi = 0x1p500, nlo = 0x1p-450, nhi = 0x1p450;
  // range_word(v): a word whose bits 30 / 31 are clear only if 2^-255 <= |v| < 2^257 (inside [nlo, nhi]; a NaN, an infinity, a
  // zero or a denormal sets one of them): (high word << 1) drops the sign and puts the exponent field at bit 21; minus 768 << 21
  // maps exponent fields [768, 1280) to [0, 512 << 21) = below bit 30.  One v_lshl_add_u32 per value and one v_or3_b32 per two
  // values instead of two compares and two scalar ORs per value (round 5).
  static __device__ __forceinline__ unsigned range_word(double v) {
    return ((unsigned)__double2hiint(v) << 1) - (768u << 21);
  }
  static __device__ __forceinline__ bool all_ones(double v) {
    return (__builtin_bit_cast(unsigned long long, v) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
  }
};
template <> struct KLim<float> {
  static constexpr float dlo = 0x1p-60f, dhi = 0x1p60f, nlo = 0x1p-50f, nhi = 0x1p50f;
  // 2^-31 <= |v| < 2^33: exponent fields [96, 160) -> [0, 64 << 24) = below bit 30
  static __device__ __forceinline__ unsigned range_word(float v) { return (__float_as_uint(v) << 1) - (96u << 24); }
  static __device__ __forceinline__ bool all_ones(float v) { return (__builtin_bit_cast(unsigned, v) & 0x7FFFFFu) == 0x7FFFFFu; }
};

// D = 5: diagonals (-1,0) (0,-1) (0,0) (0,1) (1,0); D = 9: (a, b) in {-1, 0, 1}^2 row-major
// DBG (GLA_MK_VARIANTS builds only; results are WRONG with any bit set -- timing experiments, tools/ab_mk_variants.py): 1 no
// barrier, 2 no output store, 4 no levels (the memory side alone), 8 no global loads in the line step (the compute side alone)
template <typename T, int KV, int TS, int D, int NT, int DBG = 0>
__global__ __launch_bounds__(NT) void jacobi_marchk_kernel(JacK jv, const T* __restrict__ diag, const T* __restrict__ bvec,
                                                           int64_t ldb, const T* __restrict__ x0, int64_t ldx0,
                                                           T* __restrict__ xout, int64_t ldxo, T w, int lines, int nstrips,
                                                           int64_t nlines, TileMap tm) {
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int BP = NT / KV;                      // grid points of a strip
  constexpr int LWV = (BP + 2) * KV;               // 16-byte vectors of a line in LDS: point ii at [(1 + ii) * KV + q]
  constexpr int NSL = D == 5 ? 2 : 3;              // LDS lines per level: 5-point slot = line parity, 9-point slot = line mod 3
  constexpr int NC = TS + 1, U = NC;
  constexpr unsigned FULL = (1u << D) - 1u;
  typedef T TV __attribute__((ext_vector_type(VW)));
  typedef KCoef<T, D> Coef;
  typedef KLim<T> Lim;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* buf = reinterpret_cast<TV*>(smem_raw);       // [TS][NSL][LWV]
  const int blk = map_tile(tm);
  if (blk < 0) return;
  constexpr int halo = TS, bp_eff = BP - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t J0 = (int64_t)lb * lines;
  const int L = (int)min((int64_t)lines, nlines - J0);
  const int64_t S = jv.S;
  // strip s: interior strips own the middle bp_eff points; the first starts at point 0, the last ends at point S - 1
  const int64_t P0 = max((int64_t)0, min((int64_t)strip * bp_eff, S - BP));
  const int64_t own_lo = strip == 0 ? 0 : (int64_t)strip * bp_eff + halo;
  const int64_t own_hi = strip == nstrips - 1 ? S : (int64_t)strip * bp_eff + BP - halo;
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  // the two waves that share a SIMD (NT = 512: waves w and w + 4 of the workgroup; NT = 256: of a CU's two workgroups)
  const int late = NT >= 512 ? (tid >> 8) & 1 : (((int)blockIdx.x / GLA_XCDS) >> 5) & 1;
  const T* __restrict__ val = static_cast<const T*>(jv.val);
  const int64_t gi = P0 + ii;
  const bool in_i = gi < S;                        // (false only on grids narrower than a strip)
  const bool owner = gi >= own_lo && gi < own_hi;
  const int64_t nfull = jv.n_rows / S;             // whole grid lines
  const int64_t tail = jv.n_rows - nfull * S;      // points on the ragged last line
  // the mask of a row at this point of a line when every neighbour that exists is coupled, line-independent part
  const unsigned em_i = D == 5 ? (31u & ~(gi == 0 ? 2u : 0u) & ~(gi == S - 1 ? 8u : 0u))
                               : (FULL & ~(gi == 0 ? 0x49u : 0u) & ~(gi == S - 1 ? 0x124u : 0u));
  // block-relative lines [jlo, jhi) are whole grid lines
  const int jlo = (int)max((int64_t)-64, -J0), jhi = (int)min((int64_t)L + 64, nfull - J0);
  const bool wave_in = __all(in_i) != 0;

  for (int i = tid; i < TS * NSL * LWV; i += NT) buf[i] = TV(T(0));   // (the pads stay +0: what lies beyond a line's ends)
  __syncthreads();

  auto row_clamped = [&](int jl) -> int64_t { return min(max((J0 + jl) * S + gi, (int64_t)0), jv.n_rows - 1); };
  // (bitwise, not short-circuit: hipcc 7.0 dropped the masked move of `uniform || per_lane ? xpre : 0` in one unrolled copy of the
  // line step -- x^(k) of the next line read as 0 wherever the wave-uniform term was false; profiles/r03/marchk_miscompile.txt)
  auto live_at = [&](int jl) -> bool { return in_i & (J0 + jl >= 0) & ((J0 + jl) * S + gi < jv.n_rows); };
  auto x0_load = [&](int jl) -> TV { return ldv<TV>(x0 + row_clamped(jl) * ldx0 + q * VW); };   // NOT touched until the next step
  auto coef_at = [&](int jl) -> Coef {
    const int64_t rc = row_clamped(jl);
    Coef c;
#pragma unroll
    for (int d = 0; d < D; ++d) c.v[d] = val[(int64_t)d * jv.n_rows + rc];
    c.dg = diag[rc];
    c.m = (int)jv.mask[rc];
    return c;
  };
  auto b_at = [&](int jl) -> TV { return ldv<TV>(bvec + row_clamped(jl) * ldb + q * VW); };
  // Round 5: the addresses of a line step's loads without per-lane 64-bit arithmetic.  Where the lines jl + 1 and jl + 2 are whole
  // grid lines for the whole wave (wave-uniform; everything but a block's first / last steps at the grid's ends) no row needs
  // clamping: a load's address is a SCALAR base -- the strip's first point on that line -- plus a 32-bit lane offset that never
  // changes.  The clamped per-lane form above cost ~100 of a step's 375 instructions (two 64-bit clamps, two 64-bit multiplies
  // by the leading dimensions, eleven 64-bit adds), in a pass that is bound by its instruction stream (profiles/r05/marchk_sides.txt).
  const bool fa_ok = wave_in && ldx0 < (1 << 20) && ldb < (1 << 20);
  const unsigned off_x = (unsigned)(ii * (int)ldx0 + q * VW) * (unsigned)sizeof(T);
  const unsigned off_b = (unsigned)(ii * (int)ldb + q * VW) * (unsigned)sizeof(T);
  const unsigned off_r = (unsigned)ii * (unsigned)sizeof(T), off_m = (unsigned)ii * 2u;
  auto at = [&](const auto* base, int64_t scalar_elems, unsigned lane_bytes) {
    typedef decltype(base) P;
    return reinterpret_cast<P>(reinterpret_cast<const char*>(base + scalar_elems) + lane_bytes);
  };
  auto expected = [&](int jl) -> unsigned {        // (lines inside [jlo, jhi) only)
    const int64_t gj = J0 + jl;
    if constexpr (D == 5) {
      const bool has_down = gj + 1 < nfull || (gj + 1 == nfull && gi < tail);
      return em_i & ~(gj == 0 ? 1u : 0u) & ~(has_down ? 0u : 16u);
    } else {
      unsigned e = em_i & ~(gj == 0 ? 0x7u : 0u);
      if (gj + 1 >= nfull) {                       // the line below is the ragged one, or beyond the grid
        const bool rag = gj + 1 == nfull;
#pragma unroll
        for (int bb = -1; bb <= 1; ++bb)
          if (!(rag && gi + bb < tail)) e &= ~(1u << (7 + bb));
      }
      return e;
    }
  };
  // the generic level: masked in-order sum, IEEE division, +0 for rows outside the matrix
  auto generic = [&](const Coef& c, TV cb, const TV (&xs)[D], TV ctr, int line) -> TV {
    const bool lv = live_at(line);
    const int m = lv ? (int)jv.mask[row_clamped(line)] : 0;          // (not kept per set: this path is rare)
    TV acc = TV(T(0));
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const TV nxt = acc + c.v[d] * xs[d];
      acc = ((m >> d) & 1) ? nxt : acc;
    }
    const TV xn = ctr + (w * (cb - acc)) / c.dg;
    return lv ? xn : TV(T(0));
  };
  // the fast level: unmasked sum, Markstein division through rc = RN(1 / A_ii); bad: an operand it is not proven for
  auto fast_level = [&](const Coef& c, TV cb, T rc, const TV (&xs)[D], TV ctr, unsigned& bad, TV& num_out) -> TV {
    TV acc = TV(T(0));
#pragma unroll
    for (int d = 0; d < D; ++d) acc = acc + c.v[d] * xs[d];
    const TV num = w * (cb - acc);
    TV xn;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      bad |= Lim::range_word(num[e]);                // (bits 30 / 31 set: outside the range, see KLim)
      const T q0 = num[e] * rc;
      const T rem = fma_t(-q0, c.dg, num[e]);
      xn[e] = ctr[e] + fma_t(rem, rc, q0);
    }
    num_out = num;
    return xn;
  };

  // The 5-point fast level in two halves (round 5).  Of a level's in-order sum only the LAST term -- the point below, x of the
  // level underneath on the line that level produces in this very step -- depends on this step's work; the first four terms
  // (north, west, centre, east) were all produced a step ago.  lead4() of level t + 1 is issued next to finish() of level t:
  // four dependent chains instead of two in a stream where one wave issues a dependent FP64 instruction every ~8 cycles and
  // an independent one every ~5 (tools/micro/valu_rates.hip), and the LDS reads of the next level are in flight during the
  // arithmetic of this one.  Same operations in the same order per (row, rhs): bit-identical.
  auto lead4 = [&](const Coef& c, TV north, TV west, TV ctr, TV east) -> TV {
    TV acc = TV(T(0));
    acc = acc + c.v[0] * north;
    acc = acc + c.v[1] * west;
    acc = acc + c.v[2] * ctr;
    acc = acc + c.v[3] * east;
    return acc;
  };
  auto finish5 = [&](const Coef& c, TV cb, T rc, TV acc, TV south, TV ctr, unsigned& bad) -> TV {
    acc = acc + c.v[4] * south;
    const TV num = w * (cb - acc);
    TV xn;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      bad |= Lim::range_word(num[e]);
      const T q0 = num[e] * rc;
      const T rem = fma_t(-q0, c.dg, num[e]);
      xn[e] = ctr[e] + fma_t(rem, rc, q0);
    }
    return xn;
  };

  constexpr int j_first = -(TS - 1);
  const int j_last = L - 1 + (TS - 1);
  TV h[TS][3];                                     // h[t]: x^(k+t) of this lane on the last three lines level t produced
#pragma unroll
  for (int t = 0; t < TS; ++t) h[t][0] = h[t][1] = h[t][2] = TV(T(0));
  {
    const TV a = x0_load(j_first - 1), b0 = x0_load(j_first);
    h[0][1] = live_at(j_first - 1) ? a : TV(T(0));
    h[0][2] = live_at(j_first) ? b0 : TV(T(0));
  }
  if constexpr (D == 5) {
    buf[((j_first + 64) & 1) * LWV + (1 + ii) * KV + q] = h[0][2];
  } else {
    buf[((j_first - 1 + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][1];
    buf[((j_first + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][2];
  }
  TV xpre = x0_load(j_first + 1);
  Coef c[NC];                                      // set of line j_first + s lives in c[s % NC]
  TV cb[NC];                                       // ... and the row's b
  T crc[NC];                                       // RN(1 / A_ii) (1 where the fast division is off anyway)
  bool cfast[NC];                                  // wave-uniform: a whole grid line, every row with the expected mask and a safe divisor
  c[0] = coef_at(j_first);
  cb[0] = b_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) { c[s] = c[0]; cb[s] = cb[0]; }
#pragma unroll
  for (int s = 0; s < NC; ++s) { cfast[s] = false; crc[s] = T(1); }

  // level 0 of line jl + 1 from the prefetched x, then the loads of the step: x of line jl + 2, matrix row and b of line jl + 1
  auto advance = [&](const int jl, const int u, TV& x_new) {
    const bool whole_next = wave_in && jl + 1 >= jlo && jl + 1 < jhi;
    if (fa_ok && jl + 1 >= jlo && jl + 2 < jhi) {
      x_new = xpre;
      if constexpr (!(DBG & 8)) {
        const int64_t rb1 = (J0 + jl + 1) * S + P0;                  // (scalar: the strip's first row of line jl + 1)
        xpre = ldv<TV>(at(x0, (rb1 + S) * ldx0, off_x));
        Coef& cn = c[(u + 1) % NC];
#pragma unroll
        for (int d = 0; d < D; ++d) cn.v[d] = *at(val, (int64_t)d * jv.n_rows + rb1, off_r);
        cn.dg = *at(diag, rb1, off_r);
        cn.m = (int)*at(jv.mask, rb1, off_m);
        cb[(u + 1) % NC] = ldv<TV>(at(bvec, rb1 * ldb, off_b));
      }
    } else {
      asm volatile("" ::: "memory");               // (keeps the two forms apart: merged, the clamps are computed on every step again)
      x_new = (whole_next | live_at(jl + 1)) ? xpre : TV(T(0));
      if constexpr (!(DBG & 8)) {
        xpre = x0_load(jl + 2);
        c[(u + 1) % NC] = coef_at(jl + 1);
        cb[(u + 1) % NC] = b_at(jl + 1);
      }
    }
  };
  // first use of the set of line jl (loaded a step ago): expected mask, divisor range, reciprocal
  auto classify = [&](const int jl, const int u) {
    const Coef& cu = c[u % NC];
    const T ad = __builtin_fabs(cu.dg);
    const bool dok = ad >= Lim::dlo && ad <= Lim::dhi && !Lim::all_ones(cu.dg);
    crc[u % NC] = T(1) / (dok ? cu.dg : T(1));
    const bool whole = wave_in && jl >= jlo && jl < jhi;
    cfast[u % NC] = whole && __all(((unsigned)cu.m & FULL) == expected(jl) && dok) != 0;
  };

  auto do_step = [&](const int jl, const int u, const int nact) {
    if constexpr (D == 5) {
      // 5-point: a level only needs the i -/+ 1 neighbours of ITS OWN line, which the level below wrote one line step
      // earlier: two lines per level (slot = line parity) and ONE barrier per line step (as jacobi_march5 / cheby_march5)
      if constexpr (!(DBG & 1)) __syncthreads();
      const int pj = (jl + 64) & 1;
      if (pj ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      TV* const row_e = buf + pj * LWV + ii * KV + q;          // lines of the parity of jl: [0] left, [KV] own, [2 KV] right
      TV* const row_o = buf + (1 - pj) * LWV + ii * KV + q;
#pragma unroll
      for (int t = 0; t < TS; ++t) { h[t][0] = h[t][1]; h[t][1] = h[t][2]; }     // (levels only assign h[t][2]: a step is idempotent)
      advance(jl, u, h[0][2]);
      row_o[KV] = h[0][2];                         // level 0, line jl + 1
      classify(jl, u);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DBG & 4) != 0) {              // the memory side alone: every loaded value consumed, one store per (row, vector)
        const Coef& cc = c[u % NC];
        TV sum_ = (h[0][1] + cb[u % NC]) + (row_e[0] + row_e[2 * KV]);
        sum_ = sum_ + TV((((cc.v[0] + cc.v[1]) + (cc.v[2] + cc.v[3])) + cc.v[4]) + cc.dg + (T)cc.m);
        if (nact == TS && owner) stv<TV>(xout + ((J0 + jl - (TS - 1)) * S + gi) * ldxo + q * VW, sum_);
        return;
      }
      bool fast = true;
#pragma unroll
      for (int t = 1; t <= TS; ++t)
        if (t <= nact) fast = fast && cfast[((u - (t - 1)) % NC + NC) % NC];
      unsigned bad = 0;
      if (fast) {
        // row of level t: the LDS line of level t - 1 on line jl - (t - 1) (written a step ago): [0] west, [2 KV] east
        auto row_of = [&](int t) -> TV* { return ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LWV; };
        auto set_of = [&](int t) -> int { return ((u - (t - 1)) % NC + NC) % NC; };
        // (DBG & 16, timing experiments: no LDS traffic in the levels -- the neighbours read as the own column)
        auto west_of = [&](int t) -> TV { if constexpr ((DBG & 16) != 0) return h[t - 1][0]; else return row_of(t)[0]; };
        auto east_of = [&](int t) -> TV { if constexpr ((DBG & 16) != 0) return h[t - 1][2]; else return row_of(t)[2 * KV]; };
        TV lead = lead4(c[set_of(1)], h[0][0], west_of(1), h[0][1], east_of(1));
#pragma unroll
        for (int t = 1; t <= TS; ++t) {
          if (t <= nact) {
            TV* const row = row_of(t);
            const int ci = set_of(t);
            TV lead_next = lead;
            if (t + 1 <= TS && t + 1 <= nact)
              lead_next = lead4(c[set_of(t + 1)], h[t][0], west_of(t + 1), h[t][1], east_of(t + 1));
            const TV xn = finish5(c[ci], cb[ci], crc[ci], lead, h[t - 1][2], h[t - 1][1], bad);
            if (t < TS) { h[t][2] = xn; if constexpr (!(DBG & 16)) row[2 * LWV + KV] = xn; }   // level t, same line (same parity)
            else if ((DBG & 2) ? (owner && xn[0] == T(123.456)) : owner) stv<TV>(xout + ((J0 + jl - (TS - 1)) * S + gi) * ldxo + q * VW, xn);
            lead = lead_next;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (!fast || __any(bad >> 30)) {
        asm volatile("" ::: "memory");             // (keeps this path's LDS reads from being merged with the fast path's above the branch)
#pragma unroll
        for (int t = 1; t <= TS; ++t) {
          if (t <= nact) {
            TV* const row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LWV;
            const int ci = ((u - (t - 1)) % NC + NC) % NC;
            const int line = jl - (t - 1);
            const TV xs[5] = {h[t - 1][0], row[0], h[t - 1][1], row[2 * KV], h[t - 1][2]};
            const TV xn = generic(c[ci], cb[ci], xs, h[t - 1][1], line);
            if (t < TS) { h[t][2] = xn; row[2 * LWV + KV] = xn; }
            else if (owner && live_at(line)) stv<TV>(xout + ((J0 + line) * S + gi) * ldxo + q * VW, xn);
          }
        }
      }
    } else {
      // 9-point: a level also needs the i -/+ 1 neighbours of the line BELOW, which the level underneath produces in the same
      // line step: three lines per level (slot = line mod 3), one barrier per level and line step
      if ((jl & 1) ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      const int s1 = (jl + 1 + 63) % 3, s0 = (jl + 63) % 3, s2 = (jl - 1 + 63) % 3;
      // LDS rows of the lines jl + 1, jl, jl - 1 (level 0): [0] left, [KV] own, [2 KV] right of this lane's point
      TV* const rw[3] = {buf + s1 * LWV + ii * KV + q, buf + s0 * LWV + ii * KV + q, buf + s2 * LWV + ii * KV + q};
      h[0][0] = h[0][1]; h[0][1] = h[0][2];
      advance(jl, u, h[0][2]);
      rw[0][KV] = h[0][2];                         // level 0, line jl + 1
      classify(jl, u);
#pragma unroll
      for (int t = 1; t <= TS; ++t) {
        __syncthreads();
        if (t <= nact) {
          // lines line + 1, line, line - 1 of level t - 1 (line = jl - (t - 1)): the three rows rotated by t - 1
          TV* const below = rw[(t - 1) % 3] + (t - 1) * 3 * LWV;
          TV* const same = rw[t % 3] + (t - 1) * 3 * LWV;
          TV* const above = rw[(t + 1) % 3] + (t - 1) * 3 * LWV;
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          const int line = jl - (t - 1);
          const TV xs[9] = {above[0], h[t - 1][0], above[2 * KV], same[0], h[t - 1][1], same[2 * KV],
                            below[0], h[t - 1][2], below[2 * KV]};
          TV xn;
          bool live = true;
          if (cfast[ci]) {
            unsigned bad = 0;
            TV num;
            xn = fast_level(c[ci], cb[ci], crc[ci], xs, h[t - 1][1], bad, num);
            if (__any(bad >> 30)) {                      // same sum, IEEE division
              asm volatile("" ::: "memory");
              xn = h[t - 1][1] + num / c[ci].dg;
            }
          } else {
            xn = generic(c[ci], cb[ci], xs, h[t - 1][1], line);
            live = live_at(line);
          }
          if (t < TS) {
            h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
            same[3 * LWV + KV] = xn;               // level t, same line (same slot)
          } else if (owner && live) {
            stv<TV>(xout + ((J0 + line) * S + gi) * ldxo + q * VW, xn);
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

// threads per workgroup: 512 where a row is four or more 16-byte vectors wide (strip of 64 / 128 points instead of 32 / 64)
template <int KV> struct KThreads { static constexpr int nt = KV >= 4 ? 512 : GLA_BLOCK; };

template <typename T, int KV, int TS, int D>
int marchk_blocks_per_cu(size_t lds) {
  constexpr int NT = KThreads<KV>::nt;
  static int cached = 0;
  if (!cached) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, jacobi_marchk_kernel<T, KV, TS, D, NT>, NT, lds) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 1;
    }
    const int cap = 2 * GLA_BLOCK / NT;            // two waves per SIMD: more resident workgroups only shorten the streams
    cached = nb > cap ? cap : nb;
  }
  return cached;
}

template <typename T, int KV, int TS, int D>
int launch_marchk(gla_csr* A, const T* diag, const T* b, int64_t ldb, const T* x, int64_t ldx, T* xo, int64_t ldxo, T w,
                  hipStream_t st) {
  constexpr int NT = KThreads<KV>::nt;
  constexpr int BP = NT / KV;
  constexpr int bp_eff = BP - 2 * TS;
  static_assert(bp_eff >= 1, "strip too narrow");
  JacK jv;
  jv.val = A->dia_val;
  jv.mask = A->dia_mask;
  jv.S = A->dia_S;
  jv.n_rows = A->n_rows; jv.n_cols = A->n_cols;
  const int64_t nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  // the first and the last strip own their outer halo too (nothing lies beyond the grid): S - 2 TS points to share out
  int64_t ns = (A->dia_S - 2 * TS + bp_eff - 1) / bp_eff;
  if (ns < 1) ns = 1;
  const int nstrips = (int)ns;
  const size_t lds = (size_t)TS * (D == 5 ? 2 : 3) * (BP + 2) * KV * 16;
  if (lds > 64 * 1024) {
    static bool raised = false;                    // (once per instance: the default dynamic LDS limit is 64 KiB)
    if (!raised) {
      GLA_HIP_CHECK(hipFuncSetAttribute((const void*)jacobi_marchk_kernel<T, KV, TS, D, NT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      raised = true;
    }
  }
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0)
    lines = gla_pick_block_lines(nlines, nstrips, 256 * (int64_t)marchk_blocks_per_cu<T, KV, TS, D>(lds), 2 * (TS - 1), 2 * TS);
  if (lines < 2 * TS) lines = 2 * TS;
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = (int)(((nlines + lines - 1) / lines) * nstrips);
  set_tile_mapping(tm);
#ifdef GLA_MK_VARIANTS
  if constexpr (sizeof(T) == 8 && KV == 8 && TS == 5 && D == 5) {
    const int v = g_tuning.m5_variant;
#define MKD(DBGV) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)jacobi_marchk_kernel<T, KV, TS, D, NT, DBGV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((jacobi_marchk_kernel<T, KV, TS, D, NT, DBGV>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NT), lds, st, jv, diag, b, ldb, x, ldx, xo, ldxo, w, lines, nstrips, nlines, tm); } while (0)
    if (v > 0) {
      switch (v) {
        case 1: MKD(1); break; case 2: MKD(2); break; case 3: MKD(4); break; case 4: MKD(8); break;
        case 5: MKD(9); break; case 6: MKD(5); break; case 7: MKD(10); break;
        case 8: MKD(16); break; case 9: MKD(17); break; case 10: MKD(24); break; case 11: MKD(25); break;
        default: return GLA_ERR_INVALID_ARG;
      }
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
#endif
  hipLaunchKernelGGL((jacobi_marchk_kernel<T, KV, TS, D, NT>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NT), lds,
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
      else return GLA_ERR_UNSUPPORTED;               // (9-point, four sweeps: the registers of one wave per SIMD)
    case 5:
      if constexpr (D == 5) return launch_marchk<T, KV, 5, D>(A, diag, b, ldb, x, ldx, xo, ldxo, w, st);
      else return GLA_ERR_UNSUPPORTED;
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
