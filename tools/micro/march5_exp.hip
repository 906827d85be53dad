// Experiment bench for the temporally blocked 5-point Jacobi pass (jacobi_march5_kernel of csrc/gla_rows.hip): the same
// march, stand-alone (no torch, no plan), with variants switched by template flags so that one build A/Bs them on one box:
//   V_HOIST  the i -/+ 1 neighbours of every level are read from LDS right after the barrier instead of right before their use
//   V_RCP    the IEEE division by A_ii goes through the correctly rounded reciprocal (one division per row and PASS) + one
//            Markstein correction step per level, with a wave-uniform fallback to the true division wherever that is not
//            provably the correctly rounded quotient (zero / tiny / huge / non-finite numerators, extreme or all-ones divisors)
//   V_RCPU   the same without any safety check (NOT bit-safe: only to see what the instruction count is worth)
//   V_RCPB   V_RCP with the fallback behind a real branch (an opaque asm keeps the compiler from if-converting it)
//   V_NT     coefficient streams are loaded non-temporally
//   V_AOSV   the five diagonals of a row are contiguous ([n][5]) instead of diagonal-major ([5][n])
// Every variant is compared BIT FOR BIT with TS launches of a one-row-per-lane single sweep.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/march5_exp.hip -o tools/micro/march5_exp
// Run:   tools/micro/march5_exp [N=4096] [reps=20]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <utility>

#pragma clang fp contract(off)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int BW = 256;
constexpr int V_HOIST = 1, V_RCP = 2, V_NT = 4, V_AOSV = 8, V_HD2 = 16, V_HD3 = 32, V_RCPU = 64, V_RCPB = 128;   // V_HD2/3: hoist pipelined 2 / 3 levels ahead

struct Prob {
  const double* val;       // [5][n] or [n][5]
  const uint16_t* mask;    // [n]
  const double* diag;
  const double* b;
  int64_t S, n, nlines;
  int64_t ldv;             // diagonal-major layout: elements between consecutive diagonals (>= n; skewed to de-alias the streams)
};

template <bool NT>
__device__ __forceinline__ double ldg(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// ---------------------------------------------------------------------------------------------------------------------
// reference: one sweep, one row per lane, in-order masked sum, x + (w (b - sum)) / A_ii
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep1(Prob p, const double* __restrict__ x, double* __restrict__ xo, double w, int aosv) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= p.n) return;
  const int64_t off[5] = {-p.S, -1, 0, 1, p.S};
  const int m = p.mask[r];
  double acc = 0.0;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if ((m >> d) & 1) {
      const double v = aosv ? p.val[r * 5 + d] : p.val[(int64_t)d * p.ldv + r];
      acc = acc + v * x[r + off[d]];
    }
  }
  xo[r] = x[r] + (w * (p.b[r] - acc)) / p.diag[r];
}

// ---------------------------------------------------------------------------------------------------------------------
// the march
// ---------------------------------------------------------------------------------------------------------------------
struct Coef {
  double v[5], b, dg;
  int m, lv;
};

__device__ __forceinline__ bool all_ones(double v) {
  return (__builtin_bit_cast(unsigned long long, v) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
}

template <int TS, int VAR>
__global__ __launch_bounds__(BW) void march5(Prob p, const double* __restrict__ x, double* __restrict__ xo, double w, int lines,
                                             int nstrips, int nblocks, int xcd_chunk, int tiles_per_xcd,
                                             unsigned long long* __restrict__ stamps = nullptr) {
  constexpr int HD = (VAR & V_HOIST) ? 99 : (VAR & V_HD2) ? 2 : (VAR & V_HD3) ? 3 : 0;
  constexpr bool HOIST = HD > 0, RCP = (VAR & (V_RCP | V_RCPU | V_RCPB)) != 0, RCPU = (VAR & V_RCPU) != 0, RCPB = (VAR & V_RCPB) != 0, NT = (VAR & V_NT) != 0, AOSV = (VAR & V_AOSV) != 0;
  constexpr int LW = BW + 2;
  constexpr int LV = TS, NC = LV + 1, U = NC;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* buf = reinterpret_cast<double*>(smem_raw);      // [TS][2][LW]
  int blk;
  {
    const int bid = blockIdx.x, C = xcd_chunk, xcd = bid % 8, j = bid / 8;
    blk = (j / C) * (8 * C) + xcd * C + (j % C);
    if (blk >= nblocks) return;
  }
  if (stamps && threadIdx.x == 0) stamps[2 * blk] = wall_clock64();
  struct Stamp {                                   // workgroup life time (100 MHz constant clock), written by lane 0 on every exit
    unsigned long long* p;
    __device__ ~Stamp() { if (p && threadIdx.x == 0) *p = wall_clock64(); }
  } stamp_end{stamps ? stamps + 2 * blk + 1 : nullptr};
  constexpr int halo = LV, bi_eff = BW - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t P0 = (int64_t)strip * bi_eff - halo;
  const int64_t J0 = (int64_t)lb * lines;
  const int L = (int)min((int64_t)lines, p.nlines - J0);
  const int tid = threadIdx.x;
  const int64_t gi = P0 + tid;
  const bool in_i = gi >= 0 && gi < p.S;
  const bool owner = tid >= halo && tid < BW - halo;

  auto x0_at = [&](int jl) -> double {
    int64_t flat = (J0 + jl) * p.S + gi;
    flat = min(max(flat, (int64_t)0), p.n - 1);
    return x[flat];
  };
  auto coef_at = [&](int jl) -> Coef {
    const int64_t gj = J0 + jl, r = gj * p.S + gi;
    const bool live = in_i && gj >= 0 && gj < p.nlines && r < p.n;
    const int64_t rc = live ? r : 0;
    Coef c;
#pragma unroll
    for (int d = 0; d < 5; ++d) c.v[d] = ldg<NT>(AOSV ? p.val + rc * 5 + d : p.val + (int64_t)d * p.ldv + rc);
    c.b = ldg<NT>(p.b + rc);
    c.dg = ldg<NT>(p.diag + rc);
    c.m = (int)p.mask[rc];
    c.lv = live;
    return c;
  };

  constexpr int j_first = -(LV - 1);
  const int j_last = L - 1 + (LV - 1);
  double h[LV][3];
#pragma unroll
  for (int t = 0; t < LV; ++t) h[t][0] = h[t][1] = h[t][2] = 0.0;
  h[0][1] = x0_at(j_first - 1);
  h[0][2] = x0_at(j_first);
  buf[((j_first + 64) & 1) * LW + 1 + tid] = h[0][2];
  double xpre = x0_at(j_first + 1);
  Coef c[NC];
  bool cfull[NC];
  double crc[NC];                                  // RCP: RN(1 / A_ii) of the set
  bool cdiv[NC];                                   // RCP: wave-uniform "every divisor of the set is Markstein-safe"
  c[0] = coef_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
  for (int s = 0; s < NC; ++s) { cfull[s] = false; crc[s] = 1.0; cdiv[s] = false; }

  auto rowsum = [&](const Coef& cc, bool full, double up, double left, double ctr, double right, double down) -> double {
    double acc = 0.0;
    if (full) {
      acc = acc + cc.v[0] * up;
      acc = acc + cc.v[1] * left;
      acc = acc + cc.v[2] * ctr;
      acc = acc + cc.v[3] * right;
      acc = acc + cc.v[4] * down;
    } else {
      const double xs[5] = {up, left, ctr, right, down};
      const int m = cc.lv ? cc.m : 0;
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const double nxt = acc + cc.v[d] * xs[d];
        acc = ((m >> d) & 1) ? nxt : acc;
      }
    }
    return acc;
  };
  auto jac = [&](const Coef& cc, double rc, bool divok, double ctr, double acc) -> double {
    const double num = w * (cc.b - acc);
    if constexpr (RCP) {
      // Markstein: q0 = RN(num * rc), rem = num - q0 * d (exact), q = RN(q0 + rem * rc) = RN(num / d) when rc = RN(1/d), d's
      // significand is not all ones and nothing leaves the normal range: |d| in [2^-500, 2^500] (set flag), |num| in
      // (2^-450, 2^450) (checked here: rejects +-0 -- whose sign the correction would lose --, denormals, Inf, NaN).
      bool safe = true;
      if constexpr (!RCPU) safe = divok && __all(__builtin_fabs(num) > 0x1p-450 && __builtin_fabs(num) < 0x1p450) != 0;
      if constexpr (RCPB) {
        if (!safe) {
          double nn = num;
          asm volatile("" : "+v"(nn));              // opaque: the division stays behind the branch
          return ctr + nn / cc.dg;
        }
      }
      if (safe) {
        const double q0 = num * rc;
        const double rem = __builtin_fma(-q0, cc.dg, num);
        return ctr + __builtin_fma(rem, rc, q0);
      }
    }
    return ctr + num / cc.dg;
  };

  auto do_step = [&](const int jl, const int u, const int nact) {
    __syncthreads();
    const int pj = (jl + 64) & 1;
    double* const row_e = buf + pj * LW + tid;
    double* const row_o = buf + (1 - pj) * LW + tid;
    double nl[LV], nr[LV];
    if constexpr (HOIST) {                         // level t reads what level t - 1 wrote a step ago: all reads up front
#pragma unroll
      for (int t = 1; t <= LV && t <= HD; ++t) {
        if (t <= nact) {
          const double* row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LW;
          nl[t - 1] = row[0];
          nr[t - 1] = row[2];
        }
      }
    }
    h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = xpre;
    row_o[1] = xpre;
    xpre = x0_at(jl + 2);
    c[(u + 1) % NC] = coef_at(jl + 1);
    {
      const Coef& cu = c[u % NC];
      cfull[u % NC] = __all(cu.lv && (cu.m & 0x1f) == 0x1f) != 0;
      if constexpr (RCP) {
        const double ad = __builtin_fabs(cu.dg);
        const bool ok = ad >= 0x1p-500 && ad <= 0x1p500 && !all_ones(cu.dg);
        cdiv[u % NC] = __all(ok || !cu.lv) != 0;
        crc[u % NC] = 1.0 / (ok ? cu.dg : 1.0);
      }
    }
#pragma unroll
    for (int t = 1; t <= LV; ++t) {
      if (t <= nact) {
        double* const row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LW;
        const int ci = ((u - (t - 1)) % NC + NC) % NC;
        double left, right;
        if constexpr (HOIST && HD < LV) {            // pipelined: the reads of level t + HD leave before level t computes
          if (t + HD <= LV && t + HD <= nact) {
            const double* rowp = ((t + HD - 1) % 2 == 0 ? row_e : row_o) + (t + HD - 1) * 2 * LW;
            nl[t + HD - 1] = rowp[0];
            nr[t + HD - 1] = rowp[2];
          }
        }
        if constexpr (HOIST) { left = nl[t - 1]; right = nr[t - 1]; }
        else { left = row[0]; right = row[2]; }
        const double acc = rowsum(c[ci], cfull[ci], h[t - 1][0], left, h[t - 1][1], right, h[t - 1][2]);
        const double xn = jac(c[ci], crc[ci], cdiv[ci], h[t - 1][1], acc);
        if (t < LV) {
          h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
          row[2 * LW + 1] = xn;
        }
        if (t == TS) {
          const int line = jl - (TS - 1);
          if (c[ci].lv && owner) xo[(J0 + line) * p.S + gi] = xn;
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


// ---------------------------------------------------------------------------------------------------------------------
// version 2 of the march (what the measurements above led to):
//   * positions outside the matrix (dead lanes of the edge strips, lines above / below the grid) hold +0 at every level, and a
//     row whose diagonal mask is exactly "every neighbour that exists" takes the unmasked sum: the skipped terms are then
//     (stored 0) x (+0) = +0 and acc + 0 == acc bit for bit (acc starts at +0 and a sum is -0 only if both operands are), so
//     the edge strips no longer run the masked path (they were 25 % slower than the inner strips and set the kernel time);
//   * the division by A_ii goes through RN(1 / A_ii) (one IEEE division per row and pass) and one Markstein correction per
//     level; whether that is the correctly rounded quotient is checked OFF the dependent chain (numerator in (2^-450, 2^450),
//     divisor in [2^-500, 2^500] with a significand that is not all ones) and a step in which any lane fails is redone with the
//     generic levels (masked sum + IEEE division) before the barrier -- levels are idempotent within a step;
//   * the i -/+ 1 neighbours of level t + HD are read from LDS before level t computes.
// ---------------------------------------------------------------------------------------------------------------------
struct Coef2 {
  double v[5], b, dg;
  int m;           // diagonal mask as loaded; only read at the set's first use (a step after the load)
};

// strips: BW lanes wide, interior strips own the middle BW - 2 TS points (iterate t is valid t points inside the strip edge);
// the first strip starts AT the grid's first point and the last one ends AT its last point (nothing lies beyond, so nothing
// is invalid there): no lane is ever outside the grid when S >= BW.
template <int TS, int HD, int PRIO = 0>
__global__ __launch_bounds__(BW) void march5v2(Prob p, const double* __restrict__ x, double* __restrict__ xo, double w, int lines,
                                               int nstrips, int nblocks, int xcd_chunk, int tiles_per_xcd,
                                               unsigned long long* __restrict__ stamps = nullptr) {
  constexpr int LW = BW + 2;
  constexpr int LV = TS, NC = LV + 1, U = NC;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* buf = reinterpret_cast<double*>(smem_raw);
  int blk;
  {
    const int bid = blockIdx.x, C = xcd_chunk, xcd = bid % 8, j = bid / 8;
    blk = (j / C) * (8 * C) + xcd * C + (j % C);
    if (blk >= nblocks) return;
  }
  if (stamps && threadIdx.x == 0) stamps[2 * blk] = wall_clock64();
  struct Stamp {
    unsigned long long* p;
    __device__ ~Stamp() { if (p && threadIdx.x == 0) *p = wall_clock64(); }
  } stamp_end{stamps ? stamps + 2 * blk + 1 : nullptr};
  // The SIMD arbiter serves the OLDER wave first: of the two workgroups a CU holds, the one dispatched second progresses ~15 %
  // slower and sets the kernel time.  PRIO 1: the later half of an XCD's workgroups runs at raised priority (sanity check: the
  // skew flips); PRIO 2: the two halves alternate priority every line step in antiphase.
  const int late = ((blockIdx.x / 8) >> 5) & 1;
  if constexpr (PRIO == 1) { if (late) __builtin_amdgcn_s_setprio(1); }
  constexpr int halo = LV, bi_eff = BW - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  const int64_t P0 = max((int64_t)0, min((int64_t)strip * bi_eff, p.S - BW));
  const int64_t own_lo = strip == 0 ? 0 : (int64_t)strip * bi_eff + halo;
  const int64_t own_hi = strip == nstrips - 1 ? p.S : (int64_t)strip * bi_eff + BW - halo;
  const int64_t J0 = (int64_t)lb * lines;
  const int L = (int)min((int64_t)lines, p.nlines - J0);
  const int tid = threadIdx.x;
  const int64_t gi = P0 + tid;
  const bool in_i = gi < p.S;                       // (false only on grids narrower than a strip)
  const bool owner = gi >= own_lo && gi < own_hi;
  const int em_i = 31 & ~(gi == 0 ? 2 : 0) & ~(gi == p.S - 1 ? 8 : 0);
  const bool ragged = p.nlines * p.S != p.n;
  // block-relative lines [jlo, jhi) are whole grid lines (a ragged last line is not)
  const int jlo = (int)max((int64_t)-64, -J0), jhi = (int)min((int64_t)(L + 64), p.nlines - (ragged ? 1 : 0) - J0);
  const bool wave_in = __all(in_i) != 0;

  for (int i = tid; i < TS * 2 * LW; i += BW) buf[i] = 0.0;          // (the pads either side of a line stay +0: beyond the grid)
  __syncthreads();

  auto row_clamped = [&](int jl) -> int64_t { return min(max((J0 + jl) * p.S + gi, (int64_t)0), p.n - 1); };
  auto live_at = [&](int jl) -> bool {              // (generic path and level 0 only)
    const int64_t r = (J0 + jl) * p.S + gi;
    return in_i && J0 + jl >= 0 && r < p.n;
  };
  auto coef_at = [&](int jl) -> Coef2 {
    const int64_t rc = row_clamped(jl);
    Coef2 c;
#pragma unroll
    for (int d = 0; d < 5; ++d) c.v[d] = p.val[(int64_t)d * p.ldv + rc];
    c.b = p.b[rc];
    c.dg = p.diag[rc];
    c.m = (int)p.mask[rc];
    return c;
  };

  constexpr int j_first = -(LV - 1);
  const int j_last = L - 1 + (LV - 1);
  double h[LV][3];
#pragma unroll
  for (int t = 0; t < LV; ++t) h[t][0] = h[t][1] = h[t][2] = 0.0;
  {
    const double a = x[row_clamped(j_first - 1)], b0 = x[row_clamped(j_first)];
    h[0][1] = live_at(j_first - 1) ? a : 0.0;
    h[0][2] = live_at(j_first) ? b0 : 0.0;
  }
  buf[((j_first + 64) & 1) * LW + 1 + tid] = h[0][2];
  double xpre = x[row_clamped(j_first + 1)];       // (untouched until the next step: the load stays in flight)
  Coef2 c[NC];
  double crc[NC];                                  // RN(1 / A_ii) of the set's rows (1 where the fast division is off anyway)
  bool cfast[NC];                                  // wave-uniform: a whole grid line, every row with the expected mask and a safe divisor
  c[0] = coef_at(j_first);
#pragma unroll
  for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
  for (int s = 0; s < NC; ++s) { crc[s] = 1.0; cfast[s] = false; }

  auto do_step = [&](const int jl, const int u, const int nact) {
    __syncthreads();
    const int pj = (jl + 64) & 1;
    if constexpr (PRIO == 2) { if (pj ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    if constexpr (PRIO == 3) { if (((jl + 64) >> 2 & 1) ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    double* const row_e = buf + pj * LW + tid;
    double* const row_o = buf + (1 - pj) * LW + tid;
    auto lds_row = [&](int t) -> double* { return ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LW; };   // level t - 1, line of level t
    double nl[LV], nr[LV];
#pragma unroll
    for (int t = 1; t <= LV && t <= HD; ++t)
      if (t <= nact) { nl[t - 1] = lds_row(t)[0]; nr[t - 1] = lds_row(t)[2]; }
#pragma unroll
    for (int t = 0; t < LV; ++t) { h[t][0] = h[t][1]; h[t][1] = h[t][2]; }      // (levels only assign h[t][2]: idempotent)
    const bool whole_next = wave_in && jl + 1 >= jlo && jl + 1 < jhi;            // wave-uniform: line jl + 1 exists for every lane
    h[0][2] = whole_next || live_at(jl + 1) ? xpre : 0.0;
    row_o[1] = h[0][2];
    xpre = x[row_clamped(jl + 2)];
    c[(u + 1) % NC] = coef_at(jl + 1);
    {
      const Coef2& cu = c[u % NC];
      const double ad = __builtin_fabs(cu.dg);
      const bool dok = ad >= 0x1p-500 && ad <= 0x1p500 && !all_ones(cu.dg);
      // the mask a row at this position has when every neighbour that exists is coupled
      const int em = em_i & ~(J0 + jl == 0 ? 1 : 0) & ~((J0 + jl + 2) * p.S > p.n ? 16 : 0) & ~(ragged && (J0 + jl + 1) * p.S + gi >= p.n ? 16 : 0);
      cfast[u % NC] = wave_in && jl >= jlo && jl < jhi && __all((cu.m & 0x1f) == em && dok) != 0;
      crc[u % NC] = 1.0 / (dok ? cu.dg : 1.0);
    }
    bool fast = true;
#pragma unroll
    for (int t = 1; t <= LV; ++t)
      if (t <= nact) fast = fast && cfast[((u - (t - 1)) % NC + NC) % NC];
    bool bad = false;                               // per lane: a numerator the Markstein step is not proven for
    if (fast) {
#pragma unroll
      for (int t = 1; t <= LV; ++t) {
        if (t <= nact) {
          double* const row = lds_row(t);
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          if constexpr (HD < LV) {
            if (t + HD <= LV && t + HD <= nact) { nl[t + HD - 1] = lds_row(t + HD)[0]; nr[t + HD - 1] = lds_row(t + HD)[2]; }
          }
          double left, right;
          if constexpr (HD > 0) { left = nl[t - 1]; right = nr[t - 1]; }
          else { left = row[0]; right = row[2]; }
          const Coef2& cc = c[ci];
          double acc = 0.0;
          acc = acc + cc.v[0] * h[t - 1][0];
          acc = acc + cc.v[1] * left;
          acc = acc + cc.v[2] * h[t - 1][1];
          acc = acc + cc.v[3] * right;
          acc = acc + cc.v[4] * h[t - 1][2];
          const double num = w * (cc.b - acc);
          const double an = __builtin_fabs(num);
          bad = bad || !(an > 0x1p-450 && an < 0x1p450);
          const double q0 = num * crc[ci];
          const double rem = __builtin_fma(-q0, cc.dg, num);
          const double xn = h[t - 1][1] + __builtin_fma(rem, crc[ci], q0);
          if (t < LV) { h[t][2] = xn; row[2 * LW + 1] = xn; }
          if (t == TS) { if (owner) xo[(J0 + jl - (TS - 1)) * p.S + gi] = xn; }
        }
      }
    }
    if (!fast || __any(bad)) {                      // generic levels: masked in-order sum, IEEE division, +0 outside the matrix
      asm volatile("" ::: "memory");                // (keeps this path's LDS reads from being merged with the fast path's above the branch)
#pragma unroll
      for (int t = 1; t <= LV; ++t) {
        if (t <= nact) {
          double* const row = lds_row(t);
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          const int line = jl - (t - 1);
          const Coef2& cc = c[ci];
          const bool lv = live_at(line) && J0 + line < p.nlines;
          const double xs[5] = {h[t - 1][0], row[0], h[t - 1][1], row[2], h[t - 1][2]};
          const int m = lv ? (int)p.mask[row_clamped(line)] : 0;      // (not kept per set: this path is rare)
          double acc = 0.0;
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            const double nxt = acc + cc.v[d] * xs[d];
            acc = ((m >> d) & 1) ? nxt : acc;
          }
          double xn = h[t - 1][1] + (w * (cc.b - acc)) / cc.dg;
          xn = lv ? xn : 0.0;
          if (t < LV) { h[t][2] = xn; row[2 * LW + 1] = xn; }
          if (t == TS) { if (lv && owner) xo[(J0 + line) * p.S + gi] = xn; }
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

// ---------------------------------------------------------------------------------------------------------------------
static int pick_lines(int64_t nlines, int64_t nstrips, int64_t slots, int ramp, int min_lines) {
  int64_t best_cost = -1, best_l = nlines;
  for (int rounds = 1; rounds <= 8; ++rounds) {
    int64_t blocks = rounds * slots / nstrips;
    if (blocks < 1) blocks = 1;
    int64_t l = (nlines + blocks - 1) / blocks;
    if (l < min_lines) l = min_lines;
    const int64_t nb = (nlines + l - 1) / l;
    const int64_t r = (nb * nstrips + slots - 1) / slots;
    const int64_t cost = r * (l + ramp);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_l = l; }
  }
  return (int)best_l;
}

struct Ctx {
  Prob p, p_aos;
  double *x, *out, *ref, *tmp;
  double w;
  int reps;
  unsigned long long* stamps = nullptr;
  bool verbose = false;
  bool quiet = false;                                // sel 3: only collect samples
  std::vector<std::pair<std::string, std::vector<float>>> samples;
  std::vector<double> h_ref, h_out;
};

template <typename K>
void run_kernel(Ctx& c, const char* name, int TS, K kern, const Prob& p, int aosv, int per_cu = 2, int lines_override = 0) {
  const int bi_eff = BW - 2 * TS;
  const int nstrips = (int)((p.S + bi_eff - 1) / bi_eff);
  int lines = lines_override > 0 ? lines_override : pick_lines(p.nlines, nstrips, 256 * per_cu, 2 * (TS - 1), 2 * TS);
  if (lines < 2 * TS) lines = 2 * TS;
  const int nlb = (int)((p.nlines + lines - 1) / lines);
  const int nblocks = nlb * nstrips;
  const int C = 64, per_round = 8 * C;
  const int tiles_per_xcd = ((nblocks + per_round - 1) / per_round) * C;
  const size_t lds = (size_t)TS * 2 * (BW + 2) * sizeof(double);
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, BW, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < c.reps + 3; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(tiles_per_xcd * 8), dim3(BW), lds, 0, p, c.x, c.out, c.w, lines, nstrips, nblocks, C, tiles_per_xcd,
                       (unsigned long long*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) ts.push_back(ms);
  }
  CK(hipGetLastError());
  char lifeline[400] = "";
  if (c.stamps) {                                   // life time of every workgroup of one more launch
    CK(hipMemset(c.stamps, 0, 16 * (size_t)nblocks));
    hipLaunchKernelGGL(kern, dim3(tiles_per_xcd * 8), dim3(BW), lds, 0, p, c.x, c.out, c.w, lines, nstrips, nblocks, C, tiles_per_xcd,
                       c.stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(2 * (size_t)nblocks);
    CK(hipMemcpy(st.data(), c.stamps, 16 * (size_t)nblocks, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblocks; ++b) { t0 = std::min(t0, st[2 * b]); t1 = std::max(t1, st[2 * b + 1]); }
    std::vector<double> edge, inner;
    double sum = 0;
    for (int b = 0; b < nblocks; ++b) {
      const double life = (st[2 * b + 1] - st[2 * b]) * 0.01;
      sum += life;
      const int strip = b % nstrips;
      (strip == 0 || strip == nstrips - 1 ? edge : inner).push_back(life);
    }
    std::sort(edge.begin(), edge.end());
    std::sort(inner.begin(), inner.end());
    if (c.verbose) {
      std::vector<double> bx(8, 0), bs(nstrips, 0), bl(nlb, 0);
      std::vector<int> nx(8, 0);
      for (int b = 0; b < nblocks; ++b) {
        const double life = (st[2 * b + 1] - st[2 * b]) * 0.01;
        bx[(b / 64) % 8] += life; nx[(b / 64) % 8]++;
        bs[b % nstrips] += life / nlb;
        bl[b / nstrips] += life / nstrips;
      }
      printf("      mean life per XCD:");
      for (int e = 0; e < 8; ++e) printf(" %.0f", bx[e] / std::max(nx[e], 1));
      printf("\n      mean life per strip:");
      for (int i = 0; i < nstrips; ++i) printf(" %.0f", bs[i]);
      printf("\n      mean life per line block:");
      for (int i = 0; i < nlb; ++i) printf(" %.0f", bl[i]);
      printf("\n");
    }
    snprintf(lifeline, sizeof lifeline, "      workgroup life (us): span %.1f mean %.1f | inner strips %.1f / %.1f / %.1f | edge strips %.1f / %.1f / %.1f (min / median / max)\n",
             (t1 - t0) * 0.01, sum / nblocks, inner.front(), inner[inner.size() / 2], inner.back(), edge.front(), edge[edge.size() / 2], edge.back());
  }
  // reference: TS single sweeps
  const double* src = c.x;
  double* bufs[2] = {c.ref, c.tmp};
  for (int s = 0; s < TS; ++s) {
    double* dst = bufs[(TS - 1 - s) % 2];
    hipLaunchKernelGGL(sweep1, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, 0, p, src, dst, c.w, aosv);
    src = dst;
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(c.h_ref.data(), c.ref, p.n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.h_out.data(), c.out, p.n * 8, hipMemcpyDeviceToHost));
  bool same = memcmp(c.h_ref.data(), c.h_out.data(), p.n * 8) == 0;
  if (!same) {                                      // (a NaN matches any NaN: sign / payload follow the operand order the compiler picked)
    int64_t nbad = 0, first = -1;
    for (int64_t i = 0; i < p.n; ++i)
      if (memcmp(&c.h_ref[i], &c.h_out[i], 8) && !(c.h_ref[i] != c.h_ref[i] && c.h_out[i] != c.h_out[i])) { if (first < 0) first = i; ++nbad; }
    same = nbad == 0;
    if (nbad)
    printf("      %ld mismatches; first at row %ld (line %ld, point %ld): want %a got %a\n", (long)nbad, (long)first, (long)(first / p.S),
           (long)(first % p.S), c.h_ref[first], c.h_out[first]);
  }
  if (c.quiet) {
    std::string key = std::string("TS=") + std::to_string(TS) + " " + name;
    bool found = false;
    for (auto& kv : c.samples) if (kv.first == key) { kv.second.insert(kv.second.end(), ts.begin(), ts.end()); found = true; }
    if (!found) c.samples.push_back({key, ts});
    if (!same) printf("%s: NOT bit-identical\n", key.c_str());
    return;
  }
  std::sort(ts.begin(), ts.end());
  const double med = ts[ts.size() / 2] * 1e3, mn = ts[0] * 1e3, q1 = ts[ts.size() / 4] * 1e3, mx = ts.back() * 1e3;
  const double layout = 74.0 * p.n;
  printf("TS=%d %-22s vgpr=%3d occ=%d lines=%3d blocks=%4d : median %7.1f us  min %7.1f q1 %7.1f max %7.1f us  (%5.2f TB/s on 74 B/row, %5.1f us/sweep)  bit-identical=%s\n%s",
         TS, name, fa.numRegs, occ, lines, nblocks, med, mn, q1, mx, layout / med / 1e6, med / TS, same ? "yes" : "NO", lifeline);
  fflush(stdout);
}

template <int TS, int VAR>
void run_variant(Ctx& c, const char* name, int per_cu = 2, int lines_override = 0) {
  run_kernel(c, name, TS, march5<TS, VAR>, (VAR & V_AOSV) ? c.p_aos : c.p, (VAR & V_AOSV) ? 1 : 0, per_cu, lines_override);
}
template <int TS, int HD, int PRIO = 0>
void run_v2(Ctx& c, const char* name, int per_cu = 2, int lines_override = 0) {
  run_kernel(c, name, TS, march5v2<TS, HD, PRIO>, c.p, 0, per_cu, lines_override);
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atol(argv[1]) : 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int special = argc > 3 ? atoi(argv[3]) : 0;
  const int64_t skew = argc > 5 ? atol(argv[5]) : 0;   // elements of padding between the arrays (0: every array 2^k-aligned as the allocator hands them out)
  const int sel = argc > 4 ? atoi(argv[4]) : 0;       // 0: every variant; 1: only the three kernels the counter passes compare   // 1: sprinkle zeros / tiny / huge / non-finite values (division fallback test)
  const int64_t n = N * N;
  Ctx c;
  c.reps = reps;
  c.w = 0.7;
  const int64_t ldv = n + skew;
  std::vector<double> val(5 * ldv), val_aos(5 * n), diag(n), b(n), x(n);
  std::vector<uint16_t> mask(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
  for (int64_t r = 0; r < n; ++r) {
    const int64_t j = r / N, i = r % N;
    int m = 4;
    if (j > 0) m |= 1;
    if (i > 0) m |= 2;
    if (i < N - 1) m |= 8;
    if (j < N - 1) m |= 16;
    if (special && (r % 977) == 5) m &= ~2;                    // a hole in the pattern
    mask[r] = (uint16_t)m;
    const double vv[5] = {1.0 + 0.25 * rnd(), 1.0 + 0.25 * rnd(), -4.0 - rnd(), 1.0 + 0.25 * rnd(), 1.0 + 0.25 * rnd()};
    for (int d = 0; d < 5; ++d) {
      const double v = ((m >> d) & 1) ? vv[d] : 0.0;
      val[(int64_t)d * ldv + r] = v;
      val_aos[r * 5 + d] = v;
    }
    diag[r] = vv[2];
    b[r] = rnd();
    x[r] = rnd();
    if (special) {
      const int64_t k = r % 4099;
      if (k == 1) { b[r] = 0.0; }
      if (k == 2) { x[r] = 0.0; b[r] = 0.0; }
      if (k == 3) diag[r] = 0x1.fffffffffffffp+1;              // all-ones significand
      if (k == 4) diag[r] = 0x1p-600;                          // extreme divisor
      if (k == 5) b[r] = 0x1p600;                              // huge numerator
      if (k == 6) b[r] = 0x1p-700;
      if (k == 7) b[r] = -0.0;
      if (k == 8 && (r % 8198) == 8) b[r] = __builtin_inf();
      if (k == 9 && (r % 40990) == 9) b[r] = __builtin_nan("");
      if (k == 10) diag[r] = 0x1p-1040;                        // denormal divisor
    }
  }
  if (special) {                                               // a region of exact zeros (b = 0, x = 0): numerators +-0
    for (int64_t r = n / 2; r < n / 2 + 40 * N; ++r) { b[r] = 0.0; x[r] = (r & 1) ? -0.0 : 0.0; }
  }
  double *d_val, *d_val_aos, *d_diag, *d_b;
  uint16_t* d_mask;
  CK(hipMalloc(&d_val, 5 * ldv * 8));
  CK(hipMalloc(&d_val_aos, 5 * n * 8));
  CK(hipMalloc(&d_diag, (n + 8 * skew) * 8));
  CK(hipMalloc(&d_b, (n + 8 * skew) * 8));
  d_diag += 5 * skew;
  d_b += 6 * skew;
  CK(hipMalloc(&d_mask, n * 2));
  CK(hipMalloc(&c.x, (n + 8 * skew) * 8));
  CK(hipMalloc(&c.out, (n + 8 * skew) * 8));
  c.x += 7 * skew;
  c.out += 3 * skew + skew / 2;
  CK(hipMalloc(&c.ref, n * 8));
  CK(hipMalloc(&c.tmp, n * 8));
  CK(hipMemcpy(d_val, val.data(), 5 * ldv * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val_aos, val_aos.data(), 5 * n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_diag, diag.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_mask, mask.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(c.x, x.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(c.out, 0, n * 8));
  c.p = Prob{d_val, d_mask, d_diag, d_b, N, n, N, ldv};
  c.p_aos = c.p;
  c.p_aos.val = d_val_aos;
  c.h_ref.resize(n);
  c.h_out.resize(n);
  printf("march5_exp N=%ld reps=%d special=%d\n", (long)N, reps, special);
  if (sel != 1) CK(hipMalloc(&c.stamps, 16 * 4096));
  c.verbose = sel == 2;
  if (sel == 3) {                                    // interleaved A/B: 6 rounds x reps launches each, deciles over all samples
    c.quiet = true;
    for (int round = 0; round < 6; ++round) {
      run_variant<7, 0>(c, "base");
      run_v2<7, 0>(c, "v2 hd0");
      run_v2<7, 0, 2>(c, "v2 hd0 prio-alt1");
      run_v2<7, 2>(c, "v2 hd2");
      run_v2<7, 2, 2>(c, "v2 hd2 prio-alt1");
      run_v2<6, 2>(c, "v2 hd2");
      run_v2<6, 2, 2>(c, "v2 hd2 prio-alt1");
      run_variant<6, 0>(c, "base");
      run_variant<4, 0>(c, "base");
    }
    for (auto& kv : c.samples) {
      auto& v = kv.second;
      std::sort(v.begin(), v.end());
      printf("%-28s n=%3zu  min %6.1f  p10 %6.1f  p25 %6.1f  median %6.1f  p75 %6.1f  p90 %6.1f  max %6.1f us\n", kv.first.c_str(), v.size(), v[0] * 1e3,
             v[v.size() / 10] * 1e3, v[v.size() / 4] * 1e3, v[v.size() / 2] * 1e3, v[3 * v.size() / 4] * 1e3, v[9 * v.size() / 10] * 1e3, v.back() * 1e3);
    }
    return 0;
  }
  if (sel == 2) {
    run_variant<7, 0>(c, "base");
    run_v2<7, 0>(c, "v2 hd0");
    run_v2<7, 2>(c, "v2 hd2");
    run_v2<6, 2>(c, "v2 hd2");
    run_v2<7, 0, 1>(c, "v2 hd0 prio-late");
    run_v2<7, 0, 2>(c, "v2 hd0 prio-alt1");
    run_v2<7, 0, 3>(c, "v2 hd0 prio-alt4");
    run_v2<6, 2, 2>(c, "v2 hd2 prio-alt1");
    return 0;
  }
  if (sel == 1) {
    run_variant<4, 0>(c, "base");
    run_variant<7, 0>(c, "base");
    run_v2<7, 2>(c, "v2 hd2");
    return 0;
  }
  for (int round = 0; round < 2; ++round) {
    run_variant<4, 0>(c, "base");
    run_variant<7, 0>(c, "base");
    run_variant<7, V_HD2 | V_RCPU>(c, "hd2+rcp-unchecked");
    run_v2<7, 0>(c, "v2 hd0");
    run_v2<7, 2>(c, "v2 hd2");
    run_v2<7, 3>(c, "v2 hd3");
    run_v2<7, 7>(c, "v2 hd7");
    run_v2<8, 0>(c, "v2 hd0");
    run_v2<8, 2>(c, "v2 hd2");
    run_v2<6, 2>(c, "v2 hd2");
    run_v2<6, 6>(c, "v2 hd6");
    run_v2<5, 5>(c, "v2 hd5");
    run_v2<4, 4>(c, "v2 hd4");
  }
  return 0;
}
