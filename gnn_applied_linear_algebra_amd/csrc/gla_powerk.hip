// libgla_hip.so -- temporally blocked power iteration on canonical stencil plans (gfx950 / MI355X).
//
// PowerMethodGNN(num_iter) (PowerMethodGNN.py:296-334) is, per iteration, b <- A b; n = sqrt(sum b^2); b <- b / n: a global
// norm separates the iterations, so the step-by-step path (gla_power_step) pays one pass over the matrix and the K-wide
// vector per iteration.  A is linear: m iterations started from a normalised b_0 give b_m = A^m b_0 / ||A^m b_0||, and the
// norm the reference reports after iteration i is ||A^i b_0|| / ||A^(i-1) b_0||.  So m iterations are taken in ONE pass of the
// K-wide march (as jacobi_marchk_kernel / cheby_march5_kernel): level j carries c_j = A^j b_0 UNNORMALISED, every level
// leaves sum_i c_j[i]^2 of the rows a workgroup owns as a by-product (fixed-order two-stage reduction), only c_m goes to HBM,
// and the normalisation by ||c_m|| is applied to the operand of the next pass as gla_power_step does (correctly rounded
// reciprocal + one Markstein step = the IEEE quotient).  |lambda| <= 8 (16 on the Q1 stencil) per level and at most 5 levels
// per pass: no overflow; the scale is reset every pass.  Iterates agree with the step-by-step path to rounding (the
// reference rounds b after every division, this path after every pass) -- the power method's outputs pass through global
// sums and were never bit-exact (1e-12).
// Lanes = (grid point, 16-byte rhs vector); own column of every level in registers, i -/+ 1 neighbours through LDS (5-point:
// one line per level and parity, one barrier per line step; 9-point: three lines per level, one barrier per level), static
// coefficient-set names, wave-uniform full-pattern fast path.
#include "gla_device.h"

namespace {

struct PowK {
  const void* val;                     // [D][n_rows] diagonal-major values
  const uint16_t* mask;                // [n_rows]
  int32_t S;
  int64_t n_rows, n_cols;
};

template <typename T, int D>
struct PCoef {
  T v[(D + 3) & ~3];
  int m, lv;
};

template <typename T>
__device__ __forceinline__ bool sig_all_ones(T v) {
  if constexpr (sizeof(T) == 8) return (__builtin_bit_cast(unsigned long long, v) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
  else return (__builtin_bit_cast(unsigned, v) & 0x7FFFFFu) == 0x7FFFFFu;
}

// D = 5: diagonals (-1,0) (0,-1) (0,0) (0,1) (1,0); D = 9: (a, b) in {-1, 0, 1}^2 row-major.  VW = elements per lane
// (1: a single right-hand side, lanes = points; else 16 bytes), KV = lanes per point.
// Blocks = (line block, strip), strips of one line block adjacent in the launch order: neighbouring strips run side by side on
// one XCD, so the points two strips share (2 TS of a strip's 256 / KV) are L2 hits.  (Tried: equal runs of lines cut from the
// strip-major sequence, every workgroup the same number of line steps whatever the strip count -- 14 % slower at K = 16,
// the shared points then come from HBM.)  gla_pick_block_lines takes several rounds of shorter blocks when the strip count
// quantises badly against the 512 resident workgroups.
// threads per workgroup: 512 where a row is four or more 16-byte vectors wide -- a strip of 64 / 128 points instead of 32 / 64,
// so the 2 TS halo points either side that every strip recomputes are a fifth of its work instead of almost a half (K = 16, TS = 5)
template <int KV> struct PowThreads { static constexpr int nt = KV >= 4 ? 512 : GLA_BLOCK; };

// TAIL (TS = 1 only; gla_power_rayleigh: layer 3 of the last iteration and the two Rayleigh layers, PowerMethodGNN.py:187-294,
// with the module's [b, b^2] result): level 0 is b = operand ./ nrm, level 1 is A b; b goes to yout and b^2 to y2out, A b is
// not stored; the sums are sum b^2 (first K partials) and sum b (A b) (second K).
template <typename T, int VW, int KV, int TS, int D, int NT, bool TAIL = false>
__global__ __launch_bounds__(NT) void power_march_kernel(PowK pv, const T* __restrict__ x0, int64_t ldx0,
                                                         const T* __restrict__ nrm, T* __restrict__ yout, int64_t ldy,
                                                         T* __restrict__ partials, int lines, int nstrips,
                                                         int64_t nlines, TileMap tm, T* __restrict__ y2out = nullptr,
                                                         int64_t ldy2 = 0) {
  static_assert(!TAIL || TS == 1, "the Rayleigh tail is a one-level pass");
  constexpr int K = KV * VW;
  constexpr int BP = NT / KV;                      // grid points of a strip
  constexpr int LWV = (BP + 2) * KV;               // lane vectors of a line in LDS: point ii at [(1 + ii) * KV + q]
  constexpr int NC = TS + 1, U = NC;
  constexpr unsigned FULL = (1u << D) - 1u;
  typedef T TV __attribute__((ext_vector_type(VW)));
  typedef PCoef<T, D> Coef;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  TV* buf = reinterpret_cast<TV*>(smem_raw);
  const int blk = map_tile(tm);
  if (blk < 0) return;
  constexpr int halo = TS, bp_eff = BP - 2 * halo;
  const int tid = threadIdx.x, ii = tid / KV, q = tid % KV;
  const T* __restrict__ val = static_cast<const T*>(pv.val);
  const bool owner = ii >= halo && ii < BP - halo;
  // (priority alternation between the two waves that share a SIMD: gla_march5.hip; NT = 512: waves w and w + 4 of the workgroup)
  const int late = NT >= 512 ? (tid >> 8) & 1 : ((blockIdx.x / GLA_XCDS) >> 5) & 1;

  // this lane's divisors (the absolute norm of the previous pass' result per right-hand side), as EpiPower::gx
  TV dn, drn;
  bool dslow = false;
#pragma unroll
  for (int i = 0; i < VW; ++i) {
    const T n = nrm ? nrm[q * VW + i] : T(1);
    dn[i] = n;
    drn[i] = T(1) / n;
    dslow = dslow || sig_all_ones(n);
  }
  auto normalised = [&](TV v) -> TV {
    if (!nrm) return v;
    TV o;
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      if (dslow) o[i] = v[i] / dn[i];
      else {                                       // a / b from RN(1 / b) and one Markstein step (gla_rows.hip: div_by)
        const T q0 = v[i] * drn[i];
        const T r = fma_t(-q0, dn[i], v[i]);
        o[i] = fma_t(r, drn[i], q0);
      }
    }
    return o;
  };
  auto apply = [&](const Coef& c, bool full, const TV (&xs)[D]) -> TV {      // (A c)_row: in-order masked sum
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
    return acc;
  };
  TV sq[2];                                        // sums of c_(TS-1)^2 and c_TS^2 over the rows this lane owns (fixed order);
  sq[0] = sq[1] = TV(T(0));                        // c_0 = the normalised operand

  // the workgroup's block: lines [J0, J0 + L) of strip `strip`
  auto march = [&](const int strip, const int64_t J0, const int L) {
    const int64_t P0 = (int64_t)strip * bp_eff - halo;
    const int64_t gi = P0 + ii;
    const bool in_i = gi >= 0 && gi < pv.S;
    auto x0_at = [&](int jl) -> TV {               // clamped: values of dead rows are never used
      int64_t flat = (J0 + jl) * pv.S + gi;
      flat = min(max(flat, (int64_t)0), pv.n_cols - 1);
      return ldv<TV>(x0 + flat * ldx0 + q * VW);
    };
    auto coef_at = [&](int jl) -> Coef {
      const int64_t gj = J0 + jl, r = gj * pv.S + gi;
      const bool live = in_i && gj >= 0 && gj < nlines && r < pv.n_rows;
      const int64_t rc = live ? r : 0;
      Coef c;
#pragma unroll
      for (int d = 0; d < D; ++d) c.v[d] = val[(int64_t)d * pv.n_rows + rc];
      c.m = (int)pv.mask[rc];
      c.lv = live;
      return c;
    };
    constexpr int j_first = -(TS - 1);
    const int j_last = L - 1 + (TS - 1);
    TV h[TS][3];
#pragma unroll
    for (int t = 0; t < TS; ++t) h[t][0] = h[t][1] = h[t][2] = TV(T(0));
    h[0][1] = normalised(x0_at(j_first - 1));
    h[0][2] = normalised(x0_at(j_first));
    if constexpr (D == 5) {
      buf[((j_first + 64) & 1) * LWV + (1 + ii) * KV + q] = h[0][2];
    } else {
      buf[((j_first - 1 + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][1];
      buf[((j_first + 63) % 3) * LWV + (1 + ii) * KV + q] = h[0][2];
    }
    if constexpr (TS == 1) {                       // (the level below the last is the operand itself: line 0 of the block)
      if (owner && in_i && L > 0 && J0 * pv.S + gi < pv.n_rows) {
        const TV b2 = h[0][2] * h[0][2];
        sq[0] = sq[0] + b2;
        if constexpr (TAIL) {
          stv<TV>(yout + (J0 * pv.S + gi) * ldy + q * VW, h[0][2]);
          stv<TV>(y2out + (J0 * pv.S + gi) * ldy2 + q * VW, b2);
        }
      }
    }
    // x^(k) is fetched PD line steps ahead.  PD = 3 measured no gain over 1 at K = 16 (5.41 ms per 10 iterations either way):
    // the pass is not bound by the loads in flight -- the counters (profiles/r03/pmc_k16_summary.txt) show the VALU ~46 % busy
    // and a quarter of the wave time in s_waitcnt at two waves per SIMD: dependent-latency bound, like the levels of one line step.
    constexpr int PD = 1;
    TV xq[PD];                                     // xq[i]: line jl + 1 + i (normalised when consumed: the loads stay in flight)
#pragma unroll
    for (int i = 0; i < PD; ++i) xq[i] = x0_at(j_first + 1 + i);
    Coef c[NC];
    bool cfull[NC];
    c[0] = coef_at(j_first);
#pragma unroll
    for (int s = 1; s < NC; ++s) c[s] = c[0];
#pragma unroll
    for (int s = 0; s < NC; ++s) cfull[s] = false;

    // level t of row (line, this lane's point) is xn: the last two levels are summed, the last one goes to HBM
    auto finish = [&](const Coef& cc, int t, int line, TV xn, TV ctr) {      // ctr: the row's value on the level below
      if (t >= TS - 1 && t >= 1) {
        if (cc.lv && owner && line >= 0 && line < L) {
          if constexpr (TAIL) {
            sq[1] = sq[1] + ctr * xn;                // y_i = b_i (A b)_i, n_A = sum y   (PowerMethodGNN.py:209-266)
          } else {
            sq[t - (TS - 1)] = sq[t - (TS - 1)] + xn * xn;
            if (t == TS) stv<TV>(yout + ((J0 + line) * pv.S + gi) * ldy + q * VW, xn);
          }
        }
      }
    };
    auto do_step = [&](const int jl, const int u, const int nact) {
      const TV x1 = normalised(xq[0]);
      if constexpr (TS == 1) {                     // (the level below the last is the operand itself)
        const int64_t gj = J0 + jl + 1;
        if (owner && in_i && jl + 1 >= 0 && jl + 1 < L && gj * pv.S + gi < pv.n_rows) {
          const TV b2 = x1 * x1;
          sq[0] = sq[0] + b2;
          if constexpr (TAIL) {
            stv<TV>(yout + (gj * pv.S + gi) * ldy + q * VW, x1);
            stv<TV>(y2out + (gj * pv.S + gi) * ldy2 + q * VW, b2);
          }
        }
      }
      if constexpr (D == 5) {
        __syncthreads();
        const int pj = (jl + 64) & 1;
        if (pj ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        TV* const row_e = buf + pj * LWV + ii * KV + q;          // lines of the parity of jl: [0] left, [KV] own, [2 KV] right
        TV* const row_o = buf + (1 - pj) * LWV + ii * KV + q;
        h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = x1;
        row_o[KV] = x1;                            // level 0, line jl + 1
#pragma unroll
        for (int i = 0; i + 1 < PD; ++i) xq[i] = xq[i + 1];
        xq[PD - 1] = x0_at(jl + 1 + PD);
        c[(u + 1) % NC] = coef_at(jl + 1);
        cfull[u % NC] = __all(c[u % NC].lv && ((unsigned)c[u % NC].m & FULL) == FULL) != 0;   // (at first use)
#pragma unroll
        for (int t = 1; t <= TS; ++t) {
          if (t <= nact) {
            TV* const row = ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LWV;   // level t - 1, line jl - (t - 1)
            const int ci = ((u - (t - 1)) % NC + NC) % NC;
            const TV xs[5] = {h[t - 1][0], row[0], h[t - 1][1], row[2 * KV], h[t - 1][2]};
            const TV xn = apply(c[ci], cfull[ci], xs);
            if (t < TS) {
              h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
              row[2 * LWV + KV] = xn;
            }
            finish(c[ci], t, jl - (t - 1), xn, h[t - 1][1]);
          }
        }
      } else {
        const int s1 = (jl + 1 + 63) % 3, s0 = (jl + 63) % 3, s2 = (jl - 1 + 63) % 3;
        TV* const rw[3] = {buf + s1 * LWV + ii * KV + q, buf + s0 * LWV + ii * KV + q, buf + s2 * LWV + ii * KV + q};
        h[0][0] = h[0][1]; h[0][1] = h[0][2]; h[0][2] = x1;
        rw[0][KV] = x1;
#pragma unroll
        for (int i = 0; i + 1 < PD; ++i) xq[i] = xq[i + 1];
        xq[PD - 1] = x0_at(jl + 1 + PD);
        c[(u + 1) % NC] = coef_at(jl + 1);
        cfull[u % NC] = __all(c[u % NC].lv && ((unsigned)c[u % NC].m & FULL) == FULL) != 0;
#pragma unroll
        for (int t = 1; t <= TS; ++t) {
          __syncthreads();
          if (t <= nact) {
            TV* const below = rw[(t - 1) % 3] + (t - 1) * 3 * LWV;
            TV* const same = rw[t % 3] + (t - 1) * 3 * LWV;
            TV* const above = rw[(t + 1) % 3] + (t - 1) * 3 * LWV;
            const int ci = ((u - (t - 1)) % NC + NC) % NC;
            const TV xs[9] = {above[0], h[t - 1][0], above[2 * KV], same[0], h[t - 1][1], same[2 * KV],
                              below[0], h[t - 1][2], below[2 * KV]};
            const TV xn = apply(c[ci], cfull[ci], xs);
            if (t < TS) {
              h[t][0] = h[t][1]; h[t][1] = h[t][2]; h[t][2] = xn;
              same[3 * LWV + KV] = xn;
            }
            finish(c[ci], t, jl - (t - 1), xn, h[t - 1][1]);
          }
        }
      }
    };
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
  };
  {
    const int lb = blk / nstrips, strip = blk - lb * nstrips;
    const int64_t J0 = (int64_t)lb * lines;
    march(strip, J0, (int)min((int64_t)lines, nlines - J0));
  }
  // the workgroup's two sums of squares: lanes of one rhs vector are added in a fixed order
  TV* sh = buf;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const TV tot = block_lane_group_sum<KV, TV, NT / 64>(sq[t], sh);
    if (tid < KV) {
#pragma unroll
      for (int i = 0; i < VW; ++i) partials[((int64_t)blk * 2 + t) * K + tid * VW + i] = tot[i];
    }
  }
}

struct PowGeom { int lines, nstrips, nblocks; int64_t nlines; };

template <int KV, int TS>
PowGeom pow_geom(const gla_csr* A) {
  constexpr int NT = PowThreads<KV>::nt;
  constexpr int BP = NT / KV, bp_eff = BP - 2 * TS;
  PowGeom g;
  g.nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  g.nstrips = (int)((A->dia_S + bp_eff - 1) / bp_eff);
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) lines = gla_pick_block_lines(g.nlines, g.nstrips, 256 * (2 * GLA_BLOCK / NT), 2 * (TS - 1), 2 * TS);
  if (lines < 2 * TS) lines = 2 * TS;
  g.lines = lines;
  g.nblocks = (int)(((g.nlines + lines - 1) / lines) * g.nstrips);
  return g;
}

template <typename T, int VW, int KV, int TS, int D>
int launch_power(gla_csr* A, const T* x, int64_t ldx, const T* nrm, T* y, int64_t ldy, T* sumsq, T* work, hipStream_t st) {
  constexpr int NT = PowThreads<KV>::nt;
  constexpr int BP = NT / KV, K = KV * VW;
  static_assert(BP - 2 * TS >= 1, "strip too narrow");
  PowK pv;
  pv.val = A->dia_val;
  pv.mask = A->dia_mask;
  pv.S = A->dia_S;
  pv.n_rows = A->n_rows; pv.n_cols = A->n_cols;
  const PowGeom g = pow_geom<KV, TS>(A);
  const size_t lds = (size_t)TS * (D == 5 ? 2 : 3) * (BP + 2) * KV * VW * sizeof(T);
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = g.nblocks;
  set_tile_mapping(tm);
  auto* kern = power_march_kernel<T, VW, KV, TS, D, NT>;
  if (lds > 64 * 1024) {                            // (once per instance: the default dynamic LDS limit is 64 KiB)
    static bool raised = false;
    if (!raised) { GLA_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); raised = true; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NT), lds, st,
                     pv, x, ldx, nrm, y, ldy, work, g.lines, g.nstrips, g.nlines, tm, (T*)nullptr, (int64_t)0);
  GLA_LAUNCH_CHECK();
  // second stage: one workgroup per (level, rhs), workgroup partials added in block order
  hipLaunchKernelGGL((col_dot_stage2<T>), dim3(2 * K), dim3(GLA_BLOCK), 0, st, work, g.nblocks, 2 * K, sumsq);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

// second stage of the Rayleigh tail: column k < K of the block partials -> bb_out[k] (sum b^2), column K + k -> nA_out[k]
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void tail_sums_kernel(const T* __restrict__ partial, int nblocks, int K,
                                                              T* __restrict__ bb_out, T* __restrict__ nA_out) {
  __shared__ T red[GLA_BLOCK];                   // one workgroup per column, fixed-order tree (as col_dot_stage2)
  const int tid = threadIdx.x, k = blockIdx.x;
  T acc = T(0);
  for (int b = tid; b < nblocks; b += GLA_BLOCK) acc = acc + partial[(int64_t)b * 2 * K + k];
  red[tid] = acc;
  __syncthreads();
  for (int st = GLA_BLOCK / 2; st > 0; st >>= 1) {
    if (tid < st) red[tid] = red[tid] + red[tid + st];
    __syncthreads();
  }
  if (tid == 0) (k < K ? bb_out[k] : nA_out[k - K]) = red[0];
}

template <typename T, int VW, int KV, int D>
int launch_power_tail(gla_csr* A, const T* y, int64_t ldy, const T* nrm, T* bout, int64_t ldb, T* b2out, int64_t ldb2,
                      T* nA_out, T* bb_out, T* work, int64_t work_bytes, hipStream_t st) {
  constexpr int NT = PowThreads<KV>::nt;
  constexpr int BP = NT / KV, K = KV * VW;
  static_assert(BP - 2 >= 1, "strip too narrow");
  PowK pv;
  pv.val = A->dia_val;
  pv.mask = A->dia_mask;
  pv.S = A->dia_S;
  pv.n_rows = A->n_rows; pv.n_cols = A->n_cols;
  const PowGeom g = pow_geom<KV, 1>(A);
  if ((int64_t)g.nblocks * 2 * K * (int64_t)sizeof(T) > work_bytes) return GLA_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(D == 5 ? 2 : 3) * (BP + 2) * KV * VW * sizeof(T);
  TileMap tm;
  tm.tile_row = nullptr; tm.tile_info = nullptr; tm.tile_list = nullptr;
  tm.n_tiles = g.nblocks;
  set_tile_mapping(tm);
  hipLaunchKernelGGL((power_march_kernel<T, VW, KV, 1, D, NT, true>), dim3((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), dim3(NT), lds, st,
                     pv, y, ldy, nrm, bout, ldb, work, g.lines, g.nstrips, g.nlines, tm, b2out, ldb2);
  GLA_LAUNCH_CHECK();
  hipLaunchKernelGGL((tail_sums_kernel<T>), dim3(2 * K), dim3(GLA_BLOCK), 0, st, work, g.nblocks, K, bb_out, nA_out);
  GLA_LAUNCH_CHECK();
  return GLA_OK;
}

template <typename T, int VW, int D>
int pow_tail_dispatch_kv(int kv, gla_csr* A, const T* y, int64_t ldy, const T* nrm, T* bout, int64_t ldb, T* b2out, int64_t ldb2,
                         T* nA_out, T* bb_out, T* work, int64_t work_bytes, hipStream_t st) {
  if constexpr (VW == 1) {
    return kv == 1 ? launch_power_tail<T, 1, 1, D>(A, y, ldy, nrm, bout, ldb, b2out, ldb2, nA_out, bb_out, work, work_bytes, st)
                   : GLA_ERR_UNSUPPORTED;
  } else {
    switch (kv) {
      case 1: return launch_power_tail<T, VW, 1, D>(A, y, ldy, nrm, bout, ldb, b2out, ldb2, nA_out, bb_out, work, work_bytes, st);
      case 2: return launch_power_tail<T, VW, 2, D>(A, y, ldy, nrm, bout, ldb, b2out, ldb2, nA_out, bb_out, work, work_bytes, st);
      case 4: return launch_power_tail<T, VW, 4, D>(A, y, ldy, nrm, bout, ldb, b2out, ldb2, nA_out, bb_out, work, work_bytes, st);
      case 8: return launch_power_tail<T, VW, 8, D>(A, y, ldy, nrm, bout, ldb, b2out, ldb2, nA_out, bb_out, work, work_bytes, st);
      default: return GLA_ERR_UNSUPPORTED;
    }
  }
}

// (levels per pass the register file carries at two workgroups per CU: 5 on the 5-point pattern, 3 on the 9-point one)
template <typename T, int VW, int KV, int D>
int pow_dispatch_ts(int ts, gla_csr* A, const T* x, int64_t ldx, const T* nrm, T* y, int64_t ldy, T* sumsq, T* work, hipStream_t st) {
  switch (ts) {
    case 1: return launch_power<T, VW, KV, 1, D>(A, x, ldx, nrm, y, ldy, sumsq, work, st);
    case 2: return launch_power<T, VW, KV, 2, D>(A, x, ldx, nrm, y, ldy, sumsq, work, st);
    case 3: return launch_power<T, VW, KV, 3, D>(A, x, ldx, nrm, y, ldy, sumsq, work, st);
    case 4: if constexpr (D == 5) return launch_power<T, VW, KV, 4, D>(A, x, ldx, nrm, y, ldy, sumsq, work, st); else return GLA_ERR_UNSUPPORTED;
    case 5: if constexpr (D == 5) return launch_power<T, VW, KV, 5, D>(A, x, ldx, nrm, y, ldy, sumsq, work, st); else return GLA_ERR_UNSUPPORTED;
    default: return GLA_ERR_UNSUPPORTED;
  }
}

template <int KV>
int pow_blocks_ts(int ts, const gla_csr* A) {
  switch (ts) {
    case 1: return pow_geom<KV, 1>(A).nblocks;
    case 2: return pow_geom<KV, 2>(A).nblocks;
    case 3: return pow_geom<KV, 3>(A).nblocks;
    case 4: return pow_geom<KV, 4>(A).nblocks;
    case 5: return pow_geom<KV, 5>(A).nblocks;
    default: return -1;
  }
}

int pow_blocks(int kv, int ts, const gla_csr* A) {
  switch (kv) {
    case 1: return pow_blocks_ts<1>(ts, A);
    case 2: return pow_blocks_ts<2>(ts, A);
    case 4: return pow_blocks_ts<4>(ts, A);
    case 8: return pow_blocks_ts<8>(ts, A);
    default: return -1;
  }
}

template <typename T, int VW, int D>
int pow_dispatch_kv(int kv, int ts, gla_csr* A, const T* x, int64_t ldx, const T* nrm, T* y, int64_t ldy, T* sumsq, T* work,
                    hipStream_t st) {
  switch (kv) {
    case 1: return pow_dispatch_ts<T, VW, 1, D>(ts, A, x, ldx, nrm, y, ldy, sumsq, work, st);
    case 2: if constexpr (VW > 1) return pow_dispatch_ts<T, VW, 2, D>(ts, A, x, ldx, nrm, y, ldy, sumsq, work, st); else return GLA_ERR_UNSUPPORTED;
    case 4: if constexpr (VW > 1) return pow_dispatch_ts<T, VW, 4, D>(ts, A, x, ldx, nrm, y, ldy, sumsq, work, st); else return GLA_ERR_UNSUPPORTED;
    case 8: if constexpr (VW > 1) return pow_dispatch_ts<T, VW, 8, D>(ts, A, x, ldx, nrm, y, ldy, sumsq, work, st); else return GLA_ERR_UNSUPPORTED;
    default: return GLA_ERR_UNSUPPORTED;
  }
}

bool pow_plan_ok(const gla_csr* A) {
  return g_tuning.jacobi_march && g_tuning.dia && (gla_dia_canon5(A) || gla_dia_canon9(A)) && A->dia_val && A->dia_nowrap &&
         A->dia_row_lo == 0 && A->dia_row_hi == A->n_rows && A->n_rows == A->n_cols && A->n_rows > 0;
}

// lanes per point and elements per lane for K right-hand sides (0: no kernel)
bool pow_shape(int dtype, int K, int& vw, int& kv) {
  const int w16 = dtype == GLA_F64 ? 2 : 4;
  if (K == 1) { vw = 1; kv = 1; return true; }
  if (K % w16 != 0) return false;
  vw = w16; kv = K / w16;
  return kv == 1 || kv == 2 || kv == 4 || kv == 8;
}

}  // namespace

extern "C" {

int gla_power_fused_max_iters(gla_csr_t A, int dtype, int K) {
  int vw, kv;
  if (!A || !pow_plan_ok(A) || (dtype != GLA_F32 && dtype != GLA_F64) || !pow_shape(dtype, K, vw, kv)) return 0;
  // (small problems keep one launch per iteration: a pass has a ramp of 2 (iters - 1) lines per block; same switch and
  // threshold as the temporally blocked Jacobi passes)
  if (g_tuning.jacobi_fuse != 1 || A->n_rows * (int64_t)K < g_tuning.jacobi_min_rows) return 0;
  return gla_dia_canon5(A) ? 5 : 3;
}

int64_t gla_power_fused_work_bytes(gla_csr_t A, int dtype, int K, int iters) {
  int vw, kv;
  if (!A || K < 1 || iters < 1 || iters > gla_power_fused_max_iters(A, dtype, K) || !pow_shape(dtype, K, vw, kv)) return 0;
  const int nb = pow_blocks(kv, iters, A);
  return nb < 0 ? 0 : (int64_t)nb * 2 * K * (dtype == GLA_F64 ? 8 : 4);
}

}  // extern "C"

// The Rayleigh tail of PowerMethodGNN on canonical stencil plans as a one-level pass of the march kernel: B = Y ./ nrm (as
// gla_power_step divides), B2 = B .* B, nA_out[k] = sum_i B[i,k] (A B)[i,k], bb_out[k] = sum_i B[i,k]^2.  GLA_ERR_UNSUPPORTED when the
// plan / K / alignment / work area does not qualify: gla_power_rayleigh then runs its blocked-SpMM tail.
int gla_power_tail_pass(gla_csr* A, int dtype, int K, const void* Y, int64_t ldy, const void* nrm, void* Bout, int64_t ldb,
                        void* B2out, int64_t ldb2, void* nA_out, void* bb_out, void* work, int64_t work_bytes, hipStream_t st) {
  int vw, kv;
  if (!pow_plan_ok(A) || !pow_shape(dtype, K, vw, kv) || gla_power_fused_max_iters(A, dtype, K) < 1) return GLA_ERR_UNSUPPORTED;
  if (vw > 1 && (!al16(Y) || !al16(Bout) || !al16(B2out) || ldy % vw != 0 || ldb % vw != 0 || ldb2 % vw != 0)) return GLA_ERR_UNSUPPORTED;
  const bool c5 = gla_dia_canon5(A);
  if (dtype == GLA_F64) {
    typedef double T;
    if (vw == 1) return c5 ? pow_tail_dispatch_kv<T, 1, 5>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st)
                           : pow_tail_dispatch_kv<T, 1, 9>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st);
    return c5 ? pow_tail_dispatch_kv<T, 2, 5>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st)
              : pow_tail_dispatch_kv<T, 2, 9>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st);
  }
  typedef float T;
  if (vw == 1) return c5 ? pow_tail_dispatch_kv<T, 1, 5>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st)
                         : pow_tail_dispatch_kv<T, 1, 9>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st);
  return c5 ? pow_tail_dispatch_kv<T, 4, 5>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st)
            : pow_tail_dispatch_kv<T, 4, 9>(kv, A, (const T*)Y, ldy, (const T*)nrm, (T*)Bout, ldb, (T*)B2out, ldb2, (T*)nA_out, (T*)bb_out, (T*)work, work_bytes, st);
}

extern "C" {

// `iters` power iterations in one pass: Y = A^iters (X ./ nrm_in) (nrm_in == nullptr: X as given), sumsq_out[k] =
// sum_i (A^(iters-1) (X ./ nrm_in))[i, k]^2, sumsq_out[K + k] = sum_i Y[i, k]^2.  X and Y must not alias.  work:
// gla_power_fused_work_bytes.
int gla_power_fused(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, const void* nrm_in, void* Y, int64_t ldy,
                    void* sumsq_out, void* work, int iters, gla_stream_t stream) {
  const int rc = check_plan(A, dtype, K);
  if (rc) return rc;
  if (!X || !Y || X == Y || !sumsq_out || !work || ldx < K || ldy < K || iters < 1) return GLA_ERR_INVALID_ARG;
  int vw, kv;
  if (!pow_plan_ok(A) || !pow_shape(dtype, K, vw, kv) || iters > gla_power_fused_max_iters(A, dtype, K)) return GLA_ERR_UNSUPPORTED;
  if (vw > 1 && (!al16(X) || !al16(Y) || ldx % vw != 0 || ldy % vw != 0)) return GLA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool c5 = gla_dia_canon5(A);
  if (dtype == GLA_F64) {
    typedef double T;
    if (vw == 1) return c5 ? pow_dispatch_kv<T, 1, 5>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st)
                           : pow_dispatch_kv<T, 1, 9>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st);
    return c5 ? pow_dispatch_kv<T, 2, 5>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st)
              : pow_dispatch_kv<T, 2, 9>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st);
  }
  typedef float T;
  if (vw == 1) return c5 ? pow_dispatch_kv<T, 1, 5>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st)
                         : pow_dispatch_kv<T, 1, 9>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st);
  return c5 ? pow_dispatch_kv<T, 4, 5>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st)
            : pow_dispatch_kv<T, 4, 9>(kv, iters, A, (const T*)X, ldx, (const T*)nrm_in, (T*)Y, ldy, (T*)sumsq_out, (T*)work, st);
}

}  // extern "C"
