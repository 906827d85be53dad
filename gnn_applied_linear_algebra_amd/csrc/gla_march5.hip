// libgla_hip.so -- the temporally blocked Jacobi pass for canonical 5-point plans (gfx950 / MI355X): the headline kernel.
//
// JacobiGNN.forward (JacobiGNN.py:138-148) returns only the last iterate, so TS sweeps are taken in ONE pass over the matrix:
// a workgroup's 256 lanes cover a strip of 256 grid points and march down the grid lines; every line step loads one line of
// x and of the rows' coefficients (coalesced 2 KiB segments, one step ahead of their use) and advances every level
// t = 1 .. TS by one line (level t works on line jl - (t - 1)), the last level going to HBM.  The j -/+ 1 neighbours and the
// centre are the lane's OWN values of the last three line steps (registers h[t][0..2]); LDS carries the i -/+ 1 neighbours
// (one padded line per level and line parity; a level reads what the level below wrote a step EARLIER: one barrier per line
// step); the line loop is unrolled over the TS + 1 coefficient sets so that sets and LDS slots have static names.
//
// Round 3 (tools/micro/march5_exp.hip holds the experiments, profiles/r03/ the numbers):
//   * Strips are aligned with the grid: the first strip starts AT the first point of a line and the last one ends AT the last
//     point, so no lane lies outside the grid (S >= 256).  A row whose diagonal mask is exactly "every neighbour that exists"
//     takes the unmasked sum: its missing neighbours are read as +0 (LDS pads, lines beyond the grid), their coefficients are
//     stored zeros, and acc + (0 x 0) == acc bit for bit (acc starts at +0; a sum is -0 only if both operands are).  Before,
//     the wave holding a line's first / last point ran the masked sum: the edge strips were 25 % slower than the others and
//     set the kernel time.
//   * The division by A_ii (JacobiGNN.py:119) goes through RN(1 / A_ii) -- one IEEE division per row and PASS -- and one
//     Markstein correction per level: q0 = RN(num * rc), rem = num - q0 * d (exact in an FMA), q = RN(q0 + rem * rc) = RN(num / d)
//     whenever d's significand is not all ones and nothing leaves the normal range.  That condition is checked OFF the
//     dependent chain (divisor in [2^-500, 2^500] when the set is first used, numerator in (2^-450, 2^450) per level); a line
//     step in which any lane fails is redone with the generic levels (masked in-order sum, IEEE division) before the barrier --
//     levels only assign h[t][2], their LDS line and their output row, so a step is idempotent.  An IEEE fp64 division costs
//     ~25 ns of SIMD time per wave (tools/micro/f64lat.hip), 7 of them per line step were a third of the VALU time.
//   * The SIMD arbiter serves the OLDER wave first: of the two workgroups a CU holds, the one dispatched second progressed
//     ~15 % slower and set the kernel time.  The two halves of an XCD's workgroups now alternate s_setprio every line step in
//     antiphase.
//   * The i -/+ 1 neighbours of level t + 2 are read from LDS before level t computes.
// Per row the arithmetic is that of spmv_dia_kernel + EpiJacobi (in-order masked sum over the diagonals, then
// x + (w * (b - sum)) / A_ii, product and sum rounded separately): bit-identical to TS single sweeps.
//
// Round 4, SYM: on a plan whose value set is symmetric (A_ij == A_ji bit for bit along both diagonal pairs, checked at
// gla_csr_set_values; option "sym") only the centre, east and south diagonals are read: a row's west coefficient is the east
// coefficient of the point before it (same cache lines as the east load), its north coefficient the south coefficient of the
// line above (the lane's own previous set).  24 instead of 40 bytes of values per row; the pass is bound by its bytes
// (profiles/r04/march5v2_what_bounds_the_pass.txt), so it takes a fifth less time.  Same values, same order: bit-identical.
//
// Round 4, the A_ii stream: JacobiGNN divides by the vertex attribute A_ii (JacobiGNN.py:104-119), which in every use the
// reference makes of it IS the matrix diagonal.  dv.dg_mode says where a pass takes it from: 0 = the array `diag` (stride
// dv.ldd: contiguous, or column 0 of the AoS vertex attributes), 1 = the plan's own centre diagonal (callers that pass
// diag == NULL), 2 = whichever the device flag dv.dg_flag says: the AoS first pass of gla_jacobi_sweeps_aos compares the two bit
// for bit, row by row, and raises the flag on a difference (it stores the call's number: flags are never reset); its later passes then either skip the A_ii stream
// (8 of their 58 bytes per row) or read column 0 of the AoS rows.  Branch-free: the load's ADDRESS is selected (the centre
// value is in the L1 from the row's own coefficient load), so the prefetch pattern is the same either way.
//
// MODE 1 (first pass of gla_jacobi_sweeps_aos): A_ii, b and x^(k) are columns 0, 1, 2 of the reference's row-major vertex
// attribute tensor (JacobiGNN.py:104-106); the lane that owns a row leaves contiguous copies of its A_ii and b in diag_out /
// b_out for the later passes.  MODE 2 (gla_jacobi_fused_residual): one more level forms r = b - A x^(k+TS) (-> b_out).
#include "gla_device.h"
#include <hip/hip_ext.h>

#include <type_traits>
#include <utility>

namespace {

template <int N> using IC = std::integral_constant<int, N>;
template <int N> __device__ __forceinline__ constexpr int ic_val(IC<N>) { return N; }
__device__ __forceinline__ constexpr int ic_val(int v) { return v; }

struct M5View {
  const void* val;            // [5][n_rows] diagonal-major: (-1,0) (0,-1) (0,0) (0,1) (1,0)
  const uint16_t* mask;       // [n_rows]
  int64_t S, n_rows, nlines;  // grid line length, rows (= columns), grid lines (the last one may be ragged)
  int32_t edge, nlb;          // multi-GPU interior / boundary split: height of the first / last line block (0: uniform), line blocks
#ifdef GLA_M5_VARIANTS
  unsigned long long* stamps; // experiments: per workgroup {start, end (s_memrealtime, 100 MHz), tile, HW_ID, XCC_ID, line steps}
#endif
  int32_t prio_hi;            // 1: the edge launch of a split pass (part 2) -- its waves run at raised priority throughout (below)
  int64_t aos_off, aos_rows;  // MODE 1: plan row r is row r - aos_off of an AoS tensor of aos_rows rows
  int64_t aos_wlo, aos_whi;   // MODE 1: grid lines [aos_wlo, aos_whi) get their contiguous copies of A_ii and b written by this launch
  int64_t ldd;                // MODE 0 / 2: stride of `diag` (1, or the row stride of the AoS tensor whose column 0 it points at)
  int32_t dg_mode;            // MODE 0 / 2: A_ii from 0 `diag`, 1 the centre diagonal, 2 `diag` or the centre diagonal as *dg_flag says
  int* dg_flag;               // MODE 1: set to dg_epoch when an owned row's A_ii differs from its centre coefficient; dg_mode 2: read
  int32_t dg_epoch;           // this call's number (> 0): a flag word equal to it means "raised in THIS call" -- nobody has to reset it
};

// line block lb: lines [J0, J0 + L) of the grid
__device__ __forceinline__ void m5_block(int lb, int lines, int edge, int nlb, int64_t nlines, int64_t& J0, int& L) {
  if (edge == 0) { J0 = (int64_t)lb * lines; L = (int)min((int64_t)lines, nlines - J0); }
  else if (lb == 0) { J0 = 0; L = edge; }
  else if (lb == nlb - 1) { J0 = nlines - edge; L = edge; }
  else { J0 = edge + (int64_t)(lb - 1) * lines; L = (int)min((int64_t)lines, nlines - edge - J0); }
}

template <typename T> struct M5Lim;
template <> struct M5Lim<double> {
  static constexpr double dlo = 0x1p-500, dhi = 0x1p500, nlo = 0x1p-450, nhi = 0x1p450;
  static __device__ __forceinline__ bool all_ones(double v) {
    return (__builtin_bit_cast(unsigned long long, v) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
  }
};
template <> struct M5Lim<float> {
  static constexpr float dlo = 0x1p-60f, dhi = 0x1p60f, nlo = 0x1p-50f, nhi = 0x1p50f;
  static __device__ __forceinline__ bool all_ones(float v) { return (__builtin_bit_cast(unsigned, v) & 0x7FFFFFu) == 0x7FFFFFu; }
};

template <typename T>
struct M5Coef {               // one row's matrix data
  T v[5], b, dg;
  T xv;                       // PF > 1: x^(k) of the NEXT line rides along (same prefetch distance as the coefficients)
  int m;                      // diagonal mask exactly as loaded; read once, at the set's first use (a line step after the load:
};                            // no arithmetic on it earlier, the load must stay in flight)

constexpr int M5_PLAIN = 0, M5_AOS = 1, M5_RES = 2;

// The pass (one workgroup's line block).  DGC: A_ii IS the row's centre coefficient -- neither loaded nor kept (12 registers per
// coefficient set instead of 17, one load less per line step).
// DBG (GLA_M5_VARIANTS builds only; results are WRONG with any bit set -- timing experiments): 1 no barrier, 2 no output store,
// 4 no s_setprio, 8 no levels (the memory side alone: loads, LDS line of level 0, the store stream)
template <typename T, int TS, int MODE, bool RCP, int HD, bool STATIC_IDX, bool SYM, bool FM, bool DGC, int PF = 1, int DBG = 0>
__device__ __forceinline__ void m5_body(const M5View& dv, const T* __restrict__ diag, const T* __restrict__ bvec,
                                        const T* __restrict__ x, T* __restrict__ xo, T w, int lines, int nstrips, const TileMap& tm,
                                        int tile_base0, int tile_split, int tile_base1, int64_t ldva, T* __restrict__ diag_out,
                                        T* __restrict__ b_out) {
  constexpr bool AOS = MODE == M5_AOS, RES = MODE == M5_RES;
  constexpr int BW = GLA_BLOCK, LW = BW + 2;      // a line in LDS: lane tid at [1 + tid], one pad either side (stays +0)
  constexpr int LV = TS + (RES ? 1 : 0);          // levels: TS sweeps; RES: + one level that forms r = b - A x^(k+TS)
  constexpr int NC = LV + PF, U = NC;             // coefficient sets in flight (lines jl + 1 .. jl - LV + 1); unroll: static set index
  typedef M5Lim<T> Lim;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* buf = reinterpret_cast<T*>(smem_raw);        // [LV][2][LW]: buf[t][p] = the latest line of parity p of x^(k+t)
  const int slot = map_tile(tm);
  if (slot < 0) return;
  const int blk = slot < tile_split ? tile_base0 + slot : tile_base1 + (slot - tile_split);
  // the arbiter favours the older wave: the two halves of an XCD's workgroups alternate priority in antiphase (see header)
  const int late = ((blockIdx.x / GLA_XCDS) >> 5) & 1;
  constexpr int halo = LV, bi_eff = BW - 2 * halo;
  const int lb = blk / nstrips, strip = blk - lb * nstrips;
  // strip s: interior strips own the middle bi_eff points; the first starts at point 0, the last ends at point S - 1
  const int64_t P0 = max((int64_t)0, min((int64_t)strip * bi_eff, dv.S - BW));
  const int64_t own_lo = strip == 0 ? 0 : (int64_t)strip * bi_eff + halo;
  const int64_t own_hi = strip == nstrips - 1 ? dv.S : (int64_t)strip * bi_eff + BW - halo;
  int64_t J0;
  int L;
  m5_block(lb, lines, dv.edge, dv.nlb, dv.nlines, J0, L);
  const int tid = threadIdx.x;
  const T* __restrict__ val = static_cast<const T*>(dv.val);
  bool dg_raised = false;                          // (wave-uniform: this wave has raised the A_ii flag already)
  bool dg_center = false;                          // (kernel-uniform)
  if constexpr (!AOS && !DGC) dg_center = dv.dg_mode == 1 || (dv.dg_mode == 2 && __builtin_amdgcn_readfirstlane(*dv.dg_flag) != dv.dg_epoch);
  // A_ii = dg_base[row * dg_ld]: the caller's array, or the plan's centre diagonal (the row's own v[2]: an L1 hit, no HBM bytes)
  const T* const dg_base = dg_center || !diag ? val + 2 * dv.n_rows : diag;
  const uint32_t dg_ld = dg_center || !diag ? 1u : (uint32_t)dv.ldd;
  const int64_t gi = P0 + tid;
  const bool in_i = gi < dv.S;                     // (false only on grids narrower than a strip)
  const bool owner = gi >= own_lo && gi < own_hi;
  const int em_i = 31 & ~(gi == 0 ? 2 : 0) & ~(gi == dv.S - 1 ? 8 : 0);
  const int64_t nfull = dv.n_rows / dv.S;          // whole grid lines
  const bool in_tail = gi < dv.n_rows - nfull * dv.S;          // this lane's point exists on the ragged last line
  // block-relative lines [jlo, jhi) are whole grid lines
  const int jlo = (int)max((int64_t)-64, -J0), jhi = (int)min((int64_t)L + 64, nfull - J0);
  const bool wave_in = __all(in_i) != 0;

  if (dv.prio_hi) __builtin_amdgcn_s_setprio(3);
  for (int i = tid; i < LV * 2 * LW; i += BW) buf[i] = T(0);   // (the pads stay +0: what lies beyond a line's ends)
  __syncthreads();

  // (DBG & 16, variants builds, timing only -- the results are WRONG: every array addressed as if line j of block lb were STORED at
  // line slot (j - J0) * nlb + lb, so that what all the line blocks touch in one step is contiguous; profiles/r05/march5_memory_side.txt)
  auto perm_row = [&](int jl) -> int64_t {
    int64_t sl;
    if (jl < 0) sl = (int64_t)(lines + jl) * dv.nlb + max(lb - 1, 0);
    else if (jl >= L) sl = (int64_t)(jl - L) * dv.nlb + min(lb + 1, dv.nlb - 1);
    else sl = (int64_t)jl * dv.nlb + lb;
    return min(max(sl * dv.S + gi, (int64_t)0), dv.n_rows - 1);
  };
  auto row_clamped = [&](int jl) -> int64_t {
    if constexpr ((DBG & 16) != 0) return perm_row(jl);
    else return min(max((J0 + jl) * dv.S + gi, (int64_t)0), dv.n_rows - 1);
  };
  auto live_at = [&](int jl) -> bool {             // the row (line J0 + jl, this lane's point) exists (generic path, level 0)
    return in_i & (J0 + jl >= 0) & ((J0 + jl) * dv.S + gi < dv.n_rows);
  };
  auto x0_load = [&](int jl) -> T {                // x^(k) at the clamped row; NOT touched until the next step
    const int64_t rc = row_clamped(jl);
    if constexpr (AOS) return x[min(max(rc - dv.aos_off, (int64_t)0), dv.aos_rows - 1) * ldva + 2];
    else return x[rc];
  };
  auto coef_at = [&](int jl) -> M5Coef<T> {
    const int64_t rc = row_clamped(jl);
    M5Coef<T> c;
    if constexpr (SYM) {
      c.v[2] = val[2 * dv.n_rows + rc];
      c.v[3] = val[3 * dv.n_rows + rc];
      c.v[4] = val[4 * dv.n_rows + rc];
      const T wv = val[3 * dv.n_rows + max(rc - 1, (int64_t)0)];      // west(r) = east(r - 1); a line's first point has none
      c.v[1] = gi == 0 ? T(0) : wv;
      c.v[0] = T(0);                                                  // (not kept: north(line l) = south(line l - 1), the set before)
      if (J0 + jl < 0) c.v[4] = T(0);                                 // (a line above the grid: the grid's first line has no north)
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) c.v[d] = val[(int64_t)d * dv.n_rows + rc];
    }
    if constexpr (AOS) {
      const int64_t ra = min(max(rc - dv.aos_off, (int64_t)0), dv.aos_rows - 1);
      c.dg = x[ra * ldva];
      c.b = x[ra * ldva + 1];
    } else {
      c.b = bvec[rc];
      if constexpr (DGC) c.dg = T(0);                // (A_ii IS the centre coefficient, known at launch: neither loaded nor kept)
      else c.dg = dg_base[(int64_t)((uint32_t)rc * (uint64_t)dg_ld)];      // (base and stride chosen ONCE, below: see header)
    }
    if constexpr (PF > 1) c.xv = x0_load(jl + 1); else c.xv = T(0);
    if constexpr (FM) c.m = 0;                      // (every row has the mask of its position: gla_csr.dia_full; not read)
    else c.m = (int)dv.mask[rc];
    return c;
  };

  constexpr int j_first = -(LV - 1);
  const int j_last = L - 1 + (LV - 1);
  T h[LV][3];                                      // h[t]: x^(k+t) of this lane on the last three lines level t produced
#pragma unroll
  for (int t = 0; t < LV; ++t) h[t][0] = h[t][1] = h[t][2] = T(0);
  {
    const T a = x0_load(j_first - 1), b0 = x0_load(j_first);
    h[0][1] = live_at(j_first - 1) ? a : T(0);
    h[0][2] = live_at(j_first) ? b0 : T(0);
  }
  buf[((j_first + 64) & 1) * LW + 1 + tid] = h[0][2];
  T xpre = PF > 1 ? T(0) : x0_load(j_first + 1);
  M5Coef<T> c[NC];                                 // set of line j_first + s lives in c[s % NC]
  T crc[NC];                                       // RN(1 / A_ii) of the set's rows (1 where the fast division is off anyway)
  T s_ev = T(0);                                   // SYM: south coefficient of the line whose set has just left the ring
  bool cfast[NC];                                  // wave-uniform: a whole grid line, every row with the expected mask and a safe divisor
#pragma unroll
  for (int s = 0; s < PF; ++s) c[s] = coef_at(j_first + s);
#pragma unroll
  for (int s = PF; s < NC; ++s) c[s] = c[0];
  if constexpr (SYM)                               // (the set "before" the first: its south)
    c[NC - 1].v[4] = J0 + j_first - 1 < 0 ? T(0) : val[4 * dv.n_rows + row_clamped(j_first - 1)];
#pragma unroll
  for (int s = 0; s < NC; ++s) { crc[s] = T(1); cfast[s] = false; }

  // one line step: x^(k) line jl + 1 arrives, levels 1 .. nact advance by one line (level t: line jl - (t - 1)).
  // u = (jl - j_first) mod NC names the coefficient sets (folds to constants after unrolling); the LDS slot of a line is its
  // parity: two lane addresses per step, the rest are immediate offsets.
  // u and nact are either plain ints that fold once the loops below are unrolled, or integral_constant arguments
  // (STATIC_IDX): every index into c[], crc[], cfast[] must be static -- a dynamically indexed set array lives in scratch
  // memory -- and which of the two forms the register allocator handles without spilling depends on the instance (M5Cfg)
  auto do_step = [&](const int jl, auto u_c, auto nact_c) {
    const int u = ic_val(u_c), nact = ic_val(nact_c);
    if constexpr (!(DBG & 1)) __syncthreads();
    const int pj = (jl + 64) & 1;
    // (the edge launch of a split pass keeps the raised priority it took at kernel start: its 2 x nstrips workgroups share their
    //  SIMDs with OLDER interior waves, which the arbiter serves first -- 94 us beside the interior launch against 32 us alone
    //  (tools/time_edge_blocks.py), and the next pass's exchange waits for exactly these blocks)
    if constexpr (!(DBG & 4)) { if (!dv.prio_hi) { if (pj ^ late) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
    T* const row_e = buf + pj * LW + tid;          // lines of the parity of jl:     [0] left, [1] own, [2] right
    T* const row_o = buf + (1 - pj) * LW + tid;    // lines of the other parity
    auto lds_row = [&](int t) -> T* { return ((t - 1) % 2 == 0 ? row_e : row_o) + (t - 1) * 2 * LW; };   // level t - 1, line of level t
    T nl[LV], nr[LV];
#pragma unroll
    for (int t = 1; t <= LV && t <= HD; ++t)
      if (t <= nact) { nl[t - 1] = lds_row(t)[0]; nr[t - 1] = lds_row(t)[2]; }
#pragma unroll
    for (int t = 0; t < LV; ++t) { h[t][0] = h[t][1]; h[t][1] = h[t][2]; }      // (levels only assign h[t][2]: a step is idempotent)
    const bool whole_next = wave_in & (jl + 1 >= jlo) & (jl + 1 < jhi);          // wave-uniform: line jl + 1 exists for every lane
    // (bitwise on purpose: hipcc 7.0 dropped the exec-masked move of `uniform || per_lane ? x : 0` in unrolled copies of
    // jacobi_marchk_kernel's line step -- profiles/r03/marchk_miscompile.txt; a plain v_cndmask has no such region)
    h[0][2] = (whole_next | live_at(jl + 1)) ? (PF > 1 ? c[u % NC].xv : xpre) : T(0);
    row_o[1] = h[0][2];                            // level 0, line jl + 1
    if constexpr (PF == 1 && !(DBG & 32)) xpre = x0_load(jl + 2);
    if constexpr (SYM) s_ev = c[(u + PF) % NC].v[4];   // the set about to be replaced is line jl - LV: its south is the north of the last level's line
    if constexpr (!(DBG & 32)) c[(u + PF) % NC] = coef_at(jl + PF);      // (DBG & 32: the compute side alone -- no global loads in the line step)
    {                                              // first use of the set of line jl (loaded a step ago)
      const M5Coef<T>& cu = c[u % NC];
      const int64_t gj = J0 + jl;
      // the mask a row at this position has when every neighbour that exists is coupled
      const bool has_down = gj + 1 < nfull || (gj + 1 == nfull && in_tail);
      const int em = em_i & ~(gj == 0 ? 1 : 0) & ~(has_down ? 0 : 16);
      bool dok = true;
      if constexpr (RCP) {
        const T cudg = DGC ? cu.v[2] : cu.dg;
        const T ad = __builtin_fabs(cudg);
        dok = ad >= Lim::dlo && ad <= Lim::dhi && !Lim::all_ones(cudg);
        crc[u % NC] = T(1) / (dok ? cudg : T(1));
      }
      if constexpr (FM) { (void)em; cfast[u % NC] = wave_in && jl >= jlo && jl < jhi && __all(dok) != 0; }
      else cfast[u % NC] = wave_in && jl >= jlo && jl < jhi && __all((cu.m & 0x1f) == em && dok) != 0;
      if constexpr (AOS) {                         // the row's owner leaves contiguous copies of A_ii and b
        bool dg_mis = false;
        if (jl >= 0 && jl < L && owner && live_at(jl) && gj >= dv.aos_wlo && gj < dv.aos_whi) {
          const int64_t r = gj * dv.S + gi;
          if (diag_out) diag_out[r] = cu.dg;
          b_out[r] = cu.b;
          if (dv.dg_flag) {                          // does A_ii repeat the matrix diagonal?  (bit for bit; the later passes ask)
            typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned>::type U;
            dg_mis = __builtin_bit_cast(U, cu.dg) != __builtin_bit_cast(U, cu.v[2]);
          }
        }
        if (!dg_raised && __any(dg_mis)) {           // (one atomic per WAVE and launch at most, whatever the number of rows that differ)
          if ((tid & 63) == 0) atomicExch(dv.dg_flag, dv.dg_epoch);
          dg_raised = true;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((DBG & 8) != 0) {                  // the memory side alone: every loaded value is consumed, one store per row
      const M5Coef<T>& cc = c[u % NC];
      const T sum_ = (((cc.v[1] + cc.v[2]) + (cc.v[3] + cc.v[4])) + cc.b) + (h[0][1] + row_o[0]);
      const int line = jl - (LV - 1);
      if (nact == LV && owner && line >= 0 && line < L) xo[(J0 + line) * dv.S + gi] = sum_;
      return;
    }
    bool fast = true;
#pragma unroll
    for (int t = 1; t <= LV; ++t)
      if (t <= nact) fast = fast && cfast[((u - (t - 1)) % NC + NC) % NC];
    bool bad = false;                              // per lane: a numerator the Markstein step is not proven for
    if (fast) {
#pragma unroll
      for (int t = 1; t <= LV; ++t) {
        if (t <= nact) {
          T* const row = lds_row(t);
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          if constexpr (HD > 0 && HD < LV) {
            if (t + HD <= LV && t + HD <= nact) { nl[t + HD - 1] = lds_row(t + HD)[0]; nr[t + HD - 1] = lds_row(t + HD)[2]; }
          }
          T left, right;
          if constexpr (HD > 0) { left = nl[t - 1]; right = nr[t - 1]; }
          else { left = row[0]; right = row[2]; }
          const M5Coef<T>& cc = c[ci];
          const T vnorth = !SYM ? cc.v[0] : (t == LV ? s_ev : c[(ci + NC - 1) % NC].v[4]);
          T acc = fma_t(vnorth, h[t - 1][0], T(0));   // == +0 + RN(v x) bit for bit (one rounding of an exact product; -0 + +0 = +0)
          acc = acc + cc.v[1] * left;
          acc = acc + cc.v[2] * h[t - 1][1];
          acc = acc + cc.v[3] * right;
          acc = acc + cc.v[4] * h[t - 1][2];
          const int line = jl - (t - 1);
          if (RES && t == LV) {                    // r = b - A x^(k+TS)   (GNNResidual.py:97-118, as EpiResidual)
            if (owner) b_out[(J0 + line) * dv.S + gi] = cc.b - acc;
          } else {
            const T num = w * (cc.b - acc);
            T xn;
            if constexpr (RCP) {
#ifdef GLA_CONTRACTED_FP
              // (the opt-in "contracted_fp" BUILD VARIANT, never the default and never bit-exact: the row sum above contracts into
              //  FMAs under -ffp-contract=fast and the division becomes ONE multiply-add with RN(1 / A_ii): what the bit-exact
              //  formulation costs is measured with it -- bench.py side record `contracted_fp`, parity at 1e-12 in the GPU suite)
              xn = fma_t(num, crc[ci], h[t - 1][1]);
#else
              const T an = __builtin_fabs(num);
              bad = bad || !(an > Lim::nlo && an < Lim::nhi);
              const T q0 = num * crc[ci];
              const T rem = fma_t(-q0, DGC ? cc.v[2] : cc.dg, num);
              xn = h[t - 1][1] + fma_t(rem, crc[ci], q0);
#endif
            } else {
              xn = h[t - 1][1] + num / (DGC ? cc.v[2] : cc.dg);
            }
            if (t < LV) { h[t][2] = xn; row[2 * LW + 1] = xn; }     // level t, same line (same parity)
            if (t == TS) { if ((DBG & 2) ? (owner && xn == T(123.456)) : (owner && (!RES || (line >= 0 && line < L)))) xo[(DBG & 16) ? perm_row(line) : (J0 + line) * dv.S + gi] = xn; }
          }
          __builtin_amdgcn_sched_barrier(0);       // levels stay in program order: the scheduler's own interleaving of seven
        }                                          // levels costs more registers than the two waves per SIMD leave
      }
    }
    if (!fast || __any(bad)) {                     // generic levels: masked in-order sum, IEEE division, +0 outside the matrix
      asm volatile("" ::: "memory");               // (keeps this path's LDS reads from being merged with the fast path's above the branch)
#pragma unroll
      for (int t = 1; t <= LV; ++t) {
        if (t <= nact) {
          T* const row = lds_row(t);
          const int ci = ((u - (t - 1)) % NC + NC) % NC;
          const int line = jl - (t - 1);
          const M5Coef<T>& cc = c[ci];
          const bool lv = live_at(line);
          const T xs[5] = {h[t - 1][0], row[0], h[t - 1][1], row[2], h[t - 1][2]};
          const int m = lv ? (int)dv.mask[row_clamped(line)] : 0;      // (not kept per set: this path is rare)
          T acc = T(0);
          const T vn_g = !SYM ? cc.v[0] : (t == LV ? s_ev : c[(ci + NC - 1) % NC].v[4]);
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            const T nxt = acc + (d == 0 ? vn_g : cc.v[d]) * xs[d];
            acc = ((m >> d) & 1) ? nxt : acc;
          }
          if (RES && t == LV) {
            if (lv && owner) b_out[(J0 + line) * dv.S + gi] = cc.b - acc;
          } else {
            T xn = h[t - 1][1] + (w * (cc.b - acc)) / (DGC ? cc.v[2] : cc.dg);
            xn = lv ? xn : T(0);
            if (t < LV) { h[t][2] = xn; row[2 * LW + 1] = xn; }
            if (t == TS) { if (lv && owner && (!RES || (line >= 0 && line < L))) xo[(J0 + line) * dv.S + gi] = xn; }
          }
        }
      }
    }
  };
  constexpr int u0 = (2 * (LV - 1)) % U;
  if constexpr (STATIC_IDX) {
    // ramp-up: step s = jl - j_first < 2 (LV - 1) has levels 1 .. (s + 2) / 2 active (x^(k+t) starts on line -(LV - t))
    auto ramp = [&](auto self, auto s_c) {
      constexpr int sidx = decltype(s_c)::value;
      if constexpr (sidx < 2 * (LV - 1)) {
        do_step(j_first + sidx, IC<sidx % U>{}, IC<(sidx + 2) / 2>{});
        self(self, IC<sidx + 1>{});
      }
    };
    ramp(ramp, IC<0>{});
    // steady state: lines 0 .. L - 1 leave the last level, one per step
    auto round = [&](auto self, auto q_c, const int jl0) -> bool {          // U steps; false: the block is done
      constexpr int q = decltype(q_c)::value;
      if constexpr (q < U) {
        if (jl0 + q > j_last) return false;
        do_step(jl0 + q, IC<(u0 + q) % U>{}, IC<LV>{});
        return self(self, IC<q + 1>{}, jl0);
      } else {
        return true;
      }
    };
#pragma unroll 1
    for (int jl0 = LV - 1; round(round, IC<0>{}, jl0); jl0 += U) {}
  } else {
#pragma unroll
    for (int s = 0; s < 2 * (LV - 1); ++s) do_step(j_first + s, s % U, (s + 2) / 2);
#pragma unroll 1
    for (int jl0 = LV - 1;; jl0 += U) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (jl0 + q > j_last) return;
        do_step(jl0 + q, (u0 + q) % U, LV);
      }
    }
  }
}

// DGM: where A_ii comes from.  0: `diag` / the centre diagonal through the address select of the body (every mode and layout);
// 1: the centre coefficient, said so by the caller (diag == NULL on a symmetric full-stencil plan): the DGC body; 2: the device
// flag of gla_jacobi_sweeps_aos decides, ONCE, at kernel start, between the two bodies -- the later passes of the module path
// run the DGC body without a second launch (the flag is raised only where vertex_attr[:, 0] is not the matrix diagonal).
template <typename T, int TS, int MODE, bool RCP, int HD, bool STATIC_IDX, bool SYM, bool FM, int DGM, int PF = 1, int DBG = 0>
__global__ __launch_bounds__(GLA_BLOCK, 2) void jacobi_march5_kernel(M5View dv, const T* __restrict__ diag,
                                                                     const T* __restrict__ bvec, const T* __restrict__ x,
                                                                     T* __restrict__ xo, T w, int lines, int nstrips, TileMap tm,
                                                                     int tile_base0, int tile_split, int tile_base1, int64_t ldva,
                                                                     T* __restrict__ diag_out, T* __restrict__ b_out) {
#ifdef GLA_M5_VARIANTS
  struct Stamp {
    unsigned long long* p; unsigned long long t0;
    __device__ Stamp(unsigned long long* buf) : p(buf), t0(buf ? __builtin_amdgcn_s_memrealtime() : 0) {}
    __device__ ~Stamp() {
      if (p && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* q = p + 4 * (size_t)blockIdx.x;
        q[0] = t0; q[1] = __builtin_amdgcn_s_memrealtime(); q[2] = ((unsigned long long)xcc << 32) | hw; q[3] = blockIdx.x;
      }
    }
  } stamp_(dv.stamps);
#endif
  if constexpr (DGM == 1) {
    m5_body<T, TS, MODE, RCP, HD, STATIC_IDX, SYM, FM, true, PF, DBG>(dv, diag, bvec, x, xo, w, lines, nstrips, tm, tile_base0, tile_split,
                                                           tile_base1, ldva, diag_out, b_out);
  } else if constexpr (DGM == 2) {
    if (__builtin_amdgcn_readfirstlane(*dv.dg_flag) != dv.dg_epoch)
      m5_body<T, TS, MODE, RCP, HD, STATIC_IDX, SYM, FM, true, PF, DBG>(dv, diag, bvec, x, xo, w, lines, nstrips, tm, tile_base0, tile_split,
                                                             tile_base1, ldva, diag_out, b_out);
    else
      m5_body<T, TS, MODE, RCP, HD, STATIC_IDX, SYM, FM, false, PF, DBG>(dv, diag, bvec, x, xo, w, lines, nstrips, tm, tile_base0, tile_split,
                                                              tile_base1, ldva, diag_out, b_out);
  } else {
    m5_body<T, TS, MODE, RCP, HD, STATIC_IDX, SYM, FM, false, PF, DBG>(dv, diag, bvec, x, xo, w, lines, nstrips, tm, tile_base0, tile_split,
                                                            tile_base1, ldva, diag_out, b_out);
  }
}

// Markstein division and LDS prefetch depth per level count: what fits two waves per SIMD (256 registers; the build log of
// csrc/build.sh shows the allocation of every instance)
template <typename T, int LV, int MODE, bool SYM = false> struct M5Cfg {
  // (the symmetric AoS pass of six sweeps is the one instance the Markstein division does not fit in 256 registers: IEEE division;
  //  that pass is bound by its 16-24 bytes per row of writes anyway)
#ifdef GLA_M5_VARIANTS
  static constexpr bool rcp = sizeof(T) == 4 || ((LV <= 7 || LV >= 9) && !(SYM && MODE == M5_AOS && LV == 6));     // (9, 10 levels: experiments)
#else
  static constexpr bool rcp = sizeof(T) == 4 || (LV <= 7 && !(SYM && MODE == M5_AOS && LV == 6));
#endif
  static constexpr int hd = sizeof(T) == 4 ? 2 : (LV >= 8 || (MODE == M5_AOS && LV == 6) ? 0 : 2);
  static constexpr bool static_idx = sizeof(T) == 4 || LV >= 8;
};
template <typename T, int TS, int MODE, int LV, bool SYM, bool FM, int DGM> struct M5Pick {
  typedef M5Cfg<T, LV, MODE, SYM> C;
  static constexpr auto kernel = jacobi_march5_kernel<T, TS, MODE, C::rcp, C::hd, C::static_idx, SYM, FM, DGM>;
};
#define M5_KERNEL(T, TS, MODE, LV, SYM, FM, DGM) M5Pick<T, TS, MODE, LV, SYM, FM, DGM>::kernel

#ifdef GLA_M5_VARIANTS
unsigned long long* g_m5_stamps[2] = {nullptr, nullptr};      // [0]: parts 0 / 1, [1]: part 2 (experiments: gla_debug_m5_stamps)
#endif

template <typename T, int TS, int MODE, bool SYM, bool FM, int DGM>
int m5_blocks_per_cu() {
  static int cached = 0;
  if (!cached) {
    constexpr int LV = TS + (MODE == M5_RES ? 1 : 0);
    int nb = 0;
    const size_t lds = (size_t)LV * 2 * (GLA_BLOCK + 2) * sizeof(T);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, M5_KERNEL(T, TS, MODE, LV, SYM, FM, DGM), GLA_BLOCK, lds) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 2;
    }
    cached = nb;
  }
  return cached;
}

template <typename T, int TS, int MODE, bool SYM, bool FM, int DGM = 0>
int m5_launch_sym(const gla_march5_args& a) {
  gla_csr* A = a.A;
  constexpr int LV = TS + (MODE == M5_RES ? 1 : 0);
  M5View dv;
  dv.val = A->dia_val;
  dv.mask = A->dia_mask;
  dv.S = A->dia_S; dv.n_rows = A->n_rows;
  dv.nlines = (A->n_rows + A->dia_S - 1) / A->dia_S;
  dv.aos_off = a.aos_off;
  dv.aos_rows = a.aos_rows >= 0 ? a.aos_rows : A->n_rows;
  dv.ldd = a.ldd > 0 ? a.ldd : 1;
  dv.dg_flag = a.dg_flag;
  dv.dg_epoch = a.dg_epoch;
  dv.dg_mode = MODE == M5_AOS ? 0 : (a.dg_flag ? 2 : (a.diag ? 0 : 1));
  const int64_t nlines = dv.nlines;
  // Height of the first / last line block of the multi-GPU split: ghost + LV lines keep the interior blocks off the ghost
  // lines (an interior block reads x from LV lines above its first line); at least 2 x ghost, so that the edge launch alone
  // produces the `ghost` own lines either end that the neighbours want next -- their exchange can start when the edge launch
  // ends, long before the interior launch does.
  // (a.edge_h: given by the caller -- the staggered multi-GPU passes of gla_dist.hip, whose edge blocks grow from pass to pass so
  // that an interior launch needs the previous INTERIOR launch only; never less than ghost + LV)
  const int edge_h = a.edge_h > 0 ? a.edge_h : (a.ghost + LV > 2 * a.ghost ? a.ghost + LV : 2 * a.ghost);
  // The interior launch (part 1) leaves the lines the two edge blocks READ (edge_h + TS either end, copied by the host
  // beforehand) alone: the edge launch runs concurrently on a second stream, and the two must not touch the same words of
  // diag_out / b_out
  dv.aos_wlo = a.part == 1 ? (int64_t)edge_h + TS : 0;
  dv.aos_whi = a.part == 1 ? dv.nlines - ((int64_t)edge_h + TS) : dv.nlines;
  const int bi_eff = GLA_BLOCK - 2 * LV;
  const int nstrips = (int)((A->dia_S + bi_eff - 1) / bi_eff);
  const int part = a.part, ghost = a.ghost;
  // Interior / boundary split (part 1 / 2, the multi-GPU route): the first and the last `ghost` grid lines of x arrive while
  // the interior blocks run, so no interior block may read them: the first and the last line block are edge_h lines high
  // (above), everything between is split as for part 0.
  // Grids too short for that (fewer than 2 edge blocks + 1 line) count as boundary as a whole.
  const int edge = part != 0 && nlines > 2 * (int64_t)edge_h ? edge_h : 0;
  (void)ghost;
  const int64_t inner_lines = nlines - 2 * edge;
  int lines = g_tuning.jacobi_lines;
  if (lines <= 0) {
    // block height: the fewest lines per workgroup (least serial work, but every block re-reads 2 (LV - 1) coefficient
    // lines) for which all workgroups are resident at once.  Measured (tools/ab_jacobi_march.py): two workgroups per CU beat
    // three or four even where the registers allow more -- fewer, longer streams; one per CU cannot hide a line step
    int per_cu = m5_blocks_per_cu<T, TS, MODE, SYM, FM, DGM>();
    if (per_cu > 2) per_cu = 2;
    // Interior / boundary split: the interior launch leaves 2 x nstrips workgroup slots FREE for the two edge line blocks that
    // a second stream launches once the ghost lines are in.  A march workgroup takes half a CU's registers, so on a chip full
    // of interior workgroups the edge launch (and even the small ghost-fill kernel before it) waits for the interior launch
    // to drain -- 54 us behind it instead of beside it (kernel trace of the one-GPU rehearsal, profiles/r03).
    // Round 5: slots are 32 x per_cu PER XCD, workgroup b of either launch runs on XCD b % 8, and the last slot of an XCD is not
    // reliably there for a second launch (per-workgroup time stamps, profiles/r05/edge_blocks_trace.txt: with 56 + 9 workgroups on an
    // XCD -- and even with 55 + 1 that exits at once + 9 -- one edge workgroup per XCD started ~45 us late, behind another edge
    // workgroup, and the edge launch took 90 us instead of 32; the next pass's exchange waits for exactly these blocks).  The
    // interior launch therefore takes at most 32 per_cu - 1 - ceil(2 nstrips / 8) workgroups per XCD.
    int64_t slots = 256 * (int64_t)per_cu;
    if (edge && slots > 4 * (int64_t)nstrips) {
      const int64_t edge_x = (2 * (int64_t)nstrips + GLA_XCDS - 1) / GLA_XCDS;
      const int64_t int_x = 32 * (int64_t)per_cu - (g_tuning.edge_prio ? 1 : 0) - edge_x;
      slots = g_tuning.edge_prio ? GLA_XCDS * int_x : slots - 2 * (int64_t)nstrips;
    }
    lines = gla_pick_block_lines(inner_lines, nstrips, slots, 2 * (LV - 1), 2 * LV);
    if (lines > 1024) lines = 1024;
  }
  if (lines < 2 * LV) lines = 2 * LV;
  const int nlb_all = (int)((inner_lines + lines - 1) / lines) + (edge ? 2 : 0);
  dv.edge = edge;
  dv.nlb = nlb_all;
  dv.prio_hi = (part == 2 && edge && g_tuning.edge_prio) ? 1 : 0;
#ifdef GLA_M5_VARIANTS
  dv.stamps = g_m5_stamps[part == 2 ? 1 : 0];
#endif
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
  if (part != 0 && edge && g_tuning.edge_prio) {    // the two launches of a split pass share the chip: they must fit per XCD (map_tile)
    tm.balanced = part == 1 ? 1 : 2;
    tm.tiles_per_xcd = (count + GLA_XCDS - 1) / GLA_XCDS;
  }
  const dim3 grid((unsigned)(tm.tiles_per_xcd * GLA_XCDS)), block(GLA_BLOCK);
  const size_t lds = (size_t)LV * 2 * (GLA_BLOCK + 2) * sizeof(T);
#ifdef GLA_M5_VARIANTS
  if constexpr (sizeof(T) == 8 && (TS == 7 || TS == 9 || TS == 10) && MODE == M5_PLAIN && SYM && FM && DGM == 1) {
    const int v = g_tuning.m5_variant >= 100 ? g_tuning.m5_variant % 100 : g_tuning.m5_variant;     // (1xx: with a forced sweep count, below)
#define M5V(HDV, SIV, PFV) hipLaunchKernelGGL((jacobi_march5_kernel<T, TS, MODE, true, HDV, SIV, SYM, FM, DGM, PFV>), grid, block, lds, a.st, dv, (const T*)a.diag, (const T*)a.b, (const T*)a.x, (T*)a.xo, (T)a.w, lines, nstrips, tm, base0, split, base1, a.ldva > 0 ? a.ldva : (int64_t)1, (T*)a.diag_out, (T*)a.b_out)
    if (v > 0) {
      switch (v) {
        case 1: M5V(0, false, 1); break; case 2: M5V(1, false, 1); break; case 3: M5V(3, false, 1); break;
        case 4: M5V(2, false, 2); break; case 5: M5V(0, false, 2); break; case 6: M5V(3, false, 2); break;
        case 7: M5V(2, false, 3); break; case 8: M5V(0, false, 3); break;
#define M5D(DBGV) hipLaunchKernelGGL((jacobi_march5_kernel<T, TS, MODE, true, 2, false, SYM, FM, DGM, 1, DBGV>), grid, block, lds, a.st, dv, (const T*)a.diag, (const T*)a.b, (const T*)a.x, (T*)a.xo, (T)a.w, lines, nstrips, tm, base0, split, base1, a.ldva > 0 ? a.ldva : (int64_t)1, (T*)a.diag_out, (T*)a.b_out)
        case 9: M5D(1); break; case 10: M5D(2); break; case 11: M5D(4); break; case 12: M5D(8); break;
        case 13: M5D(3); break; case 14: M5D(9); break; case 15: M5D(5); break;
        case 16: M5D(16); break; case 17: M5D(24); break; case 18: M5D(17); break;
        case 19: M5D(32); break; case 20: M5D(33); break; case 21: M5D(34); break;
        default: return GLA_ERR_INVALID_ARG;
      }
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
#endif
  hipEvent_t stop_ev = gla_m5_stop_event;
  gla_m5_stop_event = nullptr;
  if (stop_ev) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(a.st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone && g_tuning.dist_stagger != 2) {
      hipExtLaunchKernelGGL((M5_KERNEL(T, TS, MODE, LV, SYM, FM, DGM)), grid, block, lds, a.st, nullptr, stop_ev, 0, dv,
                            (const T*)a.diag, (const T*)a.b, (const T*)a.x, (T*)a.xo, (T)a.w, lines, nstrips, tm, base0, split, base1,
                            a.ldva > 0 ? a.ldva : (int64_t)1, (T*)a.diag_out, (T*)a.b_out);
      GLA_LAUNCH_CHECK();
      return GLA_OK;
    }
  }
  hipLaunchKernelGGL((M5_KERNEL(T, TS, MODE, LV, SYM, FM, DGM)), grid, block, lds, a.st, dv,
                     (const T*)a.diag, (const T*)a.b, (const T*)a.x, (T*)a.xo, (T)a.w, lines, nstrips, tm, base0, split, base1,
                     a.ldva > 0 ? a.ldva : (int64_t)1, (T*)a.diag_out, (T*)a.b_out);
  GLA_LAUNCH_CHECK();
  if (stop_ev) GLA_HIP_CHECK(hipEventRecord(stop_ev, a.st));
  return GLA_OK;
}

template <typename T, int TS, int MODE>
int m5_launch(const gla_march5_args& a) {
  if (!gla_dia_sym5(a.A)) return m5_launch_sym<T, TS, MODE, false, false>(a);
  // (the no-mask instances exist for the symmetric layout only: a plan whose values are not symmetric keeps its masks as well;
  //  ... and for up to seven levels: hipcc spills the eight-level no-mask instance, 205 registers of it)
#ifdef GLA_M5_VARIANTS
  constexpr bool fm_fits = TS + (MODE == M5_RES ? 1 : 0) <= 7 || (TS >= 9 && MODE == M5_PLAIN);
#else
  constexpr bool fm_fits = TS + (MODE == M5_RES ? 1 : 0) <= 7;
#endif
  if constexpr (fm_fits) {
    if (gla_dia_full5(a.A)) {
      if constexpr (MODE == M5_PLAIN) {             // A_ii = the centre coefficient: said so by the caller / decided by the flag
        if (a.dg_flag) return m5_launch_sym<T, TS, MODE, true, true, 2>(a);
        if (!a.diag) return m5_launch_sym<T, TS, MODE, true, true, 1>(a);
      }
      return m5_launch_sym<T, TS, MODE, true, true>(a);
    }
  }
  return m5_launch_sym<T, TS, MODE, true, false>(a);
}

template <typename T, int MODE>
int m5_dispatch_ts(const gla_march5_args& a) {
  switch (a.ts) {
    case 1: if constexpr (MODE == M5_RES) return m5_launch<T, 1, MODE>(a); else return GLA_ERR_UNSUPPORTED;
    case 2: return m5_launch<T, 2, MODE>(a);
    case 3: return m5_launch<T, 3, MODE>(a);
    case 4: return m5_launch<T, 4, MODE>(a);
    case 5: return m5_launch<T, 5, MODE>(a);
    case 6: return m5_launch<T, 6, MODE>(a);
    case 7: return m5_launch<T, 7, MODE>(a);
    case 8: if constexpr (MODE != M5_RES) return m5_launch<T, 8, MODE>(a); else return GLA_ERR_UNSUPPORTED;
#ifdef GLA_M5_VARIANTS
    case 9: if constexpr (MODE == M5_PLAIN && sizeof(T) == 8) return m5_launch<T, 9, MODE>(a); else return GLA_ERR_UNSUPPORTED;
    case 10: if constexpr (MODE == M5_PLAIN && sizeof(T) == 8) return m5_launch<T, 10, MODE>(a); else return GLA_ERR_UNSUPPORTED;
#endif
    default: return GLA_ERR_UNSUPPORTED;
  }
}

}  // namespace

// `ts` sweeps (2 .. 8; with res: 1 .. 7 sweeps + the residual) of the canonical 5-point plan A in one pass.  The caller
// (gla_rows.hip) has checked the plan (jacobi_fused_ok + gla_dia_canon5) and the operands.
// Two translation units (the instances take minutes to compile): this file holds the fp64 instances and the entry point,
// gla_march5_f32.hip includes it with GLA_M5_F32_TU defined and holds the fp32 ones.
#if defined(GLA_M5_VARIANTS) && !defined(GLA_M5_F32_TU)
extern "C" int gla_debug_m5_stamps(void* interior_buf, void* edge_buf) {
  g_m5_stamps[0] = (unsigned long long*)interior_buf;
  g_m5_stamps[1] = (unsigned long long*)edge_buf;
  return 0;
}
#endif
#ifdef GLA_M5_F32_TU
int gla_march5_pass_f32(const gla_march5_args& a) {
  if (a.res) return m5_dispatch_ts<float, M5_RES>(a);
  if (a.ldva > 0) return m5_dispatch_ts<float, M5_AOS>(a);
  return m5_dispatch_ts<float, M5_PLAIN>(a);
}
#else
thread_local hipEvent_t gla_m5_stop_event = nullptr;         // gla_march5_set_stop_event (gla_internal.h): consumed by the thread's next launch
int gla_march5_pass_f32(const gla_march5_args& a);
int gla_march5_pass(const gla_march5_args& a_in) {
  gla_march5_args a = a_in;
#ifdef GLA_M5_VARIANTS
  // experiments: option m5_variant = 100 * ts + v runs a plain 7-sweep call with ts = 9 / 10 levels instead (timing only)
  if (g_tuning.m5_variant >= 900 && a.ts == 7 && !a.res && a.ldva <= 0 && a.part == 0 && a.dtype == GLA_F64) a.ts = g_tuning.m5_variant / 100;
#endif
  if (!a.A || !gla_dia_canon5(a.A) || a.part < 0 || a.part > 2) return GLA_ERR_UNSUPPORTED;
  const int lv = a.ts + (a.res ? 1 : 0);
  if (a.ghost < lv && a.part != 0) return GLA_ERR_INVALID_ARG;
  if (a.res && (a.ldva > 0 || a.part != 0)) return GLA_ERR_UNSUPPORTED;
  if (a.dtype != GLA_F64) return gla_march5_pass_f32(a);
  if (a.res) return m5_dispatch_ts<double, M5_RES>(a);
  if (a.ldva > 0) return m5_dispatch_ts<double, M5_AOS>(a);
  return m5_dispatch_ts<double, M5_PLAIN>(a);
}
#endif
