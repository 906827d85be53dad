// Internal declarations shared by the translation units of libgla_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gla.h"

#define GLA_HIP_CHECK(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return -(1000 + (int)_e);           \
  } while (0)

#define GLA_LAUNCH_CHECK() GLA_HIP_CHECK(hipGetLastError())

constexpr int GLA_BLOCK = 256;          // 4 wavefronts of 64 lanes
constexpr int GLA_XCDS = 8;             // MI355X: 8 XCDs, each with a private L2
constexpr int GLA_LDS_ELEMS_MAX = 4096; // staged nnz per workgroup (32 KiB of f64 products)
constexpr int GLA_TILE_NNZ_MIN = 1024;
constexpr int GLA_PAD = 8;              // over-read slack (elements) behind val/col
constexpr int GLA_DIA_MAX = 16;
// rows longer than this never share an LDS tile: each is a row tile of its own (CSR-adaptive "long" bin) and is summed by
// the whole workgroup in 256 fixed chunks; shorter rows stay in stream tiles with strictly in-order sums
constexpr int GLA_LONG_ROW = GLA_LDS_ELEMS_MAX - GLA_TILE_NNZ_MIN;
// K = 1 CSR stream kernel: rows of this many nonzeros or more are summed by their wave (64 fixed chunks + xor tree; option row_sum)
constexpr int GLA_WAVE_ROW = 64;
constexpr int GLA_DG_SLOTS = 64;        // calls of gla_jacobi_sweeps_aos that may be in flight on ONE plan at a time (on different streams)
constexpr int GLA_DG_CAPTURE_SLOTS = 192;   // ... and calls captured into HIP graphs over the plan's lifetime (a flag word each)

struct gla_csr {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int32_t* rowptr = nullptr;     // [n_rows + 1]
  int32_t* col = nullptr;        // [nnz + GLA_PAD], CSR order
  int32_t* perm = nullptr;       // [nnz] CSR position -> edge id, or nullptr (identity)
  void* val = nullptr;           // [nnz + GLA_PAD] * 8 bytes, CSR order
  int val_dtype = -1;
  int32_t* tile_row = nullptr;   // [n_tiles + 1] first row of each tile
  int32_t* tile_info = nullptr;  // [4 * n_tiles] {first row, end row, first nnz, end nnz} per tile (int4)
  int32_t* tile_lists = nullptr; // [2 * n_tiles]: interior tiles then boundary tiles
  int32_t n_tiles = 0, n_interior = 0, n_boundary = 0;
  int32_t tile_nnz = 0, lds_elems = 0, max_row_len = 0;
  int32_t sorted_input = 1, long_rows = 0;   // long_rows: number of rows with more than GLA_LONG_ROW nonzeros
  int32_t* long_row_ids = nullptr;           // [long_rows] their row ids, ascending (device)
  // optional packed stream for the K = 1 kernels (option "compress"): 3 bytes per nonzero instead of 12
  int16_t* col16 = nullptr;      // [nnz + 16] column minus the first row of the nonzero's tile
  uint8_t* vcode = nullptr;      // [nnz + 16] index into vdict
  void* vdict = nullptr;         // [256] distinct values (bit patterns ascending), dtype = val_dtype
  uint16_t* rowoff16 = nullptr;  // [n_rows + n_tiles] rowptr minus the tile's first nonzero; tile t's rows at r + t
  uint32_t* col16_wide = nullptr; // bit t set: tile t has a column outside int16 range of its first row (reads col)
  int32_t packed = 0, col16_state = 0;   // col16_state: 0 not tried, 1 built, -1 most tiles do not fit int16
  int32_t col16_wide_tiles = 0;
  // optional diagonal (stencil) layout for the K = 1 kernels (option "dia"): every nonzero lies on one of D <= 15
  // diagonals col - row = dia_off[d], rows hold them in ascending order.  No column indices are streamed at all.
  int32_t dia_state = 0;         // 0 not tried, 1 built, -1 the matrix is not a stencil
  int32_t dia_D = 0;
  int32_t dia_off[16] = {0};     // ascending offsets (host copy; passed to the kernels by value)
  uint16_t* dia_mask = nullptr;  // [n_rows] bit d set: row has an entry on diagonal d
  void* dia_val = nullptr;       // [dia_D][n_rows] values, diagonal-major (0 where the row has no entry)
  int* dg_flags = nullptr;       // [GLA_DG_SLOTS] device words of gla_jacobi_sweeps_aos' "A_ii is not the matrix diagonal" flags: call k uses
  uint32_t dg_calls = 0;         // word k mod GLA_DG_SLOTS and stores k there to raise it (never reset; guarded by dg_mutex)
  int32_t dg_captured = 0;       // words [GLA_DG_SLOTS, +GLA_DG_CAPTURE_SLOTS) handed to calls captured into HIP graphs, one each
  // (the symmetry check of gla_csr_set_values is ASYNCHRONOUS: a check kernel, a copy into a pinned host word and an event on the
  //  caller's stream; the first reader -- gla_dia_sym_bits, at the first launch or gla_csr_info -- waits for the event.  No
  //  malloc / free / stream synchronise per value set: callers that refresh values every step stay asynchronous)
  int* sym_dev = nullptr;        // device word the check kernel ORs broken pairs into
  int* sym_host = nullptr;       // pinned host copy
  hipEvent_t sym_event = nullptr;
  int32_t sym_pending = 0, sym_mask = 0;
  int32_t dia_sym = 0;           // bit d set: diagonal d (offset < 0) repeats its partner of offset -off bit for bit, A[r][r + off] ==
                                 // A[r + off][r] for every row (0 where r + off < 0): the kernels need not read it (option "sym")
  int32_t dia_full = 0;          // canonical 5-point plans: 1 = EVERY row couples to every grid neighbour it has (its mask is the one its
                                 // position implies): structure only, checked at plan creation; the march kernel then reads no masks
  int32_t dia_const = 0;         // 1 (option "compress"): every diagonal holds ONE value -> no value stream either
  double dia_cval[16] = {0};     // that value per diagonal
  int64_t dia_row_lo = 0, dia_row_hi = 0;   // rows [lo, hi) fit the diagonal set (halo rows of a row block do not)
  // 2-D reading of the diagonal set for the blocked K > 1 kernel: off[d] = dia_a[d] * dia_S + dia_b[d] with a grid-line
  // stride S and small |a| <= dia_A, |b| <= dia_B (5-point: S = N, a in {-1,0,1}, b in {-1,0,1}); dia_S = 0: none
  int32_t dia_S = 0, dia_A = 0, dia_B = 0;
  int32_t dia_nowrap = 0;       // 1: no entry leaves its grid line in this reading ((r % S) + b stays in [0, S)): rows of a
                                // line depend only on rows of the lines a = -A..A -- what the temporally blocked kernel needs
  int8_t dia_a[16] = {0}, dia_b[16] = {0};
  int64_t int_row_lo = -1, int_row_hi = -1; // rows of the interior tiles when they form one contiguous run
};

struct gla_tuning {
  int spmv_variant = 0;     // experimental kernel selection for K = 1 (see launch_rows)
  int tile_nnz = 0;         // 0 = automatic (256 * average row length)
  int spmm_variant = 0;     // 0 = paired-rhs 16-byte kernel when alignment allows, 1 = scalar lanes
  int xcd_chunk = 64;       // consecutive tiles per XCD turn (1 = identity mapping, 0 = one contiguous span per XCD)
  int col16 = 1;            // K = 1 kernels read int16 tile-relative columns when the plan could build them
  int compress = 0;         // 1: gla_csr_set_values also builds the packed stream when the matrix allows it
  int row_sum = 1;          // K = 1 CSR stream kernel: 1 = a row of GLA_WAVE_ROW (64) .. 3072 nonzeros is summed by its wave (64
                            // contiguous chunks in edge order + a fixed xor tree: deterministic, a function of the row's own length --
                            // independent of tiling and row-block partition -- and equal to the in-order sum to rounding);
                            // 0 = one lane per row, strictly in edge order (bit-exact against the CPU scatter-add for any graph)
  int dia = 1;              // stencil matrices (<= 16 diagonals): K = 1 kernels use the diagonal layout
  int jacobi_fuse = 1;      // gla_jacobi_sweeps on 2-D stencil plans: 0 one sweep per pass, 1 several (temporal blocking; gla_jacobi_sweeps_schedule), 2 up to two
  int jacobi_lines = 0;     // grid lines a workgroup of the fused kernel marches through (0 = one resident round of workgroups)
  int jacobi_ts_max = 8;    // canonical 5-point plans: most sweeps per pass (2 .. 8)
  int jacobi_ts_max9 = 5;   // canonical 9-point plans: most sweeps per pass (2 .. 6)
  int sym = 1;              // stencil plans whose value set is symmetric (checked bit for bit at gla_csr_set_values): the K = 1 march
                            // kernels read the lower diagonals from their upper partners (24 instead of 40 bytes of values per 5-point row)
  int full_mask = 1;        // canonical 5-point plans whose every row has the mask of its grid position (dia_full): the symmetric march passes
                            // do not read the 2-byte row masks (0: they do)
  int jacobi_diag_flag = 1; // gla_jacobi_sweeps_aos on 5-point march plans: 1 = no A_ii copy; the first pass flags whether A_ii repeats the matrix
                            // diagonal, the later passes skip the A_ii stream if so (else read the AoS column); 0 = contiguous copies as before
  int m5_variant = 0;       // (experiments: GLA_M5_VARIANTS builds only)
  int edge_prio = 1;        // the edge launch of a split multi-GPU pass runs at raised wave priority (A/B: option edge_prio)
  int dist_stagger = 1;     // gla_dist_jacobi_temporal: interior launches of consecutive passes depend on one another only (option dist_stagger)
  int jacobi_march = 1;     // canonical 5-point plans: 1 = the register-history march kernel, 0 = the generic LDS-ring kernel
  int64_t jacobi_min_rows = 1 << 19;   // gla_jacobi_sweeps fuses only at this many rows or more (below, one sweep per launch
                                       // is faster: 512^2 takes 4.5 us per sweep single vs 5.9 fused, 768^2 8.4 vs 6.0)
};
extern gla_tuning g_tuning;

// ... or the canonical 9-point pattern (the Q1 FEM operator): all (a, b) in {-1, 0, 1}^2, row-major
inline bool gla_dia_canon9(const gla_csr* A) {
  if (A->dia_state != 1 || A->dia_S <= 0 || A->dia_D != 9 || A->dia_A != 1 || A->dia_B != 1) return false;
  for (int d = 0; d < 9; ++d)
    if (A->dia_a[d] != d / 3 - 1 || A->dia_b[d] != d % 3 - 1) return false;
  return true;
}

// Grid lines per workgroup of a line-marching kernel (temporally blocked Jacobi / Chebyshev): `nstrips` strips side by side,
// `slots` workgroups resident on the chip at once, `ramp` extra line steps every block pays.  One resident round of
// workgroups when that fills the chip (fewest blocks, least overlap); when the strip count quantises badly (171 strips on 512
// slots: 342 or 513 workgroups) several rounds of shorter blocks.  Cost model: rounds x (lines + ramp).
inline int gla_pick_block_lines(int64_t nlines, int64_t nstrips, int64_t slots, int ramp, int min_lines) {
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
  return (int)(best_l > (1 << 20) ? (1 << 20) : best_l);
}

// the 2-D reading is the canonical 5-point pattern: diagonals (-1,0) (0,-1) (0,0) (0,1) (1,0) in this order
inline bool gla_dia_canon5(const gla_csr* A) {
  static const int ca[5] = {-1, 0, 0, 0, 1}, cb[5] = {0, -1, 0, 1, 0};
  if (A->dia_state != 1 || A->dia_S <= 0 || A->dia_D != 5 || A->dia_A != 1 || A->dia_B != 1) return false;
  for (int d = 0; d < 5; ++d)
    if (A->dia_a[d] != ca[d] || A->dia_b[d] != cb[d]) return false;
  return true;
}

// march5.hip: `ts` Jacobi sweeps (res: + the residual of the result) of a canonical 5-point plan in one temporally blocked pass
struct gla_march5_args {
  gla_csr* A;
  int dtype, ts;
  bool res;                    // one more level forms r = b - A x^(k+ts) -> b_out
  const void *diag, *b, *x;    // ldva > 0: x is the AoS vertex attribute tensor [A_ii, b, x] (diag, b unused)
  void* xo;
  double w;
  hipStream_t st;
  int part;                    // 0 all line blocks, 1 interior, 2 first + last (multi-GPU split)
  int64_t ldva;
  void *diag_out, *b_out;      // AoS pass: contiguous copies of A_ii and b; res: the residual
  int ghost;                   // ghost lines either end (parts 1 / 2)
  int edge_h = 0;              // parts 1 / 2: height of the first / last line block (0: max(ghost + levels, 2 ghost)); gla_dist.hip's
                               // staggered passes grow it from pass to pass
  int64_t aos_off, aos_rows;
  int64_t ldd = 1;             // stride of `diag` (plain / residual passes): 1, or an AoS row stride
  int dg_epoch = 0;            // the call's number: `*dg_flag == dg_epoch` means raised in this call
  int* dg_flag = nullptr;      // AoS pass: raised where A_ii differs from the centre diagonal; plain passes: A_ii is `diag` if raised,
};                             // the centre diagonal otherwise.  diag == nullptr (plain passes, no flag): A_ii = the centre diagonal
// every lower diagonal of the canonical 5-point plan repeats its upper partner (and the option allows using that)
inline int32_t gla_dia_sym_bits(const gla_csr* A) {          // resolves a pending check (one event wait, once per value set)
  if (A->sym_pending) {
    gla_csr* p = const_cast<gla_csr*>(A);
    (void)hipEventSynchronize(p->sym_event);
    p->dia_sym = p->sym_mask & ~*p->sym_host;
    p->sym_pending = 0;
  }
  return A->dia_sym;
}
inline bool gla_dia_sym5(const gla_csr* A) { return g_tuning.sym && gla_dia_canon5(A) && (gla_dia_sym_bits(A) & 3) == 3; }
// ... and no row lacks a neighbour its grid position has (and the option allows using that): no mask stream
inline bool gla_dia_full5(const gla_csr* A) { return g_tuning.full_mask && gla_dia_canon5(A) && A->dia_full == 1; }
int gla_march5_pass(const gla_march5_args& a);
// The NEXT march pass this thread launches carries `ev` as its completion event (hipExtLaunchKernelGGL's stop event: the kernel's
// own packet signals it, no barrier packet between this kernel and the next one of the stream -- an event RECORD between two kernels
// of one queue costs ~7 us on this runtime, profiles/r05/config4_block_stagger.txt).  Streams being captured get a plain record.
extern thread_local hipEvent_t gla_m5_stop_event;            // (defined in gla_march5.hip)
inline void gla_march5_set_stop_event(hipEvent_t ev) { gla_m5_stop_event = ev; }
// GLA_OK iff gla_jacobi_fused_parts(_aos) would run `sweeps` sweeps on this canonical 5-point plan with these value types
// (gla_rows.hip); otherwise the status it would return.  For callers that must know BEFORE they enqueue anything (gla_dist.hip).
int gla_jacobi_parts_status(const gla_csr* A, int dtype, int sweeps);
// the A_ii flag word and number of one call, and the march passes that raise / obey it (gla_rows.hip; for gla_dist.hip)
int gla_jacobi_dg_slot(gla_csr* A, hipStream_t st, int** flag, int* epoch);
int gla_jacobi_parts_flagged(gla_csr* A, int dtype, const void* va, int64_t ldva, int64_t row_offset, int64_t n_attr_rows,
                             const void* diag, const void* B, const void* X, void* diag_out, void* b_out, void* Xout, double w,
                             int sweeps, int part, int ghost_lines, int* dg_flag, int dg_epoch, hipStream_t st);

// jacobik.hip: `sweeps` (2 .. GLA_JACOBIK_TS_MAX) Jacobi sweeps of K right-hand sides in one temporally blocked pass
constexpr int GLA_JACOBIK_TS_MAX = 5;
int gla_jacobik_pass(gla_csr* A, int dtype, int K, const void* diag, const void* B, int64_t ldb, const void* X, int64_t ldx,
                     void* Xout, int64_t ldxo, double w, int sweeps, hipStream_t st);

// powerk.hip: the Rayleigh tail of the power method as a one-level pass of the march kernel (gla_power_rayleigh tries it first)
int gla_power_tail_pass(gla_csr* A, int dtype, int K, const void* Y, int64_t ldy, const void* nrm, void* Bout, int64_t ldb,
                        void* B2out, int64_t ldb2, void* nA_out, void* bb_out, void* work, int64_t work_bytes, hipStream_t st);

// plan.hip
int gla_sort_rows_stable(int64_t nnz, int64_t n_rows, const int64_t* row, int32_t* perm_out,
                         uint32_t* keys_sorted, hipStream_t stream);
