/*
 * gla.h -- C ABI of libgla_hip.so: MI355X (gfx950) kernels for the sparse message-passing
 * hot path of sandialabs/gnn-applied-linear-algebra.
 *
 * The reference has no FFI boundary (pure Python).  Its arithmetic lives in two
 * third-party primitives -- PyG `MetaLayer` (gather x[row], x[col]) and `torch_scatter.scatter`
 * (segment reduction keyed by edgeij_pair[0]) -- composed by per-algorithm nn.Modules.  Each entry
 * point below replaces one such composition, i.e. everything between a module's `forward()` and the
 * tensors it returns; the comment on each names the reference file:line it stands for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the library never frees or
 *     keeps caller memory beyond the call (plans copy what they need);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, nothing synchronises
 *     except plan creation (gla_csr_create_from_coo) which returns sizes to the host;
 *   - dense operands are row-major [rows, K] with a leading dimension in ELEMENTS (ld >= K);
 *     K is the number of right-hand sides (1 = SpMV; the fastest K = 1 kernels need the gathered
 *     vector contiguous, ld = 1 -- a strided single column is served by the general kernel);
 *   - "edge order" = the order of the caller's COO edge list (reference tensors edgeij_pair /
 *     edge_attr); "CSR order" = stable sort of the edges by row (edge order kept inside a row, so
 *     in-order row sums equal the reference's sequential scatter-add bit for bit);
 *   - return value: 0 on success, <0 = GLA_ERR_* below, (-3000, -1000] = -(1000 + hipError_t),
 *     <= -3000 = -(3000 + ncclResult_t).
 *   - inputs are never mutated unless documented (reference modules return new tensors).
 */
#ifndef GLA_H
#define GLA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gla_csr* gla_csr_t;
typedef void* gla_stream_t;

enum { GLA_F32 = 0, GLA_F64 = 1 };

enum {
  GLA_OK = 0,
  GLA_ERR_INVALID_ARG = -1,      /* null pointer, negative size, bad dtype, ld < K ... */
  GLA_ERR_INDEX_RANGE = -2,      /* an edge index outside [0, n_rows) x [0, n_cols) */
  GLA_ERR_TOO_LARGE = -3,        /* nnz or n exceeds the int32 CSR index range */
  GLA_ERR_NO_VALUES = -4,        /* plan has no values of the requested dtype (call gla_csr_set_values) */
  GLA_ERR_UNSUPPORTED = -5,
  GLA_ERR_NO_DEVICE = -6
};

/* 100 * major + minor; entry points are only ever added within a major version (110: + gla_edge_attr,
 * gla_segment_stats, gla_edge_dot, gla_mlp_relu3 / gla_mlp_work_bytes; 120: + gla_comm_*, gla_halo_*, gla_dist_jacobi_sweeps,
 * gla_pmis_split; 130: round 3 (power / Chebyshev / dist entry points); 140: + gla_comm_probe, option "sym",
 * gla_csr_info_t.dia_sym and .dia_full appended at the end of the struct). */
int gla_version(void);
/* Tuning knobs (process-wide): "spmv_variant" (kernel flavour for K = 1), "tile_nnz" (target nnz per row
 * tile for plans created afterwards; 0 = automatic), "compress" (1: value-dependent compression at gla_csr_set_values:
 * a stencil plan whose diagonals are constant needs no value stream at all; otherwise it additionally builds a packed
 * stream -- int16 tile-relative columns + a <= 256-entry value dictionary -- when the matrix allows it; the K = 1
 * kernels then read 3 bytes per nonzero instead of 12, results are bit-identical; default 0), "col16" (1, the
 * default: plans created afterwards also keep the columns as int16 offsets from the first row of the nonzero's row
 * tile (and the row offsets as uint16 relative to the tile's first nonzero), and the K = 1 kernels read those 2 bytes
 * instead of the 4-byte column / row pointer; tiles with a column outside that range -- halo columns of a row-block
 * partition -- are read as int32; results are bit-identical), "dia" (1, the default: when every nonzero of a
 * row-sorted matrix lies on one of <= 16 diagonals col - row -- 5-point, 9-point stencils -- the values are also kept
 * diagonal-major and the K = 1 kernels stream them with NO index stream at all, x[row + offset] being a coalesced
 * load; same products, same summation order: bit-identical), "sym" (1, the default: gla_csr_set_values checks a full
 * stencil plan's diagonal pairs for A_ij == A_ji BIT FOR BIT; a lower diagonal that repeats its upper partner is then not
 * streamed by the K = 1 stencil kernels -- they read the partner's value at the row the entry points to, or keep it from the
 * line above in registers: 3 of a symmetric 5-point matrix's 5 diagonals cross HBM; per pair, bit-identical, any other
 * matrix keeps its diagonals; gla_csr_info_t.dia_sym), "full_mask" (1, the default: on a canonical 5-point plan whose every row
 * has all the neighbours its grid position has -- gla_csr_info_t.dia_full, structure only -- the symmetric march passes skip the
 * 2-byte row masks), "row_sum" (1, the default: K = 1 CSR kernel, a row of 64 nonzeros or
 * more is summed by its wave -- 64 chunks in edge order + a fixed cross-lane tree, a function of the row's own length; 0: one
 * lane per row strictly in edge order), "xcd_chunk",
 * "spmm_variant" (A/B knobs).  The environment variable GLA_OPTIONS="name=value,name=value" sets initial values when
 * the library is loaded. */
int gla_set_option(const char* name, int64_t value);
const char* gla_error_string(int code);

/* ---- plan: COO edge list -> CSR (replaces nothing 1:1; it is the index side of every
 * `torch_scatter.scatter(src, edgeij_pair[0], dim=0, dim_size=n)` call) ----------- */
/* row/col: int64 [nnz] (the two rows of the reference's edgeij_pair [2,nnz]).  Stable-sorts by row
 * when the list is not already row-sorted.  Synchronises `stream` once. */
int gla_csr_create_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                            const int64_t* row, const int64_t* col,
                            gla_stream_t stream, gla_csr_t* out);
int gla_csr_destroy(gla_csr_t plan);

/* Gather edge values into CSR order: val_csr[p] = edge_val[perm[p] * stride] (stride in elements;
 * the reference keeps A_ij in column 0 of edge_attr [nnz, n_e], so stride = n_e). */
int gla_csr_set_values(gla_csr_t plan, const void* edge_val, int64_t stride, int dtype, gla_stream_t stream);

typedef struct {
  int64_t n_rows, n_cols, nnz;
  int32_t sorted_input;     /* 1: edge order == CSR order (perm is identity, not stored) */
  int32_t max_row_len;
  int32_t n_tiles;          /* row tiles of the CSR-stream kernels */
  int32_t tile_nnz;         /* target nnz per tile */
  int32_t lds_elems;        /* staged nnz capacity per workgroup */
  int32_t long_rows;        /* 1: some row exceeds the LDS tile -> CSR-vector kernels are used */
  int32_t val_dtype;        /* GLA_F32 / GLA_F64 / -1 */
  int32_t packed;           /* option "compress": 1 = the 3-byte-per-nonzero packed CSR stream is in use for K = 1;
                             * 2 = stencil with constant diagonals: no matrix stream at all besides a 2-byte row mask */
  int32_t col16;            /* 1: int16 tile-relative columns are in use for K = 1 (option "col16") */
  int32_t col16_wide_tiles; /* row tiles that keep int32 columns (a column out of int16 range) */
  int32_t dia;              /* > 0: the diagonal (stencil) layout with this many diagonals is in use for K = 1 */
  int32_t dia_stride;       /* > 0: the offsets read as a * S + b on a 2-D grid with this line length S (the blocked
                             * multi-rhs kernel and the temporally blocked Jacobi kernel apply); 0: no 2-D reading */
  int64_t dia_row_lo, dia_row_hi;   /* rows [lo, hi) that fit the diagonal set (all rows on one GPU; a row block's
                                     * halo rows at either end do not and stay on the CSR kernels) */
  int32_t dia_A, dia_B;     /* max |a| (grid lines) and max |b| (points) of the 2-D reading */
  int32_t dia_canon5;       /* 1: the reading is the canonical 5-point pattern -S, -1, 0, +1, +S (march kernel, 2 .. 8 sweeps per
                             * pass); 9: the canonical 9-point pattern (a, b) in {-1, 0, 1}^2 (march kernel, 2 .. 6); 0: neither */
  int32_t dia_sym;          /* bit d set: diagonal d (negative offset) repeats its partner of the opposite offset bit for bit
                             * (A_ij == A_ji along it, checked at gla_csr_set_values) and option "sym" is on: kernels that say so
                             * read it from the partner -- a symmetric 5-point matrix streams 3 diagonals instead of 5 */
  int32_t dia_full;         /* 1: canonical 5-point plan in which every row couples to every grid neighbour it has (structure only,
                             * checked at plan creation) and option "full_mask" is on: the symmetric march passes read no row masks */
} gla_csr_info_t;
int gla_csr_info(gla_csr_t plan, gla_csr_info_t* info_host);

/* Copy the plan's index arrays into caller buffers (for parity tests: CSR indexing must be bit-exact).
 * rowptr_out int32 [n_rows+1], col_out int32 [nnz], perm_out int32 [nnz] (CSR position -> edge id; written as
 * the identity when sorted_input); any of them may be NULL. */
int gla_csr_export(gla_csr_t plan, int32_t* rowptr_out, int32_t* col_out, int32_t* perm_out,
                   gla_stream_t stream);

/* Restrict later launches to a subset of row tiles (multi-GPU overlap: interior tiles run while the
 * halo is in flight, boundary tiles after).  which: 0 = all, 1 = tiles whose columns are all < col_limit,
 * 2 = the others.  gla_csr_classify_tiles builds the two lists once per plan. */
int gla_csr_classify_tiles(gla_csr_t plan, int64_t col_limit, gla_stream_t stream,
                           int32_t* n_interior_host, int32_t* n_boundary_host);

/* ---- hot path ----------------------------------------------------------------------------------- */
/* Y = A X.  MatVecGNN.py:43-114 (EdgeUpdate c=A_ij*x_j :66-84, scatter-sum :60, VertexUpdate :95-114). */
int gla_spmm(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* Y, int64_t ldy,
             int which_tiles, gla_stream_t stream);

/* Y = Z + A X (Y may be Z): the coarse-grid correction x + P xc of a V-cycle (VCycle.py:228) without a separate add pass;
 * the row sum is formed first and added to Z afterwards, as torch's `x + P @ xc` rounds. */
int gla_spmm_add(gla_csr_t A, int dtype, int K, const void* Z, int64_t ldz, const void* X, int64_t ldx, void* Y,
                 int64_t ldy, int which_tiles, gla_stream_t stream);

/* Y = M X, M dense row-major [n_rows, n_cols] (leading dimension ldm), X [n_cols, K], Y [n_rows, K]; Y must not alias X.
 * The coarsest-level solve of a deep V-cycle: M = the inverse of the (small) coarsest operator, factored once at setup --
 * in place of the reference's Chebyshev coarse "solve" (VCycle.py:218-223) where the hierarchy is deep enough for an exact
 * one.  Column k of the result does not depend on K (fixed summation order per row). */
int gla_dense_apply(int dtype, int64_t n_rows, int64_t n_cols, const void* M, int64_t ldm, int K, const void* X,
                    int64_t ldx, void* Y, int64_t ldy, gla_stream_t stream);

/* R = B - A X.  GNNResidual.py:43-132 (r_i = b_i - cbar_i, :110). */
int gla_residual(gla_csr_t A, int dtype, int K, const void* B, int64_t ldb, const void* X, int64_t ldx,
                 void* R, int64_t ldr, int which_tiles, gla_stream_t stream);

/* One weighted-Jacobi sweep, Xout = X + (w*(B - A X))/diag, diag [n_rows] (one per row, shared by the K
 * columns).  JacobiGNN.py:52-123 (x_i + w*(b_i - cbar_i)/A_ii, :119).  X has n_cols rows (halo included),
 * Xout/B n_rows. */
int gla_jacobi_sweep(gla_csr_t A, int dtype, int K, const void* diag, const void* B, int64_t ldb,
                     const void* X, int64_t ldx, void* Xout, int64_t ldxo, double w,
                     int which_tiles, gla_stream_t stream);

/* n_iters sweeps ping-ponging between Xout and Xtmp so that the last sweep lands in Xout; X0 is not
 * mutated.  JacobiGNN.forward, JacobiGNN.py:138-148.  Square plans only.  On 2-D stencil plans the sweeps are taken several
 * at a time in one pass over the data (temporal blocking, bit-identical: see gla_jacobi_sweeps_schedule for K = 1; K right-hand
 * sides that fill whole 16-byte vectors take up to five sweeps per pass on canonical 5-point plans, three on 9-point plans).
 * diag == NULL: "A_ii is this plan's matrix diagonal" (what the reference always passes: JacobiGNN.py:160-170, VCycle.py:160-166).
 * Accepted for K = 1 on canonical 5- / 9-point plans whose schedule for n_iters consists of march passes only (every pass of two
 * sweeps or more, option "jacobi_march" on): the passes then take A_ii from the centre coefficient and stream no A_ii array;
 * GLA_ERR_INVALID_ARG otherwise, before anything is enqueued -- call again with the array. */
int gla_jacobi_sweeps(gla_csr_t A, int dtype, int K, const void* diag, const void* B, int64_t ldb,
                      const void* X0, int64_t ldx0, void* Xout, void* Xtmp, int64_t ldw, double w,
                      int n_iters, gla_stream_t stream);
/* JacobiGNN.forward straight from the reference's vertex attribute tensor: vertex_attr [n, 3] row-major = [A_ii, b, x]
 * (leading dimension ld >= 3; JacobiGNN.py:104-106,138-148).  Same result as splitting the columns and calling
 * gla_jacobi_sweeps, but the first temporally blocked pass reads the AoS rows itself and leaves contiguous copies of
 * A_ii and b in diag_work / b_work [n] for the later passes (16 B per row of extra writes instead of a 48 B per row
 * split pass).  The iterate after n_iters sweeps lands in Xout; Xtmp as in gla_jacobi_sweeps.  Returns
 * GLA_ERR_UNSUPPORTED when the first launch of the schedule would not be a fused pass (plan without a 2-D stencil
 * reading, n_iters = 1, option jacobi_fuse = 0): split and use gla_jacobi_sweeps then.
 * Canonical 5-point plans whose passes are all fused (option "jacobi_diag_flag", default 1): no copy of A_ii is made.  The first
 * pass compares vertex_attr[r, 0] with the row's centre coefficient bit for bit and raises a plan-owned flag on a difference; the
 * later passes take A_ii from the centre coefficient they load anyway, or -- flag raised -- from column 0 of the AoS rows.  Same
 * division either way: bit-identical.  diag_work must still hold n elements (other plans use all of it).  Up to 64 calls may be
 * in flight on ONE plan at a time on different streams (calls on one stream are ordered anyway). */
int gla_jacobi_sweeps_aos(gla_csr_t A, int dtype, const void* vertex_attr, int64_t ld, void* diag_work, void* b_work,
                          void* Xout, void* Xtmp, double w, int n_iters, gla_stream_t stream);
/* On 2-D stencil plans (diagonal layout whose offsets read as a * S + b on a grid of line length S) with one
 * contiguous right-hand side, gla_jacobi_sweeps takes several sweeps in ONE pass over the data (temporal blocking: the
 * intermediate iterates stay on chip; same arithmetic per row, bit-identical results).  Canonical 5-point plans
 * (diagonals -S, -1, 0, +1, +S) take 2 .. 8 sweeps per pass, ceil(n_iters / 8) passes of equal size +- 1 (20 = 7 + 7 + 6);
 * canonical 9-point plans ((a, b) in {-1, 0, 1}^2: the Q1 FEM operator) 2 .. 6, by default at most 5 (20 = 4 x 5, 3 = 3);
 * other stencils 5, 4 or 2 (6 = 4 + 2, 3 = 2 + 1).  Options: "jacobi_fuse" 0 off, 1 default, 2 at most two per pass;
 * "jacobi_ts_max" / "jacobi_ts_max9" most sweeps per pass on canonical 5- / 9-point plans (2 .. 8, default 8 / 2 .. 6, default 5); "jacobi_march" 0 = canonical plans use the
 * generic stencil kernel too; "jacobi_lines" grid lines per workgroup, 0 = automatic; "jacobi_min_rows": matrices with
 * fewer rows, default 2^19, keep one sweep per launch, which is faster there.
 * This reports the launches gla_jacobi_sweeps will use for n_iters sweeps, in order: counts4_host[g] launches of
 * sweeps4_host[g] sweeps each, g = 0 .. 3 (unused groups have count 0). */
int gla_jacobi_sweeps_schedule(gla_csr_t A, int dtype, int n_iters, int* counts4_host, int* sweeps4_host);
/* ONE temporally blocked launch: Xout = `sweeps` Jacobi sweeps of X (contiguous vectors, K = 1); sweeps = 2 .. 8 on
 * canonical 5-point plans, 2 .. 6 on canonical 9-point plans, 2, 4 or 5 on other 2-D stencil plans (GLA_ERR_UNSUPPORTED otherwise,
 * and when the plan has no 2-D stencil reading).
 * part: 0 = every block of grid lines.  1 / 2 = the multi-GPU split of the same pass: the first and the last `ghost_lines`
 * grid lines of X (>= sweeps) may still be in flight while part 1 runs -- part 1 is every line block that reads none of them,
 * part 2 the first and the last line block (ghost_lines + sweeps lines high each, so that they are short; on grids of
 * fewer than 2 (ghost_lines + sweeps) + 1 lines part 1 is empty and part 2 everything).  Part 1 + part 2 write exactly what
 * part 0 writes.  gla_jacobi_fused(.., part, ..) is gla_jacobi_fused_parts with ghost_lines = sweeps.
 * diag == NULL (canonical 5-point plans; GLA_ERR_INVALID_ARG elsewhere): A_ii is the plan's own centre diagonal -- what the
 * reference always passes (JacobiGNN.py:160-170) -- and the pass streams no A_ii array. */
int gla_jacobi_fused_parts(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, double w,
                           int sweeps, int part, int ghost_lines, gla_stream_t stream);
int gla_jacobi_fused(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, double w,
                     int sweeps, int part, gla_stream_t stream);
/* `sweeps` Jacobi sweeps AND the residual of the result in ONE pass: Xout = x after the sweeps, Rout = B - A Xout (as
 * gla_jacobi_sweeps followed by gla_residual, bit for bit; what a V-cycle does before it restricts, VCycle.py:196,212).
 * Canonical 5-point plans: sweeps = 1 .. 7; canonical 9-point plans: 1 .. 5; contiguous vectors, K = 1.  GLA_ERR_UNSUPPORTED
 * for other plans / counts, below "jacobi_min_rows" rows, or with "jacobi_fuse" != 1: run the two calls then. */
int gla_jacobi_fused_residual(gla_csr_t A, int dtype, const void* diag, const void* B, const void* X, void* Xout, void* Rout,
                              double w, int sweeps, gla_stream_t stream);
/* The same launch reading A_ii, b and x from the reference's AoS vertex attributes (as gla_jacobi_sweeps_aos' first pass)
 * when the plan covers MORE rows than the tensor: plan row r is row r - row_offset of vertex_attr [n_attr_rows, >= 3]
 * (multi-GPU: the plan spans the rank's rows plus ghost lines, the tensor holds the rank's own rows).  Rows outside the tensor
 * must not be needed by the chosen part (part 1 with ghost_lines covering them); the rows a launched block owns get their
 * A_ii and b copied to diag_work / b_work at plan-row positions, Xout receives the iterate at plan-row positions. */
int gla_jacobi_fused_parts_aos(gla_csr_t A, int dtype, const void* vertex_attr, int64_t ld, int64_t row_offset,
                               int64_t n_attr_rows, void* diag_work, void* b_work, void* Xout, double w, int sweeps, int part,
                               int ghost_lines, gla_stream_t stream);

/* Chebyshev, first degree (two MetaLayers fused): R = B - A X; P = R; Xout = X + alpha*R.
 * ChebyGNN.py:49-163. */
int gla_cheby_first(gla_csr_t A, int dtype, int K, const void* B, int64_t ldb, const void* X, int64_t ldx,
                    void* R, int64_t ldr, void* P, int64_t ldp, void* Xout, int64_t ldxo, double alpha,
                    int which_tiles, gla_stream_t stream);

/* Chebyshev, degree >= 2 step (two MetaLayers fused): R -= alpha_old * (A P); Pout = R + beta*P;
 * X += alpha_new * Pout (X, R updated in place; P is gathered so Pout must not alias it).
 * ChebyGNN.py:166-243. */
int gla_cheby_next(gla_csr_t A, int dtype, int K, const void* P, int64_t ldp, void* Pout, int64_t ldpo,
                   void* R, int64_t ldr, void* X, int64_t ldx, double alpha_old, double alpha_new, double beta,
                   int which_tiles, gla_stream_t stream);
/* n_steps consecutive gla_cheby_next steps enqueued from C (ChebyRelaxGNN's loop, ChebyGNN.py:322-351, without a host
 * round trip per step: a degree-768 coarse solve on a 512^2 level is launch-bound).  Step i gathers from P0 (i even) or P1
 * (i odd) and writes the other; R and X are updated in place; scalars of step i: alpha_old_host[i], alpha_new_host[i],
 * beta_host[i].  After n_steps the newest p is in P0 (n_steps even) or P1 (odd), the previous one in the other. */
int gla_cheby_steps(gla_csr_t A, int dtype, int K, int n_steps, void* P0, int64_t ldp0, void* P1, int64_t ldp1, void* R,
                    int64_t ldr, void* X, int64_t ldx, const double* alpha_old_host, const double* alpha_new_host,
                    const double* beta_host, gla_stream_t stream);

/* The first `deg` (2 .. 4) Chebyshev steps in ONE pass over the data (temporal blocking, as gla_jacobi_sweeps does for
 * Jacobi): Xout, R, P = x, r, p after step `deg` of ChebyRelaxGNN (ChebyGNN.py:287-353) started from (B, X0);
 * alpha_host[k-1] = alpha_k (k = 1 .. deg), beta_host[k-1] = beta_k (k >= 2), the scalars of the global updates
 * (ChebyGNN.py:123-139, :246-285) evaluated by the caller.  Pprev (may be NULL) receives p after step deg - 1, the vector
 * the last step gathers (the z_ij of the returned edge attributes); Bcopy (may be NULL) a copy of B (the module returns
 * [b, x, r, p] and b passes through a register anyway).  Same per-row arithmetic as gla_cheby_first +
 * gla_cheby_next: bit-identical.  2-D stencil plans with at most five diagonals, K a multiple of the 16-byte vector
 * (1, 2, 4 or 8 vectors), 16-byte aligned operands; the canonical 9-point plan with deg <= 3 (K-wide); K = 1 (the reference's
 * own single-vector call) on canonical 5- / 9-point plans, deg <= 4; otherwise GLA_ERR_UNSUPPORTED (fewer steps, or one by one). */
int gla_cheby_fused(gla_csr_t A, int dtype, int K, int deg, const void* B, int64_t ldb, const void* X0, int64_t ldx0,
                    void* Xout, int64_t ldxo, void* R, int64_t ldr, void* P, int64_t ldp, void* Pprev, int64_t ldpp,
                    void* Bcopy, int64_t ldbc, const double* alpha_host, const double* beta_host, gla_stream_t stream);
/* The continuation of gla_cheby_fused: steps k + 1 .. k + deg (2 <= deg <= 4; three K-wide on 9-point plans) in ONE pass,
 * started from (Rin, Xin, Pin) = r, x, p after step k.  alpha_prev = alpha_k; alpha_host[i], beta_host[i] = alpha, beta of
 * step k + 1 + i.  Out of place (Xout, R, P differ from the inputs: the workgroups' strips overlap); Pprev as in
 * gla_cheby_fused.  Canonical 5- / 9-point plans only (the march kernels); GLA_ERR_UNSUPPORTED otherwise. */
int gla_cheby_fused_next(gla_csr_t A, int dtype, int K, int deg, const void* Rin, int64_t ldri, const void* Xin, int64_t ldxi,
                         const void* Pin, int64_t ldpi, void* Xout, int64_t ldxo, void* R, int64_t ldr, void* P, int64_t ldp,
                         void* Pprev, int64_t ldpp, double alpha_prev, const double* alpha_host, const double* beta_host,
                         gla_stream_t stream);

/* Column sums of X .* Y over `n` rows -> out[K] (device, same dtype).  PowerMethodGNN.py:50-61
 * (torch.sum(vertex_attr[:,1]) with y = b*b :124 or y = b*cbar :233).  `work` needs gla_dot_work_bytes(n,K). */
int64_t gla_dot_work_bytes(int64_t n, int K, int dtype);
int gla_col_dot(int dtype, int64_t n, int K, const void* X, int64_t ldx, const void* Y, int64_t ldy,
                void* out, void* work, gla_stream_t stream);

/* Y[i,k] = X[i,k] / sqrt(sumsq[k]) (op=0, PowerMethodGNN.py:160-207: n = sqrt(ybar); b = b/n)
 * out[k] = num[k] / den[k] is done by the caller on K scalars. */
int gla_col_scale_rsqrt(int dtype, int64_t n, int K, const void* X, int64_t ldx, const void* sumsq,
                        void* Y, int64_t ldy, gla_stream_t stream);

/* One FUSED power-method step on K columns (PowerMethodGNN.py:86-207, the three MetaLayers of an iteration):
 *   Y = A * (X ./ nrm_in)      (nrm_in [K] = sqrt of the previous sums and rnrm_in [K] = 1 ./ nrm_in, both NULL: X
 *                               used as is; every gathered value is divided through the reciprocal with one
 *                               Markstein correction step = the correctly rounded b / n of the reference; divisors
 *                               whose significand is all ones, the theorem's one exception, take a true division)
 *   sumsq_out[k] = sum_i Y[i,k]^2   (workgroup partials in `work`, then a fixed-order second stage)
 * work: gla_power_work_bytes(A, K, dtype).  With which_tiles = 1 (interior) the second stage is deferred to the
 * which_tiles = 2 (boundary) call.  Returns GLA_ERR_UNSUPPORTED for plans with long rows or odd K > 1 (use
 * gla_spmm + gla_col_dot + gla_col_scale_rsqrt). */
int64_t gla_power_work_bytes(gla_csr_t A, int K, int dtype);
int gla_power_step(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, const void* nrm_in,
                   const void* rnrm_in, void* Y, int64_t ldy, void* sumsq_out, void* work, int which_tiles,
                   gla_stream_t stream);
/* The tail of PowerMethodGNN.forward in ONE pass (PowerMethodGNN.py:187-294): Bout = Y / nrm (layer 3 of the last iteration,
 * bit-identical to gla_col_scale_rsqrt), B2out = Bout^2, nA_out[k] = sum_i Bout[i,k] (A Bout)[i,k] and bb_out[k] = sum_i
 * Bout[i,k]^2 -- A Bout itself is never stored (unfused: a scale pass, an SpMM, two dot products and a square: 3.5 ms at
 * 4096^2 x 16, fused 1.5 ms).  nrm_in / rnrm_in [K] as for gla_power_step; work: 2 x gla_power_work_bytes.  The sums go
 * through block partials (fixed order, but not gla_col_dot's): they agree with the unfused path to rounding.
 * 2-D stencil plans with K a multiple of the 16-byte vector; GLA_ERR_UNSUPPORTED otherwise (run the unfused tail). */
int gla_power_rayleigh(gla_csr_t A, int dtype, int K, const void* Y, int64_t ldy, const void* nrm_in, const void* rnrm_in,
                       void* Bout, int64_t ldb, void* B2out, int64_t ldb2, void* nA_out, void* bb_out, void* work,
                       gla_stream_t stream);

/* Temporally blocked power iteration (PowerMethodGNN.py:296-334, the iteration block x `iters`): A is linear, so `iters`
 * iterations started from a normalised vector are ONE pass of the K-wide march with the intermediate norms as by-products:
 * Y = A^iters (X ./ nrm_in) (nrm_in == NULL: X as given; the quotient is the IEEE one, as gla_power_step),
 * sumsq_out[k] = sum_i (A^(iters-1) (X ./ nrm_in))[i,k]^2 and sumsq_out[K + k] = sum_i Y[i,k]^2: the sums of the last two
 * levels.  The norm the reference reports after the last iteration is sqrt(sumsq[K + k] / sumsq[k]) (iters == 1: the operand
 * is the normalised vector itself, sqrt(sumsq[K + k])); the next pass divides by sqrt(sumsq[K + k]).  Iterates agree with gla_power_step to rounding (the reference rounds after every division, this
 * path after every pass).  Canonical 5-point plans: iters <= 5, 9-point: <= 3 (gla_power_fused_max_iters; 0: no kernel for this
 * plan / K -- K = 1 or a multiple of the 16-byte vector up to 8 vectors).  X and Y must not alias.  work:
 * gla_power_fused_work_bytes(A, dtype, K, iters). */
int gla_power_fused_max_iters(gla_csr_t A, int dtype, int K);
int64_t gla_power_fused_work_bytes(gla_csr_t A, int dtype, int K, int iters);
int gla_power_fused(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, const void* nrm_in, void* Y, int64_t ldy,
                    void* sumsq_out, void* work, int iters, gla_stream_t stream);

/* The aggregation primitive on its own: out[i,k] = reduce over the edges e with row(e) = i of src[e,k], src in EDGE
 * order [nnz, K].  op: 0 sum (edge order, = the reference's sequential scatter-add), 1 max, 2 min, 3 mean; rows
 * without edges give 0.  torch_scatter.scatter(src, edgeij_pair[0], dim=0, dim_size=n, reduce=...), e.g.
 * JacobiGNN.py:66, SOCClassicGNN.py:69, DirectInterpGNN.py:89-92.  Only the plan's index side is used. */
int gla_segment_reduce(gla_csr_t A, int dtype, int K, const void* src, int64_t ld_src, int op, void* out,
                       int64_t ld_out, gla_stream_t stream);

/* The four row statistics of the trainable-Jacobi model in one pass: out[i, 0:K] = min, [K:2K] = mean, [2K:3K] = sum,
 * [3K:4K] = max over the edges of row i of src [nnz, K] (edge order; empty rows give 0; ld_out >= 4K).
 * TrainableJacobiDiag/TrainableJacobiGNN.py:53-70 (four torch_scatter.scatter calls + torch.cat). */
int gla_segment_stats(gla_csr_t A, int dtype, int K, const void* src, int64_t ld_src, void* out, int64_t ld_out,
                      gla_stream_t stream);

/* Edge-wise dot products on the plan's pattern: out_edge[e] = sum_k G[row(e), k] * X[col(e), k], EDGE order [nnz].
 * This is dL/dA_e of Y = A X given G = dL/dY: the backward of the `A @ Y` products the reference differentiates
 * through with torch autograd (TrainableJacobiDiag/loss.py:92-128, train.py:105).  dL/dX = A^T G is gla_spmm on a
 * plan of the flipped edge list.  Only the plan's index side is used. */
int gla_edge_dot(gla_csr_t A, int dtype, int K, const void* G, int64_t ldg, const void* X, int64_t ldx,
                 void* out_edge, gla_stream_t stream);

/* Inference of the learned-Jacobi vertex model: out[i] = W3 relu(W2 relu(W1 x_i + b1) + b2) + b3, x_i = X[i, 0:in_dim]
 * (row-major, leading dimension ldx), weights in torch.nn.Linear layout (W1 [h1, in_dim], W2 [h2, h1], W3 [1, h2]).
 * TrainableJacobiDiag/TrainableJacobiGNN.py:86-91,110-111 (Sequential(Linear(5,50), ReLU, Linear(50,20), ReLU,
 * Linear(20,1)) applied to every vertex).  Limits: in_dim <= 8, h1 <= 512, h2 <= 32.  `work`: device scratch of
 * gla_mlp_work_bytes(h1) bytes (the packed weights).  Sums use fused multiply-adds in index order (torch's GEMM
 * order is unspecified): agrees with the reference to fp rounding, not bit for bit. */
int64_t gla_mlp_work_bytes(int h1);
int gla_mlp_relu3(int dtype, int64_t n, int in_dim, int h1, int h2, const void* X, int64_t ldx, const void* W1,
                  const void* b1, const void* W2, const void* b2, const void* W3, const void* b3, void* out,
                  void* work, gla_stream_t stream);

/* Split a row-major attribute tensor src [n, a] (a <= 8, leading dimension ld) into `a` contiguous vectors
 * cols_host[c][i] = src[i*ld + c] in one pass (cols_host: HOST array of a device pointers).  The reference keeps
 * vertex attributes as AoS [A_ii, b, x] / [b, x] / ... (JacobiGNN.py:104-106); the kernels read SoA. */
int gla_split_columns(int dtype, int64_t n, int a, const void* src, int64_t ld, void* const* cols_host,
                      gla_stream_t stream);

/* Up to 8 vector copies in ONE launch: dst_i[j] = src_i[j * src_stride_i], j < count_i (HOST arrays of device pointers /
 * counts / element strides; src_stride_host = NULL: all contiguous).  The multi-GPU route's ghost-line fills and the few AoS
 * lines its edge blocks need: a dozen small hipMemcpyAsync calls (14 us each) become two launches. */
int gla_copy_batch(int dtype, int n_copies, const void* const* src_host, void* const* dst_host, const int64_t* count_host,
                   const int64_t* src_stride_host, gla_stream_t stream);

/* Edge products in EDGE order: C[e,k] = A_e * X[col_e,k] -- the `c_ij`/`z_ij` column every EdgeUpdate
 * returns (e.g. JacobiGNN.py:83-88); materialised only when a caller asks for edge_attr. */
int gla_edge_products(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* C, int64_t ldc,
                      gla_stream_t stream);

/* The edge attributes every EdgeUpdate returns, in one pass: out[e, 0] = A_e, out[e, 1 + k] = A_e * X[col(e), k],
 * EDGE order, [nnz, 1 + K] with leading dimension ldo >= 1 + K.  The reference's
 * `torch.cat([A_ij, c_ij], 1)` (MatVecGNN.py:83-84, JacobiGNN.py:88-89, ChebyGNN.py:69-70, PowerMethodGNN.py:105-106). */
int gla_edge_attr(gla_csr_t A, int dtype, int K, const void* X, int64_t ldx, void* out, int64_t ldo,
                  gla_stream_t stream);

/* Classical strength of connection on the OFF-DIAGONAL plan: v_i = max_j(-A_ij) (0 for empty rows),
 * S_e = relu(-A_e / v_row(e) - theta), S in edge order.  SOCClassicGNN.py:50-147. */
int gla_soc_classic(gla_csr_t A_od, int dtype, double theta, void* v_out, void* S_out, gla_stream_t stream);

/* Direct interpolation on the off-diagonal plan: gamma_i = sum_k A_ik / sum_k (A_ik*S_ik*C_k),
 * alpha_i = (1/A_ii)*gamma_i, w_e = (1 - C_i) * (-A_e * alpha_i); S, w in edge order, diag [n_rows], cf [n_cols]
 * (square plans: the same n; a row block whose columns are numbered [own rows | halo] passes the flags of its halo
 * columns behind its own).  DirectInterpGNN.py:50-174.  alpha_out may be NULL. */
int gla_direct_interp(gla_csr_t A_od, int dtype, const void* diag, const void* cf, const void* S,
                      void* w_out, void* alpha_out, gla_stream_t stream);

/* SA strength of connection, S_e = A_e^2 / (A_ii * A_jj) in edge order.  SOCSAGNN.py:49-71. */
int gla_soc_sa(gla_csr_t A_od, int dtype, const void* diag, void* S_out, gla_stream_t stream);

/* ---- multi-GPU: row-block partition, halo all-gather over RCCL / xGMI ------------------------------------- */
/* The reference has no distributed code; BASELINE.json (north_star, configs 4-5) defines this
 * layer: one process per GPU, contiguous row blocks, one all-gather of the halo vertices per SpMV.  A rank's X
 * buffer is [n_in | nranks * S] x K (leading dimension K): its own n_in input entries, then the all-gathered send
 * slots of every rank (S = longest send list; gnn_applied_linear_algebra_amd/dist.py builds the column
 * renumbering).  RCCL is bound at run time (dlopen librccl.so.1); without it these return GLA_ERR_UNSUPPORTED.
 * RCCL errors are returned as -(3000 + ncclResult_t). */
typedef struct gla_comm* gla_comm_t;
/* Rank 0 obtains a 128-byte id (ncclGetUniqueId) and distributes it by any means; every rank then calls
 * gla_comm_create on its own device (ncclCommInitRank + a private communication stream and two events). */
int gla_comm_unique_id(void* id128_host);
/* GLA_OK when librccl is loadable through dlopen with every symbol the gla_comm_* entry points use, GLA_ERR_UNSUPPORTED
 * otherwise (also when the environment sets GLA_RCCL_DISABLE=1: test hook).  Touches no GPU and creates nothing: a group of
 * ranks agrees on the outcome of this call BEFORE any of them enters the collective ncclCommInitRank of gla_comm_create --
 * a rank on which RCCL is missing would otherwise leave the others blocked in there (dist.py, LibComm.try_for_group). */
int gla_comm_probe(void);
int gla_comm_create(const void* id128_host, int rank, int nranks, gla_comm_t* out);
int gla_comm_destroy(gla_comm_t comm);
int gla_comm_ranks(gla_comm_t comm, int* rank_host, int* nranks_host);
/* Start the halo exchange of X: pack rows send_idx[0..n_send) into send_buf [S, K] on `stream`, then all-gather
 * send_buf into X[n_in:] on the communicator's stream.  Returns at once; gla_halo_end makes `stream` wait for the
 * gathered halo (stream-ordered, the host never blocks).  Kernels that read only X[:n_in] (interior row tiles,
 * which_tiles = 1) may be enqueued in between. */
int gla_halo_begin(gla_comm_t comm, int dtype, int K, void* X, int64_t n_in, const int32_t* send_idx,
                   int64_t n_send, int64_t S, void* send_buf, gla_stream_t stream);
int gla_halo_end(gla_comm_t comm, gla_stream_t stream);
/* All-gather of `count` values per rank (send [count] -> recv [nranks * count]) on the communicator's stream once the
 * work already enqueued on `stream` is done; returns at once, gla_halo_end makes `stream` wait for the result.  Used
 * for the ghost-line slabs of the temporally blocked multi-GPU sweep (dist.py). */
int gla_comm_all_gather_begin(gla_comm_t comm, int dtype, const void* send, void* recv, int64_t count,
                              gla_stream_t stream);
/* The same all-gather IN LINE on `stream` (no side stream, no gla_halo_end): for an exchange whose result the very next
 * kernel needs -- the [b, x] ghost lines before a V-cycle level's smoothing pass (dist_amg.py). */
int gla_comm_all_gather(gla_comm_t comm, int dtype, const void* send, void* recv, int64_t count, gla_stream_t stream);
/* In-place sum over the ranks of `count` device values, stream-ordered (power method: the K resp. 2K partial sums
 * of PowerMethodGNN.py:50-61 vertex_to_global_aggregation). */
int gla_comm_all_reduce_sum(gla_comm_t comm, int dtype, void* buf, int64_t count, gla_stream_t stream);
/* n_iters Jacobi sweeps of this rank's row block with the halo exchange of every sweep, enqueued entirely from the
 * library (JacobiGNN.py:138-148 on the rank's rows).  Xa / Xb: [n_in + nranks * S, K] buffers; Xa[:n_rows] holds x0;
 * the result is in Xa[:n_rows] for even n_iters, else in Xb[:n_rows].  overlap = 1: interior tiles run while the
 * halo is in flight (classify the plan with gla_csr_classify_tiles(plan, n_in) first). */
int gla_dist_jacobi_sweeps(gla_csr_t A, gla_comm_t comm, int dtype, int K, const void* diag, const void* B,
                           int64_t ldb, void* Xa, void* Xb, int64_t n_in, const int32_t* send_idx, int64_t n_send,
                           int64_t S, void* send_buf, double w, int n_iters, int overlap, gla_stream_t stream);

/* JacobiGNN.forward on a row block of whole grid lines extended by `ghost_lines` ghost lines either side, temporally blocked:
 * passes[i] sweeps in pass i with ONE all-gather of every rank's first / last ghost lines per pass, the whole call enqueued from
 * C on two streams (interior line blocks on `main_stream`; ghost fill, the two edge line blocks and the NEXT pass's exchange on
 * `side_stream`).  vertex_attr: the rank's own AoS rows [n_local, >= 3] = [A_ii, b, x] (plan row r = tensor row r - row_offset);
 * diag_ext / b_ext / xa / xb / x_last: extended work vectors (plan rows), the result is in x_last at plan-row positions; send:
 * 6 m elements, recv: nranks x 6 m (m = ghost_lines x grid line length).  up / down: a neighbour above / below exists (its rank:
 * up_src / down_src).  Canonical 5-point plans, passes of 2 .. ghost_lines sweeps; GLA_ERR_UNSUPPORTED otherwise.  The multi-GPU
 * form of JacobiGNN.py:138-148 (BASELINE.json north_star; the reference has no distributed code). */
int gla_dist_jacobi_temporal(gla_csr_t A, gla_comm_t c, int dtype, const void* vertex_attr, int64_t ld, int64_t row_offset,
                             int64_t n_local, void* diag_ext, void* b_ext, void* xa, void* xb, void* x_last, void* send,
                             void* recv, int ghost_lines, int up, int down, int up_src, int down_src, double w,
                             const int* passes, int n_passes, gla_stream_t main_stream, gla_stream_t side_stream);

/* The same call captured ONCE into a HIP graph (two streams owned by the graph + the communicator's stream; RCCL's all-gathers
 * and the A_ii flag's memset node included) and replayed with one launch per forward(): the eager call is ~10 enqueues and
 * ~90 us of host time per pass, which bounds a rank once its block is small (BASELINE config 4 on 8 GPUs: ~100 us of GPU time
 * per pass).  Runs the sequence once eagerly first (code loads, the plan's flag words), then captures it.  Operand ADDRESSES
 * are frozen into the graph: keep the buffers, refresh vertex_attr in place, read the result from x_last after
 * gla_graph_launch.  Every rank of the communicator must create and launch alike.  JacobiGNN.py:138-148, multi-GPU form. */
typedef struct gla_graph* gla_graph_t;
int gla_dist_jacobi_temporal_graph(gla_csr_t A, gla_comm_t c, int dtype, const void* vertex_attr, int64_t ld, int64_t row_offset,
                                   int64_t n_local, void* diag_ext, void* b_ext, void* xa, void* xb, void* x_last, void* send,
                                   void* recv, int ghost_lines, int up, int down, int up_src, int down_src, double w,
                                   const int* passes, int n_passes, gla_graph_t* out);
int gla_graph_launch(gla_graph_t g, gla_stream_t stream);
int gla_graph_destroy(gla_graph_t g);

/* ---- builders ----------------------------------------------------------------------------------- */
/* 5-point negative Laplacian on an Ny x Nx grid (reference: Ny == Nx == N), COO sorted row-major with
 * ascending columns: row/col int64 [nnz], val [nnz], nnz = 5*Nx*Ny - 2*Nx - 2*Ny.
 * UtilsGNN.py:53-67 (laplacianfun_torch).  row_offset shifts vertex ids (row blocks of a larger grid):
 * rows [row_lo, row_hi) of the global grid are emitted; pass 0, Nx*Ny for the whole matrix. */
int64_t gla_laplacian2d_nnz(int64_t Nx, int64_t Ny, int64_t row_lo, int64_t row_hi);
int gla_laplacian2d_coo(int64_t Nx, int64_t Ny, int64_t row_lo, int64_t row_hi,
                        int64_t* row, int64_t* col, void* val, int dtype, gla_stream_t stream);

/* Order-preserving removal of self loops.  UtilsGNN.py:69-72 (PyG remove_self_loops).
 * Step 1 returns the kept count (synchronises); step 2 writes the compacted arrays.
 * `work` needs gla_compact_work_bytes(nnz). */
int64_t gla_compact_work_bytes(int64_t nnz);
int gla_remove_diag_count(int64_t nnz, const int64_t* row, const int64_t* col, void* work,
                          int64_t* kept_host, gla_stream_t stream);
int gla_remove_diag_apply(int64_t nnz, const int64_t* row, const int64_t* col, const void* val,
                          int64_t val_stride, int dtype, const void* work,
                          int64_t* row_out, int64_t* col_out, void* val_out, gla_stream_t stream);

/* ---- V-cycle setup (device-side sparse algebra) ----------------------------------------------------- */
/* Coalesce a COO edge list: sort by (row, col), add duplicates in their original order.
 * UtilsGNN.py:74-78 (coo_to_gnn_input -> A.coalesce()).  Outputs need room for nnz entries; the number of
 * unique entries is returned through nnz_out_host (synchronises). */
int gla_coo_coalesce(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* row, const int64_t* col,
                     const void* val, int64_t val_stride, int dtype, int64_t* row_out, int64_t* col_out,
                     void* val_out, int64_t* nnz_out_host, gla_stream_t stream);

/* C = A * B for two plans holding values of `dtype`, as a coalesced COO list (row-major, ascending columns).
 * VCycle.py:209 (Ac = P.t() @ (A @ P)), also R = P^T via a plan of the flipped edge list.
 * Step 1 returns the number of scalar products (the size the output buffers need); step 2 fills
 * row_out / col_out / val_out [n_products] and returns the number of entries of C.  Both synchronise.
 * Duplicates are added in a defined order: A's CSR order, then B's. */
int gla_spgemm_count(gla_csr_t A, gla_csr_t B, int64_t* n_products_host, gla_stream_t stream);
/* gla_coo_coalesce / gla_spgemm* / gla_pmis_split keep their scratch in one grow-only pool per device; the setup
 * entry points serialise on one lock (safe to call from several threads / for several devices).  This frees the pools. */
int gla_release_scratch(void);

/* Deterministic PMIS-style C/F splitting on the OFF-DIAGONAL plan; strong_edge[e] != 0 says vertex row(e) strongly
 * depends on col(e) (EDGE order, e.g. S_ij > 0 of gla_soc_classic); split_out int64 [n], 1 = C point.  Stands where
 * the reference calls pyamg.classical.split.CLJP (VCycle.py:103-116, DirectInterpGNN.py:194) -- upstream is randomised,
 * so the rule is this library's own (csrc/gla_sparse.hip; a CPU restatement lives with the test oracle):
 * weight = (influence count, fixed hash of the vertex id), rounds of "local weight maxima of the undecided strong graph
 * become C, undecided vertices depending on a C point become F".  Synchronises once per round (rounds_host, may be
 * NULL, receives their number). */
int gla_pmis_split(gla_csr_t A_od, const uint8_t* strong_edge, int64_t* split_out, int max_rounds,
                   int* rounds_host, gla_stream_t stream);
int gla_spgemm(gla_csr_t A, gla_csr_t B, int dtype, int64_t n_products, int64_t* row_out, int64_t* col_out,
               void* val_out, int64_t* nnz_out_host, gla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLA_H */
