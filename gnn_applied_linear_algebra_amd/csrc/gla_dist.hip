// libgla_hip.so -- multi-GPU halo exchange and the library-side distributed sweep loop (gfx950 / MI355X only).
//
// The reference has no distributed code; BASELINE.json defines this layer: contiguous row blocks, one
// process per GPU, and ONE exchange step per SpMV -- an RCCL all-gather of every rank's send slots straight into the
// tail of the X buffer the kernels gather from (layout in gnn_applied_linear_algebra_amd/dist.py).
//
// Why a library-side loop: driven from Python, a sweep costs index_select + all_gather + two ctypes launches +
// work.wait() on the host -- more than the ~0.12 ms a rank's sweep takes at BASELINE config 4 on 8 GPUs.  Here a sweep
// is eight runtime calls from C on two streams with event ordering, and nothing synchronises the host:
//
//   main stream   pack(t) ──ev_pack──┐  interior(t) ───────────── wait ev_halo ── boundary(t) ── pack(t+1) ...
//   comm stream                      └─ wait ── ncclAllGather(t) ──ev_halo──┘
//
// RCCL is bound at run time (dlopen of librccl.so.1: the instance the process already has when torch is loaded), so
// single-GPU hosts of libgla_hip.so carry no RCCL dependency.
#include "gla_internal.h"
#include <vector>

#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

// the slice of the RCCL API used here (signatures of rccl.h; ncclUniqueId is a 128-byte opaque struct passed by value)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_comm_destroy)(Comm);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0;      // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* off = getenv("GLA_RCCL_DISABLE");      // test hook: this process behaves as if librccl could not be loaded
  if (off && off[0] == '1') return;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.handle, "ncclAllGather");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.error_string = (fn_error_string)dlsym(g_rccl.handle, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather &&
              g_rccl.all_reduce;
}

inline bool rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  return g_rccl.ok;
}

#define GLA_RCCL_CHECK(expr)                     \
  do {                                           \
    int _r = (expr);                             \
    if (_r != 0) return -(3000 + _r);            \
  } while (0)

// send[i, :] = X[send_idx[i], :] for the n_send send slots (K contiguous values each)
template <typename T>
__global__ __launch_bounds__(GLA_BLOCK) void halo_pack_kernel(const T* __restrict__ X, int64_t ldx,
                                                              const int32_t* __restrict__ send_idx, int64_t n_send,
                                                              int K, T* __restrict__ send) {
  const int64_t total = n_send * K;
  for (int64_t i = (int64_t)blockIdx.x * GLA_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GLA_BLOCK) {
    const int64_t s = i / K;
    const int k = (int)(i - s * K);
    send[i] = X[(int64_t)send_idx[s] * ldx + k];
  }
}

}  // namespace

struct gla_comm {
  Comm comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // gla_dist_jacobi_temporal: compute stream <-> second stream
};

// Rows [lo0, hi0) and [lo1, hi1) of an extended block's A_ii copy against the plan's centre diagonal, bit for bit: the rows the
// two edge line blocks read (hand copies of own rows + the neighbours' ghost lines), which the AoS interior launch does not see.
// A difference raises the call's flag (gla_jacobi_dg_slot); the march passes then read the copy instead of the coefficient.
template <typename U>
__global__ __launch_bounds__(256) void dg_check_kernel(const U* __restrict__ d_ext, const U* __restrict__ centre, int64_t lo0,
                                                       int64_t hi0, int64_t lo1, int64_t hi1, int* __restrict__ flag, int epoch) {
  const int64_t n0 = hi0 - lo0, n1 = hi1 - lo1;
  bool mis = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i < n0 ? lo0 + i : lo1 + (i - n0);
    mis = mis || d_ext[r] != centre[r];
  }
  if (__any(mis) && (threadIdx.x & 63) == 0) atomicExch(flag, epoch);
}

// The flag of the call is raised: rows [lo, hi) of the A_ii copy are filled from column 0 of the AoS rows after all (the AoS
// interior launch of a flagged call does not write them); otherwise every workgroup leaves at once.
template <typename U>
__global__ __launch_bounds__(256) void dg_repair_kernel(const U* __restrict__ va, int64_t ldva, int64_t off, U* __restrict__ d_ext,
                                                        int64_t lo, int64_t hi, const int* __restrict__ flag, int epoch) {
  if (__builtin_amdgcn_readfirstlane(*flag) != epoch) return;
  for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) d_ext[r] = va[(r - off) * ldva];
}

extern "C" {

int gla_comm_unique_id(void* id_host) {
  if (!id_host) return GLA_ERR_INVALID_ARG;
  if (!rccl_ready()) return GLA_ERR_UNSUPPORTED;
  UniqueId id;
  GLA_RCCL_CHECK(g_rccl.get_unique_id(&id));
  memcpy(id_host, &id, sizeof(id));
  return GLA_OK;
}

int gla_comm_probe(void) { return rccl_ready() ? GLA_OK : GLA_ERR_UNSUPPORTED; }

int gla_comm_create(const void* id_host, int rank, int nranks, gla_comm_t* out) {
  if (!id_host || !out || nranks < 1 || rank < 0 || rank >= nranks) return GLA_ERR_INVALID_ARG;
  if (!rccl_ready()) return GLA_ERR_UNSUPPORTED;
  // Everything that can fail locally comes BEFORE the collective ncclCommInitRank (a rank that left early would leave the
  // others blocked in it), and a failure after it releases what was created, the communicator included.
  int device = 0;
  GLA_HIP_CHECK(hipGetDevice(&device));
  gla_comm* c = new gla_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    gla_comm_destroy(c);
    return -(1000 + (int)e);
  }
  UniqueId id;
  memcpy(&id, id_host, sizeof(id));
  const int rc = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);
  if (rc != 0) {
    c->comm = nullptr;
    gla_comm_destroy(c);
    return -(3000 + rc);
  }
  *out = c;
  return GLA_OK;
}

int gla_comm_destroy(gla_comm_t c) {
  if (!c) return GLA_OK;
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm && g_rccl.ok) g_rccl.comm_destroy(c->comm);
  if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return GLA_OK;
}

int gla_comm_ranks(gla_comm_t c, int* rank_host, int* nranks_host) {
  if (!c) return GLA_ERR_INVALID_ARG;
  if (rank_host) *rank_host = c->rank;
  if (nranks_host) *nranks_host = c->nranks;
  return GLA_OK;
}

// X: [n_in + nranks * S, K] row-major, leading dimension K.  Packs this rank's send slots (rows send_idx[0..n_send)
// of X, padded to S) and starts the all-gather into X[n_in:] on the communicator's stream; returns at once.
int gla_halo_begin(gla_comm_t c, int dtype, int K, void* X, int64_t n_in, const int32_t* send_idx, int64_t n_send,
                   int64_t S, void* send_buf, gla_stream_t stream) {
  if (!c || !X || K < 1 || n_in < 0 || n_send < 0 || S < n_send || (dtype != GLA_F32 && dtype != GLA_F64))
    return GLA_ERR_INVALID_ARG;
  if (S == 0) return GLA_OK;
  if (!send_buf || (n_send > 0 && !send_idx)) return GLA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == GLA_F64 ? 8 : 4;
  if (n_send > 0) {
    int64_t blocks = (n_send * K + GLA_BLOCK - 1) / GLA_BLOCK;
    if (blocks > 1024) blocks = 1024;
    if (dtype == GLA_F64)
      hipLaunchKernelGGL((halo_pack_kernel<double>), dim3((unsigned)blocks), dim3(GLA_BLOCK), 0, st, (const double*)X,
                         (int64_t)K, send_idx, n_send, K, (double*)send_buf);
    else
      hipLaunchKernelGGL((halo_pack_kernel<float>), dim3((unsigned)blocks), dim3(GLA_BLOCK), 0, st, (const float*)X,
                         (int64_t)K, send_idx, n_send, K, (float*)send_buf);
    GLA_LAUNCH_CHECK();
  }
  GLA_HIP_CHECK(hipEventRecord(c->ev_pack, st));
  GLA_HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
  GLA_RCCL_CHECK(g_rccl.all_gather(send_buf, (char*)X + (size_t)n_in * K * esz, (size_t)S * K,
                                   dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm, c->comm_stream));
  GLA_HIP_CHECK(hipEventRecord(c->ev_halo, c->comm_stream));
  return GLA_OK;
}

int gla_comm_all_gather_begin(gla_comm_t c, int dtype, const void* send, void* recv, int64_t count,
                              gla_stream_t stream) {
  if (!c || !send || !recv || count < 0 || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (count == 0) return GLA_OK;
  GLA_HIP_CHECK(hipEventRecord(c->ev_pack, (hipStream_t)stream));
  GLA_HIP_CHECK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
  GLA_RCCL_CHECK(g_rccl.all_gather(send, recv, (size_t)count, dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm,
                                   c->comm_stream));
  GLA_HIP_CHECK(hipEventRecord(c->ev_halo, c->comm_stream));
  return GLA_OK;
}

int gla_comm_all_gather(gla_comm_t c, int dtype, const void* send, void* recv, int64_t count, gla_stream_t stream) {
  if (!c || !send || !recv || count < 0 || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (count == 0) return GLA_OK;
  GLA_RCCL_CHECK(g_rccl.all_gather(send, recv, (size_t)count, dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm,
                                   (hipStream_t)stream));
  return GLA_OK;
}

// Makes `stream` wait for the all-gather started by the last gla_halo_begin / gla_comm_all_gather_begin (stream-ordered;
// the host does not block).
int gla_halo_end(gla_comm_t c, gla_stream_t stream) {
  if (!c) return GLA_ERR_INVALID_ARG;
  GLA_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, c->ev_halo, 0));
  return GLA_OK;
}

// Sum of `count` values over the ranks, in place, stream-ordered on `stream` (power method: K resp. 2K partial sums).
int gla_comm_all_reduce_sum(gla_comm_t c, int dtype, void* buf, int64_t count, gla_stream_t stream) {
  if (!c || !buf || count < 0 || (dtype != GLA_F32 && dtype != GLA_F64)) return GLA_ERR_INVALID_ARG;
  if (count == 0) return GLA_OK;
  GLA_RCCL_CHECK(g_rccl.all_reduce(buf, buf, (size_t)count, dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, NCCL_SUM,
                                   c->comm, (hipStream_t)stream));
  return GLA_OK;
}

// n_iters Jacobi sweeps of a row block (JacobiGNN.py:138-148 on this rank's rows) with the halo all-gather of every
// sweep, all enqueued from here.  Xa / Xb: [n_in + nranks * S, K] buffers (ld = K); Xa[:n_rows] holds x0; the result
// is in Xa[:n_rows] when n_iters is even, else in Xb[:n_rows].  overlap = 1: interior tiles run while the halo is in
// flight (the plan must have been classified with gla_csr_classify_tiles(col_limit = n_in)); 0: exchange, then all tiles.
int gla_dist_jacobi_sweeps(gla_csr_t A, gla_comm_t c, int dtype, int K, const void* diag, const void* B, int64_t ldb,
                           void* Xa, void* Xb, int64_t n_in, const int32_t* send_idx, int64_t n_send, int64_t S,
                           void* send_buf, double w, int n_iters, int overlap, gla_stream_t stream) {
  if (!A || !c || !Xa || !Xb || Xa == Xb || n_iters < 0) return GLA_ERR_INVALID_ARG;
  void* cur = Xa;
  void* nxt = Xb;
  const bool exchange = S > 0;      // (a 1-rank communicator with S > 0 gathers its own slots: the self-halo test)
  for (int it = 0; it < n_iters; ++it) {
    int rc;
    if (!exchange) {
      rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, cur, K, nxt, K, w, 0, stream);
      if (rc) return rc;
    } else {
      rc = gla_halo_begin(c, dtype, K, cur, n_in, send_idx, n_send, S, send_buf, stream);
      if (rc) return rc;
      if (overlap) {
        rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, cur, K, nxt, K, w, 1, stream);
        if (rc) return rc;
      }
      rc = gla_halo_end(c, stream);
      if (rc) return rc;
      rc = gla_jacobi_sweep(A, dtype, K, diag, B, ldb, cur, K, nxt, K, w, overlap ? 2 : 0, stream);
      if (rc) return rc;
    }
    void* t = cur;
    cur = nxt;
    nxt = t;
  }
  return GLA_OK;
}

// JacobiGNN.forward (JacobiGNN.py:138-148) on a row block of whole grid lines EXTENDED by `ghost` ghost lines either side
// (where a neighbour exists), temporally blocked: passes[i] sweeps in pass i, ONE all-gather of every rank's first and last
// `ghost` lines per pass.  The whole call is enqueued from here on two streams (dist.py, RowBlockJacobi._forward_temporal
// describes the same sequence driven from Python; ~45 us of interpreter time per pass are what this saves):
//
//   main   [pack + all-gather(0)]  interior(0) ........... wait  interior(1) ........... wait  ...
//   side            wait: ghosts(0) edge(0) pack + all-gather(1) |     ghosts(1) edge(1) pack + all-gather(2) | ...
//
// The first pass reads A_ii, b, x from the AoS rows `va` (own rows only: plan row r = tensor row r - off) and leaves the
// contiguous copies d_ext / b_ext behind; the few lines its edge blocks touch are copied by hand first.  The edge blocks are
// 2 x ghost lines high, so they alone produce the own lines the neighbours want next: the next exchange starts when they end.
// xa / xb: extended scratch iterates (xa receives the edge window of x first); x_last: the extended buffer the LAST pass writes.
// send: 6 m elements, recv: nranks x 6 m (m = ghost x line length).  Canonical 5-point plans, every pass of 2 sweeps or more,
// blocks of more than 2 (2 ghost + passes[0]) lines; GLA_ERR_UNSUPPORTED otherwise (the caller drives the general route).
int gla_dist_jacobi_temporal(gla_csr_t A, gla_comm_t c, int dtype, const void* va, int64_t ldva, int64_t off, int64_t n_local,
                             void* d_ext, void* b_ext, void* xa, void* xb, void* x_last, void* send, void* recv, int ghost,
                             int up, int down, int up_src, int down_src, double w, const int* passes, int n_passes,
                             gla_stream_t main_stream, gla_stream_t side_stream) {
  if (!A || !c || !va || !d_ext || !b_ext || !xa || !xb || !x_last || !send || !recv || !passes || n_passes < 1 || ldva < 3 ||
      ghost < 1 || (dtype != GLA_F32 && dtype != GLA_F64) || off < 0 || n_local < 1 || xa == xb || x_last == xa || x_last == xb)
    return GLA_ERR_INVALID_ARG;
  if (!gla_dia_canon5(A) || A->dia_S <= 0) return GLA_ERR_UNSUPPORTED;
  const int64_t S = A->dia_S, n_ext = A->n_rows, m = (int64_t)ghost * S;
  if (n_ext % S != 0 || off + n_local > n_ext || (up && off != m) || (!up && off != 0) ||
      (down && n_ext - off - n_local != m) || (!down && n_ext != off + n_local) || n_local < 2 * m)
    return GLA_ERR_UNSUPPORTED;
  for (int i = 0; i < n_passes; ++i)
    if (passes[i] < 2 || passes[i] > ghost) return GLA_ERR_UNSUPPORTED;
  if (n_ext / S <= 2 * (int64_t)(2 * ghost + passes[0])) return GLA_ERR_UNSUPPORTED;
  if ((up && (up_src < 0 || up_src >= c->nranks)) || (down && (down_src < 0 || down_src >= c->nranks))) return GLA_ERR_INVALID_ARG;
  // Everything a pass launch could decline is asked BEFORE the first enqueue: a "declined" (UNSUPPORTED / NO_VALUES) that came
  // back from the middle of the sequence would leave an all-gather on this rank that the other ranks never match, and the
  // side stream forked.  (What is left after this point are runtime errors; `bail` joins the streams for those.)
  for (int i = 0; i < n_passes; ++i) {
    const int st = gla_jacobi_parts_status(A, dtype, passes[i]);
    if (st != GLA_OK) return st;
  }
  if (!rccl_ready()) return GLA_ERR_UNSUPPORTED;
  hipStream_t mn = (hipStream_t)main_stream, sd = (hipStream_t)side_stream;
  bool forked = false;
  auto bail = [&](int err) -> int {                 // an error after the fork: the main stream still waits for the side stream
    if (forked && hipEventRecord(c->ev_join, sd) == hipSuccess) (void)hipStreamWaitEvent(mn, c->ev_join, 0);
    (void)hipGetLastError();
    return err;
  };
  const size_t esz = dtype == GLA_F64 ? 8 : 4;
  auto at = [&](const void* p, int64_t elems) -> const void* { return (const char*)p + (size_t)elems * esz; };
  auto atw = [&](void* p, int64_t elems) -> void* { return (char*)p + (size_t)elems * esz; };
  int rc;
  void* cur = xa;
  void* nxt = n_passes == 1 ? x_last : xb;
  // A_ii = the matrix diagonal (what JacobiGNN.py:160-170 passes) is found out inside pass 0 -- the AoS interior launch compares
  // its rows, dg_check_kernel the rows the edge blocks read -- and every later launch then runs the body that neither loads nor
  // keeps A_ii (gla_march5.hip, DGM 2).  The copy d_ext is what a raised flag falls back to: its edge windows and ghost lines are
  // always filled (by hand, below), its interior only by dg_repair_kernel, after pass 0, when the flag IS raised.
  // Round 5, STAGGERED passes (option dist_stagger, default on; blocks tall enough).  With one edge height for every pass the
  // interior launch of pass i + 1 reads lines the EDGE launch of pass i wrote: the main stream waited for the second stream once
  // per pass, and a wait across two hardware queues costs ~12 us on this runtime even when the event fired long ago -- 12-19 us
  // between consecutive interior launches in the kernel trace of config 4's 8-rank block, three times per step of 0.41 ms
  // (profiles/r05/config4_block_stagger.txt).  Now the first / last line block of pass i is H_i lines high with
  //     H_0 = max(ghost + ts_0, 2 ghost),    H_1 = H_0 + ts_0 + ts_1,    H_{i+1} = H_i + max(ts_i, ts_{i+1}):
  // the interior launch of pass i + 1 reads lines >= H_{i+1} - ts_{i+1} >= H_i, all written by the interior launch of pass i (same
  // stream, no wait), and writes lines >= H_{i+1} >= H_i + ts_i, past everything the edge launch of pass i still reads from that
  // buffer.  (H_1: the AoS interior launch leaves the copies of A_ii and b of the H_0 + ts_0 lines either end to the hand copies
  // of the second stream -- the edge blocks of pass 0 read them while it runs -- so the later interior launches stay off those.)
  // The main stream runs its interior launches with an event record between them, nothing else, and waits for the second stream
  // ONCE, at the end of the call; the second stream (pack, all-gather, ghost fill, edge blocks) waits for the main stream once
  // per pass, off the critical path.  Each stream has its own A_ii flag word: the interior launches compute lines >= H_0 only,
  // all compared by the AoS interior launch; the edge blocks compute lines < H_last only, all inside the windows
  // dg_check_kernel compares (and that the hand copies of A_ii cover: win_lines).
  std::vector<int> H((size_t)n_passes);
  H[0] = ghost + passes[0] > 2 * ghost ? ghost + passes[0] : 2 * ghost;
  for (int i = 1; i < n_passes; ++i)
    H[(size_t)i] = i == 1 ? H[0] + passes[0] + passes[1] : H[(size_t)i - 1] + (passes[i - 1] > passes[i] ? passes[i - 1] : passes[i]);
  const int win0 = 2 * ghost + passes[0];                      // lines either end the edge blocks of pass 0 read
  int win_lines = win0;                                        // ... and of every pass (H_i + ts_i at most) when staggered
  // (not while the caller's stream is being captured -- by gla_dist_jacobi_temporal_graph or by torch: replayed from a graph the
  // staggered form is the slower one)
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)main_stream, &cap_st) != hipSuccess) { (void)hipGetLastError(); cap_st = hipStreamCaptureStatusNone; }
  const bool stag = g_tuning.dist_stagger && cap_st == hipStreamCaptureStatusNone && n_passes >= 2 &&
                    n_ext / S >= 2 * (int64_t)(H[(size_t)n_passes - 1] + passes[n_passes - 1]) + 32 && H[(size_t)n_passes - 1] < 0x7fff;
  if (stag)
    for (int i = 0; i < n_passes; ++i) win_lines = H[(size_t)i] + passes[i] > win_lines ? H[(size_t)i] + passes[i] : win_lines;
  auto gh = [&](int i) -> int { return stag ? (ghost | (H[(size_t)i] << 16)) : ghost; };   // (ghost lines | explicit edge height)
  int* dg_flag = nullptr;                                      // the main stream's word (the only one when not staggered)
  int dg_epoch = 0;
  int* dg_flag_s = nullptr;                                    // the second stream's
  int dg_epoch_s = 0;
  if (g_tuning.jacobi_diag_flag && gla_dia_sym5(A) && gla_dia_full5(A) && A->dia_val) {
    rc = gla_jacobi_dg_slot(A, (hipStream_t)main_stream, &dg_flag, &dg_epoch);
    if (rc) return rc;
    dg_flag_s = dg_flag; dg_epoch_s = dg_epoch;
    if (stag) {
      rc = gla_jacobi_dg_slot(A, (hipStream_t)main_stream, &dg_flag_s, &dg_epoch_s);
      if (rc) return rc;
    }
  }
  auto parts = [&](const void* dg, const void* bb, const void* xin, void* xout, int ts, int part, int pass, hipStream_t st) -> int {
    int* const fl = part == 2 ? dg_flag_s : dg_flag;
    const int ep = part == 2 ? dg_epoch_s : dg_epoch;
    // (staggered interior launches whose flag is raised read A_ii from the caller's AoS rows -- plan row r = tensor row r - off, and
    // they touch own rows only -- so no contiguous copy has to be repaired for them between the passes)
    if (fl && stag && part == 1)
      return gla_jacobi_parts_flagged(A, dtype, nullptr, ldva, 0, 0, (const void*)((uintptr_t)va - (uintptr_t)((size_t)off * (size_t)ldva * esz)), bb,
                                      xin, nullptr, nullptr, xout, w, ts, part, gh(pass), fl, ep, st);
    if (fl) return gla_jacobi_parts_flagged(A, dtype, nullptr, 0, 0, 0, dg, bb, xin, nullptr, nullptr, xout, w, ts, part, gh(pass), fl, ep, st);
    return gla_jacobi_fused_parts(A, dtype, dg, bb, xin, xout, w, ts, part, gh(pass), st);
  };
  // ---- pass 0 ---------------------------------------------------------------------------------------------------------------
  {
    const int ts = passes[0];
    const int64_t E = (int64_t)win0 * S;                      // rows the first / last edge block of THIS pass reads
    // ... of every pass: what the hand copy of A_ii covers when the passes run on the flag (the AoS interior launch writes no copy
    // of A_ii then, and an edge block of a later pass whose stream's flag is raised reads it up there)
    const int64_t Ed = (int64_t)(stag && dg_flag ? win_lines : win0) * S;
    void* dst3[3] = {d_ext, b_ext, cur};
    const void* src[8]; void* dst[8]; int64_t cnt[8], strd[8];
    int k = 0;
    for (int wdx = 0; wdx < 2; ++wdx) {
      for (int col = 0; col < 3; ++col) {
        const int64_t Ec = col == 0 ? Ed : E;
        const int64_t w0 = wdx == 0 ? 0 : n_ext - Ec, w1 = wdx == 0 ? Ec : n_ext;
        const int64_t a = w0 > off ? w0 : off;
        const int64_t b = w1 < off + n_local ? w1 : off + n_local;
        if (a >= b) continue;
        src[k] = at(va, (a - off) * ldva + col); dst[k] = atw(dst3[col], a); cnt[k] = b - a; strd[k] = ldva; ++k;
      }
    }
    // Two streams, no third: the exchange runs ON the second stream -- its producer (the pack) and its consumer (the ghost fill)
    // are there anyway, so a communicator stream of its own bought nothing, and a capture of three streams that wait for one
    // another (main -> {side, comm}, comm -> side, side -> comm) dies inside hipStreamEndCapture on ROCm 7.0 (round 5,
    // tools/capture_variants.py: every two-stream form captures, every three-stream form crashes).
    GLA_HIP_CHECK(hipEventRecord(c->ev_fork, mn));
    GLA_HIP_CHECK(hipStreamWaitEvent(sd, c->ev_fork, 0));
    forked = true;
    // (staggered: the interior launch carries the event the second stream's NEXT edge launch waits for as its completion signal)
    if (stag) gla_march5_set_stop_event(c->ev_pack);
    if (dg_flag) rc = gla_jacobi_parts_flagged(A, dtype, va, ldva, off, n_local, nullptr, nullptr, nullptr, nullptr, b_ext, nxt, w, ts, 1, gh(0), dg_flag, dg_epoch, mn);
    else rc = gla_jacobi_fused_parts_aos(A, dtype, va, ldva, off, n_local, d_ext, b_ext, nxt, w, ts, 1, gh(0), mn);
    gla_march5_set_stop_event(nullptr);
    if (rc) return bail(rc);
    // (the interior launch reads the AoS rows and leaves the edge windows of d_ext / b_ext alone: the hand copies may run beside it)
    if (k) { rc = gla_copy_batch(dtype, k, src, dst, cnt, strd, sd); if (rc) return bail(rc); }
    k = 0;                                                    // send = per column [first m own rows | last m own rows]
    for (int col = 0; col < 3; ++col) {
      src[k] = at(va, col); dst[k] = atw(send, 2 * m * col); cnt[k] = m; strd[k] = ldva; ++k;
      src[k] = at(va, (n_local - m) * ldva + col); dst[k] = atw(send, 2 * m * col + m); cnt[k] = m; strd[k] = ldva; ++k;
    }
    rc = gla_copy_batch(dtype, k, src, dst, cnt, strd, sd); if (rc) return bail(rc);
    { const int r_ = g_rccl.all_gather(send, recv, (size_t)(6 * m), dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm, sd);
      if (r_ != 0) return bail(-(3000 + r_)); }
    k = 0;
    for (int col = 0; col < 3; ++col) {                       // the neighbour above's LAST lines / below's FIRST lines
      if (up) { src[k] = at(recv, (int64_t)up_src * 6 * m + 2 * m * col + m); dst[k] = dst3[col]; cnt[k] = m; strd[k] = 1; ++k; }
      if (down) { src[k] = at(recv, (int64_t)down_src * 6 * m + 2 * m * col); dst[k] = atw(dst3[col], off + n_local); cnt[k] = m; strd[k] = 1; ++k; }
    }
    if (k) { rc = gla_copy_batch(dtype, k, src, dst, cnt, strd, sd); if (rc) return bail(rc); }
    if (dg_flag) {                                            // the rows the edge blocks read: [0, Ed) and [n_ext - Ed, n_ext)
      const int64_t nchk = 2 * Ed;
      const unsigned grid = (unsigned)((nchk + 255) / 256 < 1024 ? (nchk + 255) / 256 : 1024);
      if (dtype == GLA_F64)
        hipLaunchKernelGGL((dg_check_kernel<unsigned long long>), dim3(grid), dim3(256), 0, sd, (const unsigned long long*)d_ext,
                           (const unsigned long long*)A->dia_val + 2 * n_ext, (int64_t)0, Ed, n_ext - Ed, n_ext, dg_flag_s, dg_epoch_s);
      else
        hipLaunchKernelGGL((dg_check_kernel<unsigned>), dim3(grid), dim3(256), 0, sd, (const unsigned*)d_ext,
                           (const unsigned*)A->dia_val + 2 * n_ext, (int64_t)0, Ed, n_ext - Ed, n_ext, dg_flag_s, dg_epoch_s);
      if (hipGetLastError() != hipSuccess) return bail(GLA_ERR_INVALID_ARG);
    }
    rc = parts(d_ext, b_ext, cur, nxt, ts, 2, 0, sd); if (rc) return bail(rc);
  }
  // ---- passes 1 .., staggered ---------------------------------------------------------------------------------------------
  if (stag) {
    for (int i = 1; i < n_passes; ++i) {
      // second stream: the own lines the neighbours want are complete (edge launch of pass i - 1) -> exchange of pass i
      {
        const void* src[2] = {at(nxt, off), at(nxt, off + n_local - m)};
        void* dst[2] = {send, atw(send, m)};
        const int64_t cnt[2] = {m, m};
        rc = gla_copy_batch(dtype, 2, src, dst, cnt, nullptr, sd); if (rc) return bail(rc);
        const int r_ = g_rccl.all_gather(send, recv, (size_t)(2 * m), dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm, sd);
        if (r_ != 0) return bail(-(3000 + r_));
      }
      // ... its edge blocks also read what the interior launch of pass i - 1 wrote (whose completion event is ev_pack)
      GLA_HIP_CHECK(hipStreamWaitEvent(sd, c->ev_pack, 0));
      { void* t = cur; cur = nxt; nxt = t; }
      if (i == 1) nxt = xb == cur ? xa : xb;
      if (i == n_passes - 1) nxt = x_last;
      const int ts = passes[i];
      if (i + 1 < n_passes) gla_march5_set_stop_event(c->ev_pack);
      rc = parts(d_ext, b_ext, cur, nxt, ts, 1, i, mn);
      gla_march5_set_stop_event(nullptr);
      if (rc) return bail(rc);
      const void* src[2]; void* dst[2]; int64_t cnt[2];
      int k = 0;
      if (up) { src[k] = at(recv, (int64_t)up_src * 2 * m + m); dst[k] = cur; cnt[k] = m; ++k; }
      if (down) { src[k] = at(recv, (int64_t)down_src * 2 * m); dst[k] = atw(cur, off + n_local); cnt[k] = m; ++k; }
      if (k) { rc = gla_copy_batch(dtype, k, src, dst, cnt, nullptr, sd); if (rc) return bail(rc); }
      rc = parts(d_ext, b_ext, cur, nxt, ts, 2, i, sd); if (rc) return bail(rc);
    }
    GLA_HIP_CHECK(hipEventRecord(c->ev_join, sd));
    GLA_HIP_CHECK(hipStreamWaitEvent(mn, c->ev_join, 0));
    return GLA_OK;
  }
  // ---- passes 1 .. ----------------------------------------------------------------------------------------------------------
  for (int i = 1; i <= n_passes; ++i) {
    // on the second stream: the result of pass i - 1 is complete at its ends -> start the exchange of pass i, then join
    if (i < n_passes) {
      const void* src[2] = {at(nxt, off), at(nxt, off + n_local - m)};
      void* dst[2] = {send, atw(send, m)};
      const int64_t cnt[2] = {m, m};
      rc = gla_copy_batch(dtype, 2, src, dst, cnt, nullptr, sd); if (rc) return bail(rc);
    }
    GLA_HIP_CHECK(hipEventRecord(c->ev_join, sd));            // (BEFORE the all-gather: the main stream waits for the edge blocks only)
    GLA_HIP_CHECK(hipStreamWaitEvent(mn, c->ev_join, 0));
    forked = false;
    if (i == n_passes) break;
    { const int r_ = g_rccl.all_gather(send, recv, (size_t)(2 * m), dtype == GLA_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32, c->comm, sd);
      if (r_ != 0) { forked = true; return bail(-(3000 + r_)); } }      // (the second stream holds work the main stream has not joined)
    if (i == 1 && dg_flag) {                                  // (both streams joined: pass 0 is complete, the flag final)
      const int64_t E0 = (int64_t)(2 * ghost + passes[0]) * S;
      const int64_t lo = E0 > off ? E0 : off, hi = n_ext - E0 < off + n_local ? n_ext - E0 : off + n_local;
      if (hi > lo) {
        if (dtype == GLA_F64)
          hipLaunchKernelGGL((dg_repair_kernel<unsigned long long>), dim3(1024), dim3(256), 0, mn, (const unsigned long long*)va, ldva, off,
                             (unsigned long long*)d_ext, lo, hi, dg_flag, dg_epoch);
        else
          hipLaunchKernelGGL((dg_repair_kernel<unsigned>), dim3(1024), dim3(256), 0, mn, (const unsigned*)va, ldva, off, (unsigned*)d_ext, lo, hi,
                             dg_flag, dg_epoch);
        if (hipGetLastError() != hipSuccess) return GLA_ERR_INVALID_ARG;
      }
    }
    { void* t = cur; cur = nxt; nxt = t; }
    if (i == 1) nxt = xb == cur ? xa : xb;                    // (pass 0 wrote xb or x_last; from now on ping-pong xa / xb)
    if (i == n_passes - 1) nxt = x_last;
    const int ts = passes[i];
    GLA_HIP_CHECK(hipEventRecord(c->ev_fork, mn));
    GLA_HIP_CHECK(hipStreamWaitEvent(sd, c->ev_fork, 0));
    forked = true;
    rc = parts(d_ext, b_ext, cur, nxt, ts, 1, i, mn); if (rc) return bail(rc);
    const void* src[2]; void* dst[2]; int64_t cnt[2];
    int k = 0;
    if (up) { src[k] = at(recv, (int64_t)up_src * 2 * m + m); dst[k] = cur; cnt[k] = m; ++k; }
    if (down) { src[k] = at(recv, (int64_t)down_src * 2 * m); dst[k] = atw(cur, off + n_local); cnt[k] = m; ++k; }
    if (k) { rc = gla_copy_batch(dtype, k, src, dst, cnt, nullptr, sd); if (rc) return bail(rc); }
    rc = parts(d_ext, b_ext, cur, nxt, ts, 2, i, sd); if (rc) return bail(rc);
  }
  return GLA_OK;
}

// ---- the same call as ONE HIP graph ------------------------------------------------------------------------------------------
// gla_dist_jacobi_temporal is ~10 enqueues per pass (pack launch, event record / wait, ncclAllGather, interior launch, ghost fill,
// edge launch, fork / join): 89 us of host time per pass, next to ~100 us of GPU time per pass on the block of one of 8 ranks of
// BASELINE config 4 (1024 lines x 8192 points).  Captured once -- on two streams the graph owns, RCCL's all-gathers on the
// communicator's stream included, the call's A_ii flag word a capture word with its memset node (gla_jacobi_dg_slot) -- and
// replayed with one hipGraphLaunch per forward().  Operand ADDRESSES are frozen into the graph: the caller keeps the buffers and
// refreshes vertex_attr in place.  Every rank of the communicator must capture and launch alike (the all-gathers are collective).
struct gla_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t mn = nullptr, sd = nullptr;
};

int gla_dist_jacobi_temporal_graph(gla_csr_t A, gla_comm_t c, int dtype, const void* va, int64_t ldva, int64_t off, int64_t n_local,
                                   void* d_ext, void* b_ext, void* xa, void* xb, void* x_last, void* send, void* recv, int ghost,
                                   int up, int down, int up_src, int down_src, double w, const int* passes, int n_passes,
                                   gla_graph_t* out) {
  if (!out) return GLA_ERR_INVALID_ARG;
  *out = nullptr;
  if (!A || !c) return GLA_ERR_INVALID_ARG;
  // everything a capture may not do comes first: the plan's flag words (hipMalloc), the kernels' occupancy queries and code
  // loads (one eager pass through the whole sequence on the graph's own streams -- it is also the check that the sequence runs)
  gla_graph* g = new gla_graph();
  hipError_t e = hipStreamCreateWithFlags(&g->mn, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->sd, hipStreamNonBlocking);
  if (e != hipSuccess) { (void)hipGetLastError(); gla_graph_destroy(g); return -(1000 + (int)e); }
  // (the PLAIN form is what gets captured: replayed from a graph the staggered passes are slower than the plain ones -- 0.60 against
  // 0.49 ms on config 4's block, profiles/r05/config4_block_stagger.txt -- the opposite of the eager queues; same results bit for bit)
  const int stagger_was = g_tuning.dist_stagger;
  g_tuning.dist_stagger = 0;
  int rc = gla_dist_jacobi_temporal(A, c, dtype, va, ldva, off, n_local, d_ext, b_ext, xa, xb, x_last, send, recv, ghost, up, down,
                                    up_src, down_src, w, passes, n_passes, g->mn, g->sd);
  if (rc == GLA_OK && hipStreamSynchronize(g->mn) != hipSuccess) rc = GLA_ERR_INVALID_ARG;
  if (rc != GLA_OK) { g_tuning.dist_stagger = stagger_was; (void)hipGetLastError(); gla_graph_destroy(g); return rc; }
  e = hipStreamBeginCapture(g->mn, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) { g_tuning.dist_stagger = stagger_was; (void)hipGetLastError(); gla_graph_destroy(g); return -(1000 + (int)e); }
  rc = gla_dist_jacobi_temporal(A, c, dtype, va, ldva, off, n_local, d_ext, b_ext, xa, xb, x_last, send, recv, ghost, up, down,
                                up_src, down_src, w, passes, n_passes, g->mn, g->sd);
  g_tuning.dist_stagger = stagger_was;
  e = hipStreamEndCapture(g->mn, &g->graph);           // (also after a failed enqueue: the stream must leave capture mode)
  if (rc != GLA_OK || e != hipSuccess || !g->graph) {
    (void)hipGetLastError();
    gla_graph_destroy(g);
    return rc != GLA_OK ? rc : -(1000 + (int)e);
  }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); gla_graph_destroy(g); return -(1000 + (int)e); }
  *out = g;
  return GLA_OK;
}

int gla_graph_launch(gla_graph_t g, gla_stream_t stream) {
  if (!g || !g->exec) return GLA_ERR_INVALID_ARG;
  GLA_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return GLA_OK;
}

int gla_graph_destroy(gla_graph_t g) {
  if (!g) return GLA_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  if (g->mn) { (void)hipStreamSynchronize(g->mn); (void)hipStreamDestroy(g->mn); }
  if (g->sd) { (void)hipStreamSynchronize(g->sd); (void)hipStreamDestroy(g->sd); }
  delete g;
  return GLA_OK;
}

}  // extern "C"
