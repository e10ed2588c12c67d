// agp_capi.hip -- C ABI of libagp_hip.so (see include/agp_hip.h).  Host-side orchestration of the gfx950 kernels in
// agp_linalg.h / agp_cavi.h: one process per GPU, everything enqueued on the caller's HIP stream.
#include "../../include/agp_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "agp_cavi.h"
#include "agp_comm.h"
#include "agp_hyper.h"
#include "agp_kmeans.h"
#include "agp_linalg.h"

using namespace agp;

struct agp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int32_t* dag_flags = nullptr;   // tile / x-ready / abort flags of the task-graph factorisation (k_chol_dag), epoch-stamped
  int64_t dag_cap = 0;
  int32_t dag_epoch = 0;
  // sentinel-filled hand-over area of the task graph (set 0; see Dirty below for how it gets refilled)
  void* hset[2] = {nullptr, nullptr};
  size_t hbytes = 0;
  int htype = -1;  // sizeof(T) the sets were filled for
  // lazy refill: a launch leaves its set "dirty"; the next fused kappa' diag(w) kappa launch on the same stream refills it with
  // rider workgroups (no extra launch, stream or event); whoever needs a dirty set before that refills it inline
  struct Dirty {
    bool on = false;
    int64_t used = 0, stride = 0;
    int nb = 0;
  } h_dirty[2];
  int h_step_set = 0;  // the CAVI-step launches with a prologue alternate between the sets and refill each other's (ProArgs::fill)
  void* tri_scratch = nullptr;    // n x n scratch of the recursive-doubling triangular inverse
  size_t tri_bytes = 0;
  // fallback of the task-graph factorisation (k_chol_safe): grid-barrier words, retry counter, number of CUs; once a retry has
  // been seen by the host (any synchronising call) the task graph is not used again on this context
  void* kmm_scratch = nullptr;  // scaled copy + squared norms of the Y side of a kernel matrix whose Y is not a cached Z
  size_t kmm_bytes = 0;
  void* bal_ws = nullptr;       // partial tiles of the balanced triangular product (k_xtx_bal)
  size_t bal_bytes = 0;
  unsigned* safe_bar = nullptr;
  int32_t* safe_retries = nullptr;
  int n_cu = 0;
  bool dag_off = false;
  int64_t dag_retries_seen = 0;
  // probation: after a lost dependency the context factors by plain launches for `dag_cooldown` CAVI steps, then tries the task
  // graph again (whoever shared the GPU may be gone); every further loss makes the next pause four times longer
  int64_t dag_cooldown = 0, dag_backoff = 512;
  // blocked factorisation of large matrices: side stream of the look-ahead (trailing update of the far columns next to the next
  // group's diagonal block and panel), fork / join events, inverses of the current group's diagonal tiles
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void* chol_li = nullptr;
  size_t chol_li_bytes = 0;
  // split task-graph launches (k_chol_dag ROLE 1 / 2): the chain kernels' own high-priority stream, the word the tile kernel
  // releases them with (signal memory) and the event this context's stream waits on behind every split launch
  hipStream_t chain_stream = nullptr;
  int32_t* chain_go = nullptr;
  int32_t* chain_ctr = nullptr;      // device word: chain workgroups that have exited (DagSync::done); chain_exits = what it will reach
  int32_t chain_exits = 0;
  int chain_state = 0;  // 0 not tried, 1 usable, -1 not available (the two streams do not run kernels side by side) / switched off
  int32_t chain_seq = 0;
};

#define HIPCHK(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#expr) + " : " + hipGetErrorString(_e);                          \
      return AGP_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)

#define LAUNCHCHK(ctx)                                                                          \
  do {                                                                                          \
    hipError_t _e = hipGetLastError();                                                          \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->err = std::string("kernel launch : ") + hipGetErrorString(_e) + " @" + std::to_string(__LINE__); \
      return AGP_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)

#define AGPCHK(expr)                   \
  do {                                 \
    agp_status _s = (expr);            \
    if (_s != AGP_OK) return _s;       \
  } while (0)

// Every entry point runs with the ctx's device current and restores the caller's on the way out: the caller's thread may
// have another device selected (two models on two GPUs in one process, torch.cuda.set_device between calls), and a library
// must not change it behind the caller's back.
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DevGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};

static inline int64_t rup64(int64_t x) { return (x + 63) / 64 * 64; }
static inline dim3 grid1(int64_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }
static inline dim3 grid2(int64_t rows, int64_t cols) { return dim3((unsigned)((cols + 15) / 16), (unsigned)((rows + 15) / 16)); }
static const dim3 blk2(16, 16);

template <typename T>
static agp_status dmalloc(agp_ctx* c, T** p, int64_t n) {
  *p = nullptr;
  if (n <= 0) n = 1;
  hipError_t e = hipMalloc((void**)p, (size_t)n * sizeof(T));
  if (e != hipSuccess) {
    c->err = std::string("hipMalloc : ") + hipGetErrorString(e);
    return AGP_ERR_NOMEM;
  }
  return AGP_OK;
}

// ---- linear-algebra drivers on padded matrices ---------------------------------------------------------------
// Cholesky (lower, in place; diagonal factors in Dg) of the n x n (n = nt*64) matrix A with `ne` extension row blocks
static agp_status tri_scratch_ensure(agp_ctx* c, size_t need) {
  if (c->tri_bytes < need) {
    if (c->tri_scratch) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(c->tri_scratch);
    }
    c->tri_scratch = nullptr;
    c->tri_bytes = 0;
    if (hipMalloc(&c->tri_scratch, need) != hipSuccess) return AGP_ERR_NOMEM;
    c->tri_bytes = need;
  }
  return AGP_OK;
}

// E <- E L^-T (augmented Cholesky): nt launches of k_chol_step; do_x adds X = L^-1 (one extra row launch per column).
// X = L^-1 (all off-diagonal tiles) from L and the diagonal inverses already in X: recursive doubling, 2 launches per level
template <typename T>
static agp_status trtri_levels(agp_ctx* c, const T* A, int64_t ld, T* X, int64_t ldx, int64_t nt) {
  if (nt <= 1) return AGP_OK;
  const int64_t n = nt * TILE;
  AGPCHK(tri_scratch_ensure(c, sizeof(T) * (size_t)n * (size_t)n));
  T* S = (T*)c->tri_scratch;
  for (int64_t bs = 1; bs < nt; bs *= 2) {
    const int64_t pairs = (nt + 2 * bs - 1) / (2 * bs);
    dim3 g((unsigned)bs, (unsigned)bs, (unsigned)pairs);
    hipLaunchKernelGGL((k_trtri_level<T>), g, dim3(NTHREADS), 0, c->stream, A, ld, X, ldx, S, n, nt, bs, 0);
    hipLaunchKernelGGL((k_trtri_level<T>), g, dim3(NTHREADS), 0, c->stream, A, ld, X, ldx, S, n, nt, bs, 1);
  }
  LAUNCHCHK(c);
  return AGP_OK;
}

// hand-over area for the next task-graph launch (`elems` elements of T): waits for the pending refill of the set, returns it;
// dag_handover_release() schedules the refill behind the launch
template <typename T>
static agp_status dag_handover_acquire(agp_ctx* c, int64_t elems, int set, T** out) {
  const size_t need = sizeof(T) * (size_t)elems;
  if (c->hbytes < need || c->htype != (int)sizeof(T)) {
    (void)hipStreamSynchronize(c->stream);
    for (int q = 0; q < 2; ++q) {
      if (c->hset[q]) (void)hipFree(c->hset[q]);
      c->hset[q] = nullptr;
      c->h_dirty[q].on = false;
    }
    c->hbytes = 0;
    const size_t cap = need + need / 4;
    c->hbytes = cap;
    c->htype = (int)sizeof(T);
  }
  if (!c->hset[set]) {  // set 1 only exists once a step launch with a prologue asks for it (they alternate between the sets)
    if (hipMalloc(&c->hset[set], c->hbytes) != hipSuccess) return AGP_ERR_NOMEM;
    hipLaunchKernelGGL((k_fill_sent<T>), dim3(2048), dim3(256), 0, c->stream, (T*)c->hset[set], (int64_t)(c->hbytes / sizeof(T)),
                       (int64_t)0);
    c->h_dirty[set].on = false;
  }
  if (c->h_dirty[set].on) {  // nobody refilled it in passing: do it now, on this stream
    const auto& d = c->h_dirty[set];
    hipLaunchKernelGGL((k_fill_sent<T>), dim3((unsigned)std::max<int64_t>(1, 512 / d.nb), (unsigned)d.nb), dim3(256), 0, c->stream,
                       (T*)c->hset[set], d.used, d.stride);
    c->h_dirty[set].on = false;
  }
  *out = (T*)c->hset[set];
  return AGP_OK;
}
template <typename T>
static agp_status dag_handover_release(agp_ctx* c, int64_t used, int64_t stride, int nb, int set) {
  // what the launch could have written (the first `used` elements of each of the nb problem regions) must hold the sentinel
  // again before the set's next use: left to the next fused syrk launch (riders) or, failing that, to the next acquire
  c->h_dirty[set].on = true;
  c->h_dirty[set].used = used;
  c->h_dirty[set].stride = stride;
  c->h_dirty[set].nb = nb;
  return AGP_OK;
}

// One-launch task graph (k_chol_dag) or one launch per block column (k_chol_step)?  Measured on MI355X, whole CAVI step:
// m = 1024 f64 0.39 vs 0.52 ms, m = 2048 f32 0.80 vs 1.05 ms, m = 4096 f64 11.6 vs 8.3 ms -- the task graph removes launch
// gaps and re-reads from the latency-bound chain, but its tiles stream their operands past the L2s (coherent loads), which
// costs more than it saves once the trailing updates dominate.  AGP_CHOL_DAG=0 / 1 forces one or the other.
// Column bound: the chain of a task graph waits for feeder tiles with HIGHER workgroup indices -- tile (k+1, k) in block column k
// and tile (k+1, k+1), the first workgroup of column k + 1.  Progress does not depend on them being resident early: the chain
// publishes X_k before it blocks on them (k_chol_dag, "late_feed"), so every resident workgroup -- all of them belong to block
// columns <= k of their problem -- can finish on what the chains have published, retires, and the in-order dispatch reaches the
// feeders.  (Rounds 1-2 published X_k after that wait and therefore needed a whole block column of every problem resident,
// nb * (nt + ne + 1) <= 208; 8 problems of 34 tiles stalled.)  What remains is a performance matter: a feeder that gets its slot only
// when the column before it retires applies its k pending updates on the chain's critical path.  Up to 288 tiles per column of all
// problems the launch is still well ahead of per-column launches (8 x 34: 0.62 ms against 2 x 0.40 ms for 4 + 4).
constexpr int64_t DAG_MAX_NT = 32, DAG_MAX_COLUMN_TILES = 288;
static bool chol_use_dag(const agp_ctx* c, int64_t nt, int64_t ne = 0, int64_t nb = 1) {
  static const int v = []() {
    const char* e = getenv("AGP_CHOL_DAG");
    return e ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  if (c->dag_off) return false;  // a lost dependency was seen on this context (two processes sharing the device): stay safe
  if (nb * (nt + ne + 1) > DAG_MAX_COLUMN_TILES) return false;
  return v < 0 ? nt <= DAG_MAX_NT : v == 1;
}

__global__ void k_set_i32(int32_t* p, int32_t v) { *p = v; }
// ... visible to a polling kernel of another stream (signal memory, system scope)
__global__ void k_set_sig(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// ... and the arrival word of a column group of the all-reduced statistics (comm_allreduce_groups): the collective's kernel before
// this one on the same stream has ended, i.e. its writes are in memory
__global__ void k_set_arrive(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

static agp_status comm_allreduce_groups(agp_comm* cm, void* base, int esz, const int64_t* off, const int64_t* cnt, int ng,
                                        int32_t dtype, int32_t* arrive, int32_t epoch);  // (defined with agp_comm_allreduce)

// grid-barrier words, retry counter and the CU count of the fallback (allocated on first use)
template <typename T>
static agp_status ensure_safe_words(agp_ctx* c) {
  if (!c->safe_bar) {
    if (hipMalloc((void**)&c->safe_bar, 2 * sizeof(unsigned)) != hipSuccess) return AGP_ERR_NOMEM;
    if (hipMalloc((void**)&c->safe_retries, sizeof(int32_t)) != hipSuccess) return AGP_ERR_NOMEM;
    HIPCHK(c, hipMemsetAsync(c->safe_bar, 0, 2 * sizeof(unsigned), c->stream));
    HIPCHK(c, hipMemsetAsync(c->safe_retries, 0, sizeof(int32_t), c->stream));
    hipDeviceProp_t pr;
    HIPCHK(c, hipGetDeviceProperties(&pr, c->device));
    c->n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 64;
  }
  return AGP_OK;
}

// Workgroups the grid-barrier fallback may count on being resident together.  One per CU -- minus the CUs a chain kernel of the
// NEXT split launch may be sitting on: with the host ahead, that kernel is already in flight on its own stream, polls for a tile
// kernel that is enqueued BEHIND the fallback, and holds most of its CU's LDS while it does (a fallback workgroup cannot share the
// CU).  A fallback grid of n_cu workgroups would then wait at its first barrier for a workgroup that can never be placed.
static int64_t safe_grid_cap(const agp_ctx* c) {
  // test hook (AGP_DAG_TEST_OVERSUBSCRIBE=1, with AGP_DAG_TEST_ABORT=1): a fallback grid that CANNOT be resident at once (four
  // workgroups of ~110 KB LDS per CU), i.e. the situation the bounded grid barrier exists for -- the step must end in status -3
  // (AGP_ERR_HIP from agp_svgp_check_status) after the barrier's limit instead of hanging (tests/test_gpu_round6.py)
  static const bool over = []() {
    const char* e = getenv("AGP_DAG_TEST_OVERSUBSCRIBE");
    return e && e[0] == '1';
  }();
  if (over) return 4 * (int64_t)c->n_cu;
  return std::max<int64_t>(1, (int64_t)c->n_cu - (c->chain_state == 1 ? CHOL_MAXB : 0));
}
// the fallback behind a task-graph launch (see k_chol_safe): one launch that returns at once unless the latch reads -1
template <typename T>
static agp_status launch_chol_safe(agp_ctx* c, const CholBatch<T>& bt, const SafeSrc<T>& src, int nb, int64_t ld, int64_t ldx,
                                   int64_t lde, int64_t ne, int64_t nt, int32_t* info_dev, int64_t nvalid) {
  AGPCHK(ensure_safe_words<T>(c));
  // one workgroup per CU at most (each needs ~110 KB of LDS, so one fits per CU): all of them become resident, whatever else runs
  const int64_t most = (nt + ne + nt * (nt + 1) / 2 + ne * nt) * nb;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(safe_grid_cap(c), most));
  hipLaunchKernelGGL((k_chol_safe<T>), dim3(grid), dim3(CHOL_THREADS), 0, c->stream, bt, src, nb, ld, ldx, lde, ne, nt, info_dev,
                     nvalid, c->safe_bar, c->safe_retries);
  LAUNCHCHK(c);
  return AGP_OK;
}
// The kernel behind the task graph of a single-latent CAVI step: the fallback of k_chol_safe (a no-op unless the latch reads -1)
// and then the row statistics + local update, in ONE launch -- the separate k_chol_safe launch cost the step ~5 us of kernel and a
// launch gap on its critical path.  grid <= n_cu workgroups of 512 threads (all resident: the fallback uses grid barriers); the
// rows are taken wave by wave, grid-stride.
template <typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_safe_rowstats(CholBatch<T> bt, SafeSrc<T> src, int64_t ld, int64_t ldx, int64_t lde,
                                                                int64_t ne, int64_t nt, int32_t* __restrict__ info, int64_t nvalid,
                                                                unsigned* __restrict__ bar, int32_t* __restrict__ retries,
                                                                int64_t B, int nslices, RowstatsBatch<T> rb, int64_t ldp,
                                                                int64_t ldw, int64_t cols, T jitter, T rho, LikParams<T> lp,
                                                                const T* __restrict__ y, const int64_t* __restrict__ idx,
                                                                T* __restrict__ Kt, T* __restrict__ muf, T* __restrict__ varf,
                                                                T* __restrict__ cb, T* __restrict__ theta, T* __restrict__ r,
                                                                T* __restrict__ w, int* __restrict__ flags,
                                                                const T* __restrict__ lam, T* __restrict__ gamma,
                                                                int rows_done = 0, const int32_t* __restrict__ pf_word = nullptr,
                                                                int32_t pf_want = 0) {
  __shared__ __attribute__((aligned(16))) T sm[3 * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[SC_ELEMS];
  __shared__ T piv[TILE];
  const bool ran = chol_safe_body<T>(bt, src, 1, ld, ldx, lde, ne, nt, info, nvalid, bar, retries, sm, sc, piv);
  // (after a fallback the last grid barrier of the column loop has made every workgroup's tiles visible)
  // rows_done (round 3): the task-graph launch finished its rows itself (EpiArgs, agp_chol.h) -- unless it was aborted and re-run here
  if (ran || !rows_done) {
    const int64_t wpb = CHOL_THREADS / 64, nwave = (int64_t)gridDim.x * wpb;
    for (int64_t i = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); i < B; i += nwave)
      rowstats_row<T>(i, threadIdx.x & 63, 0, B, nslices, rb, ldp, ldw, cols, jitter, rho, lp, y, idx, Kt, muf, varf, cb, theta, r,
                      w, (int64_t)0, flags, lam, gamma);
  }
  // pf_word (round 3): this launch was deferred to the head of the NEXT step and also carries that step's wait for its look-ahead
  // (one wave polls the look-ahead's "done" word; in the steady state it is set long before)
  if (pf_word && blockIdx.x == 0 && threadIdx.x == 0) {
    long spins = 0;
    while ((int32_t)(__hip_atomic_load(pf_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - pf_want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1L << 27)) {
        atomicExch(info, -2);
        break;
      }
    }
  }
}

// The same for the launch behind materialize() (the factorisation of -2 eta2 with its inverse and Sigma = X' X, round 5): the
// fallback, then mu = Sigma eta1 and v = X eta1 wave by wave (k_symv_trmv's rows) -- one launch less in the hyper-parameter iteration.
template <typename T>
__global__ __launch_bounds__(CHOL_THREADS) void k_safe_symv_trmv(CholBatch<T> bt, SafeSrc<T> src, int64_t ld, int64_t ldx, int64_t lde,
                                                                 int64_t ne, int64_t nt, int32_t* __restrict__ info, int64_t nvalid,
                                                                 unsigned* __restrict__ bar, int32_t* __restrict__ retries,
                                                                 const T* __restrict__ S, const T* __restrict__ X, int64_t ldm,
                                                                 int64_t n, const T* __restrict__ x, T* __restrict__ ys,
                                                                 T* __restrict__ yt) {
  __shared__ __attribute__((aligned(16))) T sm[3 * TILE * LDP];
  __shared__ __attribute__((aligned(16))) T sc[SC_ELEMS];
  __shared__ T piv[TILE];
  (void)chol_safe_body<T>(bt, src, 1, ld, ldx, lde, ne, nt, info, nvalid, bar, retries, sm, sc, piv);
  // (after a fallback its last grid barrier has made X and Sigma visible to every workgroup)
  const int64_t wpb = CHOL_THREADS / 64, nwave = (int64_t)gridDim.x * wpb;
  for (int64_t w = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); w < 2 * n; w += nwave)
    symv_trmv_row<T>(w, threadIdx.x & 63, S, X, ldm, n, x, ys, yt);
}

// Factorisation by plain launches (matrices beyond the task graph, and the task graph's fallback).  From 8 block columns on it is
// blocked (agp_chol.h, k_chol_panel): groups of G block columns -- the G x G diagonal block by G small launches, the rows below it
// by one panel-solve launch, everything to the right by one trailing launch per group; the part of the trailing update that the
// next group does not need runs on a side stream, next to the next group's diagonal block and panel (look-ahead of one group).
// AGP_CHOL_GROUP = 1 gives the plain per-column right-looking sequence, AGP_CHOL_LOOKAHEAD = 0 keeps everything on one stream.
constexpr int CHOL_GMAX = 8;
static int chol_group() {
  static const int g = []() {
    const char* e = getenv("AGP_CHOL_GROUP");
    const int v = e ? atoi(e) : 8;
    return v < 1 ? 1 : v > CHOL_GMAX ? CHOL_GMAX : v;
  }();
  return g;
}
// blocked from 96 block rows on (extension included): measured on MI355X, a plain 4096 x 4096 matrix (64 block rows) is still
// quicker column by column (2.6 vs 3.0 ms, the group's serial launches dominate), 8192 and the C5 step (64 + 65 rows) are not
static bool chol_blocked(int64_t nt, int64_t ne) { return chol_group() > 1 && nt >= 8 && nt + ne >= 96; }
// kernel launches of one factorisation by plain launches (what the HIP-event timing of the sequence is divided by)
static int64_t chol_launch_count(int64_t nt, int64_t ne) {
  if (!chol_blocked(nt, ne)) return nt;
  const int64_t G = chol_group();
  int64_t n = 0;
  for (int64_t k0 = 0; k0 < nt; k0 += G) {
    const int64_t k1 = (k0 + G < nt) ? k0 + G : nt, kn = (k1 + G < nt) ? k1 + G : nt;
    n += (k1 - k0) + (nt - k1 + ne > 0 ? 1 : 0) + (k1 < nt ? 1 : 0) + (k1 < nt && kn < nt ? 1 : 0);
  }
  return n;
}
static bool chol_lookahead() {
  static const bool on = []() {
    const char* e = getenv("AGP_CHOL_LOOKAHEAD");
    return !(e && e[0] == '0');
  }();
  return on;
}
template <typename T>
static agp_status chol_columns(agp_ctx* c, const CholBatch<T>& bt, int nb, int64_t ld, int64_t ldx, int64_t lde, int64_t ne,
                               int do_x, int64_t nt, int32_t* info_dev, int64_t nvalid) {
  const int64_t G = chol_group();
  if (!chol_blocked(nt, ne)) {
    for (int64_t k = 0; k < nt; ++k) {
      const int64_t nP = nt - k + ne;
      const int64_t nU = chol_nU(k, 0, nt, nt, ne);
      hipLaunchKernelGGL((k_chol_step<T>), dim3((unsigned)(nP + nU), (unsigned)nb), dim3(CHOL_THREADS), 0, c->stream, bt, ld, ldx,
                         lde, ne, do_x, k, nt, info_dev, nvalid, (int64_t)0, (int64_t)-1, (T*)nullptr, (int64_t)0);
    }
    return AGP_OK;
  }
  const int64_t li_stride = G * TILE * TILE;
  const size_t li_need = sizeof(T) * (size_t)(li_stride * nb);
  if (c->chol_li_bytes < li_need) {
    if (c->chol_li) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->chol_li);
    }
    c->chol_li = nullptr;
    c->chol_li_bytes = 0;
    if (hipMalloc(&c->chol_li, li_need) != hipSuccess) return AGP_ERR_NOMEM;
    c->chol_li_bytes = li_need;
  }
  T* li = (T*)c->chol_li;
  const bool look = chol_lookahead();
  if (look && !c->side) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  auto trail = [&](hipStream_t st, int64_t k0, int64_t k1, int64_t j_lo, int64_t j_hi) {
    const int64_t n = chol_trail_tiles(j_lo, j_hi, nt, ne);
    if (n > 0)
      hipLaunchKernelGGL((k_chol_trail<T>), dim3((unsigned)n, (unsigned)nb), dim3(CHOL_THREADS), 0, st, bt, ld, lde, ne, k0, k1,
                         nt, j_lo, j_hi);
  };
  bool side_busy = false;
  for (int64_t k0 = 0; k0 < nt; k0 += G) {
    const int64_t k1 = (k0 + G < nt) ? k0 + G : nt;
    const int g = (int)(k1 - k0);
    // D: the diagonal block on its own (block rows k0 .. k1-1 only), leaving the inverses of its diagonal tiles in li
    for (int64_t k = k0; k < k1; ++k) {
      const int64_t nP = k1 - k;
      const int64_t nU = chol_nU(k, k0, k1, k1, 0);
      hipLaunchKernelGGL((k_chol_step<T>), dim3((unsigned)(nP + nU), (unsigned)nb), dim3(CHOL_THREADS), 0, c->stream, bt, ld, ldx,
                         lde, (int64_t)0, do_x, k, k1, info_dev, nvalid, k0, k1, li, li_stride);
    }
    // P: block rows k1 .. nt-1 and the extension rows against the block
    const int64_t rows = nt - k1 + ne;
    if (rows > 0) {
      const dim3 grid((unsigned)rows, (unsigned)nb);
#define AGP_PANEL(GG)                                                                                                       \
  hipLaunchKernelGGL((k_chol_panel<T, GG>), grid, dim3(CHOL_THREADS), 0, c->stream, bt, ld, lde, ne, k0, nt, (const T*)li, \
                     li_stride, g)
      if (G <= 2) AGP_PANEL(2);
      else if (G <= 4) AGP_PANEL(4);
      else AGP_PANEL(8);
#undef AGP_PANEL
    }
    if (k1 >= nt) break;
    // T: the far columns were last written by the previous group's side-stream launch: order behind it
    if (side_busy) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    const int64_t kn = (k1 + G < nt) ? k1 + G : nt;  // the next group's columns [k1, kn) are needed first
    trail(c->stream, k0, k1, k1, kn);
    if (kn < nt) {
      if (look) {
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        trail(c->side, k0, k1, kn, nt);
        HIPCHK(c, hipEventRecord(c->ev_join, c->side));
        side_busy = true;
      } else {
        trail(c->stream, k0, k1, kn, nt);
      }
    }
  }
  if (side_busy) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return AGP_OK;
}


static void dag_pause(agp_ctx* c) {
  c->dag_off = true;
  c->dag_cooldown = c->dag_backoff;
  c->dag_backoff = std::min<int64_t>(c->dag_backoff * 4, (int64_t)1 << 20);
}
// once per CAVI step: end of the probation?
static void dag_tick(agp_ctx* c) {
  if (c->dag_off && --c->dag_cooldown <= 0) c->dag_off = false;
}
// host side of the latch: called where the stream has just been synchronised anyway
static void dag_retry_check(agp_ctx* c) {
  if (!c->safe_retries || c->dag_off) return;
  int32_t r = 0;
  if (hipMemcpy(&r, c->safe_retries, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) return;
  if (r > c->dag_retries_seen) {
    fprintf(stderr,
            "[agp_hip] warning: %d task-graph factorisation(s) lost a tile dependency (is another process using this GPU?) and were "
            "re-run by the in-stream fallback; this context uses plain launches for the next %lld steps\n",
            (int)(r - c->dag_retries_seen), (long long)c->dag_backoff);
    c->dag_retries_seen = r;
    dag_pause(c);
  }
}

// Split task-graph launches: worth it when the launch queues far more tiles than the chip has workgroup slots (C3: 1584, C4: 3264);
// the small launches (C2: 408 tiles, and its merged step with the prologue) stay one kernel.  AGP_CHAIN_SPLIT=0 / 1 forces.
static bool chain_split_wanted(int64_t tiles, bool with_prologue = false, bool f64 = true) {
  static const int v = []() {
    const char* e = getenv("AGP_CHAIN_SPLIT");
    return e ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  // fp64 launches with the prologue stay merged unless forced: their tile workgroups stage four LDS tiles (135 KB: one workgroup
  // per CU whatever the registers), and the measured case lost (C2: 0.3167 ms split, 0.3103 ms merged).  In fp32 the same tile
  // kernel fits two workgroups per CU (68 KB, 128 VGPRs) and wins: m = 1024, B = 2048 fp32 0.325 -> 0.270 ms per step.
  // Round 5 made the fp32 form opt-in as well after two findings of the stress runs (docs/DESIGN_LOG.md section 14): the chain kernel
  // took tile (0, 0)'s eta2 step, so an ABORTED launch could leave eta2 half-stepped (repaired in round 5: the chain's place in the
  // tile kernel takes it and parks the tile), and about one split launch in 10 000 lost a dependency on its own -- the tile kernel
  // filled every CU before the chain kernel was resident (repaired in round 6: DagSync::here / k_wait_here).  Default again in fp32.
  if (with_prologue && f64 && v < 0) return false;
  return v < 0 ? tiles >= 600 : v == 1;
}
// the chain stream, its release word and the proof that kernels of the two streams run at the same time (k_handshake: where
// dispatches are serialised -- rocprofv3 --pmc, AMD_SERIALIZE_KERNEL -- a chain kernel polling for the tile kernel behind it in
// the device's single queue would never be released; such a context keeps the merged kernel)
static bool chain_split_ready(agp_ctx* c) {
  if (c->chain_state != 0) return c->chain_state == 1;
  c->chain_state = -1;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return false;
  if (hipStreamCreateWithPriority(&c->chain_stream, hipStreamNonBlocking, hi) != hipSuccess) {
    c->chain_stream = nullptr;
    (void)hipGetLastError();
    return false;
  }
  bool ok = hipExtMallocWithFlags((void**)&c->chain_go, 8, hipMallocSignalMemory) == hipSuccess && hipMemset(c->chain_go, 0, 8) == hipSuccess &&
            hipMalloc((void**)&c->chain_ctr, sizeof(int32_t)) == hipSuccess && hipMemset(c->chain_ctr, 0, sizeof(int32_t)) == hipSuccess;
  int32_t* hs = nullptr;
  ok = ok && hipMalloc((void**)&hs, 4 * sizeof(int32_t)) == hipSuccess && hipMemset(hs, 0, 4 * sizeof(int32_t)) == hipSuccess;
  if (ok) {
    (void)hipStreamSynchronize(c->stream);
    hipLaunchKernelGGL(k_handshake, dim3(1), dim3(64), 0, c->chain_stream, hs, (const int32_t*)(hs + 1), hs + 2);
    hipLaunchKernelGGL(k_handshake, dim3(1), dim3(64), 0, c->stream, hs + 1, (const int32_t*)hs, hs + 3);
    int32_t res[4] = {0, 0, 0, 0};
    ok = hipStreamSynchronize(c->chain_stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
         hipMemcpy(res, hs, sizeof(res), hipMemcpyDeviceToHost) == hipSuccess && res[2] == 1 && res[3] == 1;
  }
  if (hs) (void)hipFree(hs);
  (void)hipGetLastError();
  if (ok) c->chain_state = 1;
  return ok;
}
// Behind a split launch NO event joins the two streams: the tile kernel cannot end before the chain's last publish, after which the
// chain writes nothing (its diagonal factors are write-through stores issued before that publish), so whatever follows on this
// context's stream -- and a host synchronisation of it -- sees a finished factorisation.  The one exception, an aborted launch whose
// chain is still inside a tile factorisation, is handled where it matters: the fallback waits for the chain workgroups' exit count
// (DagSync::done, SafeSrc::chain_done).  (An event join -- record on the chain stream, wait on this one -- was the first version and
// DEADLOCKED with the host several steps ahead, the look-ahead on and the prologue inside the split launch; removed in round 5,
// the account is in docs/DESIGN_LOG.md.)
// what a split launch hands its kernels / its fallback about the chain kernel (nb chain workgroups)
static void chain_split_arm(agp_ctx* c, DagSync& ds, int nb) {
  ds.go = c->chain_go;
  ds.go_val = ++c->chain_seq;
  ds.done = c->chain_ctr;
  ds.here = c->chain_go + 1;  // (second word of the same signal-memory allocation)
  c->chain_exits += nb;
}
// ... enqueued on the step's stream between the chain kernel's launch (chain stream) and the tile kernel's: every chain workgroup
// of this launch -- and of all launches before it: the count is cumulative, like chain_exits -- is resident (DagSync::here)
static void chain_split_wait_here(agp_ctx* c) {
  hipLaunchKernelGGL(k_wait_here, dim3(1), dim3(64), 0, c->stream, (const int32_t*)(c->chain_go + 1), c->chain_exits);
}

// what a CAVI step hands to its factorisation about the look-ahead stream (see DagSync, agp_chol.h)
struct StepSync {
  DagSync ds{};
  bool used = false;  // out: the task-graph launch took `ds`
};

// The pending natural-gradient step that a CAVI step's task-graph launch takes along as its prologue (ProArgs, agp_chol.h)
template <typename T>
struct ProHost {
  const T* kap = nullptr;
  int64_t ldk = 0, Kdim = 0;
  const T *w = nullptr, *r = nullptr;
  T* eta2 = nullptr;
  const T* Kinv = nullptr;
  int64_t ldm = 0;
  T* eta1 = nullptr;
  const T* kinv_mu0 = nullptr;
  T lr = T(0);
  const T *packed = nullptr, *tred = nullptr;  // batch-parallel step: reduced statistics instead of (kap, w, r)  (ProArgs)
  const int32_t* arrive = nullptr;             // ... arriving in block-column groups (AGP_SPLIT_OVERLAP)
  int32_t arrive_want = 0;
  unsigned char grp[32] = {};
  T* Cout = nullptr;                           // ProArgs::Cout
};
// k-slices per block column of the prologue's product: a tile of block column c has to be there when the chain reaches the
// column (about tau * c after the start, tau = 17.8 us f64 / 14 us f32 per block column), a 64-row chunk of the product costs a
// workgroup about tc = 2.7 / 1.6 us; columns 0 and 1 feed the chain at once and are split as far as it pays (8).
static void pro_ks_table(int64_t nt, int64_t nq, bool f64, unsigned char* ks, unsigned char* kf) {
  const double tc = f64 ? 2.7 : 1.6, tau = f64 ? 17.8 : 14.0;
  for (int64_t c = 0; c < nt && c < 32; ++c) {
    int want;
    if (c == 0) want = 8;
    else if (c == 1) want = f64 ? 4 : 8;
    else want = (int)std::ceil((double)nq * (tc + 0.3) / (tau * (double)c - 12.0));
    want = std::max(1, std::min<int>(want, (int)std::min<int64_t>(8, nq)));
    // the tiles next to the diagonal (ProArgs::kf) take the same split as their column (a finer one was measured at 32 block
    // columns, docs/DESIGN_LOG.md, and not adopted)
    ks[c] = kf[c] = (unsigned char)want;
  }
}
// test hook (AGP_DAG_TEST_ABORT=1): pretend every task-graph launch of a CAVI step lost a dependency, so that the in-stream
// fallback runs behind each of them
static bool dag_test_abort() {
  static const bool on = []() {
    const char* e = getenv("AGP_DAG_TEST_ABORT");
    return e && e[0] == '1';
  }();
  return on;
}

template <typename T>
static agp_status potrf_fused(agp_ctx* c, T* A, int64_t ld, int64_t n, T* X, int64_t ldx, T* Dg, T* E, int64_t lde,
                              int64_t ne, int do_x, int32_t* info_dev, int64_t nvalid, const T* erow = nullptr,
                              bool want_l = true, SafeSrc<T>* safe = nullptr, bool* defer_safe = nullptr,
                              StepSync* ssync = nullptr, const ProHost<T>* pro = nullptr, const EpiArgs<T>* epi = nullptr,
                              T* Pout = nullptr, bool* p_done = nullptr, double* ld_out = nullptr, int32_t* ld_status = nullptr) {
  // Pout (with do_x; leading dimension ldx): P = X' X is wanted next -- K^-1 at a kernel refresh, Sigma for the hyper-gradient.  On
  // the task graph it is formed by product workgroups at the end of the same launch (ProdArgs, agp_chol.h) and *p_done says so;
  // otherwise the caller forms it (xtx_padded)
  if (p_done) *p_done = false;
  // ssync (CAVI step next to a look-ahead stream): the step's task-graph instantiation stores its `started` number (`used` is set)
  // defer_safe (in: the caller can run the fallback itself, k_safe_rowstats; out: whether it has to -- the task graph was used)
  const bool can_defer = defer_safe && *defer_safe;
  if (defer_safe) *defer_safe = false;
  // safe: sources the inputs can be restored from (A = -2 eta2, E = [kappa ; eta1' ; 0]): the in-stream fallback k_chol_safe is
  // then enqueued behind the task graph; without it a lost dependency surfaces as an error at the caller's next check
  // want_l = false: the caller never reads the factor L itself (only E L^-T, X, Dg): the task graph skips those stores
  // erow: the last extension block is [erow' ; 0] (not yet written to E: the task graph reads it in place; the per-column
  // path needs it in E first)
  const int64_t nt = n / TILE;
  const bool use_dag = chol_use_dag(c, nt, ne);
  bool split_used = false;  // the launch went out as chain kernel + tile kernel: its fallback waits for the chain's exit count
  if (pro && !(use_dag && X && !want_l && nt <= 32)) {
    c->err = "potrf_fused: a pending natural-gradient step can only ride on the CAVI step's task-graph launch";
    return AGP_ERR_INVALID;
  }
  if (use_dag && X) {
    const int64_t nx = (do_x && nt > 1) ? nt : 0;  // the full inverse rides along as nt identity block rows
    // prologue (pro): helper workgroups, their flags and hand-over slots
    ProArgs<T> pa{};
    int64_t nhelp = 0;
    if (pro) {
      if (pro->packed)
        for (int64_t cc = 0; cc < nt; ++cc) pa.ks[cc] = pa.kf[cc] = 1;  // nothing to compute: no helpers
      else
        pro_ks_table(nt, pro->Kdim / TILE, sizeof(T) == 8, pa.ks, pa.kf);
      for (int64_t cc = 0; cc < nt; ++cc) nhelp += pro_nhelp(nt, cc, pa.ks[cc], pa.kf[cc]);
    }
    const int64_t nf = ((nt + ne + nx) * nt + 3 * nt + 1 + nhelp) * DAG_FS;
    if (c->dag_cap < nf) {
      if (c->dag_flags) (void)hipFree(c->dag_flags);
      c->dag_flags = nullptr;
      c->dag_cap = 0;
      if (hipMalloc((void**)&c->dag_flags, sizeof(int32_t) * (size_t)(nf + 1024)) != hipSuccess) return AGP_ERR_NOMEM;
      c->dag_cap = nf + 1024;
      HIPCHK(c, hipMemsetAsync(c->dag_flags, 0, sizeof(int32_t) * (size_t)c->dag_cap, c->stream));
      c->dag_epoch = 0;
    }
    c->dag_epoch += 1;
    const int64_t ntiles = nt * (nt + 1) / 2 + ne * nt + (nx ? nt * (nt + 1) / 2 : 0);
    const int64_t hstride = ((2 * nt + ne) * nt + 3 * nt + nhelp) * TILE * TILE;
    const int64_t hused = (3 * nt + (nt + ne + nx) * nt + nhelp) * TILE * TILE;
    T* H = nullptr;
    const bool with_p = Pout != nullptr && nx > 0 && p_done != nullptr;
    const int64_t nprod = with_p ? nt * (nt + 1) / 2 : 0;
    // launches with a prologue alternate between the two hand-over sets and refill each other's; so do the launches with product
    // workgroups (the symmetric-product launches whose riders refilled set 0 behind them are gone from their path)
    const int hs = (pro || with_p) ? c->h_step_set : 0;
    AGPCHK(dag_handover_acquire<T>(c, hstride, hs, &H));
    ProdArgs<T> pd{};
    if (with_p) {
      pd.out = Pout;
      pd.ld = ldx;
      pd.ld_out = ld_out;  // (log det of the factor rides on the last product workgroup)
      pd.status = ld_status;
      if (!pro) {
        const int other = hs ^ 1;
        if (c->hset[other] && c->h_dirty[other].on && c->h_dirty[other].nb == 1 && c->htype == (int)sizeof(T)) {
          pd.fill = (T*)c->hset[other];
          pd.fill_n = c->h_dirty[other].used;
          c->h_dirty[other].on = false;
        }
        c->h_step_set = other;
      }
      if (safe) {
        safe->pout = Pout;
        safe->ldpo = ldx;
        safe->ld_out = ld_out;
        safe->ld_status = ld_status;
        safe->ld_n = nvalid;
      }
      *p_done = true;
    }
    unsigned long long* const trace = nullptr;  // (per-tile timestamps: the TRACE instantiation of k_chol_dag, a development aid)
    const bool step_inst = nx == 0 && !do_x && !want_l;
    DagSync ds{};
    if (ssync) {
      if (step_inst) {
        ds = ssync->ds;
        ssync->used = true;
      }
    }
    CholBatch<T> one{};
    one.A[0] = A;
    one.X[0] = X;
    one.Dg[0] = Dg;
    one.E[0] = E;
    if (pro) {  // ... with the pending natural-gradient step as its prologue (the CAVI step's launch, or --
                                        // hyper-parameter iteration -- the factorisation of the updated -2 eta2 with its inverse)
      pa.kap = pro->kap;
      pa.ldk = pro->ldk;
      pa.Kdim = pro->Kdim;
      pa.w = pro->w;
      pa.r = pro->r;
      pa.eta2 = pro->eta2;
      pa.Kinv = pro->Kinv;
      pa.ldm = pro->ldm;
      pa.eta1 = pro->eta1;
      pa.kinv_mu0 = pro->kinv_mu0;
      pa.lr = pro->lr;
      pa.packed = pro->packed;
      pa.Cout = pro->Cout;
      pa.tred = pro->tred;
      pa.arrive = pro->arrive;
      pa.arrive_want = pro->arrive_want;
      memcpy(pa.grp, pro->grp, sizeof(pa.grp));
      pa.HS = H + (3 * nt + (nt + ne + nx) * nt) * TILE * TILE;
      pa.sflags = c->dag_flags + ((nt + ne + nx) * nt + 3 * nt + 1) * DAG_FS;
      const int other = hs ^ 1;
      if (c->hset[other] && c->h_dirty[other].on && c->h_dirty[other].nb == 1 && c->htype == (int)sizeof(T)) {
        pa.fill = (T*)c->hset[other];  // the set the launch before this one used: refilled in this launch's shadow
        pa.fill_n = c->h_dirty[other].used;
        pa.nfill = 64;
        c->h_dirty[other].on = false;
      }
      unsigned long long* const ptrace = nullptr;  // (wall-clock stamps of the prologue, PRO_TS in agp_chol.h: a development aid)
      constexpr unsigned lds_pad = 0;
      if (step_inst && chain_split_wanted(ntiles + nhelp, true, sizeof(T) == 8) && chain_split_ready(c)) {  // chain kernel + tile kernel (k_chol_dag, ROLE)
        // No event joins the chain stream behind a split launch (see chain_split_arm): correct only as long as the chain kernel
        // stores nothing after its last publish.  C = S + K^-1 / 4 (ProArgs::Cout) is a plain store of tile (0, 0)'s workgroup --
        // in a split launch that would be the chain kernel, whose plain stores only become visible when THAT kernel ends.  Cout
        // exists for launches with the inverse (do_x), which never split; enforced here rather than assumed.
        if (pa.Cout) {
          c->err = "potrf_fused: C = S + K^-1/4 (ProArgs::Cout) cannot ride on a split (chain kernel + tile kernel) launch";
          return AGP_ERR_INVALID;
        }
        chain_split_arm(c, ds, 1);
        split_used = true;
        hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true, true, 1>), dim3(1), dim3(CHOL_THREADS), 0, c->chain_stream, one, 1,
                           (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, ptrace, H, hstride, nx, erow, 0,
                           ds, pa, epi ? *epi : EpiArgs<T>{});
        chain_split_wait_here(c);
        hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true, true, 2>), dim3((unsigned)(ntiles + nhelp + pa.nfill)),
                           dim3(CHOL_THREADS), lds_pad, c->stream, one, 1, (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid,
                           c->dag_flags, c->dag_epoch, ptrace, H, hstride, nx, erow, 0, ds, pa, epi ? *epi : EpiArgs<T>{});
        LAUNCHCHK(c);
      } else if (step_inst)
        hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true, true>), dim3((unsigned)(ntiles + nhelp + pa.nfill)),
                           dim3(CHOL_THREADS), lds_pad, c->stream, one, 1, (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid,
                           c->dag_flags, c->dag_epoch, ptrace, H, hstride, nx, erow, 0, ds, pa, epi ? *epi : EpiArgs<T>{});
      else {
        pd.base = ntiles + nhelp + pa.nfill;
        hipLaunchKernelGGL((k_chol_dag<T, true, false, false, false, true>), dim3((unsigned)(ntiles + nhelp + pa.nfill + nprod)),
                           dim3(CHOL_THREADS), lds_pad, c->stream, one, 1, (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid,
                           c->dag_flags, c->dag_epoch, ptrace, H, hstride, nx, erow, (int)(do_x && nx == 0) | (want_l ? 2 : 0),
                           DagSync{}, pa, EpiArgs<T>{}, pd);
      }
      c->h_step_set = other;
    } else if (step_inst && chain_split_wanted(ntiles) && chain_split_ready(c)) {
      // ... as two kernels: the chain workgroup on its own stream (enqueued first), every other tile on this one
      chain_split_arm(c, ds, 1);
      split_used = true;
      hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true, false, 1>), dim3(1), dim3(CHOL_THREADS), 0, c->chain_stream, one, 1,
                         (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, trace, H, hstride, nx, erow,
                         0, ds);
      chain_split_wait_here(c);
      hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true, false, 2>), dim3((unsigned)ntiles), dim3(CHOL_THREADS), 0, c->stream,
                         one, 1, (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, trace, H, hstride, nx,
                         erow, 0, ds);
      LAUNCHCHK(c);
    } else if (step_inst)  // the CAVI step's launch: specialised instantiation
      hipLaunchKernelGGL((k_chol_dag<T, true, false, false, true>), dim3((unsigned)ntiles), dim3(CHOL_THREADS), 0, c->stream, one, 1,
                         (int64_t)0, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, trace, H, hstride, nx, erow,
                         0, ds);
    else {
      pd.base = ntiles;
      hipLaunchKernelGGL((k_chol_dag<T, true>), dim3((unsigned)(ntiles + nprod)), dim3(CHOL_THREADS), 0, c->stream, one, 1, (int64_t)0,
                         ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, trace, H, hstride, nx, erow,
                         (int)(do_x && nx == 0) | (want_l ? 2 : 0), DagSync{}, ProArgs<T>{}, EpiArgs<T>{}, pd);
    }
    LAUNCHCHK(c);
    AGPCHK(dag_handover_release<T>(c, hused, hstride, 1, hs));
    const bool test_abort = dag_test_abort();
    if (safe) {
      safe->chain_done = split_used ? c->chain_ctr : nullptr;
      safe->chain_want = c->chain_exits;
    }
    if (safe && (!do_x || safe->want_x)) {
      if (test_abort) hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, c->stream, info_dev, -1);
      if (can_defer) *defer_safe = true;
      else AGPCHK(launch_chol_safe<T>(c, one, *safe, 1, ld, ldx, lde, ne, nt, info_dev, nvalid));
    }
    return AGP_OK;  // X = L^-1 came out of the same launch
  }
  if (erow && ne > 0)
    hipLaunchKernelGGL((k_set_ext_rows<T>), dim3((unsigned)((TILE * n + 255) / 256)), dim3(256), 0, c->stream,
                       E + (ne - 1) * TILE * lde, lde, n, erow);
  CholBatch<T> bt{};
  bt.A[0] = A;
  bt.X[0] = X;
  bt.Dg[0] = Dg;
  bt.E[0] = E;
  AGPCHK(chol_columns<T>(c, bt, 1, ld, ldx, lde, ne, do_x, nt, info_dev, nvalid));
  LAUNCHCHK(c);
  if (do_x) AGPCHK(trtri_levels<T>(c, (const T*)A, ld, X, ldx, nt));
  return AGP_OK;
}

// Task-graph launches whose inputs cannot be restored on the device (the factor is written in place: K_ZZ, the building blocks;
// or the inverse rides along) are checked on the host instead: synchronise, and if the latch reads -1 stop using the task graph
// on this context and tell the caller to rebuild its input and factor again (now with per-column launches).
static agp_status dag_lost_dependency(agp_ctx* c, int32_t* info_dev, bool* lost) {
  *lost = false;
  int32_t info = 0;
  HIPCHK(c, hipMemcpyAsync(&info, info_dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (info == -1) {
    HIPCHK(c, hipMemsetAsync(info_dev, 0, sizeof(int32_t), c->stream));
    if (!c->dag_off)
      fprintf(stderr, "[agp_hip] warning: a task-graph factorisation lost a tile dependency (is another process using this GPU?); "
                      "re-running it with plain launches, which this context uses for the next %lld steps\n",
              (long long)c->dag_backoff);
    dag_pause(c);
    *lost = true;
  }
  return AGP_OK;
}

// nb <= DAG_MAX_NB independent problems of identical shape as ONE interleaved task-graph launch (see k_chol_dag): their chains
// run side by side on nb CUs (workgroup index = tile * nb + problem: with 8 problems each one lives on its own XCD).
constexpr int DAG_MAX_NB = 8;
template <typename T>
static agp_status potrf_dag_batch(agp_ctx* c, const CholBatch<T>& bt, int nb, int64_t ld, int64_t n, int64_t ldx, int64_t lde,
                                  int64_t ne, int32_t* info_dev, int64_t nvalid, SafeSrc<T>* safe = nullptr) {
  const int64_t nt = n / TILE;
  bool split_used = false;
  const int64_t fstride = ((nt + ne) * nt + 3 * nt + 1) * DAG_FS, nf = fstride * nb;
  if (c->dag_cap < nf) {
    if (c->dag_flags) (void)hipFree(c->dag_flags);
    c->dag_flags = nullptr;
    c->dag_cap = 0;
    if (hipMalloc((void**)&c->dag_flags, sizeof(int32_t) * (size_t)(nf + 1024)) != hipSuccess) return AGP_ERR_NOMEM;
    c->dag_cap = nf + 1024;
    HIPCHK(c, hipMemsetAsync(c->dag_flags, 0, sizeof(int32_t) * (size_t)c->dag_cap, c->stream));
    c->dag_epoch = 0;
  }
  c->dag_epoch += 1;
  const int64_t ntiles = nt * (nt + 1) / 2 + ne * nt;
  const int64_t hstride = ((2 * nt + ne) * nt + 3 * nt) * TILE * TILE;
  T* H = nullptr;
  const int hs = 0;
  AGPCHK(dag_handover_acquire<T>(c, hstride * nb, hs, &H));
  if (chain_split_wanted(ntiles * nb) && chain_split_ready(c)) {  // the nb chains as one kernel, all other tiles as another
    DagSync ds{};
    chain_split_arm(c, ds, nb);
    split_used = true;
    hipLaunchKernelGGL((k_chol_dag<T, true, true, false, true, false, 1>), dim3((unsigned)nb), dim3(CHOL_THREADS), 0, c->chain_stream, bt,
                       nb, fstride, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, (unsigned long long*)nullptr, H,
                       hstride, (int64_t)0, (const T*)nullptr, 0, ds);
    chain_split_wait_here(c);
    hipLaunchKernelGGL((k_chol_dag<T, true, true, false, true, false, 2>), dim3((unsigned)(ntiles * nb)), dim3(CHOL_THREADS), 0,
                       c->stream, bt, nb, fstride, ld, ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch,
                       (unsigned long long*)nullptr, H, hstride, (int64_t)0, (const T*)nullptr, 0, ds);
    LAUNCHCHK(c);
  } else
    hipLaunchKernelGGL((k_chol_dag<T, true, true, false, true>), dim3((unsigned)(ntiles * nb)), dim3(CHOL_THREADS), 0, c->stream, bt, nb, fstride, ld,
                       ldx, lde, ne, nt, info_dev, nvalid, c->dag_flags, c->dag_epoch, (unsigned long long*)nullptr, H, hstride,
                       (int64_t)0, (const T*)nullptr, 0, DagSync{});
  LAUNCHCHK(c);
  AGPCHK(dag_handover_release<T>(c, (3 * nt + (nt + ne) * nt) * TILE * TILE, hstride, nb, hs));
  if (safe) {
    safe->chain_done = split_used ? c->chain_ctr : nullptr;
    safe->chain_want = c->chain_exits;
    if (dag_test_abort()) hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, c->stream, info_dev, -1);
    AGPCHK(launch_chol_safe<T>(c, bt, *safe, nb, ld, ldx, lde, ne, nt, info_dev, nvalid));
  }
  return AGP_OK;
}

// the same factorisation for nb <= CHOL_MAXB independent problems of identical shape in shared launches (no X = L^-1)
template <typename T>
static agp_status potrf_fused_batch(agp_ctx* c, const CholBatch<T>& bt, int nb, int64_t ld, int64_t n, int64_t ldx,
                                    int64_t lde, int64_t ne, int32_t* info_dev, int64_t nvalid) {
  const int64_t nt = n / TILE;
  AGPCHK(chol_columns<T>(c, bt, nb, ld, ldx, lde, ne, 0, nt, info_dev, nvalid));
  LAUNCHCHK(c);
  return AGP_OK;
}

// Up to this many C tiles a GEMM / symmetric-product launch uses two k-groups per workgroup (512 threads, two waves per SIMD):
// one four-wave workgroup reaches about half of a CU's MFMA rate, and up to ~4 workgroups per CU the second k-group is worth
// more than the extra tiles in flight (measured, step times with 320 -> 1100: fp32 m = B = 2048 0.821 -> 0.789 ms, fp64 m = B =
// 1536 0.703 -> 0.687 ms, 2048 1.37 -> 1.33 ms; C2's 256 / 136 tiles were below the old limit already).
static constexpr int64_t kg2_limit() { return 1100; }
static constexpr int64_t syrk_kg2_limit() { return kg2_limit(); }
static int ctx_cus(agp_ctx* c) {
  if (c->n_cu <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || v <= 0) v = 64;
    c->n_cu = v;
  }
  return c->n_cu;
}
static agp_status ensure_bal_ws(agp_ctx* c, size_t need) {
  if (c->bal_bytes < need) {
    if (c->bal_ws) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->bal_ws);
    }
    c->bal_ws = nullptr;
    c->bal_bytes = 0;
    HIPCHK(c, hipMalloc(&c->bal_ws, need));
    c->bal_bytes = need;
  }
  return AGP_OK;
}

// S = A' diag(w) A (lower tiles mirrored), two k-groups per workgroup when the tile count underfills the chip
template <typename T, int MODE>
static agp_status syrk_tn(agp_ctx* c, const T* A, int64_t lda, int64_t n, int64_t Kdim, const T* w, int lower_a, T* out,
                          int64_t ldo, T* eta2, const T* Kinv, int64_t ldm, T lr, const T* rvec = nullptr,
                          T* eta1 = nullptr, const T* kinv_mu0 = nullptr) {
  // rvec: nt rider workgroups also step eta1 (see k_syrk_tn); a dirty hand-over set of the task-graph Cholesky is refilled by
  // further riders (from the fused step, MODE == SY_ETA2 with rvec, and from the packed statistics of the batch-parallel step,
  // MODE == SY_PACK with rvec, whose riders store t = A' rvec into `eta1` instead of stepping it)
  const int64_t nt = n / TILE, tiles = nt * (nt + 1) / 2, nrider = rvec ? nt : 0;
  T* fillp = nullptr;
  int64_t fused_used = 0, fstride = 0, nfill = 0;
  int fnb = 0;
  // (round 3: any symmetric-product launch refills a dirty set, e.g. X'X behind a factorisation with its inverse: the inline refill
  //  in front of the NEXT task graph was a 6 us launch of its own on the hyper-parameter iteration's path)
  if (c->h_dirty[0].on && c->htype == (int)sizeof(T) && out != nullptr) {
    fillp = (T*)c->hset[0];
    fused_used = c->h_dirty[0].used;
    fstride = c->h_dirty[0].stride;
    fnb = c->h_dirty[0].nb;
    nfill = 96;
    c->h_dirty[0].on = false;
  }
  // up to 160 tiles (C2: 136 on 256 CUs, one workgroup per CU) four k-groups: 16 waves per CU instead of 8 -- 58 -> 52 us at C2
  // (step 0.379 -> 0.3735 ms)
  const int kg = (tiles <= 160 && Kdim >= 8 * BK) ? 4 : (tiles <= syrk_kg2_limit() && Kdim >= 4 * BK) ? 2 : 1;
  const int64_t grid = tiles + nrider + nfill;
  if (kg == 4)
    hipLaunchKernelGGL((k_syrk_tn<T, MODE, 4>), dim3((unsigned)grid), dim3(4 * NTHREADS), 0, c->stream, A, lda, Kdim, w,
                       lower_a, out, ldo, eta2, Kinv, ldm, lr, tiles, rvec, eta1, kinv_mu0, nrider, fillp, fused_used, fstride,
                       fnb);
  else if (kg == 2)
    hipLaunchKernelGGL((k_syrk_tn<T, MODE, 2>), dim3((unsigned)grid), dim3(2 * NTHREADS), 0, c->stream, A, lda, Kdim, w,
                       lower_a, out, ldo, eta2, Kinv, ldm, lr, tiles, rvec, eta1, kinv_mu0, nrider, fillp, fused_used, fstride,
                       fnb);
  else
    hipLaunchKernelGGL((k_syrk_tn<T, MODE, 1>), dim3((unsigned)grid), dim3(NTHREADS), 0, c->stream, A, lda, Kdim, w,
                       lower_a, out, ldo, eta2, Kinv, ldm, lr, tiles, rvec, eta1, kinv_mu0, nrider, fillp, fused_used, fstride,
                       fnb);
  LAUNCHCHK(c);
  return AGP_OK;
}

// out = X' X for lower-triangular X  (A^-1 from its inverse Cholesky factor): the symmetric product with the k range of every
// tile starting at its first row, k-groups chosen like everywhere else (it ran with one k-group on 136 tiles: 60 us at m = 1024)
// Round 4: from 8 block rows on, the balanced form (k_xtx_bal: units of at most ch k-blocks, partial tiles added by the last arriver
// in unit order): 42 -> ~15 us at m = 1024 (below 8 block rows the one-workgroup-per-tile product stays).
// (Dg ...: log det from the diagonal factors rides on the reduction launch; *rider_done says whether it did)
template <typename T>
static agp_status xtx_padded(agp_ctx* c, const T* X, int64_t ld, int64_t n, T* out, int64_t ldo, const T* Dg = nullptr,
                             int64_t nvalid = 0, double* ld_out = nullptr, int32_t* status = nullptr, bool* rider_done = nullptr) {
  if (rider_done) *rider_done = false;
  const int64_t nt = n / TILE;
  if (nt < 8)
    return syrk_tn<T, SY_STORE>(c, X, ld, n, n, (const T*)nullptr, 1, out, ldo, (T*)nullptr, (const T*)nullptr, (int64_t)0, T(0));
  const int ch = (int)std::max<int64_t>(2, (nt + XTX_MAXU - 1) / XTX_MAXU);  // at most XTX_MAXU units per tile
  const int64_t nunits = xtx_bal_units(nt, ch), ntri = nt * (nt + 1) / 2;
  AGPCHK(ensure_bal_ws(c, sizeof(T) * (size_t)nunits * TILE * TILE));
  T* fillp = nullptr;
  int64_t fused_used = 0, fstride = 0, nfill = 0;
  int fnb = 0;
  if (c->h_dirty[0].on && c->htype == (int)sizeof(T)) {  // hand-over refill riders, as in syrk_tn()
    fillp = (T*)c->hset[0];
    fused_used = c->h_dirty[0].used;
    fstride = c->h_dirty[0].stride;
    fnb = c->h_dirty[0].nb;
    nfill = 96;
    c->h_dirty[0].on = false;
  }
  hipLaunchKernelGGL((k_xtx_bal<T, 1>), dim3((unsigned)(nunits + nfill)), dim3(NTHREADS), 0, c->stream, X, ld, n, out, ldo,
                       (T*)c->bal_ws, ch, nunits, fillp, fused_used, fstride, fnb);
  const bool rider = Dg != nullptr && ld_out != nullptr;
  hipLaunchKernelGGL((k_xtx_bal_reduce<T>), dim3((unsigned)(ntri + (rider ? 1 : 0))), dim3(NTHREADS), 0, c->stream, n, out, ldo,
                     (const T*)c->bal_ws, ch, Dg, nvalid, ld_out, status);
  LAUNCHCHK(c);
  if (rider && rider_done) *rider_done = true;
  return AGP_OK;
}

template <typename T, int EPI>
static agp_status gemm_nt(agp_ctx* c, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                          int tri_b, T* C, int64_t ldc, const T* E, int64_t lde, const T* v, T* p0, T* p1,
                          int64_t ldp, const HkArgs<T>* hk = nullptr) {
  dim3 g((unsigned)(N / TILE), (unsigned)(M / TILE));
  const HkArgs<T> hka = hk ? *hk : HkArgs<T>{};  // (EPI_HK only)
  // round 6: 128 x 64 C tiles (k_gemm_nt_tall) for the large fp64 products -- more 64-tiles than two k-groups are used for (C5's
  // 4096-tile kappa GEMM).  AGP_GEMM_TALL=0 / 1 forces (1: wherever the shape allows, M a multiple of 128).
  if constexpr (EPI == EPI_STORE || EPI == EPI_KAPPA) {
    static const int tall = []() {
      const char* e = getenv("AGP_GEMM_TALL");
      return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    const bool big = (N / TILE) * (M / TILE) > kg2_limit();
    if (M % (2 * TILE) == 0 && K >= BK && (tall == 1 || (tall < 0 && big && sizeof(T) == 8))) {
      dim3 gt((unsigned)(N / TILE), (unsigned)(M / (2 * TILE)));
      hipLaunchKernelGGL((k_gemm_nt_tall<T, EPI>), gt, dim3(NTHREADS), 0, c->stream, A, lda, B, ldb, K, tri_b, C, ldc, E, lde, p0, p1,
                         ldp);
      LAUNCHCHK(c);
      return AGP_OK;
    }
  }
  // fewer tiles than ~1.25 waves of CUs: two k-groups per workgroup (2 waves per SIMD) instead of idle SIMD slots
  if ((N / TILE) * (M / TILE) <= kg2_limit() && K >= 4 * BK)
    hipLaunchKernelGGL((k_gemm_nt<T, EPI, 2>), g, dim3(2 * NTHREADS), 0, c->stream, A, lda, B, ldb, K, tri_b, C, ldc, E,
                       lde, v, p0, p1, ldp, hka);
  else
    hipLaunchKernelGGL((k_gemm_nt<T, EPI, 1>), g, dim3(NTHREADS), 0, c->stream, A, lda, B, ldb, K, tri_b, C, ldc, E, lde,
                       v, p0, p1, ldp, hka);
  LAUNCHCHK(c);
  return AGP_OK;
}


// kernelmatrix launch: the MFMA form (k_kernelmatrix_mma) up to D = KMM_MAXD, the direct-difference VALU kernel beyond (or with
// AGP_KERNELMATRIX_VALU=1).  Same arguments as the kernels; `cgroups` = number of column groups a fused row-dot is split into
// (<= 0: one group per column tile, like the VALU kernel; 1: the whole row in one workgroup -- streaming prediction).  Returns
// the number of partial slices the row-dot consumer has to sum.
// The MFMA kernel reads the Y side as ready-made tiles (scaled, zero-padded, with squared norms: k_scale_rows).  Callers whose Y
// is a latent's inducing points pass the cached copy (ysc / ysn, see Svgp::ensure_zsc); otherwise the copy is made here into a
// per-context scratch on the same stream (c may be null only together with a cached copy).
static inline int kmm_dp(int64_t D) { return (int)((D + 7) / 8 * 8); }
static inline bool kmm_usable(int64_t D) {
  static const bool force_valu = []() {
    const char* e = getenv("AGP_KERNELMATRIX_VALU");
    return e && e[0] == '1';
  }();
  return D <= KMM_MAXD && !force_valu;
}
template <typename T>
static int launch_kernelmatrix(agp_ctx* c, hipStream_t stream, const T* X, int64_t ldx, const int64_t* idx, int64_t n, const T* Y,
                               int64_t ldy, int64_t p, int64_t D, const T* scales, int kind, T variance, T* out, int64_t ldo,
                               int64_t n_out, int64_t p_out, int sym, T diag_add, const T* alpha, T* part, int64_t ldp,
                               int64_t cgroups = 0, const T* ysc = nullptr, const T* ysn = nullptr) {
  const int64_t nct = (p_out + TILE - 1) / TILE, nrt = (n_out + TILE - 1) / TILE;
  if (!kmm_usable(D)) {
    hipLaunchKernelGGL((k_kernelmatrix<T>), dim3((unsigned)nct, (unsigned)nrt), dim3(NTHREADS), 0, stream, X, ldx, idx, n, Y, ldy, p,
                       D, scales, kind, variance, out, ldo, n_out, p_out, sym, diag_add, alpha, part, ldp);
    return (int)nct;
  }
  const int Dp = kmm_dp(D);
  const int64_t p_pad = nct * TILE;
  if (!ysc) {
    const size_t need = sizeof(T) * (size_t)(p_pad * Dp + p_pad);
    if (c->kmm_bytes < need) {
      if (c->kmm_scratch) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(c->kmm_scratch);
      }
      c->kmm_scratch = nullptr;
      c->kmm_bytes = 0;
      if (hipMalloc(&c->kmm_scratch, need + need / 4) != hipSuccess) return -1;
      c->kmm_bytes = need + need / 4;
    }
    T* sc0 = (T*)c->kmm_scratch;
    hipLaunchKernelGGL((k_scale_rows<T>), dim3((unsigned)((p_pad + 3) / 4)), dim3(256), 0, stream, Y, ldy, p, p_pad, D, Dp, scales, sc0,
                       sc0 + p_pad * Dp);
    ysc = sc0;
    ysn = sc0 + p_pad * Dp;
  }
  const size_t sh = kmm_smem_bytes<T>(Dp);
  const int64_t groups = cgroups <= 0 ? nct : std::min<int64_t>(cgroups, nct);
  const int64_t ctiles = (nct + groups - 1) / groups;
  const int64_t g_eff = (nct + ctiles - 1) / ctiles;
  const dim3 grid((unsigned)g_eff, (unsigned)nrt);
#define AGP_KMM_LAUNCH(KIND)                                                                                                  \
  do {                                                                                                                        \
    if (sh > 64 * 1024) { /* more than 64 KB of dynamic LDS has to be requested once per kernel */                            \
      static size_t asked = 0;                                                                                                \
      if (sh > asked) {                                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kernelmatrix_mma<T, KIND>),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kernelmatrix_mma<T, KIND, 1>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kernelmatrix_mma<T, KIND, 2>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kernelmatrix_mma<T, KIND, 3>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);                                       \
        asked = sh;                                                                                                           \
      }                                                                                                                       \
    }                                                                                                                         \
    if (out == nullptr && !sym && alpha != nullptr)                                                                           \
      hipLaunchKernelGGL((k_kernelmatrix_mma<T, KIND, 1>), grid, dim3(NTHREADS), sh, stream, X, ldx, idx, n, ysc, ysn, p, D, \
                         Dp, scales, variance, out, ldo, n_out, p_out, sym, diag_add, alpha, part, ldp, ctiles);              \
    else if (out != nullptr && !sym && alpha == nullptr)                                                                      \
      hipLaunchKernelGGL((k_kernelmatrix_mma<T, KIND, 2>), grid, dim3(NTHREADS), sh, stream, X, ldx, idx, n, ysc, ysn, p, D, \
                         Dp, scales, variance, out, ldo, n_out, p_out, sym, diag_add, alpha, part, ldp, ctiles);              \
    else if (out != nullptr && sym && alpha == nullptr)                                                                       \
      hipLaunchKernelGGL((k_kernelmatrix_mma<T, KIND, 3>), grid, dim3(NTHREADS), sh, stream, X, ldx, idx, n, ysc, ysn, p, D, \
                         Dp, scales, variance, out, ldo, n_out, p_out, sym, diag_add, alpha, part, ldp, ctiles);              \
    else                                                                                                                      \
      hipLaunchKernelGGL((k_kernelmatrix_mma<T, KIND>), grid, dim3(NTHREADS), sh, stream, X, ldx, idx, n, ysc, ysn, p, D,    \
                         Dp, scales, variance, out, ldo, n_out, p_out, sym, diag_add, alpha, part, ldp, ctiles);              \
  } while (0)
  switch (kind) {
    case AGP_K_SQEXP: AGP_KMM_LAUNCH(K_SQEXP); break;
    case AGP_K_MATERN52: AGP_KMM_LAUNCH(K_MATERN52); break;
    case AGP_K_MATERN32: AGP_KMM_LAUNCH(K_MATERN32); break;
    default: AGP_KMM_LAUNCH(K_EXPONENTIAL); break;
  }
#undef AGP_KMM_LAUNCH
  return (int)g_eff;
}

// ---- model ---------------------------------------------------------------------------------------------------
struct KernelHost {
  int kind = AGP_K_SQEXP;
  double variance = 1.0;
  bool ard = false;
  bool has_variance = true, has_transform = true;  // structure of the kernel object (agp_kernel_desc): what update_kernel! steps
  std::vector<double> scales;  // length D
};

struct SvgpBase {
  agp_ctx* ctx = nullptr;
  agp_svgp_desc desc{};
  virtual ~SvgpBase() {}
  virtual agp_status init() = 0;
  virtual agp_status set_kernel(int l, const agp_kernel_desc* k) = 0;
  virtual agp_status set_Z(int l, const void* z, int64_t ldz) = 0;
  virtual agp_status get_Z(int l, void* z, int64_t ldz) = 0;
  virtual agp_status set_mu0(int l, const void* mu0) = 0;
  virtual agp_status refresh_K() = 0;
  virtual agp_status step_local(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho,
                                bool fresh) = 0;
  virtual agp_status prefetch(const void* x, int64_t ldx, const int64_t* idx, int64_t B) = 0;
  virtual agp_status set_multioutput(int n_task, const agp_lik_desc* liks, const double* A_host, double eta, double b1,
                                     double b2, double eps) = 0;
  virtual agp_status get_A(double* A_host) = 0;
  virtual agp_status mo_shard(int q_total) = 0;
  virtual agp_status mo_fbuf_ptr(void** p, int64_t* n) = 0;
  virtual agp_status mo_mix() = 0;
  virtual agp_status mo_refresh_f() = 0;
  virtual agp_status mo_predict_from_f(int64_t nt, int mode, void* o0, void* o1, const double* nodes,
                                       const double* weights, int nn) = 0;
  virtual agp_status hyper_rule(int k_rule, double k_rho, int z_rule, double z_rho) = 0;
  virtual agp_status hyper_configure(int opt_k, double k_eta, int opt_z, double z_eta, double b1, double b2,
                                     double eps) = 0;
  virtual agp_status hypergrad(int l, double* dvar, double* dscale, void* dZ) = 0;
  virtual agp_status hyper_step() = 0;
  virtual agp_status get_kernel(int l, double* var, double* scales) = 0;
  virtual agp_status lsm_gamma() = 0;
  virtual agp_status lsm_alpha() = 0;
  virtual agp_status lsm_local_all() = 0;  // both rounds and the final theta, r, w of a handle holding every latent
  virtual agp_status lsm_gsum_ptr(void** p, int64_t* n) = 0;
  virtual agp_status step_stats(bool fused) = 0;
  virtual agp_status stats_ptr(void** p, int64_t* n) = 0;
#ifdef AGP_DEBUG_PTRS  // (development builds only: device addresses of the step's state, tools/tmp)
  virtual void* debug_ptr(int what) { (void)what; return nullptr; }
#endif
  virtual agp_status step_global(bool fused) = 0;
  // tail of agp_svgp_cavi_step: the fused natural-gradient step now, or left pending for the next step's task-graph launch
  virtual agp_status step_finish() = 0;
  // applies a pending natural-gradient step with the stand-alone kernel (every entry point other than the CAVI step calls this
  // first, so eta, Sigma, predictions, the ELBO and the hyper-gradient never see a half-taken step)
  virtual agp_status flush() = 0;
  virtual agp_status check_status() = 0;
  virtual agp_status elbo(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho,
                          int fresh, double* out) = 0;
  virtual agp_status elbo_terms(double* out3) = 0;
  virtual agp_status elbo_enqueue(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho, int fresh,
                                  int32_t* ticket) = 0;
  virtual agp_status elbo_fetch(int32_t ticket, int wait, double* out, int32_t* ready) = 0;
  virtual agp_status set_batch_shard(int rank, int world) = 0;
  virtual agp_status get_state(int l, void* mu, void* sigma, void* eta1, void* eta2) = 0;
  virtual agp_status set_state(int l, const void* eta1, const void* eta2) = 0;
  virtual agp_status get_matrix(int l, int which, void* out, int64_t ldo, int64_t cap) = 0;
  virtual int64_t last_batch() = 0;
  virtual agp_status init_state() = 0;
  virtual agp_status invalidate_data() = 0;
  virtual agp_status predict_f_cov(const void* xt, int64_t ldx, int64_t nt, void* mu, void* cov) = 0;
  virtual agp_status refresh_K_explicit() = 0;
  // multi-GPU (agp_comm.h)
  virtual agp_status cavi_step_multi(agp_comm* cm, int mode, const void* x, int64_t ldx, const void* y, const int64_t* idx,
                                     int64_t B, double rho) = 0;
  virtual agp_status elbo_multi(agp_comm* cm, int mode, double* out) = 0;
  virtual agp_status hyper_step_multi(agp_comm* cm, int tied) = 0;
  virtual agp_status predict_multi(agp_comm* cm, int what, const void* xt, int64_t ldx, int64_t nt, void* mu, void* var,
                                   const double* nodes, const double* weights, int nn) = 0;
  virtual agp_status predict_f(const void* xt, int64_t ldx, int64_t nt, void* mu, void* var) = 0;
  virtual agp_status predict_y(const void* xt, int64_t ldx, int64_t nt, void* out) = 0;
  virtual agp_status proba_y(const void* xt, int64_t ldx, int64_t nt, const double* nodes, const double* weights,
                             int nn, void* o0, void* o1) = 0;
  virtual agp_status set_quadrature(const double* nodes, const double* weights, int nn) = 0;
  virtual agp_status hyper_state(int l, int set, double* k_m, double* k_v, int32_t* k_step) = 0;
  virtual agp_status hyper_apply(int l, const double* dvar, const double* dscale, const void* dZ) = 0;
  virtual agp_status set_lsm_alpha(const void* a, int64_t n) = 0;
  virtual agp_status set_online_prior(int l, const void* za, int64_t ldza, int64_t ma, const void* invDa, int64_t ldi,
                                      const void* peta1, double prevLa) = 0;
  virtual agp_status online_snapshot(int l, void* invDa_out, int64_t ldi, void* eta1_out, double* prevLa_host) = 0;
  virtual agp_status adopt_local(SvgpBase* src) = 0;
  virtual agp_status get_lik_param(double* out) = 0;
  virtual agp_status set_lik_param(double v) = 0;
  int64_t n_opt = 1;  // RobbinsMonro counter (optimisers.jl:12)
  bool in_cavi_step = false;  // step_local is running as the first half of agp_svgp_cavi_step (its tail may then be deferred)
  int64_t n_prologue = 0;  // CAVI steps whose natural-gradient part rode on the next step's task-graph launch (agp_svgp_step_counters)
  int64_t n_steps = 0;     // agp_svgp_cavi_step calls
  int64_t n_hgrad = 0, n_gk_fused = 0;  // hyper-gradient evaluations / those with the one-product G_K (agp_svgp_hyper_counters)
  // HIP-event timing of the dominant kernel sequence (agp_svgp_timing_*)
  bool timing = false;
  int timing_every = 1, timing_ctr = 0;  // every n-th sequence is bracketed (the two event records cost the step ~16 us at C2)
  bool timing_now = false;
  std::vector<hipEvent_t> ev;
  std::vector<int64_t> ev_launches;
  size_t ev_used = 0;
  agp_status timing_begin() {
    timing_now = timing && (timing_ctr++ % timing_every) == 0;
    if (!timing_now) return AGP_OK;
    if (ev_used + 2 > ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreate(&e));
        ev.push_back(e);
      }
    }
    HIPCHK(ctx, hipEventRecord(ev[ev_used], ctx->stream));
    return AGP_OK;
  }
  agp_status timing_end(int64_t launches) {
    if (!timing_now) return AGP_OK;
    timing_now = false;
    HIPCHK(ctx, hipEventRecord(ev[ev_used + 1], ctx->stream));
    ev_used += 2;
    ev_launches.push_back(launches);
    return AGP_OK;
  }
  agp_status timing_read(int64_t* n, double* ms) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n = 0;
    *ms = 0.0;
    for (size_t i = 0; i + 1 < ev_used; i += 2) {
      float t = 0;
      HIPCHK(ctx, hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      *ms += t;
      *n += ev_launches[i / 2];
    }
    ev_used = 0;
    ev_launches.clear();
    return AGP_OK;
  }
};

// the last line of the ELBO on the device (agp_svgp_elbo_enqueue): rho E_data - KL(q(u) || p(u)) - rho KL_aug from the five partial
// results of the evaluation and log det K, written to mapped host memory -- no stream synchronisation on the host's side
__global__ void k_elbo_combine(const double* __restrict__ sc, const double* __restrict__ half_logdetK, double m, double rho,
                               double* __restrict__ out) {
  const double kl = 0.5 * (2.0 * half_logdetK[0] + 2.0 * sc[2] + sc[3] + sc[4] - m);  // log det Sigma = -2 sc[2]
  *out = rho * sc[0] - kl - rho * sc[1];
  __threadfence_system();
}

struct agp_svgp {
  SvgpBase* impl;
};

// teardown helpers: a failing free / destroy is a bug in this library, so it is reported, never swallowed
static void dcheck(hipError_t e, int line) {
  if (e != hipSuccess) fprintf(stderr, "[agp_hip] teardown: %s (agp_capi.hip:%d)\n", hipGetErrorString(e), line);
}
#define dfree(p) dcheck(hipFree(p), __LINE__)

template <typename T>
struct Svgp : SvgpBase {
  struct Latent {
    KernelHost k;
    T* scales = nullptr;  // device, D
    T* Z = nullptr;       // m x D
    T *Zsc = nullptr, *zn = nullptr;  // scaled, tile-padded copy of Z (mp x Dp) and its squared norms: the Y side of the MFMA
    bool zsc_valid = false;           // kernel matrix (k_scale_rows); redone when Z or the scales change
    T* L = nullptr;       // mp x mp : K then chol(K) (lower)
    T* Xk = nullptr;      // L^-1
    T* Kinv = nullptr;
    T* mu0 = nullptr;
    T* kinv_mu0 = nullptr;
    T* eta1 = nullptr;
    T* eta2 = nullptr;
    T* La = nullptr;      // -2*eta2 then its Cholesky factor
    T* Xa = nullptr;      // La^-1   (Sigma = Xa' Xa)
    T* v = nullptr;       // Xa eta1 (mu = Xa' v)
    T* Sigma = nullptr;
    T* mu = nullptr;
    T* Knm = nullptr;
    T* kappa = nullptr;
    T* Wbuf = nullptr;    // (Bp + 64) x mp : [kappa ; eta1'] -> [W ; v'] by the augmented Cholesky
    T* pk = nullptr;      // K~ partial slices of this latent [2*mp/64][ldp]
    T *Knm_alt = nullptr, *kappa_alt = nullptr, *Wbuf_alt = nullptr, *pk_alt = nullptr;  // prefetch targets
    T* kappa_old = nullptr;  // third kappa buffer: the minibatch whose natural-gradient step is still pending (see Pending)
    T* DgK = nullptr;     // diagonal 64x64 factors of chol(K)      (mp x 64)
    T* DgA = nullptr;     // diagonal 64x64 factors of chol(-2 eta2)
    T* Apred = nullptr;   // K^-1 - K^-1 Sigma K^-1
    // hyper-parameter iteration: C = kappa' diag(w) kappa + K^-1 / 4 of the natural-gradient step that the factorisation launch took
    // as its prologue (ProArgs::Cout); valid until the next local step / natural-gradient step (hypergrad, k_hyper_gK_fused)
    T* Cmat = nullptr;
    bool C_valid = false;
    const T* C_kap = nullptr;  // the kappa buffer it was formed from
    T* apred = nullptr;   // K^-1 mu
    bool K_stale = true, post_valid = false, pred_valid = false, predvar_valid = false, kappa_valid = false;
    bool kappa_eval = false;  // kappa / Knm / pk hold the kernel matrices of the last FRESH evaluation batch (ev_x ..., round 6)
    bool keep_last = false;  // this step reuses kappa / K~ of the previous full-batch step
    bool via_inverse = false;  // this step gets W, v from the available inverse factor instead of a new factorisation
    // hyper-parameter optimiser state (ADAM): kernel parameters on the host, Z on the device
    std::vector<double> k_m, k_v;  // (host staging of the moments for agp_svgp_hyper_opt_state; the live ones are kadam)
    int k_step = 0, z_step = 0;
    double* kadam = nullptr;       // device: ADAM moments of the kernel parameters [m (1 + D) | v (1 + D)]
    bool host_params_stale = false;  // the device parameter array moved (device ADAM): k.variance / k.scales lag behind
    double *z_am = nullptr, *z_av = nullptr;
    // La holds: 0 = -2*eta2 (unfactored), 1 = its Cholesky factor ; xa_valid: Xa = La^-1 is current
    int la_state = 0;
    bool xa_valid = false;
    // v = Xa eta1 formed by materialize() next to mu belongs to the inverse with this number (aug_factor counts them): a step that
    // takes W, v from the inverse uses it instead of launching the triangular mat-vec itself
    int64_t xa_epoch = 0, v_epoch = -1;
    int64_t sigma_epoch = -1;  // xa_epoch of the inverse whose Sigma = Xa' Xa the factorisation launch itself has left in Sigma (ProdArgs)
    bool v_ready = false;
    double half_logdetK = 0.0;
    bool logdet_pending = false;  // half_logdetK still sits in logdetK_dev (asynchronous K refresh)
    // AGP_FLAG_STALE_K: the step-side copies of (inv(K), L^-1, K\mu0, logdet K) frozen at the first hyper step of a train! --
    // what the reference keeps using until train! ends (training.jl:187-208); the members above are always the fresh ones
    T *sKinv = nullptr, *sXk = nullptr, *skinv_mu0 = nullptr;
    double s_half_logdetK = 0.0;
    bool stale_on = false;
    // OnlineSVGP streaming prior (onlinetraining.jl:170-180, latentgp.jl:217-237): the previous posterior enters through
    // Z_a, invD_a = Sigma_a^-1 - K_a^-1, eta1_a ; kappa_a = K_ab K^-1, K~_a = K_a - kappa_a K_ab'
    bool on = false, on_dirty = false, on_first = false;
    int64_t ma = 0, map = 0;
    T *Za = nullptr, *invDa = nullptr, *peta1 = nullptr, *kappa_a = nullptr, *Kab = nullptr, *Kta = nullptr, *onT = nullptr,
      *onQ = nullptr, *kinv_mu0_on = nullptr, *Kinv_on = nullptr, *ov0 = nullptr, *ov1 = nullptr, *oh1 = nullptr,
      *oh2 = nullptr, *oh3 = nullptr;
    double prevLa = 0.0;
  };
  std::vector<Latent> lat;
  int64_t m = 0, mp = 0, D = 0, Bmax = 0, Bp = 0;  // Bp = padded max batch
  int nl = 0;
  double jitter = 1e-4;
  LikParams<T> lp{};
  // shared batch buffers
  T *pw0 = nullptr, *pw1 = nullptr;  // row-statistic scratch [ldp]
  // hyper-parameter step (update_hyperparameters!, autotuning.jl:86-140)
  bool hy_k = false, hy_z = false;
  double hy_keta = 0.01, hy_zeta = 0.001, hy_b1 = 0.9, hy_b2 = 0.999, hy_eps = 1e-8;
  int hy_krule = AGP_OPT_ADAM, hy_zrule = AGP_OPT_ADAM;  // agp_svgp_hyper_rule: ADAM / Descent / Momentum (opt_rule_delta)
  double hy_krho = 0.0, hy_zrho = 0.0;
  T *hyKap = nullptr, *hyKnm = nullptr;  // kappa (Knm) under the fresh inv(K) (kernel) for the hyper-gradient, AGP_FLAG_STALE_K
  T* hy_pZ2 = nullptr;
  double *hy_pvar2 = nullptr, *hy_pscale2 = nullptr;
  T* hy_upart = nullptr;                 // two per tile row: partial column sums of kappa' g_mu (k_gemm_nt<EPI_HK> -> k_hyper_gK_fused)
  T *hyH1 = nullptr, *hyH2 = nullptr, *hyH3 = nullptr, *hy_gmu = nullptr, *hy_gs = nullptr, *hy_muf = nullptr,
    *hy_pZ = nullptr, *hy_dZ = nullptr;
  double *hy_pvar = nullptr, *hy_pscale = nullptr, *hy_g = nullptr;
  // multi-output mode (MOSVGP): n_task likelihoods over A-mixed latents
  bool mo = false;
  MoCfg<T> mocfg{};
  int nT = 0;
  int64_t ystride = 0;
  T* A_dev = nullptr;
  double *gradA_dev = nullptr, *am_dev = nullptr, *av_dev = nullptr;
  int a_step = 0;
  double a_eta = 0, a_b1 = 0.9, a_b2 = 0.999, a_eps = 1e-8;
  T *mo_mixm = nullptr, *mo_mixv = nullptr, *mo_th = nullptr, *mo_cc = nullptr, *mo_th_save = nullptr;
  // latent-sharded multi-output model (one handle per GPU holds the latents [qlo, qlo + nl) of qtot): the mixing needs every
  // latent's (mean_f, var_f) on the batch, exchanged through `fall` = T[2][qtot][Bp] (own rows filled, the rest zero, so the
  // sum over ranks is the all-gather of SURVEY.md section 8e).  fall_state: 0 empty, 1 = pre-update values published by
  // step_local (-> mo_mix), 2 = current-posterior values published by mo_refresh_f (-> elbo / hypergrad), 3 = consumed.
  bool mo_sharded = false;
  int qtot = 0, qlo = 0, fall_state = 0;
  T* fall = nullptr;
  int Qa() const { return mo_sharded ? qtot : nl; }
  // prefetch of the next minibatch's Knm / kappa on a second stream (overlaps the latency-bound factorisation)
  hipStream_t pf_stream = nullptr;
  static constexpr int PF_SIDE = 3;  // further look-ahead streams of a handle with several latents (prefetch())
  hipStream_t pf_side[PF_SIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t pf_fork = nullptr, pf_join[PF_SIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t pf_done = nullptr, step_done[2] = {nullptr, nullptr};
  int step_parity = 0;
  bool pf_valid = false;
  // hand-over word with the look-ahead stream (DagSync): sig[0] = "started" (written by the step's task graph).  The release of
  // a step's kappa buffers -- what the NEXT look-ahead but one waits for -- is either an event recorded on the stream
  // (slot_kind 0) or, when the step after it starts with a task graph that stores its `started` number, that number (slot_kind
  // 1: nothing is enqueued on the stream).  Which one is only known when the next step is enqueued, hence rel_pending.
  // sig[1] = "look-ahead done" (round 3): written by a one-thread kernel behind the look-ahead's GEMM, polled by a one-wave kernel
  // at the head of the step that adopts the buffers -- the cross-stream event wait cost the step's queue ~6 us of idling
  int32_t* sig[2] = {nullptr, nullptr};
  int32_t pf_seq = 0;
  int sig_state = 0;  // 0 not tried, 1 usable, -1 not available / switched off (AGP_PF_INKERNEL=0)
  int32_t started_seq = 0;
  bool rel_pending = false;
  int rel_slot = 0, slot_kind[2] = {0, 0};
  int32_t slot_seq[2] = {0, 0};
  const void* pf_x = nullptr;
  const int64_t* pf_idx = nullptr;
  int64_t pf_B = 0, pf_ldx = 0;
  int64_t ldp = 0;
  T *Kt = nullptr, *muf = nullptr, *varf = nullptr, *cbuf = nullptr, *theta = nullptr, *gamma = nullptr,
    *rbuf = nullptr, *wbuf = nullptr;              // [nl][Bp]
  T *alpha = nullptr, *beta = nullptr, *gsum = nullptr, *alpha_save = nullptr;  // [Bp]
  T *emuf = nullptr, *evarf = nullptr;             // ELBO-time mean_f / var_f [nl][Bp]
  T* stats = nullptr;                              // [nl][mp + nt(nt+1)/2 * 64*64]: kappa' r, then the lower tiles of kappa' diag(w) kappa
  int64_t stats_stride() const { return mp + (mp / TILE) * (mp / TILE + 1) / 2 * TILE * TILE; }
  T* Tw = nullptr;                                 // mp x mp scratch
  T* Tw2 = nullptr;                                // mp x mp scratch (predict)
  int tw2_kis_of = -1;                             // latent whose K^-1 Sigma the scratch holds, -1: something else
  T* tmpv = nullptr;                               // mp
  T* lr_dev = nullptr;
  int32_t* info_dev = nullptr;   // failure latch of the factorisations of -2*eta2 (asynchronous steps)
  int32_t* infoK_dev = nullptr;  // ... and of K_ZZ (read where it is produced: refresh_K synchronises)
  int* flags_dev = nullptr;
  double* scal_dev = nullptr;                      // 16 doubles
  // prediction workspace
  T *Kstar = nullptr, *ppm = nullptr, *ppv = nullptr, *pmu = nullptr, *pvar = nullptr;
  int64_t pred_chunk = 0, pred_nt_cap = 0;
  double* gh_dev = nullptr;
  int gh_cap = 0, gh_n = 0;  // Gauss-Hermite nodes | weights (agp_svgp_set_quadrature / proba_y)
  T* lam_dev = nullptr;       // Poisson / heteroscedastic lambda (state re-estimated by every local update); Gaussian noise sigma2
                              // when it is optimised (LikParams::noise_dev)
  double* noise_adam = nullptr;
  double noise_eta = 0.0;
  double* lam_part = nullptr; // per-workgroup partial sums of the lambda update
  double* frob_part = nullptr;  // row partial sums of frob_dot
  int64_t frob_cap = 0;
  agp_status frob_dot(const T* A, const T* Bm, int64_t ld, int64_t n, double* out) {
    if (n > frob_cap) {
      if (frob_part) (void)hipFree(frob_part);
      frob_part = nullptr;
      frob_cap = 0;
      AGPCHK(dmalloc(ctx, &frob_part, n));
      frob_cap = n;
    }
    hipLaunchKernelGGL((k_frob_dot_rows<T>), dim3((unsigned)n), dim3(256), 0, st(), A, Bm, ld, n, frob_part);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1024), 0, st(), (const double*)frob_part, n, out);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  // The natural-gradient step of the last CAVI step, not taken yet: the next step's task-graph launch takes it as its prologue
  // (ProArgs, agp_chol.h) -- the 53 us symmetric product then no longer sits between two factorisation chains.  Everything else
  // that touches the posterior calls flush() first.  kap is the physical kappa buffer of that minibatch (it survives the rotation of
  // the look-ahead buffers: kappa has three of them).
  struct Pending {
    bool on = false;
    int64_t Bq = 0;
    T lr = T(0);
    const T* kap = nullptr;
    const T *Kinv = nullptr, *kinv_mu0 = nullptr;
    const T *r = nullptr, *w = nullptr;  // rho grad_E_mu / rho grad_E_Sigma of that minibatch (the r / w pair current at the time)
  } pend;
  // ... and the batch-parallel form of it (agp_svgp_cavi_step_multi, AGP_SHARD_BATCH over several ranks): the statistics have been
  // all-reduced into `stats` = [t | packed lower tiles]; the eta step from them rides on the next launch as well
  struct PendingPacked {
    bool on = false;
    T lr = T(0);
    const T *Kinv = nullptr, *kinv_mu0 = nullptr;
    // AGP_SPLIT_OVERLAP: the statistics are still arriving, group by group, on the communicator's stream (arrive words at `epoch`)
    bool overlap = false;
    int32_t epoch = 0;
    agp_comm* cm = nullptr;
  } pendp;
  // arrival words of the block-column groups (device memory, ARRIVE_STRIDE apart), the step counter they carry, the groups
  int32_t* arrive_dev = nullptr;
  int32_t arrive_epoch = 0;
  int ov_ng = 0;
  int64_t ov_nt = 0;
  int64_t ov_off[8] = {}, ov_cnt[8] = {};
  unsigned char ov_grp[32] = {};
  // Second (r, w) pair: a task-graph launch with the row-statistics EPILOGUE (EpiArgs) writes this step's r, w while late workgroups
  // of its own prologue may still read the previous step's -- the two pairs alternate.
  T *rbuf2 = nullptr, *wbuf2 = nullptr;
  // The launch behind such a task graph -- its in-stream fallback (k_chol_safe), which also redoes the rows if it has to run -- is
  // DEFERRED to the head of the next step, where it also carries that step's wait for its look-ahead: one tiny kernel between two
  // task graphs instead of two (k_safe_rowstats + k_wait_ge_fast; every kernel costs the in-order queue ~5 us).  Anything else that
  // needs the step's results runs it first (flush()).
  struct SafeDeferred {
    bool on = false;
    CholBatch<T> bt{};
    SafeSrc<T> src{};
    RowstatsBatch<T> rb{};
    int64_t ne = 0, nt = 0, B = 0;
    int ns = 0;
    T rho = T(0);
    const T* y = nullptr;
    const int64_t* idx = nullptr;
    T *r = nullptr, *w = nullptr;
    unsigned grid = 1;
  } sdef;
  agp_status run_deferred_safe(const int32_t* pf_word = nullptr, int32_t pf_want = 0) {
    if (!sdef.on) return AGP_OK;
    sdef.on = false;
    hipLaunchKernelGGL((k_safe_rowstats<T>), dim3(sdef.grid), dim3(CHOL_THREADS), 0, st(), sdef.bt, sdef.src, mp, mp, mp, sdef.ne,
                       sdef.nt, info_dev, m, ctx->safe_bar, ctx->safe_retries, sdef.B, sdef.ns, sdef.rb, ldp, mp, mp, (T)jitter,
                       sdef.rho, lp, sdef.y, sdef.idx, Kt, muf, varf, cbuf, theta, sdef.r, sdef.w, flags_dev, (const T*)lam_dev,
                       gamma, 1, pf_word, pf_want);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  bool epi_allowed() const {
    static const bool off = []() {
      const char* e = getenv("AGP_STEP_EPILOGUE");
      return e && e[0] == '0';
    }();
    if (off) return false;
    const int k = lp.kind;  // likelihoods whose local update is complete in rowstats_finish (no further kernel reads mean_f / var_f)
    return (k == AGP_LIK_GAUSSIAN && !lp.noise_dev) || k == AGP_LIK_LOGISTIC || k == AGP_LIK_STUDENTT || k == AGP_LIK_LAPLACE ||
           k == AGP_LIK_BAYESIANSVM || k == AGP_LIK_NEGBINOMIAL;
  }
  bool pro_allowed() const {
    static const bool off = []() {
      const char* e = getenv("AGP_STEP_PROLOGUE");
      return e && e[0] == '0';
    }();
    if (off || nl != 1 || mo || mo_sharded) return false;
    // only where the launch is not bound by workgroup slots: at 32 block columns (C3) the task graph already queues 1584 tile
    // workgroups through 256 slots, and the product's 27 ms of CU time inside it costs more than the kernel of its own (measured:
    // 0.70 -> 0.86-0.96 ms per step with any k-split table)
    if (mp / TILE > 16) return false;
    const int k = lp.kind;
    return k == AGP_LIK_GAUSSIAN || k == AGP_LIK_LOGISTIC || k == AGP_LIK_STUDENTT || k == AGP_LIK_LAPLACE ||
           k == AGP_LIK_BAYESIANSVM || k == AGP_LIK_NEGBINOMIAL || k == AGP_LIK_POISSON;
  }
  agp_status flush() override {
    AGPCHK(run_deferred_safe());
    if (!pend.on && !pendp.on) return AGP_OK;
    Latent& g = lat[0];
    g.C_valid = false;
    if (pend.on) {
      pend.on = false;
      AGPCHK((syrk_tn<T, SY_ETA2>(ctx, pend.kap, mp, mp, pend.Bq, pend.w, 0, g.La, mp, g.eta2, pend.Kinv, mp, pend.lr, pend.r,
                                  g.eta1, pend.kinv_mu0)));
    } else {
      pendp.on = false;
      if (pendp.overlap) {  // the stand-alone step reads the whole statistic: join the collective's stream
        pendp.overlap = false;
        HIPCHK(ctx, hipStreamWaitEvent(st(), pendp.cm->ev_out, 0));
      }
      const int64_t ntri = (mp / TILE) * (mp / TILE + 1) / 2;
      hipLaunchKernelGGL((k_eta2_from_packed<T>), dim3((unsigned)(4 * ntri + (mp + 255) / 256)), dim3(256), 0, st(), stats + mp, mp,
                         g.eta2, pendp.Kinv, g.La, pendp.lr, ntri, (const T*)stats, pendp.kinv_mu0, g.eta1, mp);
    }
    LAUNCHCHK(ctx);
    g.la_state = 0;  // the epilogue left La = -2 eta2
    g.xa_valid = false;
    return AGP_OK;
  }
  agp_status step_finish() override {
    if (pro_allowed() && chol_use_dag(ctx, mp / TILE, rup64(B_last) / TILE + 1, 1) && mp / TILE <= 32) {
      Latent& g = lat[0];
      g.C_valid = false;
      pend.on = true;
      pend.Bq = rup64(B_last);
      pend.lr = (T)cur_lr();
      pend.kap = g.kappa;
      pend.Kinv = kinv_step(g);
      pend.kinv_mu0 = kinv_mu0_step(g);
      pend.r = rbuf;
      pend.w = wbuf;
      // the bookkeeping of step_global: the posterior changes (as soon as the step is taken)
      g.la_state = 1;  // La does not hold -2 eta2: whoever wants it rebuilds it from eta2 (after flush())
      g.xa_valid = false;
      g.post_valid = false;
      g.pred_valid = g.predvar_valid = false;
      n_opt += 1;
      return kappa_released();
    }
    AGPCHK(run_deferred_safe());
    AGPCHK(step_stats(true));
    return step_global(true);
  }
  // last step
  const void* x_last = nullptr;
  // identity of the batch of the last fresh evaluation (ELBO(model, X, y), ELBO.jl:32-47) whose kappa is still in the buffers: a
  // handle that only evaluates -- the shadow handle of a SideObjective, same points every check, kernels and Z fixed -- recomputes
  // neither K_nm nor kappa = K_nm K^-1 (17 GF at 8192 x 1024) per check.  Same contract as the full-batch kappa cache: pointer
  // identity; a host that rewrites X / idx in place says so (agp_svgp_invalidate_data).  Any training step, K refresh, set_Z /
  // set_kernel clears it.
  const void* ev_x = nullptr;
  const int64_t* ev_idx = nullptr;
  int64_t ev_B = 0, ev_ldx = 0;
  bool eval_cache_ok = false;
  const void* y_last = nullptr;
  const int64_t* idx_last = nullptr;
  int64_t B_last = 0, ldx_last = 0;
  double rho_last = 1.0;

  hipStream_t st() { return ctx->stream; }
  // (round 4 measured the hyper-gradient's small launches on a side stream next to the products: 1063 -> 1077 us, every cross-stream
  //  event costs the waiting stream 7 - 12 us; removed in round 5, docs/DESIGN_LOG.md)

  agp_status init() override {
    m = desc.m;
    D = desc.D;
    nl = desc.n_latent;
    Bmax = desc.max_batch;
    if (m <= 0 || D <= 0 || nl <= 0 || Bmax <= 0) {
      ctx->err = "agp_svgp_create: m, D, n_latent, max_batch must be positive";
      return AGP_ERR_INVALID;
    }
    mp = rup64(m);
    Bp = rup64(Bmax);
    jitter = desc.jitter > 0 ? desc.jitter : (sizeof(T) == 8 ? 1e-4 : 1e-3);
    lp.kind = desc.lik.kind;
    lp.p0 = (T)desc.lik.p0;
    lp.p1 = (T)desc.lik.p1;
    if (lp.kind < 0 || lp.kind > AGP_LIK_HETEROSCEDASTIC) {
      ctx->err = "likelihood not implemented for AnalyticVI on this path";
      return AGP_ERR_UNSUPPORTED;
    }
    if (lp.kind == AGP_LIK_GAUSSIAN && !(desc.lik.p0 > 0)) return AGP_ERR_INVALID;
    if (lp.kind == AGP_LIK_GAUSSIAN && desc.lik.p1 > 0) {  // GaussianLikelihood(sigma2; opt_noise = ADAM(p1)), gaussian.jl:18-23
      lp.noise_dev = 1;
      noise_eta = desc.lik.p1;
    }
    if (lp.kind == AGP_LIK_STUDENTT && !(desc.lik.p0 > 0.5)) {
      ctx->err = "nu should be greater than 0.5";  // studentt.jl:28
      return AGP_ERR_INVALID;
    }
    if ((lp.kind == AGP_LIK_LAPLACE || lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_NEGBINOMIAL ||
         lp.kind == AGP_LIK_HETEROSCEDASTIC) && !(desc.lik.p0 > 0)) {
      ctx->err = "likelihood parameter (beta / lambda / r) must be positive";
      return AGP_ERR_INVALID;
    }
    if (lp.kind == AGP_LIK_HETEROSCEDASTIC) {  // n_latent(::HeteroscedasticGaussianLikelihood) = 2  heteroscedastic.jl:47
      if (nl != 2 || desc.latent_offset != 0) return AGP_ERR_INVALID;
    } else if (lp.kind != AGP_LIK_LOGISTICSOFTMAX && lp.kind != AGP_LIK_MULTIOUTPUT && nl != 1) {
      return AGP_ERR_INVALID;
    }
    if (desc.stochastic && !(desc.rm_kappa > 0.5 && desc.rm_kappa <= 1.0 && desc.rm_tau > 0)) {
      ctx->err = "RobbinsMonro: kappa in (0.5,1], tau > 0";  // optimisers.jl:7-8
      return AGP_ERR_INVALID;
    }
    lat.resize(nl);
    const int64_t mm = mp * mp;
    for (auto& g : lat) {
      g.k.scales.assign(D, 1.0);
      AGPCHK(dmalloc(ctx, &g.scales, D + 1));  // [scale_0 .. scale_{D-1} | variance]: the kernel's device parameter array
      AGPCHK(dmalloc(ctx, &g.Z, m * D));
      AGPCHK(dmalloc(ctx, &g.L, mm));
      AGPCHK(dmalloc(ctx, &g.Xk, mm));
      AGPCHK(dmalloc(ctx, &g.Kinv, mm));
      AGPCHK(dmalloc(ctx, &g.eta1, mp));
      AGPCHK(dmalloc(ctx, &g.eta2, mm));
      AGPCHK(dmalloc(ctx, &g.La, mm));
      AGPCHK(dmalloc(ctx, &g.Xa, mm));
      AGPCHK(dmalloc(ctx, &g.v, mp));
      AGPCHK(dmalloc(ctx, &g.Sigma, mm));
      AGPCHK(dmalloc(ctx, &g.mu, mp));
      AGPCHK(dmalloc(ctx, &g.Knm, Bp * mp));
      AGPCHK(dmalloc(ctx, &g.kappa, Bp * mp));
      AGPCHK(dmalloc(ctx, &g.Wbuf, (Bp + TILE) * mp));
      AGPCHK(dmalloc(ctx, &g.pk, (2 * mp / TILE) * Bp));
      AGPCHK(dmalloc(ctx, &g.DgK, mp * TILE));
      AGPCHK(dmalloc(ctx, &g.DgA, mp * TILE));
      AGPCHK(upload_scales(g));
      AGPCHK(reset_posterior(g));
    }
    ldp = Bp;
    const int ns = (int)(2 * mp / TILE);
    (void)ns;
    AGPCHK(dmalloc(ctx, &pw0, 2 * ldp > mp ? 2 * ldp : mp));
    AGPCHK(dmalloc(ctx, &pw1, 2 * ldp > mp ? 2 * ldp : mp));
    T** bv[] = {&Kt, &muf, &varf, &cbuf, &theta, &gamma, &rbuf, &wbuf, &emuf, &evarf};
    for (auto p : bv) {
      AGPCHK(dmalloc(ctx, p, nl * Bp));
      HIPCHK(ctx, hipMemsetAsync(*p, 0, sizeof(T) * nl * Bp, st()));
    }
    AGPCHK(dmalloc(ctx, &rbuf2, nl * Bp));
    AGPCHK(dmalloc(ctx, &wbuf2, nl * Bp));
    HIPCHK(ctx, hipMemsetAsync(rbuf2, 0, sizeof(T) * nl * Bp, st()));
    HIPCHK(ctx, hipMemsetAsync(wbuf2, 0, sizeof(T) * nl * Bp, st()));
    AGPCHK(dmalloc(ctx, &alpha, Bp));
    AGPCHK(dmalloc(ctx, &beta, Bp));
    AGPCHK(dmalloc(ctx, &gsum, Bp));
    AGPCHK(dmalloc(ctx, &alpha_save, Bp));
    AGPCHK(dmalloc(ctx, &stats, nl * stats_stride()));
    AGPCHK(dmalloc(ctx, &Tw, mm));
    AGPCHK(dmalloc(ctx, &Tw2, mm));
    AGPCHK(dmalloc(ctx, &tmpv, mp));
    AGPCHK(dmalloc(ctx, &lr_dev, 1));
    AGPCHK(dmalloc(ctx, &info_dev, 4));  // [info | infoK | flags | -]: one block, so that check_status fetches them with one copy
    infoK_dev = info_dev + 1;
    flags_dev = (int*)(info_dev + 2);
    AGPCHK(dmalloc(ctx, &scal_dev, 64));
    AGPCHK(dmalloc(ctx, &lam_dev, 1));
    AGPCHK(dmalloc(ctx, &lam_part, 2 * (Bp / 256 + 1)));
    AGPCHK(dmalloc(ctx, &noise_adam, 6));  // ADAM state [m, v, t] of the noise optimiser, + a scratch state for fresh evaluations
    HIPCHK(ctx, hipMemsetAsync(noise_adam, 0, sizeof(double) * 6, st()));
    hipLaunchKernelGGL((k_fill<T>), dim3(1), dim3(64), 0, st(), lam_dev, (int64_t)1, (T)(desc.lik.p0 > 0 ? desc.lik.p0 : 1.0));
    HIPCHK(ctx, hipMemsetAsync(info_dev, 0, sizeof(int32_t), st()));
    HIPCHK(ctx, hipMemsetAsync(infoK_dev, 0, sizeof(int32_t), st()));
    HIPCHK(ctx, hipMemsetAsync(flags_dev, 0, sizeof(int), st()));
    HIPCHK(ctx, hipMemsetAsync(info_dev + 3, 0, sizeof(int32_t), st()));  // orderK (k_logdiag_sum)
    // LogisticSoftMax state: alpha = beta = K (total classes)  logisticsoftmax.jl:43-53
    const T kk = (T)(lp.kind == AGP_LIK_LOGISTICSOFTMAX ? desc.lik.n_class : 1);
    hipLaunchKernelGGL((k_fill<T>), grid1(Bp), dim3(256), 0, st(), alpha, Bp, kk);
    hipLaunchKernelGGL((k_fill<T>), grid1(Bp), dim3(256), 0, st(), beta, Bp, kk);
    LAUNCHCHK(ctx);
    n_opt = 1;
    return AGP_OK;
  }

  ~Svgp() override {
    for (auto e : ev) dcheck(hipEventDestroy(e), __LINE__);
    for (auto& g : lat) {
      T* ps[] = {g.Zsc, g.zn, g.scales, g.Z, g.L, g.Xk, g.Kinv, g.mu0, g.kinv_mu0, g.eta1, g.eta2, g.La, g.Xa, g.v,
                 g.Sigma, g.mu, g.Knm, g.kappa, g.Apred, g.apred, g.Wbuf, g.DgK, g.DgA, g.pk, g.Knm_alt, g.kappa_alt, g.Wbuf_alt, g.pk_alt, g.kappa_old};
      for (T* p : ps)
        if (p) dfree(p);
    }
    if (pf_stream && sig[0]) {
      // a look-ahead still waiting for a step that was never launched must not outlive the handle
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipMemcpy(sig[0], &started_seq, sizeof(int32_t), hipMemcpyHostToDevice);
      (void)hipStreamSynchronize(pf_stream);
    }
    for (int q = 0; q < PF_SIDE; ++q) {
      if (pf_side[q]) {
        (void)hipStreamSynchronize(pf_side[q]);
        dcheck(hipStreamDestroy(pf_side[q]), __LINE__);
      }
      if (pf_join[q]) dcheck(hipEventDestroy(pf_join[q]), __LINE__);
    }
    if (pf_fork) dcheck(hipEventDestroy(pf_fork), __LINE__);
    if (elbo_pin) {
      dcheck(hipHostFree(elbo_pin), __LINE__);
      for (auto& e : elbo_ev)
        if (e) dcheck(hipEventDestroy(e), __LINE__);
    }
    if (pf_stream) dcheck(hipStreamDestroy(pf_stream), __LINE__);
    for (auto q : sig)
      if (q) dcheck(hipFree(q), __LINE__);
    if (arrive_dev) dcheck(hipFree(arrive_dev), __LINE__);
    if (pf_done) dcheck(hipEventDestroy(pf_done), __LINE__);
    for (auto e : step_done)
      if (e) dcheck(hipEventDestroy(e), __LINE__);
    T* ps[] = {rbuf2, wbuf2, pw0, pw1, Kt, muf, varf, cbuf, theta, gamma, rbuf, wbuf, alpha, beta, gsum, alpha_save, emuf,
               evarf, stats, Tw, Tw2, tmpv, lr_dev, Kstar, ppm, ppv, pmu, pvar};
    for (T* p : ps)
      if (p) dfree(p);
    T* hps[] = {hyH1, hyH2, hyH3, hy_gmu, hy_gs, hy_muf, hy_pZ, hy_dZ, hyKap, hyKnm, hy_upart, hy_pZ2};
    for (T* p : hps)
      if (p) dfree(p);
    double* hds[] = {hy_pvar, hy_pscale, hy_g, hy_tied, hy_pvar2, hy_pscale2};
    for (double* p : hds)
      if (p) dfree(p);
    for (auto& g : lat) {
      free_online(g);
      T* sp[] = {g.sKinv, g.sXk, g.skinv_mu0, g.Cmat};
      for (T* q : sp)
        if (q) dfree(q);
      if (g.kadam) dfree(g.kadam);
      if (g.z_am) dfree(g.z_am);
      if (g.z_av) dfree(g.z_av);
    }
    T* mops[] = {A_dev, mo_mixm, mo_mixv, mo_th, mo_cc, mo_th_save, mo_pmu, mo_pvar, fall};
    for (T* p : mops)
      if (p) dfree(p);
    double* dps[] = {gradA_dev, am_dev, av_dev};
    for (double* p : dps)
      if (p) dfree(p);
    if (info_dev) dfree(info_dev);  // (infoK_dev, flags_dev point into the same block)
    if (scal_dev) dfree(scal_dev);
    if (gh_dev) dfree(gh_dev);
    if (lam_dev) dfree(lam_dev);
    if (lam_part) dfree(lam_part);
    if (noise_adam) dfree(noise_adam);
    if (frob_part) dfree(frob_part);
    if (logdetK_dev) dfree(logdetK_dev);
  }

  // host copy of the kernel parameters -> device array [scales | variance] (set_kernel; the training loop never comes here: the
  // hyper step's ADAM runs on the device, k_adam_kernel_params)
  agp_status upload_scales(Latent& g) {
    std::vector<T> h(D + 1);
    for (int64_t d = 0; d < D; ++d) h[d] = (T)g.k.scales[d];
    h[D] = (T)g.k.variance;
    HIPCHK(ctx, hipMemcpyAsync(g.scales, h.data(), sizeof(T) * (D + 1), hipMemcpyHostToDevice, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));  // h goes out of scope
    g.host_params_stale = false;
    return AGP_OK;
  }
  // device -> host copy, only where host logic needs the values (get_kernel, prediction set-up, the online model's hand-over):
  // synchronises
  agp_status params_to_host(Latent& g) {
    if (!g.host_params_stale) return AGP_OK;
    std::vector<T> h(D + 1);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), g.scales, sizeof(T) * (D + 1), hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    for (int64_t d = 0; d < D; ++d) g.k.scales[d] = (double)h[d];
    g.k.variance = (double)h[D];
    g.host_params_stale = false;
    return AGP_OK;
  }
  // the variance argument of the kernels that take (scales, variance): the host value, or -1 = "read element D of the device
  // parameter array" while the host copy is stale (between a device-side ADAM step and the next params_to_host)
  T kvar(const Latent& g) const { return g.host_params_stale ? T(-1) : (T)g.k.variance; }

  // VarPosterior{T}(dim): mu = 0, Sigma = I, eta1 = 0, eta2 = -I/2   (posterior.jl:29-37)
  agp_status reset_posterior(Latent& g) {
    HIPCHK(ctx, hipMemsetAsync(g.eta1, 0, sizeof(T) * mp, st()));
    HIPCHK(ctx, hipMemsetAsync(g.v, 0, sizeof(T) * mp, st()));
    hipLaunchKernelGGL((k_set_identity<T>), grid2(mp, mp), blk2, 0, st(), g.eta2, mp, mp, T(-0.5));
    hipLaunchKernelGGL((k_set_identity<T>), grid2(mp, mp), blk2, 0, st(), g.La, mp, mp, T(1));
    hipLaunchKernelGGL((k_set_identity<T>), grid2(mp, mp), blk2, 0, st(), g.Xa, mp, mp, T(1));
    LAUNCHCHK(ctx);
    g.post_valid = false;
    g.pred_valid = g.predvar_valid = false;
    g.la_state = 0;
    g.xa_valid = false;
    return AGP_OK;
  }

  agp_status set_kernel(int l, const agp_kernel_desc* k) override {
    if (l < 0 || l >= nl || !k) return AGP_ERR_INVALID;
    if (k->kind < 0 || k->kind > AGP_K_EXPONENTIAL || !(k->variance > 0)) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    g.k.kind = k->kind;
    g.k.variance = k->variance;
    g.k.ard = k->ard != 0;
    g.k.has_variance = k->has_variance != 0;
    g.k.has_transform = k->has_transform != 0 || k->ard != 0;
    for (int64_t d = 0; d < D; ++d) g.k.scales[d] = k->ard ? k->ard_scales_host[d] : k->scale;
    g.K_stale = true;
    g.kappa_valid = g.kappa_eval = false;
    g.zsc_valid = false;
    pf_valid = false;
    g.pred_valid = g.predvar_valid = false;
    return upload_scales(g);
  }

  agp_status set_Z(int l, const void* z, int64_t ldz) override {
    if (l < 0 || l >= nl || !z || ldz < D) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    HIPCHK(ctx, hipMemcpy2DAsync(g.Z, sizeof(T) * D, z, sizeof(T) * ldz, sizeof(T) * D, m, hipMemcpyDeviceToDevice,
                                 st()));
    g.K_stale = true;
    g.kappa_valid = g.kappa_eval = false;
    g.zsc_valid = false;
    pf_valid = false;
    g.pred_valid = g.predvar_valid = false;
    return AGP_OK;
  }

  agp_status get_Z(int l, void* z, int64_t ldz) override {
    if (l < 0 || l >= nl || !z || ldz < D) return AGP_ERR_INVALID;
    HIPCHK(ctx, hipMemcpy2DAsync(z, sizeof(T) * ldz, lat[l].Z, sizeof(T) * D, sizeof(T) * D, m,
                                 hipMemcpyDeviceToDevice, st()));
    return AGP_OK;
  }

  agp_status set_mu0(int l, const void* mu0) override {
    if (l < 0 || l >= nl) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    if (!mu0) {
      if (g.mu0) (void)hipFree(g.mu0);
      if (g.kinv_mu0) (void)hipFree(g.kinv_mu0);
      g.mu0 = g.kinv_mu0 = nullptr;
      return AGP_OK;
    }
    if (!g.mu0) {
      AGPCHK(dmalloc(ctx, &g.mu0, mp));
      AGPCHK(dmalloc(ctx, &g.kinv_mu0, mp));
    }
    HIPCHK(ctx, hipMemsetAsync(g.mu0, 0, sizeof(T) * mp, st()));
    HIPCHK(ctx, hipMemcpyAsync(g.mu0, mu0, sizeof(T) * m, hipMemcpyDeviceToDevice, st()));
    g.K_stale = true;  // K^-1 mu0 must be refreshed
    return AGP_OK;
  }

  // compute_K : cholesky(kernelmatrix(k, Z) + jitt*I) ; inv(K)      latentgp.jl:205-207, analyticVI.jl:179
  agp_status refresh_K() override {
    bool any = false;
    if (pend.on)  // the pending natural-gradient step belongs to the kernel matrices it was computed with
      for (auto& g : lat)
        if (g.K_stale) {
          AGPCHK(flush());
          break;
        }
    for (auto& g : lat) {
      if (!g.K_stale) continue;
      any = true;
      // the factorisations of K_ZZ latch their failures in a word of their own (infoK_dev): what an earlier asynchronous step
      // latched (K~ <= 0, a non-SPD -2*eta2) stays in info_dev / flags_dev and is reported as what it is by check_status
      {
        AGPCHK(ensure_zsc(g));
        (void)launch_kernelmatrix<T>(ctx, st(), (const T*)g.Z, D, (const int64_t*)nullptr, m, (const T*)g.Z, D, m, D,
                                     (const T*)g.scales, g.k.kind, kvar(g), g.L, mp, mp, mp, 1, (T)jitter,
                                     (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)g.Zsc, (const T*)g.zn);
        LAUNCHCHK(ctx);
        // the task graph's in-stream fallback recomputes K from the inducing points (the factor overwrote it) and forms L^-1 itself
        SafeSrc<T> src{};
        src.want_x = 1;
        src.kz = g.Z;
        src.ldz = D;
        src.mz = m;
        src.Dz = D;
        src.kscales = g.scales;
        src.kkind = g.k.kind;
        src.kvariance = kvar(g);
        src.kjitter = (T)jitter;
        if (!logdetK_dev) AGPCHK(dmalloc(ctx, &logdetK_dev, nl));
        const int li = (int)(&g - lat.data());
        bool kinv_done = false;  // K^-1 = X' X (and log det K) by product workgroups of the same launch (task graph), else below
        AGPCHK(potrf_fused<T>(ctx, g.L, mp, mp, g.Xk, mp, g.DgK, (T*)nullptr, 0, 0, 1, infoK_dev, m, (const T*)nullptr, true, &src,
                              nullptr, nullptr, nullptr, nullptr, g.Kinv, &kinv_done, logdetK_dev + li, info_dev));
        bool ld_done = kinv_done;
        if (!kinv_done)
          AGPCHK(xtx_padded<T>(ctx, g.Xk, mp, mp, g.Kinv, mp, (const T*)g.DgK, m, logdetK_dev + li, info_dev, &ld_done));
        if (!ld_done)
          hipLaunchKernelGGL((k_logdiag_sum<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.DgK, m, logdetK_dev + li, info_dev);
        LAUNCHCHK(ctx);
        g.logdet_pending = true;
        if (!refresh_lazy) {
          // log det K and the status, read now: one synchronisation per refreshed latent.  (refresh_lazy: a refresh issued from
          // inside the training loop -- the hyper step moved the kernel -- leaves both on the device: log det K is fetched when an
          // ELBO asks for it, a non-SPD K_ZZ stays latched in infoK_dev and is reported by agp_svgp_check_status)
          int32_t info = 0;
          HIPCHK(ctx, hipMemcpyAsync(&info, infoK_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st()));
          AGPCHK(resolve_logdet(g));  // synchronises
          if (info != 0) {
            HIPCHK(ctx, hipMemsetAsync(infoK_dev, 0, sizeof(int32_t), st()));
            for (auto& q : lat) q.K_stale = true;
            // a failure an earlier asynchronous step latched (K~ <= 0 makes everything after it NaN, kernel parameters included)
            // is the root cause and is what the reference would have thrown first
            AGPCHK(check_status());
            ctx->err = info < 0 ? std::string("task-graph factorisation of K_ZZ aborted: a tile dependency never arrived")
                                : "PosDefException: K_ZZ + jitter*I is not positive definite; leading minor " + std::to_string(info);
            return info < 0 ? AGP_ERR_HIP : AGP_ERR_NOT_POSDEF;
          }
        }
      }
      if (g.mu0) {
        hipLaunchKernelGGL((k_symv<T>), grid1(mp * 64), dim3(256), 0, st(), (const T*)g.Kinv, mp, mp, (const T*)g.mu0,
                           g.kinv_mu0);
        LAUNCHCHK(ctx);
      }
      g.K_stale = false;
      g.C_valid = false;  // (C = kappa' diag(w) kappa + K^-1 / 4 of a step under the kernel matrices before this refresh)
      if (tw2_kis_of == (int)(&g - lat.data())) tw2_kis_of = -1;
      // (under AGP_FLAG_STALE_K a full-batch run keeps the step's kernel matrices across the refresh of the fresh set)
      g.kappa_eval = false;
      if (!(g.stale_on && !desc.stochastic)) g.kappa_valid = false;
      g.pred_valid = g.predvar_valid = false;
    }
    for (auto& g : lat)
      if (g.on && (g.on_dirty || any)) AGPCHK(online_refresh(g));
    return AGP_OK;
  }

  // ---- OnlineSVGP streaming prior ------------------------------------------------------------------------------------
  void free_online(Latent& g) {
    T* ps[] = {g.Za, g.invDa, g.peta1, g.kappa_a, g.Kab, g.Kta, g.onT, g.onQ, g.kinv_mu0_on, g.Kinv_on, g.ov0, g.ov1,
               g.oh1, g.oh2, g.oh3};
    for (T* p : ps)
      if (p) dfree(p);
    g.Za = g.invDa = g.peta1 = g.kappa_a = g.Kab = g.Kta = g.onT = g.onQ = g.kinv_mu0_on = g.Kinv_on = g.ov0 = g.ov1 = nullptr;
    g.oh1 = g.oh2 = g.oh3 = nullptr;
    g.on = false;
  }

  agp_status set_online_prior(int l, const void* za, int64_t ldza, int64_t ma, const void* invDa, int64_t ldi,
                              const void* peta1, double prevLa) override {
    if (l < 0 || l >= nl || ma <= 0 || !invDa || !peta1 || ldi < ma || (za && ldza < D)) return AGP_ERR_INVALID;
    if (desc.stochastic) {  // set_rho!(model, ...) in the stochastic branch of onlinetraining.jl:52 references an undefined name
      ctx->err = "OnlineSVGP runs with AnalyticVI() (full batches); the reference's stochastic branch is broken";
      return AGP_ERR_UNSUPPORTED;
    }
    if (!za && ma != m) return AGP_ERR_INVALID;  // first batch: kappa_a = I needs the sizes to agree
    Latent& g = lat[l];
    free_online(g);
    g.ma = ma;
    g.map = rup64(ma);
    const int64_t map = g.map;
    AGPCHK(dmalloc(ctx, &g.invDa, map * map));
    AGPCHK(dmalloc(ctx, &g.peta1, map));
    AGPCHK(dmalloc(ctx, &g.kappa_a, map * mp));
    AGPCHK(dmalloc(ctx, &g.Kab, map * mp));
    AGPCHK(dmalloc(ctx, &g.Kta, map * map));
    AGPCHK(dmalloc(ctx, &g.onT, map * mp));
    AGPCHK(dmalloc(ctx, &g.onQ, std::max(map * map, mp * mp)));
    AGPCHK(dmalloc(ctx, &g.kinv_mu0_on, mp));
    AGPCHK(dmalloc(ctx, &g.Kinv_on, mp * mp));
    AGPCHK(dmalloc(ctx, &g.ov0, map));
    AGPCHK(dmalloc(ctx, &g.ov1, map));
    AGPCHK(dmalloc(ctx, &g.oh1, map * mp));
    AGPCHK(dmalloc(ctx, &g.oh2, map * mp));
    AGPCHK(dmalloc(ctx, &g.oh3, map * mp));
    hipLaunchKernelGGL((k_copy2d_zero<T>), grid2(map, map), blk2, 0, st(), (const T*)invDa, ldi, ma, ma, g.invDa, map, map, map);
    HIPCHK(ctx, hipMemsetAsync(g.peta1, 0, sizeof(T) * map, st()));
    HIPCHK(ctx, hipMemcpyAsync(g.peta1, peta1, sizeof(T) * ma, hipMemcpyDeviceToDevice, st()));
    if (za) {
      AGPCHK(dmalloc(ctx, &g.Za, ma * D));
      HIPCHK(ctx, hipMemcpy2DAsync(g.Za, sizeof(T) * D, za, sizeof(T) * ldza, sizeof(T) * D, ma, hipMemcpyDeviceToDevice, st()));
    }
    LAUNCHCHK(ctx);
    g.on_first = za == nullptr;
    g.prevLa = prevLa;
    g.on = true;
    g.on_dirty = true;
    return AGP_OK;
  }

  agp_status online_refresh(Latent& g) {
    const int64_t map = g.map, ma = g.ma;
    if (g.on_first) {  // compute_kappa with empty Z_a (latentgp.jl:220-223): K_ab = 0, kappa_a = I, K~_a = 0
      HIPCHK(ctx, hipMemsetAsync(g.Kab, 0, sizeof(T) * map * mp, st()));
      HIPCHK(ctx, hipMemsetAsync(g.kappa_a, 0, sizeof(T) * map * mp, st()));
      HIPCHK(ctx, hipMemsetAsync(g.Kta, 0, sizeof(T) * map * map, st()));
      hipLaunchKernelGGL((k_add_diag<T>), grid1(ma), dim3(256), 0, st(), g.kappa_a, mp, ma, T(1));
    } else {
      dim3 gk((unsigned)(mp / TILE), (unsigned)(map / TILE));
      (void)launch_kernelmatrix<T>(ctx, st(), (const T*)g.Za, D, (const int64_t*)nullptr, ma,
                         (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), g.Kab, mp, map, mp, 0, T(0),
                         (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)g.Zsc, (const T*)g.zn);
      LAUNCHCHK(ctx);
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.Kab, mp, g.Kinv, mp, map, mp, mp, 0, g.kappa_a, mp, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));
      dim3 ga((unsigned)(map / TILE), (unsigned)(map / TILE));
      (void)launch_kernelmatrix<T>(ctx, st(), (const T*)g.Za, D, (const int64_t*)nullptr, ma,
                         (const T*)g.Za, D, ma, D, (const T*)g.scales, g.k.kind, kvar(g), g.Kta, map, map, map, 0,
                         T(0), (const T*)nullptr, (T*)nullptr, (int64_t)0);
      hipLaunchKernelGGL((k_add_diag<T>), grid1(ma), dim3(256), 0, st(), g.Kta, map, ma, (T)jitter);
      LAUNCHCHK(ctx);
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.kappa_a, mp, g.Kab, mp, map, map, mp, 0, g.onQ, map, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));
      hipLaunchKernelGGL((k_sub2d<T>), grid2(ma, ma), blk2, 0, st(), g.Kta, (const T*)g.onQ, map, ma);
    }
    LAUNCHCHK(ctx);
    // eta1 target gains kappa_a' eta1_a ; -eta2 target gains kappa_a' invD_a kappa_a / 2  (analyticVI.jl:197-201): folded into
    // the K^-1 mu0 and K^-1 operands of the step kernels
    hipLaunchKernelGGL((k_gemv_cols<T>), grid1(mp), dim3(256), 0, st(), (const T*)g.kappa_a, mp, ma, mp, (const T*)g.peta1,
                       g.kinv_mu0_on);
    hipLaunchKernelGGL((k_axpby<T>), grid1(mp), dim3(256), 0, st(), mp, T(1), (const T*)g.kinv_mu0_on, T(1),
                       (const T*)g.kinv_mu0, g.kinv_mu0_on);
    {
      dim3 g1((unsigned)(mp / TILE), (unsigned)(map / TILE));  // T = invD_a kappa_a  (invD_a symmetric)
      hipLaunchKernelGGL((k_gemm_tn<T, 1>), g1, dim3(NTHREADS), 0, st(), (const T*)g.invDa, map, (const T*)g.kappa_a, mp, map,
                         g.onT, mp);
      dim3 g2((unsigned)(mp / TILE), (unsigned)(mp / TILE));   // kappa_a' T
      hipLaunchKernelGGL((k_gemm_tn<T, 1>), g2, dim3(NTHREADS), 0, st(), (const T*)g.kappa_a, mp, (const T*)g.onT, mp, map,
                         g.onQ, mp);
    }
    HIPCHK(ctx, hipMemcpyAsync(g.Kinv_on, g.Kinv, sizeof(T) * mp * mp, hipMemcpyDeviceToDevice, st()));
    hipLaunchKernelGGL((k_add_sym<T>), grid2(m, m), blk2, 0, st(), (const T*)g.Kinv, (const T*)g.onQ, mp, m, g.Kinv_on);
    LAUNCHCHK(ctx);
    g.on_dirty = false;
    return AGP_OK;
  }

  // (invD_a, eta1_a, L_a) of save_old_gp! (onlinetraining.jl:170-180) from the current posterior
  agp_status online_snapshot(int l, void* invDa_out, int64_t ldi, void* eta1_out, double* prevLa_host) override {
    if (l < 0 || l >= nl || !invDa_out || !eta1_out || !prevLa_host || ldi < m) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    AGPCHK(refresh_K());
    AGPCHK(materialize(g));
    hipLaunchKernelGGL((k_axpby<T>), grid1(mp * mp), dim3(256), 0, st(), mp * mp, T(-2), (const T*)g.eta2, T(-1),
                       (const T*)g.Kinv, Tw);
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipMemcpy2DAsync(invDa_out, sizeof(T) * ldi, Tw, sizeof(T) * mp, sizeof(T) * m, m, hipMemcpyDeviceToDevice,
                                 st()));
    HIPCHK(ctx, hipMemcpyAsync(eta1_out, g.eta1, sizeof(T) * m, hipMemcpyDeviceToDevice, st()));
    hipLaunchKernelGGL((k_logdiag_sum<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.DgA, m, scal_dev + 2);
    hipLaunchKernelGGL((k_dot<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.mu, (const T*)g.eta1, m, scal_dev + 3);
    LAUNCHCHK(ctx);
    double h[2];
    HIPCHK(ctx, hipMemcpyAsync(h, scal_dev + 2, sizeof(double) * 2, hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    // (-logdet Sigma + logdet K - mu'eta1)/2 with logdet Sigma = -2 sum log diag chol(-2 eta2)
    AGPCHK(resolve_logdet(g));
    *prevLa_host = 0.5 * (2.0 * h[0] + 2.0 * g.half_logdetK - h[1]);
    return AGP_OK;
  }

  // local variables and expectation gradients of `src` (same likelihood, same batch capacity) become this handle's:
  // first iteration of a new streaming batch (onlinetraining.jl:78-104: local update under the OLD inducing points)
  agp_status adopt_local(SvgpBase* srcb) override {
    Svgp<T>* src = dynamic_cast<Svgp<T>*>(srcb);
    if (!src || src == this || src->nl != nl || src->Bp != Bp || src->lp.kind != lp.kind ||
        src->ctx->device != ctx->device || src->ctx->stream != ctx->stream)
      return AGP_ERR_INVALID;  // both handles must enqueue on the same stream: the copies are ordered by it
    AGPCHK(src->lsm_finish());
    const size_t nb = sizeof(T) * nl * Bp;
    T* d[] = {cbuf, theta, gamma, rbuf, wbuf};
    T* s0[] = {src->cbuf, src->theta, src->gamma, src->rbuf, src->wbuf};
    for (int i = 0; i < 5; ++i) HIPCHK(ctx, hipMemcpyAsync(d[i], s0[i], nb, hipMemcpyDeviceToDevice, st()));
    HIPCHK(ctx, hipMemcpyAsync(alpha, src->alpha, sizeof(T) * Bp, hipMemcpyDeviceToDevice, st()));
    HIPCHK(ctx, hipMemcpyAsync(lam_dev, src->lam_dev, sizeof(T), hipMemcpyDeviceToDevice, st()));
    return AGP_OK;
  }

  // extraKL (KLdivergences.jl:30-54)
  agp_status extra_kl(Latent& g, double* out) {
    const int64_t map = g.map, ma = g.ma;
    AGPCHK(materialize(g));
    hipLaunchKernelGGL((k_gemv_rows<T>), grid1(ma * 64), dim3(256), 0, st(), (const T*)g.kappa_a, mp, ma, mp, (const T*)g.mu,
                       g.ov0);                                                     // kappa_a mu
    AGPCHK(frob_dot((const T*)g.invDa, (const T*)g.Kta, map, ma,
                    scal_dev + 48));  // tr(invD_a K~_a)
    LAUNCHCHK(ctx);
    AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.kappa_a, mp, g.Sigma, mp, map, mp, mp, 0, g.onT, mp, nullptr, 0, nullptr, nullptr,
                                  nullptr, 0)));                                 // kappa_a Sigma
    AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.onT, mp, g.kappa_a, mp, map, map, mp, 0, g.onQ, map, nullptr, 0, nullptr, nullptr,
                                  nullptr, 0)));                                 // (kappa_a Sigma) kappa_a'
    AGPCHK(frob_dot((const T*)g.invDa, (const T*)g.onQ, map, ma, scal_dev + 49));
    hipLaunchKernelGGL((k_dot<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.peta1, (const T*)g.ov0, ma, scal_dev + 50);
    hipLaunchKernelGGL((k_symv<T>), grid1(ma * 64), dim3(256), 0, st(), (const T*)g.invDa, map, ma, (const T*)g.ov0, g.ov1);
    hipLaunchKernelGGL((k_dot<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.ov0, (const T*)g.ov1, ma, scal_dev + 51);
    LAUNCHCHK(ctx);
    double h[4];
    HIPCHK(ctx, hipMemcpyAsync(h, scal_dev + 48, sizeof(double) * 4, hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    *out = g.prevLa - 0.5 * (h[0] + h[1]) + h[2] - 0.5 * h[3];
    return AGP_OK;
  }

  // the ready-made Y tiles of every kernel matrix against this latent's inducing points (k_kernelmatrix_mma)
  agp_status ensure_zsc(Latent& g) {
    if (g.zsc_valid || !kmm_usable(D)) return AGP_OK;
    const int Dp = kmm_dp(D);
    if (!g.Zsc) {
      AGPCHK(dmalloc(ctx, &g.Zsc, mp * Dp));
      AGPCHK(dmalloc(ctx, &g.zn, mp));
    }
    hipLaunchKernelGGL((k_scale_rows<T>), dim3((unsigned)((mp + 3) / 4)), dim3(256), 0, st(), (const T*)g.Z, D, m, mp, D, Dp,
                       (const T*)g.scales, g.Zsc, g.zn);
    LAUNCHCHK(ctx);
    g.zsc_valid = true;
    return AGP_OK;
  }

  // which K-derived matrices the STEP uses (online prior folded in > frozen copies of AGP_FLAG_STALE_K > the fresh ones)
  const T* kinv_kappa(const Latent& g) const { return g.stale_on ? g.sKinv : g.Kinv; }
  const T* kinv_step(const Latent& g) const { return g.on ? g.Kinv_on : (g.stale_on ? g.sKinv : g.Kinv); }
  const T* kinv_mu0_step(const Latent& g) const { return g.on ? g.kinv_mu0_on : (g.stale_on ? g.skinv_mu0 : g.kinv_mu0); }
  bool stale_mode() const { return (desc.flags & AGP_FLAG_STALE_K) != 0; }

  // compute_K where train! starts and ends (training.jl:41-43,107): with AGP_FLAG_STALE_K this is the only thing that ends
  // the staleness; without the flag it is plain refresh_K
  agp_status refresh_K_explicit() override {
    for (auto& g : lat) g.stale_on = false;
    return refresh_K();
  }

  agp_status check_batch(int64_t B) {
    if (B <= 0 || B > Bmax) {  // training.jl:27-29
      ctx->err = "The size of mini-batch " + std::to_string(B) + " is incorrect (negative or bigger than max_batch)";
      return AGP_ERR_BAD_BATCH;
    }
    return AGP_OK;
  }

  // compute_kappa + mean_f/var_f + local update
  agp_status step_local(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho,
                        bool fresh) override {
    AGPCHK(check_batch(B));
    if (!x || !y || ldx < D) return AGP_ERR_INVALID;
    lsm_finished = false;
    for (auto& g : lat) g.C_valid = false;  // the local variables (theta, w) and kappa it belongs to are about to change
    dag_tick(ctx);
    refresh_lazy = !fresh;  // a refresh issued from inside the training loop (the hyper step moved the kernel) does not synchronise
    const agp_status rks = refresh_K();
    refresh_lazy = false;
    AGPCHK(rks);
    const int64_t Bq = rup64(B);
    const int ns = (int)(2 * mp / TILE);
    const bool reuse = !desc.stochastic && !fresh && x == x_last && idx == idx_last && B == B_last && ldx == ldx_last;
    const bool reuse_eval = fresh && eval_cache_ok && x == ev_x && idx == ev_idx && B == ev_B && ldx == ev_ldx;
    if (!fresh)
      for (auto& g : lat) g.kappa_eval = false;  // (a training step takes the buffers)
    const bool prefetched = pf_valid && !fresh && x == pf_x && idx == pf_idx && B == pf_B && ldx == pf_ldx;
    // how many problems one task-graph launch may take (0: none fits, plain launches)
    int dag_nb = 0;
    for (int q = DAG_MAX_NB; q >= 1 && !dag_nb; --q)
      if (chol_use_dag(ctx, mp / TILE, Bq / TILE + 1, q)) dag_nb = q;
    // single latent on the task graph: the launch itself tells the look-ahead stream that the step
    // before has released its kappa buffers (DagSync) -- no event record on this stream
    StepSync ssync{};
    const bool sync_step = sig_state == 1 && !fresh && nl == 1 && dag_nb > 0 &&
                           !(lat[0].la_state == 1 && lat[0].xa_valid);
    if (sync_step) {
      ssync.ds.started = sig[0];
      ssync.ds.seq = started_seq + 1;
    } else if (rel_pending) {
      // the release of the previous step's kappa buffers becomes an event here, before anything of this step is enqueued (the
      // look-ahead waiting for it is meant to run next to this step's factorisation).  The previous step's deferred fallback goes
      // first: if it really re-runs it rewrites that step's Wbuf and reads its pk -- the buffers this event releases
      AGPCHK(run_deferred_safe());
      slot_kind[rel_slot] = 0;
      HIPCHK(ctx, hipEventRecord(step_done[rel_slot], st()));
      rel_pending = false;
    }
    // a pending natural-gradient step rides on this step's task-graph launch when this is the steady state of a training loop
    // (kappa of the minibatch already there -- look-ahead or kept --, one latent on the task graph); otherwise it is taken now
    bool use_pro = false;
    if (pend.on || pendp.on) {
      use_pro = !fresh && nl == 1 && dag_nb > 0 && (prefetched || (reuse && lat[0].kappa_valid)) && pro_allowed() &&
                mp / TILE <= 32 && (pendp.on || pend.Bq >= TILE);
      if (!use_pro) AGPCHK(flush());
    }
    // the row statistics of this step as the epilogue of its task-graph launch, the launch's fallback deferred to the next step
    const bool use_epi = use_pro && in_cavi_step && epi_allowed();
    if (prefetched) {  // kappa of this minibatch was produced on the prefetch stream: adopt those buffers
      if (sig_state == 1) {  // (the look-ahead's completion word is polled in-stream; without signal memory: an event wait)
        if (sdef.on) {  // the previous step's deferred fallback launch carries this step's wait for its look-ahead
          AGPCHK(run_deferred_safe((const int32_t*)sig[1], pf_seq));
        } else {
          hipLaunchKernelGGL(k_wait_ge_fast, dim3(1), dim3(64), 0, st(), (const int32_t*)sig[1], pf_seq, info_dev);
          LAUNCHCHK(ctx);
        }
      } else {
        AGPCHK(run_deferred_safe());
        HIPCHK(ctx, hipStreamWaitEvent(st(), pf_done, 0));
      }
      for (auto& g : lat) {
        std::swap(g.Knm, g.Knm_alt);
        std::swap(g.Wbuf, g.Wbuf_alt);
        std::swap(g.pk, g.pk_alt);
        // kappa rotates through three buffers: the one just left may still be read by this step's prologue (pending step), so
        // the next look-ahead target is the one before it
        T* left = g.kappa;
        g.kappa = g.kappa_alt;
        g.kappa_alt = g.kappa_old;
        g.kappa_old = left;
      }
      pf_valid = false;
    }
    AGPCHK(run_deferred_safe());  // (full-batch steps: no look-ahead to wait for)
    CholBatch<T> merged_bt{};  // single latent on the task graph: fallback + row statistics share a launch (k_safe_rowstats)
    SafeSrc<T> merged_src{};
    bool merged_safe = false;
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      const bool keep = (reuse && g.kappa_valid) || (reuse_eval && g.kappa_eval);
      g.keep_last = keep;
      if (prefetched) {
        // nothing to compute
      } else if (!keep) {
        AGPCHK(ensure_zsc(g));
        dim3 gk((unsigned)(mp / TILE), (unsigned)(Bq / TILE));
        (void)launch_kernelmatrix<T>(ctx, st(), (const T*)x, ldx, idx, B, (const T*)g.Z,
                           D, m, D, (const T*)g.scales, g.k.kind, kvar(g), g.Knm, mp, Bq, mp, 0, T(0),
                           (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)g.Zsc, (const T*)g.zn);
        LAUNCHCHK(ctx);
        AGPCHK((gemm_nt<T, EPI_KAPPA>(ctx, g.Knm, mp, kinv_kappa(g), mp, Bq, mp, mp, 0, g.kappa, mp, g.Knm, mp, nullptr,
                                      g.pk, g.Wbuf, ldp)));
        g.kappa_valid = !desc.stochastic && !fresh;
        g.kappa_eval = fresh && eval_cache_ok;
      } else if (!(g.la_state == 1 && g.xa_valid)) {  // (W = kappa Xa' below writes Wbuf itself)
        HIPCHK(ctx, hipMemcpyAsync(g.Wbuf, g.kappa, sizeof(T) * Bq * mp, hipMemcpyDeviceToDevice, st()));
      }
      // When the factor of the CURRENT -2*eta2 and its inverse X_a are still around (a materialize() since the last
      // global update: ELBO callback, hyper-parameter step, prediction), W = kappa X_a' and v = X_a eta1 are one GEMM and
      // one triangular mat-vec -- no second 16-launch factorisation chain of the same matrix.
      g.via_inverse = g.la_state == 1 && g.xa_valid;
      if (g.via_inverse) {
        AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.kappa, mp, g.Xa, mp, Bq, mp, mp, 1, g.Wbuf, mp, nullptr, 0, nullptr, nullptr,
                                      nullptr, 0)));
        g.v_ready = g.v_epoch == g.xa_epoch;  // materialize() left v = Xa eta1 of this very inverse in g.v
        if (!g.v_ready)
          hipLaunchKernelGGL((k_trmv_lower<T>), grid1(mp * 64), dim3(256), 0, st(), (const T*)g.Xa, mp, mp, (const T*)g.eta1,
                             g.Wbuf + Bq * mp);
        LAUNCHCHK(ctx);
        continue;
      }
      g.v_ready = false;
      // pre-factorisation part of aug_factor: -2*eta2 back into La if it holds a factor, extension rows [eta1' ; 0]
      if (g.la_state != 0 && !use_pro) {  // (with the prologue, -2 eta2 is formed inside the launch and never stored)
        hipLaunchKernelGGL((k_copy2d<T>), grid2(mp, mp), blk2, 0, st(), (const T*)g.eta2, mp, mp, mp, g.La, mp, mp, mp,
                           T(1), T(-2));
      }
      if (nl > 1 && dag_nb == 0)  // the task graphs take [eta1' ; 0] from eta1 itself (erow / CholBatch::R): only the plain launches need it in E
        hipLaunchKernelGGL((k_set_ext_rows<T>), grid1(TILE * mp), dim3(256), 0, st(), g.Wbuf + Bq * mp, mp, mp,
                           (const T*)g.eta1);
      LAUNCHCHK(ctx);
    }
    // the augmented Cholesky factorisations of the remaining latents share their launches (independent chains overlap)
    {
      std::vector<int> todo;
      for (int l = 0; l < nl; ++l)
        if (!lat[l].via_inverse) todo.push_back(l);
      if (!todo.empty()) AGPCHK(timing_begin());
      // chunking: task graphs take as many problems per launch as the residency bound allows (balanced chunks, e.g. 8 latents
      // as 4 + 4); if not even one fits, all of them share per-column launches
      const int64_t ntl = mp / TILE, nel = Bq / TILE + 1;
      size_t chunk = CHOL_MAXB;
      if (dag_nb > 0 && !todo.empty()) {
        const size_t nchunks = (todo.size() + dag_nb - 1) / dag_nb;
        chunk = (todo.size() + nchunks - 1) / nchunks;
      }
      int64_t launches = 0;
      for (size_t l0 = 0; l0 < todo.size(); l0 += chunk) {
        const int nb = (int)std::min<size_t>(chunk, todo.size() - l0);
        CholBatch<T>& bt = merged_bt;
        bt = CholBatch<T>{};
        for (int q = 0; q < nb; ++q) {
          Latent& g = lat[todo[l0 + q]];
          bt.A[q] = g.La;
          bt.X[q] = g.Xa;
          bt.Dg[q] = g.DgA;
          bt.E[q] = g.Wbuf;
          bt.R[q] = g.eta1;
          g.la_state = 1;
          g.xa_valid = false;
        }
        SafeSrc<T>& src = merged_src;  // where the in-stream fallback of the task graph finds the inputs again
        src = SafeSrc<T>{};
        src.Bq = Bq;
        for (int q = 0; q < nb; ++q) {
          Latent& g = lat[todo[l0 + q]];
          src.kappa[q] = g.kappa;
          src.eta1[q] = g.eta1;
          src.eta2[q] = g.eta2;
        }
        if (nb == 1) {  // also writes the [eta1' ; 0] block when it falls back to per-column launches
          bool defer = nl == 1;  // single latent: the row-statistics launch below carries the fallback (k_safe_rowstats)
          ProHost<T> ph{};
          if (use_pro) {
            Latent& g0 = lat[0];
            ph.ldk = mp;
            ph.eta2 = g0.eta2;
            ph.ldm = mp;
            ph.eta1 = g0.eta1;
            if (pendp.on) {  // reduced statistics of a batch-parallel step
              ph.packed = stats + mp;
              ph.tred = stats;
              if (pendp.overlap) {
                ph.arrive = arrive_dev;
                ph.arrive_want = pendp.epoch;
                memcpy(ph.grp, ov_grp, sizeof(ph.grp));
              }
              ph.Kdim = 0;
              ph.Kinv = pendp.Kinv;
              ph.kinv_mu0 = pendp.kinv_mu0;
              ph.lr = pendp.lr;
            } else {
              ph.kap = pend.kap;
              ph.Kdim = pend.Bq;
              ph.w = pend.w;
              ph.r = pend.r;
              ph.Kinv = pend.Kinv;
              ph.kinv_mu0 = pend.kinv_mu0;
              ph.lr = pend.lr;
            }
          }
          EpiArgs<T> ea{};
          if (use_epi) {
            Latent& g0 = lat[0];
            // this launch writes r, w of THIS minibatch while its prologue reads the pending step's pair: alternate
            std::swap(rbuf, rbuf2);
            std::swap(wbuf, wbuf2);
            ea.on = 1;
            ea.B = B;
            ea.nslices = ns;
            ea.pk = g0.pk;
            ea.ldp = ldp;
            ea.kdiag = kvar(g0);
            ea.kd_ptr = g0.scales + D;
            ea.use_kt = g0.keep_last ? 1 : 0;
            ea.jitter = (T)jitter;
            ea.rho = (T)rho;
            ea.lp = lp;
            ea.y = (const T*)y;
            ea.idx = idx;
            ea.Kt = Kt;
            ea.muf = muf;
            ea.varf = varf;
            ea.cb = cbuf;
            ea.theta = theta;
            ea.r = rbuf;
            ea.w = wbuf;
            ea.flags = flags_dev;
            ea.lam = lam_dev;
            ea.gamma = gamma;
          }
          AGPCHK(potrf_fused<T>(ctx, bt.A[0], mp, mp, bt.X[0], mp, bt.Dg[0], bt.E[0], mp, nel, 0, info_dev, m,
                                (const T*)lat[todo[l0]].eta1, false, &src, &defer, sync_step ? &ssync : nullptr,
                                use_pro ? &ph : nullptr, use_epi ? &ea : nullptr));
          if (use_pro) {  // the launch has taken the pending step (had it been refused, the step would still be pending for flush())
            pend.on = pendp.on = pendp.overlap = false;
            n_prologue += 1;
          }
          merged_safe = defer;
          launches += dag_nb > 0 ? 1 : chol_launch_count(ntl, nel);
        } else if (dag_nb > 0) {
          AGPCHK(potrf_dag_batch<T>(ctx, bt, nb, mp, mp, mp, mp, nel, info_dev, m, &src));
          launches += 1;
        } else {
          AGPCHK(potrf_fused_batch<T>(ctx, bt, nb, mp, mp, mp, mp, nel, info_dev, m));
          launches += chol_launch_count(ntl, nel);
        }
      }
      if (!todo.empty()) AGPCHK(timing_end(launches));
    }
    if (ssync.used) started_seq = ssync.ds.seq;
    if (rel_pending) {  // the release of the previous step's kappa buffers: this step's `started` number, or an event now
      slot_kind[rel_slot] = ssync.used ? 1 : 0;
      slot_seq[rel_slot] = ssync.ds.seq;
      if (!ssync.used) HIPCHK(ctx, hipEventRecord(step_done[rel_slot], st()));
      rel_pending = false;
    }
    for (int l0 = 0; l0 < nl; l0 += ROWSTATS_MAXB) {  // row statistics + local update of all latents in one launch
      const int nb = std::min(ROWSTATS_MAXB, nl - l0);
      RowstatsBatch<T> rb{};
      for (int q = 0; q < nb; ++q) {
        Latent& g = lat[l0 + q];
        rb.pk[q] = g.pk;
        rb.W[q] = g.Wbuf;
        rb.v[q] = (g.via_inverse && g.v_ready) ? g.v : g.Wbuf + Bq * mp;
        rb.kdiag[q] = kvar(g);
        rb.kd_ptr[q] = g.scales + D;
        rb.use_kt[q] = g.keep_last ? 1 : 0;
      }
      if (merged_safe) {  // nl == 1
        AGPCHK(ensure_safe_words<T>(ctx));
        const int64_t nt_ = mp / TILE, ne_ = Bq / TILE + 1;
        const int64_t most = std::max<int64_t>(nt_ + ne_ + nt_ * (nt_ + 1) / 2 + ne_ * nt_, (B + 7) / 8);
        const unsigned g1 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(safe_grid_cap(ctx), most));
        if (use_epi) {  // the rows are done inside the task graph: the fallback launch waits for the head of the next step
          sdef.on = true;
          sdef.bt = merged_bt;
          sdef.src = merged_src;
          sdef.rb = rb;
          sdef.ne = ne_;
          sdef.nt = nt_;
          sdef.B = B;
          sdef.ns = ns;
          sdef.rho = (T)rho;
          sdef.y = (const T*)y;
          sdef.idx = idx;
          sdef.r = rbuf;
          sdef.w = wbuf;
          sdef.grid = g1;
          continue;
        }
        hipLaunchKernelGGL((k_safe_rowstats<T>), dim3(g1), dim3(CHOL_THREADS), 0, st(), merged_bt, merged_src, mp, mp, mp, ne_, nt_,
                           info_dev, m, ctx->safe_bar, ctx->safe_retries, B, ns, rb, ldp, mp, mp, (T)jitter, (T)rho, lp, (const T*)y,
                           idx, Kt, muf, varf, cbuf, theta, rbuf, wbuf, flags_dev, (const T*)lam_dev, gamma);
        LAUNCHCHK(ctx);
        continue;
      }
      const dim3 grid((unsigned)((B * 64 + 255) / 256), (unsigned)nb);
      hipLaunchKernelGGL((k_rowstats_local<T>), grid, dim3(256), 0, st(), B, ns, rb, ldp, mp, mp, (T)jitter, (T)rho, lp,
                         (const T*)y, idx, Kt + l0 * Bp, muf + l0 * Bp, varf + l0 * Bp, cbuf + l0 * Bp, theta + l0 * Bp,
                         rbuf + l0 * Bp, wbuf + l0 * Bp, Bp, flags_dev, (const T*)lam_dev, gamma + l0 * Bp);
      LAUNCHCHK(ctx);
    }
    if (lp.kind == AGP_LIK_GAUSSIAN && lp.noise_dev) {  // sigma2 step, then theta / gradients with the new sigma2  gaussian.jl:56-72
      const int nb = (int)((B + 255) / 256);
      // a fresh evaluation (external ELBO: new local variables, ELBO.jl:32-47) steps sigma2 from a NEW optimiser state and throws
      // that state away, as the reference does; the training state is left alone
      double* ad = fresh ? noise_adam + 3 : noise_adam;
      if (fresh) HIPCHK(ctx, hipMemsetAsync(ad, 0, sizeof(double) * 3, st()));
      hipLaunchKernelGGL((k_noise_partial<T>), dim3(nb), dim3(256), 0, st(), B, (const T*)y, idx, (const T*)muf, (const T*)varf,
                         lam_part);
      if (lam_deferred) {
        hipLaunchKernelGGL(k_lambda_reduce, dim3(1), dim3(256), 0, st(), nb, 1, (const double*)lam_part, (double)B, scal_dev + 60);
      } else {
        hipLaunchKernelGGL((k_noise_finish<T>), dim3(1), dim3(256), 0, st(), nb, (const double*)lam_part, (double)B,
                           (const double*)nullptr, noise_eta, 0.9, 0.999, 1e-8, ad, lam_dev);
        hipLaunchKernelGGL((k_gauss_grads<T>), dim3(nb), dim3(256), 0, st(), B, (T)rho, (const T*)y, idx, (const T*)lam_dev, theta,
                           cbuf, rbuf, wbuf);
      }
      LAUNCHCHK(ctx);
    } else if (lp.kind == AGP_LIK_POISSON) {  // lambda <- sum(y) / sum E[logistic(f)]   poisson.jl:78
      if (gh_n <= 0) {
        ctx->err = "PoissonLikelihood: install the Gauss-Hermite rule first (agp_svgp_set_quadrature)";
        return AGP_ERR_INVALID;
      }
      const int nb = (int)((B + 255) / 256);
      hipLaunchKernelGGL((k_poisson_partial<T>), dim3(nb), dim3(256), 0, st(), B, (const T*)y, idx, (const T*)muf,
                         (const T*)varf, gh_n, (const double*)gh_dev, (const double*)(gh_dev + gh_n), lam_part);
      if (lam_deferred)  // batch-sharded: the sums travel first (cavi_step_multi), lambda_finish_reduced() follows
        hipLaunchKernelGGL(k_lambda_reduce, dim3(1), dim3(256), 0, st(), nb, 2, (const double*)lam_part, (double)B, scal_dev + 60);
      else
        hipLaunchKernelGGL((k_lambda_finish<T>), dim3(1), dim3(256), 0, st(), nb, 2, (const double*)lam_part, 0, (double)B,
                           lam_dev);
      LAUNCHCHK(ctx);
    } else if (lp.kind == AGP_LIK_HETEROSCEDASTIC) {  // heteroscedastic.jl:71-129
      const int nb = (int)((B + 255) / 256);
      hipLaunchKernelGGL((k_hetero_local<T>), dim3(nb), dim3(256), 0, st(), B, Bp, (const T*)y, idx, (const T*)muf,
                         (const T*)varf, (const T*)lam_dev, cbuf, gamma, theta, lam_part);
      if (lam_deferred) {
        hipLaunchKernelGGL(k_lambda_reduce, dim3(1), dim3(256), 0, st(), nb, 1, (const double*)lam_part, (double)B, scal_dev + 60);
      } else {
        hipLaunchKernelGGL((k_lambda_finish<T>), dim3(1), dim3(256), 0, st(), nb, 1, (const double*)lam_part, 1, (double)B,
                           lam_dev);
        hipLaunchKernelGGL((k_hetero_grads<T>), dim3(nb), dim3(256), 0, st(), B, Bp, (T)rho, (const T*)y, idx,
                           (const T*)lam_dev, (const T*)gamma, theta, rbuf, wbuf);
      }
      LAUNCHCHK(ctx);
    }
    x_last = x;
    y_last = y;
    idx_last = idx;
    B_last = B;
    ldx_last = ldx;
    rho_last = rho;
    if (fresh) {  // (whose kappa the buffers hold now: see ev_x)
      ev_x = x;
      ev_idx = idx;
      ev_B = B;
      ev_ldx = ldx;
    }
    if (lp.kind == AGP_LIK_MULTIOUTPUT) {
      if (!mo) {
        ctx->err = "multi-output handle: call agp_svgp_set_multioutput first";
        return AGP_ERR_INVALID;
      }
      if (mo_sharded) return publish_f(muf, varf, 1);  // the driver all-reduces `fall`, then agp_svgp_mo_mix
      AGPCHK(mo_local(y, idx, B, rho, !fresh, muf, varf));
    }
    return AGP_OK;
  }

  // ---- hyper-parameter / inducing-point gradient (see agp_hyper.h) ------------------------------------------------
  agp_status hyper_rule(int k_rule, double k_rho, int z_rule, double z_rho) override {
    for (int r : {k_rule, z_rule})
      if (r != AGP_OPT_ADAM && r != AGP_OPT_DESCENT && r != AGP_OPT_MOMENTUM) {
        ctx->err = "agp_svgp_hyper_rule: unknown optimiser rule";
        return AGP_ERR_INVALID;
      }
    hy_krule = k_rule;
    hy_krho = k_rho;
    hy_zrule = z_rule;
    hy_zrho = z_rho;
    return AGP_OK;
  }
  agp_status hyper_configure(int opt_k, double k_eta, int opt_z, double z_eta, double b1, double b2,
                             double eps) override {
    hy_k = opt_k != 0;
    hy_z = opt_z != 0;
    hy_keta = k_eta;
    hy_zeta = z_eta;
    hy_b1 = b1;
    hy_b2 = b2;
    hy_eps = eps;
    return AGP_OK;
  }

  agp_status hyper_alloc() {
    if (hyH1) return AGP_OK;
    if (D > HB_MAXD) {
      ctx->err = "hyper-gradient: input dimension above HB_MAXD";
      return AGP_ERR_UNSUPPORTED;
    }
    // (partial sums per RT x 64 tile of a backward pass: k_kernel_backward, HB_RT rows per workgroup)
    const int64_t tiles = std::max((Bp / HB_RT) * (mp / TILE), (mp / HB_RT) * (mp / TILE));
    const int64_t rowt = std::max(Bp / HB_RT, mp / HB_RT);
    AGPCHK(dmalloc(ctx, &hyH1, Bp * mp));
    AGPCHK(dmalloc(ctx, &hyH2, Bp * mp));
    AGPCHK(dmalloc(ctx, &hyH3, Bp * mp));
    AGPCHK(dmalloc(ctx, &hy_gmu, Bp));
    AGPCHK(dmalloc(ctx, &hy_gs, Bp));
    AGPCHK(dmalloc(ctx, &hy_muf, Bp));
    AGPCHK(dmalloc(ctx, &hy_pZ, rowt * mp * D));
    AGPCHK(dmalloc(ctx, &hy_dZ, m * D));
    AGPCHK(dmalloc(ctx, &hy_pvar, tiles));
    AGPCHK(dmalloc(ctx, &hy_pscale, tiles * D));
    AGPCHK(dmalloc(ctx, &hy_g, 1 + D));
    // second set of partial sums: both backward passes are reduced by ONE launch (k_hyper_reduce2)
    AGPCHK(dmalloc(ctx, &hy_pZ2, rowt * mp * D));
    AGPCHK(dmalloc(ctx, &hy_pvar2, tiles));
    AGPCHK(dmalloc(ctx, &hy_pscale2, tiles * D));
    return AGP_OK;
  }

  // gradient of the hyper objective w.r.t. (variance, per-dimension scales, Z) of latent l, on the batch of the last step
  agp_status hypergrad(int l, double* dvar, double* dscale, void* dZ_out) override {
    if (l < 0 || l >= nl || !x_last || B_last <= 0) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    if (g.k.kind == AGP_K_EXPONENTIAL) {
      ctx->err = "hyper-gradient: ExponentialKernel is not differentiable at zero distance";
      return AGP_ERR_UNSUPPORTED;
    }
    AGPCHK(hyper_alloc());
    const int64_t B = B_last, Bq = rup64(B);
    const T rho = (T)rho_last;
    // round 4: H = G_kappa K^-1 comes from ONE product, kappa (Sigma K^-1), with the K^-1 Sigma that is formed on the way to Apred
    // (k_hyper_hk) -- not kappa Sigma followed by (.) K^-1.  The heteroscedastic model needs kappa Sigma itself (var_f under the
    // current posterior) and keeps the two products.
    const bool one_product = lp.kind != AGP_LIK_HETEROSCEDASTIC;
    // ... and G_K from ONE more, C (Sigma K^-1), when the factorisation launch inside materialize() took the pending natural-gradient
    // step as its prologue and left C = kappa' diag(w) kappa + K^-1 / 4 behind (aug_factor, k_hyper_gK_fused): no kappa' H, no Apred
    AGPCHK(refresh_K());
    AGPCHK(materialize(g));
    const bool gk_fused = one_product && g.C_valid && g.C_kap == g.kappa && !g.stale_on && !(g.on && !g.on_first);
    n_hgrad += 1;
    n_gk_fused += gk_fused ? 1 : 0;
    // K^-1 mu rides on the mean_f / g_mu / g_sigma launch (k_hyper_muf_gvec) when that launch exists and nothing else needs it first
    const bool fuse_a = one_product && !mo && !g.pred_valid;
    struct PredGuard {  // an error return between here and the fused launch must not leave K^-1 mu marked valid
      bool* flag = nullptr;
      ~PredGuard() {
        if (flag) *flag = false;
      }
    } pred_guard;
    if (fuse_a) {
      if (!g.apred) AGPCHK(dmalloc(ctx, &g.apred, mp));
      g.pred_valid = true;  // (ensure_pred below then skips its own launch)
      pred_guard.flag = &g.pred_valid;
    }
    AGPCHK(ensure_pred(g, !gk_fused));  // Sigma, mu, K^-1 mu (, Apred = K^-1 - K^-1 Sigma K^-1)
    // AGP_FLAG_STALE_K: the step's kappa mixes the new Knm with the frozen inv(K); the differentiated ELBO recomputes the
    // kernel matrices (ELBO.jl:15-21), so the gradient takes kappa = Knm K^-1 with the FRESH inverse
    if (g.stale_on && lp.kind == AGP_LIK_HETEROSCEDASTIC) {
      ctx->err = "reference_compat_stale_K is not wired for the heteroscedastic hyper-gradient";
      return AGP_ERR_UNSUPPORTED;
    }
    // (a full-batch AnalyticVI run does not even recompute Knm after a hyper step -- its kappa cache stays valid,
    //  training.jl:196-204 -- so there the gradient also needs Knm under the current kernel and Z)
    auto knm_for_grad = [&](Latent& q) -> const T* {
      if (!q.stale_on || desc.stochastic) return q.Knm;
      if (!hyKnm && dmalloc(ctx, &hyKnm, Bp * mp) != AGP_OK) return nullptr;
      dim3 gk((unsigned)(mp / TILE), (unsigned)(Bq / TILE));
      (void)launch_kernelmatrix<T>(ctx, st(), (const T*)x_last, ldx_last, idx_last, B,
                         (const T*)q.Z, D, m, D, (const T*)q.scales, q.k.kind, kvar(q), hyKnm, mp, Bq, mp, 0, T(0),
                         (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)q.Zsc, (const T*)q.zn);
      return hyKnm;
    };
    auto kappa_for_grad = [&](Latent& q) -> const T* {
      if (!q.stale_on) return q.kappa;
      const T* kn = knm_for_grad(q);
      if (!kn) return nullptr;
      if (!hyKap && dmalloc(ctx, &hyKap, Bp * mp) != AGP_OK) return nullptr;
      if (gemm_nt<T, EPI_STORE>(ctx, kn, mp, q.Kinv, mp, Bq, mp, mp, 0, hyKap, mp, nullptr, 0, nullptr, nullptr, nullptr,
                                0) != AGP_OK)
        return nullptr;
      return hyKap;
    };
    const T* kap = kappa_for_grad(g);
    if (!kap) return AGP_ERR_NOMEM;
    // mean_f with the current posterior, then g_mu / g_sigma from the step's local variables
    if (!one_product || mo)
      hipLaunchKernelGGL((k_gemv_rows<T>), grid1(B * 64), dim3(256), 0, st(), kap, mp, B, mp, (const T*)g.mu, hy_muf);
    if (!one_product)
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, kap, mp, g.Sigma, mp, Bq, mp, mp, 0, hyH1, mp, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));
    if (mo) {
      // mixed means under the current posterior need every latent's mean_f on this batch
      if (mo_sharded && fall_state != 2) {
        ctx->err = "latent-sharded multi-output hyper-gradient: call agp_svgp_mo_refresh_f and all-reduce the exchange "
                   "buffer first";
        return AGP_ERR_INVALID;
      }
      for (int q = 0; q < nl && !mo_sharded; ++q) {
        AGPCHK(materialize(lat[q]));
        const T* kq = q == l ? kap : kappa_for_grad(lat[q]);
        if (!kq) return AGP_ERR_NOMEM;
        hipLaunchKernelGGL((k_gemv_rows<T>), grid1(B * 64), dim3(256), 0, st(), kq, mp, B, mp, (const T*)lat[q].mu,
                           emuf + q * Bp);
      }
      if (g.stale_on && nl > 1 && !mo_sharded) {  // the scratch now holds another latent's kappa
        kap = kappa_for_grad(g);
        if (!kap) return AGP_ERR_NOMEM;
      }
      hipLaunchKernelGGL((k_mo_hyper_gvec<T>), grid1(B), dim3(256), 0, st(), B, Qa(), Bp, mocfg, (const T*)A_dev,
                         (const T*)y_last, ystride, idx_last, mo_sharded ? (const T*)fall : (const T*)emuf,
                         (const T*)mo_th, l + qlo,
                         (int)(desc.elbo_mode == AGP_ELBO_REFERENCE), hy_gmu, hy_gs);
      LAUNCHCHK(ctx);
    } else {
      int gmode = 0;
      const bool refm = desc.elbo_mode == AGP_ELBO_REFERENCE;
      if (refm && (lp.kind == AGP_LIK_LOGISTIC || lp.kind == AGP_LIK_NEGBINOMIAL)) gmode = 1;
      if (refm && lp.kind == AGP_LIK_BAYESIANSVM) gmode = 2;
      if (lp.kind == AGP_LIK_HETEROSCEDASTIC && l == 0) {
        gmode = 3;  // needs var_f under the current posterior: rowdot(kappa Sigma, kappa) + K~
        hipLaunchKernelGGL((k_hyper_varf<T>), grid1(B * 64), dim3(256), 0, st(), B, mp, mp, (const T*)hyH1,
                           kap, (const T*)(Kt + l * Bp), pw0);
      }
      if (one_product) {
        hipLaunchKernelGGL((k_hyper_muf_gvec<T>), grid1((B + (fuse_a ? mp : 0)) * 64), dim3(256), 0, st(), B, mp, mp, rho, gmode,
                           kap, (const T*)g.mu, (const T*)(rbuf + l * Bp), (const T*)(theta + l * Bp), (const T*)y_last, idx_last,
                           hy_muf, hy_gmu, hy_gs, fuse_a ? (const T*)g.Kinv : (const T*)nullptr, g.apred);
        pred_guard.flag = nullptr;  // K^-1 mu is enqueued
      } else
        hipLaunchKernelGGL((k_hyper_gvec<T>), grid1(B), dim3(256), 0, st(), B, rho, gmode, (const T*)(rbuf + l * Bp),
                           (const T*)(theta + l * Bp), (const T*)hy_muf, (const T*)y_last, idx_last, (const T*)pw0,
                           (const T*)gamma, (const T*)lam_dev, hy_gmu, hy_gs);
      LAUNCHCHK(ctx);
    }
    const T* knm = knm_for_grad(g);  // (the scratch is per handle: recomputed after the loop over the other latents)
    if (!knm) return AGP_ERR_NOMEM;
    if (one_product) {
      if (tw2_kis_of != l) {  // Apred was still valid (a prediction since the last step): K^-1 Sigma is not in the scratch
        AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.Kinv, mp, g.Sigma, mp, mp, mp, mp, 0, Tw2, mp, nullptr, 0, nullptr, nullptr,
                                      nullptr, 0)));
        tw2_kis_of = l;
      }
      if (gk_fused) {
        // kappa (K^-1 Sigma)' = kappa Sigma K^-1 with the element-wise pass behind it as the product's epilogue (EPI_HK): G_Knm and
        // the partial column sums of u = kappa' g_mu, two rows of partials per tile row (T1 itself is never stored)
        if (!hy_upart) AGPCHK(dmalloc(ctx, &hy_upart, 2 * (Bp / TILE) * mp));
        HkArgs<T> hk{};
        hk.gmu = hy_gmu;
        hk.gs = hy_gs;
        hk.rho = rho;
        hk.B = B;
        AGPCHK((gemm_nt<T, EPI_HK>(ctx, kap, mp, Tw2, mp, Bq, mp, mp, 0, (T*)nullptr, mp, kap, mp, (const T*)g.apred, hy_upart, hyH3,
                                   mp, &hk)));
      } else {
        AGPCHK((gemm_nt<T, EPI_STORE>(ctx, kap, mp, Tw2, mp, Bq, mp, mp, 0, hyH1, mp, nullptr, 0, nullptr, nullptr, nullptr,
                                      0)));  // kappa (K^-1 Sigma)' = kappa Sigma K^-1
        hipLaunchKernelGGL((k_hyper_hk<T>), grid2(Bq, mp), blk2, 0, st(), B, Bq, mp, mp, rho, (const T*)hy_gmu,
                           (const T*)hy_gs, (const T*)g.apred, (const T*)hyH1, kap, hyH2, hyH3);
      }
      LAUNCHCHK(ctx);
    } else {
      hipLaunchKernelGGL((k_hyper_gkappa<T>), grid2(Bq, mp), blk2, 0, st(), B, Bq, mp, mp, rho, (const T*)hy_gmu,
                         (const T*)hy_gs, (const T*)g.mu, knm, hyH1);
      LAUNCHCHK(ctx);
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, hyH1, mp, g.Kinv, mp, Bq, mp, mp, 0, hyH2, mp, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));
      hipLaunchKernelGGL((k_hyper_gknm<T>), grid2(Bq, mp), blk2, 0, st(), B, Bq, mp, mp, rho, (const T*)hy_gs,
                         (const T*)hyH2, kap, hyH3);
    }
    if (gk_fused) {
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.Cmat, mp, Tw2, mp, mp, mp, mp, 0, Tw, mp, nullptr, 0, nullptr, nullptr, nullptr,
                                    0)));  // C (K^-1 Sigma)' = C Sigma K^-1
      hipLaunchKernelGGL((k_hyper_gK_fused<T>), dim3((unsigned)(mp / 32), (unsigned)(mp / 32)), dim3(256), 0, st(), m, mp,
                         (const T*)Tw, (const T*)g.Cmat,
                         (const T*)g.Kinv, (const T*)g.apred, (const T*)g.kinv_mu0, (const T*)hy_upart, (int)(2 * (Bq / TILE)), mp, rho,
                         Tw2);
    } else {
      dim3 gt((unsigned)(mp / TILE), (unsigned)(mp / TILE));
      if ((mp / TILE) * (mp / TILE) <= 320)
        hipLaunchKernelGGL((k_gemm_tn<T, 2>), gt, dim3(2 * NTHREADS), 0, st(), kap, mp, (const T*)hyH2, mp, Bq, Tw, mp);
      else
        hipLaunchKernelGGL((k_gemm_tn<T, 1>), gt, dim3(NTHREADS), 0, st(), kap, mp, (const T*)hyH2, mp, Bq, Tw, mp);
      hipLaunchKernelGGL((k_hyper_gK<T>), grid2(mp, mp), blk2, 0, st(), m, mp, (const T*)Tw, (const T*)g.Apred,
                         (const T*)g.apred, Tw2, (T)(1.0 / (double)bs_world), (const T*)g.kinv_mu0);
    }
    tw2_kis_of = -1;
    // backward through kernelmatrix(k, x, Z)  (gradient w.r.t. the second argument); its reduction initialises the gradient and
    // adds the kdiag term of the variance (rho sum_i g_sigma,i), which used to be a memset in front and a kernel behind.  Unless the
    // streaming model's extra passes follow, the reductions of both backward passes are ONE launch behind the second pass
    // (k_hyper_reduce2, second set of partial sums).
    const bool online_x = g.on && !g.on_first;
    const bool one_reduce = !online_x;
    if (!one_reduce) {  // (one_reduce: this pass shares the launch of the pass through K_ZZ below, k_kernel_backward2)
      dim3 gk((unsigned)(mp / TILE), (unsigned)(Bq / HB_RT));
      const int64_t tiles = (int64_t)gk.x * gk.y;
      hipLaunchKernelGGL((k_kernel_backward<T>), gk, dim3(NTHREADS), 0, st(), (const T*)x_last, ldx_last, idx_last, B,
                         (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), (const T*)hyH3, mp,
                         hy_pvar, hy_pscale, hy_pZ, mp);
      hipLaunchKernelGGL((k_hyper_reduce<T>), dim3((unsigned)(D + 1 + (m * D + 255) / 256)), dim3(256), 0, st(), tiles, D,
                           (const double*)hy_pvar, (const double*)hy_pscale, hy_g, 1.0, 1, (int64_t)gk.y, m, mp, (const T*)hy_pZ,
                           hy_dZ, T(1), (const T*)hy_gs, B, (double)rho);
    }
    if (online_x && bs_world > 1) {
      ctx->err = "hyper-gradient of a streaming (online) model on a batch-sharded handle is not wired";
      return AGP_ERR_UNSUPPORTED;
    }
    if (online_x) {
      // -extraKL (KLdivergences.jl:30-54) is part of the differentiated ELBO; its kernel matrices K_ab, kappa_a, K~_a are
      // recomputed with the candidate kernel / Z by compute_kappa(::OnlineVarLatent):
      //   G_kappa_a = D kappa_a (Sigma + mu mu') - eta_a mu' - D K_ab/2 ; G_Kab = G_kappa_a K^-1 - D kappa_a/2 ;
      //   G_K -= sym(kappa_a' G_kappa_a K^-1) ; G_Ka = D/2
      const int64_t map = g.map, ma = g.ma;
      if (map / TILE > std::max(Bp / TILE, mp / TILE)) {
        ctx->err = "online hyper-gradient: more old inducing points than the scratch was sized for";
        return AGP_ERR_UNSUPPORTED;
      }
      dim3 gam((unsigned)(mp / TILE), (unsigned)(map / TILE));
      hipLaunchKernelGGL((k_add_outer<T>), grid2(mp, mp), blk2, 0, st(), (const T*)g.Sigma, (const T*)g.mu, mp, m, Tw);
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.kappa_a, mp, Tw, mp, map, mp, mp, 0, g.oh1, mp, nullptr, 0, nullptr, nullptr, nullptr,
                                    0)));                                                       // kappa_a (Sigma + mu mu')
      hipLaunchKernelGGL((k_gemm_tn<T, 1>), gam, dim3(NTHREADS), 0, st(), (const T*)g.invDa, map, (const T*)g.oh1, mp, map,
                         g.oh2, mp);                                                              // D (.)
      hipLaunchKernelGGL((k_gemm_tn<T, 1>), gam, dim3(NTHREADS), 0, st(), (const T*)g.invDa, map, (const T*)g.Kab, mp, map,
                         g.oh3, mp);                                                              // D K_ab
      hipLaunchKernelGGL((k_online_gkappa<T>), grid2(map, mp), blk2, 0, st(), ma, m, map, mp, mp, (const T*)g.peta1,
                         (const T*)g.mu, (const T*)g.oh3, g.oh2);                                 // G_kappa_a
      LAUNCHCHK(ctx);
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.oh2, mp, g.Kinv, mp, map, mp, mp, 0, g.oh1, mp, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));                                              // G_kappa_a K^-1
      {
        dim3 gmm((unsigned)(mp / TILE), (unsigned)(mp / TILE));
        hipLaunchKernelGGL((k_gemm_tn<T, 1>), gmm, dim3(NTHREADS), 0, st(), (const T*)g.kappa_a, mp, (const T*)g.oh1, mp, map,
                           Tw, mp);                                                               // kappa_a' (G_kappa_a K^-1)
      }
      hipLaunchKernelGGL((k_sub_sym<T>), grid2(m, m), blk2, 0, st(), Tw2, (const T*)Tw, mp, m);
      hipLaunchKernelGGL((k_gemm_tn<T, 1>), gam, dim3(NTHREADS), 0, st(), (const T*)g.invDa, map, (const T*)g.kappa_a, mp, map,
                         g.oh3, mp);                                                              // D kappa_a
      hipLaunchKernelGGL((k_axpy2d<T>), grid2(map, mp), blk2, 0, st(), map, mp, mp, T(-0.5), (const T*)g.oh3, g.oh1);  // G_Kab
      LAUNCHCHK(ctx);
    }
    // backward through kernelmatrix(k, Z) : both arguments are Z and G_K is symmetric -> twice the second-argument part
    {
      dim3 gk((unsigned)(mp / TILE), (unsigned)(mp / HB_RT));
      const int64_t tiles = (int64_t)gk.x * gk.y;
      if (one_reduce) {
        KbPass<T> pa{(const T*)x_last, ldx_last, idx_last, B, (const T*)hyH3, mp, hy_pvar, hy_pscale, hy_pZ};
        KbPass<T> pb{(const T*)g.Z, D, (const int64_t*)nullptr, m, (const T*)Tw2, mp, hy_pvar2, hy_pscale2, hy_pZ2};
        const int64_t ny1 = Bq / HB_RT;
        hipLaunchKernelGGL((k_kernel_backward2<T>), dim3(gk.x, (unsigned)(ny1 + gk.y)), dim3(NTHREADS), 0, st(), pa, pb, ny1,
                           (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), mp);
      } else
        hipLaunchKernelGGL((k_kernel_backward<T>), gk, dim3(NTHREADS), 0, st(), (const T*)g.Z, D, (const int64_t*)nullptr,
                           m, (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), (const T*)Tw2, mp, hy_pvar, hy_pscale,
                           hy_pZ, mp);
      if (one_reduce && hy_grad_on_device_only && hy_k && hy_z && D + 1 <= 256) {
        // the training loop's hyper step: the reduction rides on the optimiser launch (k_hyper_reduce2_adam, hyper_apply_one)
        red_def.on = true;
        red_def.latent = l;
        red_def.nt1 = (int64_t)(mp / TILE) * (Bq / HB_RT);
        red_def.nrow1 = (int64_t)(Bq / HB_RT);
        red_def.nt2 = tiles;
        red_def.nrow2 = (int64_t)gk.y;
        red_def.B = B;
        red_def.rho = (double)rho;
      } else if (one_reduce)
        hipLaunchKernelGGL((k_hyper_reduce2<T>), dim3((unsigned)(D + 1 + (m * D + 255) / 256)), dim3(256), 0, st(), D, hy_g, m, mp,
                           hy_dZ, (int64_t)(mp / TILE) * (Bq / HB_RT), (const double*)hy_pvar, (const double*)hy_pscale,
                           (int64_t)(Bq / HB_RT), (const T*)hy_pZ, tiles, (const double*)hy_pvar2, (const double*)hy_pscale2,
                           (int64_t)gk.y, (const T*)hy_pZ2, (const T*)hy_gs, B, (double)rho);
      else
        hipLaunchKernelGGL((k_hyper_reduce<T>), dim3((unsigned)(D + 1 + (m * D + 255) / 256)), dim3(256), 0, st(), tiles, D,
                           (const double*)hy_pvar, (const double*)hy_pscale, hy_g, 1.0, 0, (int64_t)gk.y, m, mp, (const T*)hy_pZ,
                           hy_dZ, T(2), (const T*)nullptr, (int64_t)0, 0.0);
    }
    if (online_x) {
      // K_ab = k(Z_a, Z): gradient w.r.t. the kernel parameters and the second argument ; K_a = k(Z_a, Z_a): parameters only
      const int64_t map = g.map, ma = g.ma;
      dim3 gk((unsigned)(mp / TILE), (unsigned)(map / HB_RT));
      hipLaunchKernelGGL((k_kernel_backward<T>), gk, dim3(NTHREADS), 0, st(), (const T*)g.Za, D, (const int64_t*)nullptr, ma,
                         (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), (const T*)g.oh1, mp, hy_pvar,
                         hy_pscale, hy_pZ, mp);
      hipLaunchKernelGGL((k_hyper_reduce_scalar<T>), dim3((unsigned)(D + 1)), dim3(256), 0, st(), (int64_t)gk.x * gk.y, D,
                         (const double*)hy_pvar, (const double*)hy_pscale, hy_g, 1.0);
      hipLaunchKernelGGL((k_hyper_reduce_Z<T>), grid1(m * D), dim3(256), 0, st(), (int64_t)gk.y, m, mp, D, (const T*)hy_pZ,
                         hy_dZ, T(1), 1);
      dim3 ga((unsigned)(map / TILE), (unsigned)(map / HB_RT));
      hipLaunchKernelGGL((k_kernel_backward<T>), ga, dim3(NTHREADS), 0, st(), (const T*)g.Za, D, (const int64_t*)nullptr, ma,
                         (const T*)g.Za, D, ma, D, (const T*)g.scales, g.k.kind, kvar(g), (const T*)g.invDa, map,
                         hy_pvar, hy_pscale, hy_pZ, map);
      hipLaunchKernelGGL((k_hyper_reduce_scalar<T>), dim3((unsigned)(D + 1)), dim3(256), 0, st(), (int64_t)ga.x * ga.y, D,
                         (const double*)hy_pvar, (const double*)hy_pscale, hy_g, 0.5);
      LAUNCHCHK(ctx);
    }
    LAUNCHCHK(ctx);
    if (dZ_out) HIPCHK(ctx, hipMemcpyAsync(dZ_out, hy_dZ, sizeof(T) * m * D, hipMemcpyDeviceToDevice, st()));
    if (hy_grad_on_device_only) return AGP_OK;  // the training loop: the gradient stays in hy_g / hy_dZ for the device-side ADAM
    std::vector<double> hg(1 + D);
    HIPCHK(ctx, hipMemcpyAsync(hg.data(), hy_g, sizeof(double) * (1 + D), hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    if (dvar) *dvar = hg[0];
    if (dscale)
      for (int64_t d = 0; d < D; ++d) dscale[d] = hg[1 + d];
    hy_last = hg;
    return AGP_OK;
  }
  std::vector<double> hy_last;
  bool hy_grad_on_device_only = false;  // set around hypergrad() by hyper_step(): no download, no synchronisation
  struct RedDeferred {  // the gradient's final reduction, left to the optimiser launch (k_hyper_reduce2_adam)
    bool on = false;
    int latent = -1;
    int64_t nt1 = 0, nrow1 = 0, nt2 = 0, nrow2 = 0, B = 0;
    double rho = 0.0;
  } red_def;

  // ADAM moments of the kernel-parameter optimiser (1 + D entries: variance, scales; a ScaleTransform uses entry 1 only).  They
  // live on the device (Latent::kadam); this call copies them out / in (handle re-creation, the online model's chain of handles,
  // checkpoints) and synchronises.
  agp_status ensure_kadam(Latent& g) {
    if (!g.kadam) {
      AGPCHK(dmalloc(ctx, &g.kadam, 2 * (1 + D)));
      HIPCHK(ctx, hipMemsetAsync(g.kadam, 0, sizeof(double) * 2 * (1 + D), st()));
      g.k_step = 0;
    }
    return AGP_OK;
  }
  agp_status hyper_state(int l, int set, double* k_m, double* k_v, int32_t* k_step) override {
    if (l < 0 || l >= nl || !k_m || !k_v || !k_step) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    const size_t np = 1 + (g.k.ard ? (size_t)D : 1);
    AGPCHK(ensure_kadam(g));
    std::vector<double> h(2 * (1 + D), 0.0);
    if (set) {
      for (size_t i = 0; i < np; ++i) {
        h[i] = k_m[i];
        h[1 + D + i] = k_v[i];
      }
      HIPCHK(ctx, hipMemcpyAsync(g.kadam, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st()));
      HIPCHK(ctx, hipStreamSynchronize(st()));
      g.k_step = *k_step;
    } else {
      HIPCHK(ctx, hipMemcpyAsync(h.data(), g.kadam, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st()));
      HIPCHK(ctx, hipStreamSynchronize(st()));
      for (size_t i = 0; i < np; ++i) {
        k_m[i] = h[i];
        k_v[i] = h[1 + D + i];
      }
      *k_step = g.k_step;
    }
    return AGP_OK;
  }

  // ADAM ascent on latent l (update_kernel!, autotuning_utils.jl:47-67; update_Z!, :70-76) -- ON THE DEVICE, from the gradient in
  // hy_g = [dvariance, dscale_0..D-1] (doubles, w.r.t. the parameters themselves) and dZ_dev (m x D).  hg_host != nullptr: a
  // caller-supplied gradient (tied / all-reduced) is uploaded into hy_g first.  Nothing synchronises; the host copy of the kernel
  // parameters goes stale until somebody asks for it (params_to_host).
  agp_status hyper_apply_one(int l, const std::vector<double>* hg_host, const T* dZ_dev) {
    Latent& g = lat[l];
    const bool red_here = red_def.on && red_def.latent == l && !hg_host && dZ_dev == (const T*)hy_dZ;
    if (red_def.on && !red_here) {
      ctx->err = "hyper step: a deferred gradient reduction was not taken by its optimiser launch";
      red_def.on = false;
      return AGP_ERR_INVALID;
    }
    if (red_here) {  // reduction of both backward passes + both optimiser steps: ONE launch
      red_def.on = false;
      AGPCHK(hyper_alloc());
      AGPCHK(ensure_kadam(g));
      if (!g.z_am) {
        AGPCHK(dmalloc(ctx, &g.z_am, m * D));
        AGPCHK(dmalloc(ctx, &g.z_av, m * D));
        HIPCHK(ctx, hipMemsetAsync(g.z_am, 0, sizeof(double) * m * D, st()));
        HIPCHK(ctx, hipMemsetAsync(g.z_av, 0, sizeof(double) * m * D, st()));
      }
      g.k_step += 1;
      g.z_step += 1;
      const int64_t nzb = (m * D + HYPER_RA_THREADS - 1) / HYPER_RA_THREADS;
      hipLaunchKernelGGL((k_hyper_reduce2_adam<T>), dim3((unsigned)(nzb + 1)), dim3(HYPER_RA_THREADS), 0, st(), D, hy_g, m, mp, hy_dZ, red_def.nt1,
                         (const double*)hy_pvar, (const double*)hy_pscale, red_def.nrow1, (const T*)hy_pZ, red_def.nt2,
                         (const double*)hy_pvar2, (const double*)hy_pscale2, red_def.nrow2, (const T*)hy_pZ2, (const T*)hy_gs, red_def.B,
                         red_def.rho, g.Z, g.z_am, g.z_av, g.z_step, hy_zeta, hy_zrule, hy_zrho, g.k.ard ? 1 : 0,
                         g.k.has_variance ? 1 : 0, g.k.has_transform ? 1 : 0, g.scales, g.kadam, g.kadam + (1 + D), g.k_step, hy_keta,
                         hy_krule, hy_krho, hy_b1, hy_b2, hy_eps);
      LAUNCHCHK(ctx);
      g.host_params_stale = true;
      return AGP_OK;
    }
    if (hy_k && hy_z && dZ_dev && D + 1 <= 256) {  // both steps in one launch (k_adam_z_and_params)
      AGPCHK(hyper_alloc());
      AGPCHK(ensure_kadam(g));
      if (hg_host) {
        hy_up = *hg_host;  // staging that outlives the asynchronous copy
        HIPCHK(ctx, hipMemcpyAsync(hy_g, hy_up.data(), sizeof(double) * (1 + D), hipMemcpyHostToDevice, st()));
      }
      if (!g.z_am) {
        AGPCHK(dmalloc(ctx, &g.z_am, m * D));
        AGPCHK(dmalloc(ctx, &g.z_av, m * D));
        HIPCHK(ctx, hipMemsetAsync(g.z_am, 0, sizeof(double) * m * D, st()));
        HIPCHK(ctx, hipMemsetAsync(g.z_av, 0, sizeof(double) * m * D, st()));
      }
      g.k_step += 1;
      g.z_step += 1;
      const int64_t nzb = (m * D + 255) / 256;
      hipLaunchKernelGGL((k_adam_z_and_params<T>), dim3((unsigned)(nzb + 1)), dim3(256), 0, st(), m * D, nzb, g.Z, dZ_dev, g.z_am,
                         g.z_av, g.z_step, hy_zeta, hy_zrule, hy_zrho, (int)D, g.k.ard ? 1 : 0, g.k.has_variance ? 1 : 0,
                         g.k.has_transform ? 1 : 0, (const double*)hy_g, g.scales, g.kadam, g.kadam + (1 + D), g.k_step, hy_keta,
                         hy_krule, hy_krho, hy_b1, hy_b2, hy_eps);
      LAUNCHCHK(ctx);
      g.host_params_stale = true;
      return AGP_OK;
    }
    if (hy_k) {
      AGPCHK(hyper_alloc());
      AGPCHK(ensure_kadam(g));
      if (hg_host) {
        hy_up = *hg_host;  // staging that outlives the asynchronous copy
        HIPCHK(ctx, hipMemcpyAsync(hy_g, hy_up.data(), sizeof(double) * (1 + D), hipMemcpyHostToDevice, st()));
      }
      g.k_step += 1;
      hipLaunchKernelGGL((k_adam_kernel_params<T>), dim3(1), dim3(256), 0, st(), (int)D, g.k.ard ? 1 : 0, g.k.has_variance ? 1 : 0,
                         g.k.has_transform ? 1 : 0, (const double*)hy_g, g.scales, g.kadam, g.kadam + (1 + D), g.k_step, hy_keta,
                         hy_b1, hy_b2, hy_eps, hy_krule, hy_krho);
      LAUNCHCHK(ctx);
      g.host_params_stale = true;
    }
    if (hy_z && dZ_dev) {
      if (!g.z_am) {
        AGPCHK(dmalloc(ctx, &g.z_am, m * D));
        AGPCHK(dmalloc(ctx, &g.z_av, m * D));
        HIPCHK(ctx, hipMemsetAsync(g.z_am, 0, sizeof(double) * m * D, st()));
        HIPCHK(ctx, hipMemsetAsync(g.z_av, 0, sizeof(double) * m * D, st()));
      }
      g.z_step += 1;
      hipLaunchKernelGGL((k_adam_ascent<T>), grid1(m * D), dim3(256), 0, st(), m * D, g.Z, dZ_dev, g.z_am, g.z_av, g.z_step,
                         hy_zeta, hy_b1, hy_b2, hy_eps, hy_zrule, hy_zrho);
      LAUNCHCHK(ctx);
    }
    return AGP_OK;
  }
  std::vector<double> hy_up;

  agp_status hyper_finish() {
    for (auto& g : lat) {
      if (stale_mode() && !g.stale_on && !g.on && !g.K_stale) {
        // reference_compat_stale_K: freeze what the step uses before the kernel / Z move
        const int64_t mm = mp * mp;
        if (!g.sKinv) {
          AGPCHK(dmalloc(ctx, &g.sKinv, mm));
          AGPCHK(dmalloc(ctx, &g.sXk, mm));
        }
        HIPCHK(ctx, hipMemcpyAsync(g.sKinv, g.Kinv, sizeof(T) * mm, hipMemcpyDeviceToDevice, st()));
        HIPCHK(ctx, hipMemcpyAsync(g.sXk, g.Xk, sizeof(T) * mm, hipMemcpyDeviceToDevice, st()));
        if (g.kinv_mu0) {
          if (!g.skinv_mu0) AGPCHK(dmalloc(ctx, &g.skinv_mu0, mp));
          HIPCHK(ctx, hipMemcpyAsync(g.skinv_mu0, g.kinv_mu0, sizeof(T) * mp, hipMemcpyDeviceToDevice, st()));
        }
        AGPCHK(resolve_logdet(g));
        g.s_half_logdetK = g.half_logdetK;
        g.stale_on = true;
      }
      g.K_stale = true;  // (the kernel parameters were stepped in place on the device: nothing to upload)
      g.zsc_valid = false;
      // (the reference's full-batch path keeps its kernel matrices across a hyper step, training.jl:196-204)
      g.kappa_eval = false;
      if (!(g.stale_on && !desc.stochastic)) g.kappa_valid = false;
      g.pred_valid = g.predvar_valid = false;
    }
    pf_valid = false;
    return AGP_OK;
  }

  // update_hyperparameters!(m, state, x, y): ADAM ASCENT; positive kernel parameters are stepped in log space
  // (update_kernel!, autotuning_utils.jl:63-67), Z directly (update_Z!, :70-76).  K is refreshed before the next step.
  bool hyper_multi_ok = false;
  agp_status hyper_step() override {
    if (!hy_k && !hy_z) return AGP_OK;
    if (bs_world > 1 && !hyper_multi_ok) {
      // every rank would step its kernel / Z with the gradient of its own shard and the replicas would drift apart
      ctx->err = "batch-sharded handle: take the hyper step through agp_svgp_hyper_step_multi (the gradient is all-reduced)";
      return AGP_ERR_INVALID;
    }
    for (int l = 0; l < nl; ++l) {
      hy_grad_on_device_only = true;
      const agp_status hs = hypergrad(l, nullptr, nullptr, nullptr);
      hy_grad_on_device_only = false;
      //  hypergrad() marks the reduction as deferred BEFORE its last fallible calls (the online model's extra passes, a
      // launch check, a copy): an error exit must not leave the mark behind for the next, unrelated hyper_apply to trip over
      if (hs != AGP_OK) red_def.on = false;
      AGPCHK(hs);
      AGPCHK(hyper_apply_one(l, nullptr, (const T*)hy_dZ));
    }
    return hyper_finish();
  }

  // the optimiser step with a caller-supplied gradient (e.g. summed over latents / all-reduced over ranks: tied-Z mode)
  agp_status hyper_apply(int l, const double* dvar, const double* dscale, const void* dZ) override {
    if (l < 0 || l >= nl || !dvar || !dscale) return AGP_ERR_INVALID;
    if (!hy_k && !hy_z) return AGP_OK;
    std::vector<double> hg(1 + D);
    hg[0] = *dvar;
    for (int64_t d = 0; d < D; ++d) hg[1 + d] = dscale[d];
    AGPCHK(hyper_apply_one(l, &hg, (const T*)dZ));
    return hyper_finish();
  }

  agp_status get_kernel(int l, double* var, double* scales) override {
    if (l < 0 || l >= nl) return AGP_ERR_INVALID;
    AGPCHK(params_to_host(lat[l]));
    if (var) *var = lat[l].k.variance;
    if (scales)
      for (int64_t d = 0; d < D; ++d) scales[d] = lat[l].k.scales[d];
    return AGP_OK;
  }

  // MOSVGP(kernel, likelihoods, inference, Zs; Aoptimiser)  src/models/MOSVGP.jl:33-115 : install the task likelihoods,
  // the mixing weights A (n_task x n_latent, rows normalised by the caller like MOSVGP.jl:101-104) and the A optimiser.
  agp_status set_multioutput(int n_task, const agp_lik_desc* liks, const double* A_host, double eta, double b1,
                             double b2, double eps) override {
    if (lp.kind != AGP_LIK_MULTIOUTPUT || n_task <= 0 || n_task > MO_MAXT || !liks || !A_host) return AGP_ERR_INVALID;
    for (int t = 0; t < n_task; ++t) {
      const int tk = liks[t].kind;
      if (!(tk == AGP_LIK_GAUSSIAN || tk == AGP_LIK_LOGISTIC || tk == AGP_LIK_STUDENTT || tk == AGP_LIK_LAPLACE ||
            tk == AGP_LIK_BAYESIANSVM || tk == AGP_LIK_NEGBINOMIAL)) {
        ctx->err = "multi-output tasks support the Gaussian / Logistic / StudentT likelihoods on this path";
        return AGP_ERR_UNSUPPORTED;
      }
      if (tk == AGP_LIK_GAUSSIAN && liks[t].p1 > 0) {
        ctx->err = "GaussianLikelihood(...; opt_noise) is not wired as a multi-output task likelihood";
        return AGP_ERR_UNSUPPORTED;
      }
      mocfg.kind[t] = liks[t].kind;
      mocfg.p0[t] = (T)liks[t].p0;
      mocfg.p1[t] = (T)liks[t].p1;
    }
    nT = n_task;
    mocfg.nT = n_task;
    ystride = n_task;  // targets are point-major: y[i * n_task + t]
    a_eta = eta;
    a_b1 = b1;
    a_b2 = b2;
    a_eps = eps;
    a_step = 0;
    if (!A_dev) {
      AGPCHK(dmalloc(ctx, &A_dev, (int64_t)MO_MAXT * Qa()));
      AGPCHK(dmalloc(ctx, &gradA_dev, (int64_t)MO_MAXT * Qa()));
      AGPCHK(dmalloc(ctx, &am_dev, (int64_t)MO_MAXT * Qa()));
      AGPCHK(dmalloc(ctx, &av_dev, (int64_t)MO_MAXT * Qa()));
      if (mo_sharded) AGPCHK(dmalloc(ctx, &fall, (int64_t)2 * qtot * Bp));
      T** bv[] = {&mo_mixm, &mo_mixv, &mo_th, &mo_cc, &mo_th_save};
      for (auto p : bv) AGPCHK(dmalloc(ctx, p, (int64_t)MO_MAXT * Bp));
    }
    std::vector<T> ha((size_t)n_task * Qa());
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = (T)A_host[i];
    HIPCHK(ctx, hipMemcpyAsync(A_dev, ha.data(), sizeof(T) * ha.size(), hipMemcpyHostToDevice, st()));
    HIPCHK(ctx, hipMemsetAsync(am_dev, 0, sizeof(double) * MO_MAXT * Qa(), st()));
    HIPCHK(ctx, hipMemsetAsync(av_dev, 0, sizeof(double) * MO_MAXT * Qa(), st()));
    HIPCHK(ctx, hipMemsetAsync(mo_cc, 0, sizeof(T) * MO_MAXT * Bp, st()));
    // local variables before the first step (init_local_vars): theta = 1/sigma2 (Gaussian) or 0
    for (int t = 0; t < n_task; ++t)
      hipLaunchKernelGGL((k_fill<T>), grid1(Bp), dim3(256), 0, st(), mo_th + (int64_t)t * Bp, Bp,
                         liks[t].kind == AGP_LIK_GAUSSIAN ? (T)(1.0 / liks[t].p0) : T(0));
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(st()));
    mo = true;
    return AGP_OK;
  }

  agp_status get_A(double* A_host) override {
    if (!mo || !A_host) return AGP_ERR_INVALID;
    std::vector<T> ha((size_t)nT * Qa());
    HIPCHK(ctx, hipMemcpyAsync(ha.data(), A_dev, sizeof(T) * ha.size(), hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    for (size_t i = 0; i < ha.size(); ++i) A_host[i] = (double)ha[i];
    return AGP_OK;
  }

  // update_A! then the mixed local updates / gradients (update_parameters!(::MOSVGP), training.jl:153-158)
  // fm / fv: T[Qa][Bp] mean_f / var_f of ALL latents (the handle's own muf / varf, or the exchanged `fall` when sharded:
  // update_A! is then replicated on every rank, the gradients r / w are written for the owned latents only)
  agp_status mo_local(const void* y, const int64_t* idx, int64_t B, double rho, bool update_A, const T* fm,
                      const T* fv) {
    const int Q = Qa();
    if (update_A && a_eta > 0) {
      a_step += 1;
      hipLaunchKernelGGL((k_mo_gradA<T>), dim3((unsigned)Q, (unsigned)nT), dim3(256), 0, st(), B, Q, Bp, mocfg,
                         (const T*)A_dev, (const T*)y, ystride, idx, fm, fv, (const T*)mo_th, gradA_dev);
      hipLaunchKernelGGL((k_mo_applyA<T>), dim3(1), dim3(64), 0, st(), nT, Q, A_dev, (const double*)gradA_dev, am_dev,
                         av_dev, a_step, a_eta, a_b1, a_b2, a_eps);
    }
    hipLaunchKernelGGL((k_mo_local<T>), grid1(B), dim3(256), 0, st(), B, Q, Bp, mocfg, (const T*)A_dev, (T)rho,
                       (const T*)y, ystride, idx, fm, fv, mo_mixm, mo_mixv, mo_th, mo_cc, rbuf, wbuf, 1, qlo, nl);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }

  agp_status mo_shard(int q_total) override {
    if (lp.kind != AGP_LIK_MULTIOUTPUT || mo || q_total < desc.latent_offset + nl || desc.latent_offset < 0) {
      ctx->err = "agp_svgp_mo_shard: call before agp_svgp_set_multioutput with q_total >= latent_offset + n_latent";
      return AGP_ERR_INVALID;
    }
    mo_sharded = true;
    qtot = q_total;
    qlo = desc.latent_offset;
    return AGP_OK;
  }
  // own rows of (mean_f, var_f) into the exchange buffer, zeros elsewhere
  agp_status publish_f(const T* fm, const T* fv, int state) {
    HIPCHK(ctx, hipMemsetAsync(fall, 0, sizeof(T) * 2 * qtot * Bp, st()));
    HIPCHK(ctx, hipMemcpyAsync(fall + (int64_t)qlo * Bp, fm, sizeof(T) * nl * Bp, hipMemcpyDeviceToDevice, st()));
    HIPCHK(ctx, hipMemcpyAsync(fall + (int64_t)(qtot + qlo) * Bp, fv, sizeof(T) * nl * Bp, hipMemcpyDeviceToDevice, st()));
    fall_state = state;
    return AGP_OK;
  }
  agp_status mo_fbuf_ptr(void** p, int64_t* n) override {
    if (!mo_sharded || !fall || !p || !n) return AGP_ERR_INVALID;
    *p = fall;
    *n = (int64_t)2 * qtot * Bp;
    return AGP_OK;
  }
  agp_status mo_mix() override {
    if (!mo_sharded || !mo || fall_state != 1) {
      ctx->err = "agp_svgp_mo_mix: needs a latent-sharded multi-output handle right after agp_svgp_step_local";
      return AGP_ERR_INVALID;
    }
    fall_state = 3;
    return mo_local(y_last, idx_last, B_last, rho_last, true, fall, fall + (int64_t)qtot * Bp);
  }
  agp_status mo_refresh_f() override {
    if (!mo_sharded || !mo || B_last <= 0) return AGP_ERR_INVALID;
    AGPCHK(posterior_f(B_last));
    return publish_f(emuf, evarf, 2);
  }
  // mean_f / var_f of the owned latents on the last batch with the CURRENT posterior -> emuf / evarf (analyticVI.jl:260-266)
  agp_status posterior_f(int64_t B) {
    const int64_t Bq = rup64(B);
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      HIPCHK(ctx, hipMemcpyAsync(g.Wbuf, g.kappa, sizeof(T) * Bq * mp, hipMemcpyDeviceToDevice, st()));
      AGPCHK(aug_factor(g, Bq, 1));
      HIPCHK(ctx, hipMemcpyAsync(g.v, g.Wbuf + Bq * mp, sizeof(T) * mp, hipMemcpyDeviceToDevice, st()));
      hipLaunchKernelGGL((k_w_rowstats<T>), grid1(B * 64), dim3(256), 0, st(), (const T*)g.Wbuf, mp, B, mp,
                         (const T*)g.v, pw0, pw1);
      hipLaunchKernelGGL((k_meanvar_finish<T>), grid1(B), dim3(256), 0, st(), B, 1, (const T*)pw0, (const T*)pw1, ldp,
                         (const T*)(Kt + l * Bp), emuf + l * Bp, evarf + l * Bp);
      LAUNCHCHK(ctx);
    }
    return AGP_OK;
  }

  // Knm / kappa of the NEXT minibatch on a second stream.  The targets are the alternate buffers, last read by the
  // step before the most recently enqueued one, so the prefetch only has to wait for that older step.
  agp_status prefetch(const void* x, int64_t ldx, const int64_t* idx, int64_t B) override {
    AGPCHK(check_batch(B));
    if (!x || !idx || ldx < D) return AGP_ERR_INVALID;
    for (auto& g : lat)
      if (g.K_stale) return AGP_OK;  // nothing sensible to prefetch against
    if (!pf_stream) {
      {
        // one latent: lowest priority -- the look-ahead GEMM must not take CUs from the latency-bound factorisation chain (C2 0.309
        // -> 0.336 ms with the highest).  Several latents: highest -- the batched task graph holds every CU with mostly waiting
        // workgroups for 0.6 ms and nl look-ahead pairs have to get through next to it; a starved look-ahead is what the main stream
        // then waits for (C4, 8 latents: 1.34 -> 1.21 ms).
        int lo = 0, hi = 0;
        HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        int pr = nl > 1 ? hi : lo;
#ifdef AGP_DEV_KNOBS  // (development builds: A/B of the look-ahead's priority, round 6 -- C3 experiment)
        if (const char* e = getenv("AGP_DEV_PF_PRIO")) pr = e[0] == 'h' ? hi : e[0] == 'n' ? 0 : lo;
#endif
        HIPCHK(ctx, hipStreamCreateWithPriority(&pf_stream, hipStreamNonBlocking, pr));
      }
      HIPCHK(ctx, hipEventCreateWithFlags(&pf_done, hipEventDisableTiming));
      for (auto& e : step_done) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      HIPCHK(ctx, hipEventRecord(step_done[0], st()));
      HIPCHK(ctx, hipEventRecord(step_done[1], st()));
      {
        // hand-over word (AGP_PF_INKERNEL=0 keeps the event)
        const char* e = getenv("AGP_PF_INKERNEL");
        sig_state = -1;
        if (!(e && e[0] == '0')) {
          bool ok = true;
          for (auto& q : sig)
            ok = ok && hipExtMallocWithFlags((void**)&q, 8, hipMallocSignalMemory) == hipSuccess && hipMemset(q, 0, 8) == hipSuccess;
          if (ok) {
            // both streams must be able to run kernels at the same time (k_handshake, agp_chol.h)
            int32_t* hs = nullptr;
            ok = hipMalloc((void**)&hs, 4 * sizeof(int32_t)) == hipSuccess && hipMemset(hs, 0, 4 * sizeof(int32_t)) == hipSuccess;
            if (ok) {
              (void)hipStreamSynchronize(st());
              hipLaunchKernelGGL(k_handshake, dim3(1), dim3(64), 0, pf_stream, hs, (const int32_t*)(hs + 1), hs + 2);
              hipLaunchKernelGGL(k_handshake, dim3(1), dim3(64), 0, st(), hs + 1, (const int32_t*)hs, hs + 3);
              int32_t res[4] = {0, 0, 0, 0};
              ok = hipStreamSynchronize(pf_stream) == hipSuccess && hipStreamSynchronize(st()) == hipSuccess &&
                   hipMemcpy(res, hs, sizeof(res), hipMemcpyDeviceToHost) == hipSuccess && res[2] == 1 && res[3] == 1;
            }
            if (hs) (void)hipFree(hs);
          }
          if (ok) sig_state = 1;
          else
            for (auto& q : sig) {
              if (q) (void)hipFree(q);
              q = nullptr;
            }
        }
        (void)hipGetLastError();
      }
      for (auto& g : lat) {
        AGPCHK(dmalloc(ctx, &g.Knm_alt, Bp * mp));
        AGPCHK(dmalloc(ctx, &g.kappa_alt, Bp * mp));
        AGPCHK(dmalloc(ctx, &g.kappa_old, Bp * mp));
        AGPCHK(dmalloc(ctx, &g.Wbuf_alt, (Bp + TILE) * mp));
        AGPCHK(dmalloc(ctx, &g.pk_alt, (2 * mp / TILE) * Bp));
      }
    }
    // the alternate buffers were "current" in the step before the last enqueued one
    if (slot_kind[step_parity] == 1) {
      hipLaunchKernelGGL(k_wait_ge, dim3(1), dim3(64), 0, pf_stream, (const int32_t*)sig[0], slot_seq[step_parity], info_dev);
      LAUNCHCHK(ctx);
    } else
      HIPCHK(ctx, hipStreamWaitEvent(pf_stream, step_done[step_parity], 0));
    const int64_t Bq = rup64(B);
    hipStream_t keep_stream = ctx->stream;
    agp_status rc = AGP_OK;
    // several latents (round 4): their (K_nm, kappa) pairs are independent, and one in-order stream runs them as 2 nl kernels of
    // exactly one workgroup per CU each, next to a task graph that holds most CUs -- every kernel boundary drains.  They are spread
    // over two streams (fork / join by events on the look-ahead's own streams, none on the step's); more streams take too much of
    // the chip from the task graph (measured at C4, 8 latents: 1 stream 1.186, 2: 1.088, 3: 1.41, 4: 1.34 ms).
    constexpr int pf_ways = 2;
    // (only next to the batched task graph: at C5 -- blocked factorisation, GEMMs of 4096 tiles -- two streams cost 13.5 -> 14.5 ms)
    const int ways = (nl > 1 && chol_use_dag(ctx, mp / TILE, Bq / TILE + 1, 2)) ? std::min<int>(pf_ways, nl) : 1;
    if (ways > 1) {
      if (!pf_side[0]) {
        int lo = 0, hi = 0;
        HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        const int pr = hi;
        for (int q = 0; q < PF_SIDE; ++q) {
          HIPCHK(ctx, hipStreamCreateWithPriority(&pf_side[q], hipStreamNonBlocking, pr));
          HIPCHK(ctx, hipEventCreateWithFlags(&pf_join[q], hipEventDisableTiming));
        }
        HIPCHK(ctx, hipEventCreateWithFlags(&pf_fork, hipEventDisableTiming));
      }
      HIPCHK(ctx, hipEventRecord(pf_fork, pf_stream));
      for (int q = 0; q < ways - 1; ++q) HIPCHK(ctx, hipStreamWaitEvent(pf_side[q], pf_fork, 0));
    }
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      hipStream_t sl = (ways > 1 && l % ways) ? pf_side[l % ways - 1] : pf_stream;
      ctx->stream = sl;  // reuse the launch helpers on the look-ahead streams
      dim3 gk((unsigned)(mp / TILE), (unsigned)(Bq / TILE));
      (void)launch_kernelmatrix<T>(ctx, sl, (const T*)x, ldx, idx, B, (const T*)g.Z,
                         D, m, D, (const T*)g.scales, g.k.kind, kvar(g), g.Knm_alt, mp, Bq, mp, 0, T(0),
                         (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)g.Zsc, (const T*)g.zn);
      rc = gemm_nt<T, EPI_KAPPA>(ctx, g.Knm_alt, mp, kinv_kappa(g), mp, Bq, mp, mp, 0, g.kappa_alt, mp, g.Knm_alt, mp, nullptr,
                                 g.pk_alt, g.Wbuf_alt, ldp);
      if (rc != AGP_OK) break;
    }
    ctx->stream = keep_stream;
    AGPCHK(rc);
    for (int q = 0; q < ways - 1; ++q) {
      HIPCHK(ctx, hipEventRecord(pf_join[q], pf_side[q]));
      HIPCHK(ctx, hipStreamWaitEvent(pf_stream, pf_join[q], 0));
    }
    HIPCHK(ctx, hipEventRecord(pf_done, pf_stream));
    if (sig_state == 1) {
      pf_seq += 1;
      hipLaunchKernelGGL(k_set_sig, dim3(1), dim3(1), 0, pf_stream, sig[1], pf_seq);
      LAUNCHCHK(ctx);
    }
    pf_valid = true;
    pf_x = x;
    pf_idx = idx;
    pf_B = B;
    pf_ldx = ldx;
    return AGP_OK;
  }

  agp_status lsm_gamma() override {
    if (lp.kind != AGP_LIK_LOGISTICSOFTMAX) return AGP_OK;
    hipLaunchKernelGGL((k_lsm_gamma<T>), grid1(B_last), dim3(256), 0, st(), B_last, nl, Bp, (const T*)muf,
                       (const T*)cbuf, (const T*)alpha, (const T*)beta, gamma, gsum);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  agp_status lsm_alpha() override {
    if (lp.kind != AGP_LIK_LOGISTICSOFTMAX) return AGP_OK;
    hipLaunchKernelGGL((k_lsm_alpha<T>), grid1(B_last), dim3(256), 0, st(), B_last, (const T*)gsum, alpha);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  agp_status lsm_gsum_ptr(void** p, int64_t* n) override {
    *p = gsum;
    *n = B_last;
    return AGP_OK;
  }
  // k_elbo_terms is one instantiation per likelihood (no scratch under the 1024-thread register budget)
  agp_status launch_elbo_terms(int64_t B, int nlat, const LikParams<T>& lk, int lat_off, int add_global, const T* yv,
                               const int32_t* ycls, const int64_t* idx, const T* mf, const T* vf, const T* cv, const T* thv,
                               const T* gv, const T* av, const T* bv, double* outp, int64_t ystr) {
#define AGP_ELBO_CASE(K)                                                                                                        \
  case K:                                                                                                                       \
    hipLaunchKernelGGL((k_elbo_terms<T, K>), dim3(1), dim3(1024), 0, st(), B, nlat, Bp, lk, desc.elbo_mode, lat_off, add_global, \
                       yv, ycls, idx, mf, vf, cv, thv, gv, av, bv, outp, ystr, (const T*)lam_dev, (int)shard_once);              \
    break;
    switch (lk.kind) {
      AGP_ELBO_CASE(LIK_GAUSSIAN)
      AGP_ELBO_CASE(LIK_LOGISTIC)
      AGP_ELBO_CASE(LIK_STUDENTT)
      AGP_ELBO_CASE(LIK_LSM)
      AGP_ELBO_CASE(LIK_LAPLACE)
      AGP_ELBO_CASE(LIK_BSVM)
      AGP_ELBO_CASE(LIK_POISSON)
      AGP_ELBO_CASE(LIK_NEGBIN)
      AGP_ELBO_CASE(LIK_HETERO)
      default:
        ctx->err = "ELBO data terms: unknown likelihood kind";
        return AGP_ERR_INVALID;
    }
#undef AGP_ELBO_CASE
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  bool lsm_finished = false;  // k_lsm_fused has written theta, r, w of this step already
  agp_status lsm_local_all() override {
    if (lp.kind != AGP_LIK_LOGISTICSOFTMAX) return AGP_OK;
    if (nl > LSM_FUSED_MAXL) {
      for (int it = 0; it < 2; ++it) {  // logisticsoftmax.jl:65
        AGPCHK(lsm_gamma());
        AGPCHK(lsm_alpha());
      }
      return AGP_OK;
    }
    // one lane per (point, latent), LG = 2^ceil(log2 nl) lanes per point, 64 threads per workgroup: the 8 x 1024 values of a C4
    // update spread over 128 workgroups (the kernel is latency, not bandwidth)
#define AGP_LSM_LAUNCH(LG)                                                                                                   \
  hipLaunchKernelGGL((k_lsm_fused<T, LG>), dim3((unsigned)((B_last * LG + 63) / 64)), dim3(64), 0, st(), B_last, nl, Bp,      \
                     desc.latent_offset, (T)rho_last, (const int32_t*)y_last, idx_last, (const T*)muf, (const T*)cbuf, alpha, \
                     (const T*)beta, gamma, gsum, theta, rbuf, wbuf, (int)desc.lik.n_class, flags_dev)
    if (nl <= 1) AGP_LSM_LAUNCH(1);
    else if (nl <= 2) AGP_LSM_LAUNCH(2);
    else if (nl <= 4) AGP_LSM_LAUNCH(4);
    else if (nl <= 8) AGP_LSM_LAUNCH(8);
    else AGP_LSM_LAUNCH(16);
#undef AGP_LSM_LAUNCH
    LAUNCHCHK(ctx);
    lsm_finished = true;
    return AGP_OK;
  }
  agp_status lsm_finish() {
    if (lp.kind != AGP_LIK_LOGISTICSOFTMAX) return AGP_OK;
    if (lsm_finished) {
      lsm_finished = false;
      return AGP_OK;
    }
    hipLaunchKernelGGL((k_lsm_finish<T>), grid1(B_last), dim3(256), 0, st(), B_last, nl, Bp, desc.latent_offset,
                       (T)rho_last, (const int32_t*)y_last, idx_last, (const T*)cbuf, (const T*)gamma, theta, rbuf,
                       wbuf, (int)desc.lik.n_class, flags_dev);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }

  double* logdetK_dev = nullptr;  // [nl] log det of chol(K) diagonals (half log det K), fetched lazily (resolve_logdet)
  bool refresh_lazy = false;      // set around refresh_K() by the training loop's step_local
  agp_status resolve_logdet(Latent& g) {
    if (!g.logdet_pending) return AGP_OK;
    double hl = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hl, logdetK_dev + (&g - lat.data()), sizeof(double), hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    g.half_logdetK = hl;
    g.logdet_pending = false;
    return AGP_OK;
  }
  bool lam_deferred = false;  // set around step_local by the batch-sharded driver (see cavi_step_multi)
  agp_status lambda_finish_reduced() {
    if (lp.kind == AGP_LIK_GAUSSIAN) {  // the reduced sum and batch size: the same ADAM step on every rank
      const int nb = (int)((B_last + 255) / 256);
      hipLaunchKernelGGL((k_noise_finish<T>), dim3(1), dim3(256), 0, st(), 1, (const double*)(scal_dev + 60), 0.0,
                         (const double*)(scal_dev + 62), noise_eta, 0.9, 0.999, 1e-8, noise_adam, lam_dev);
      hipLaunchKernelGGL((k_gauss_grads<T>), dim3(nb), dim3(256), 0, st(), B_last, (T)rho_last, (const T*)y_last, idx_last,
                         (const T*)lam_dev, theta, cbuf, rbuf, wbuf);
      LAUNCHCHK(ctx);
      return AGP_OK;
    }
    const int mode = lp.kind == AGP_LIK_POISSON ? 0 : 1;
    hipLaunchKernelGGL((k_lambda_finish_red<T>), dim3(1), dim3(64), 0, st(), (const double*)(scal_dev + 60), mode, lam_dev);
    if (lp.kind == AGP_LIK_HETEROSCEDASTIC) {
      const int nb = (int)((B_last + 255) / 256);
      hipLaunchKernelGGL((k_hetero_grads<T>), dim3(nb), dim3(256), 0, st(), B_last, Bp, (T)rho_last, (const T*)y_last, idx_last,
                         (const T*)lam_dev, (const T*)gamma, theta, rbuf, wbuf);
    }
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  double cur_lr() const {  // optimisers.jl:14-19 ; Descent(1.0) for AnalyticVI
    return desc.stochastic ? 1.0 / std::pow(desc.rm_tau + (double)n_opt, desc.rm_kappa) : 1.0;
  }

  // batch statistics ; fused: eta1/eta2 are stepped directly, otherwise stats = [t | S] for an all-reduce
  agp_status step_stats(bool fused) override {
    AGPCHK(lsm_finish());
    const int64_t Bq = rup64(B_last);
    const T lr = (T)cur_lr();
    if (fused && nl > 1 && nl <= SYRK_MAXB) {
      // all latents of the handle in ONE launch (blockIdx.y = latent): a single latent's tiles fill about half the chip
      SyrkBatch<T> b{};
      for (int l = 0; l < nl; ++l) {
        Latent& g = lat[l];
        b.A[l] = g.kappa;
        b.w[l] = wbuf + l * Bp;
        b.out[l] = g.La;
        b.eta2[l] = g.eta2;
        b.Kinv[l] = kinv_step(g);
        b.rvec[l] = rbuf + l * Bp;
        b.eta1[l] = g.eta1;
        b.kinv_mu0[l] = kinv_mu0_step(g);
      }
      const int64_t nt = mp / TILE, tiles = nt * (nt + 1) / 2, nrider = nt;
      T* fillp = nullptr;
      int64_t fused_used = 0, fstride = 0, nfill = 0;
      int fnb = 0;
      if (ctx->h_dirty[0].on && ctx->htype == (int)sizeof(T)) {  // hand-over refill riders, as in syrk_tn()
        fillp = (T*)ctx->hset[0];
        fused_used = ctx->h_dirty[0].used;
        fstride = ctx->h_dirty[0].stride;
        fnb = ctx->h_dirty[0].nb;
        nfill = 96;
        ctx->h_dirty[0].on = false;
      }
      const bool kg2 = tiles * nl <= kg2_limit() && Bq >= 4 * BK;
      dim3 grid((unsigned)(tiles + nrider + nfill), (unsigned)nl);
      if (kg2)
        hipLaunchKernelGGL((k_syrk_eta_batch<T, 2>), grid, dim3(2 * NTHREADS), 0, st(), b, mp, Bq, mp, mp, lr, tiles, nrider, fillp,
                           fused_used, fstride, fnb);
      else
        hipLaunchKernelGGL((k_syrk_eta_batch<T, 1>), grid, dim3(NTHREADS), 0, st(), b, mp, Bq, mp, mp, lr, tiles, nrider, fillp,
                           fused_used, fstride, fnb);
      LAUNCHCHK(ctx);
      return AGP_OK;
    }
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      T* sl = stats + l * stats_stride();
      if (fused) {  // one launch: eta2 step on the lower tiles, eta1 step on nt rider workgroups
        AGPCHK((syrk_tn<T, SY_ETA2>(ctx, g.kappa, mp, mp, Bq, wbuf + l * Bp, 0, g.La, mp, g.eta2,
                                    kinv_step(g), mp, lr, (const T*)(rbuf + l * Bp), g.eta1, kinv_mu0_step(g))));
      } else {
        // ONE launch: the packed lower tiles of kappa' diag(w) kappa, t = kappa' (rho g1) by rider workgroups (the same riders
        // as the fused step, so a one-rank run sums in the same order), and the hand-over refill in its shadow
        AGPCHK((syrk_tn<T, SY_PACK>(ctx, g.kappa, mp, mp, Bq, wbuf + l * Bp, 0, sl + mp, mp, (T*)nullptr,
                                    (const T*)nullptr, (int64_t)0, T(0), (const T*)(rbuf + l * Bp), sl, (const T*)nullptr)));
      }
      LAUNCHCHK(ctx);
    }
    if (!fused) AGPCHK(kappa_released());
    return AGP_OK;
  }
  agp_status stats_ptr(void** p, int64_t* n) override {
    *p = stats;
    *n = (int64_t)nl * stats_stride();
    return AGP_OK;
  }
#ifdef AGP_DEBUG_PTRS
  void* debug_ptr(int what) override {
    Latent& g = lat[0];
    switch (what) {
      case 0: return g.eta2;
      case 1: return g.eta1;
      case 2: return wbuf;
      case 3: return rbuf;
      case 4: return wbuf2;
      case 5: return rbuf2;
      case 6: return g.kappa;
      case 7: return g.Wbuf;
      case 8: return theta;
      default: return nullptr;
    }
  }
#endif

  // AGP_HYPER_GK_FUSED=0: the hyper-gradient forms G_K from kappa' H and Apred as before round 4 (A/B, tests)
  static bool gk_fused_on() {
    static const bool on = [] {
      const char* e = getenv("AGP_HYPER_GK_FUSED");
      return !(e && e[0] == '0');
    }();
    return on;
  }
  // Augmented Cholesky of -2*eta2 with the extension rows [kappa (Bq rows, already in Wbuf) ; eta1'] :
  //   Wbuf <- [kappa L^-T ; (L^-1 eta1)']   i.e. W and v of mean_f = W v, var_f = rowsum(W^2) + K~.
  // with_x additionally forms Xa = L^-1 (needed only for Sigma / mu export, ELBO and prediction).
  // tail (materialize): the caller's next launch carries the task graph's fallback itself (k_safe_symv_trmv) -- filled when it has to
  struct SafeTail {
    bool on = false;
    CholBatch<T> bt{};
    SafeSrc<T> src{};
    int64_t ne = 0, nt = 0;
  };
  agp_status aug_factor(Latent& g, int64_t Bq, int with_x, SafeTail* tail = nullptr) {
    // a pending natural-gradient step (only the hyper step's wrapper leaves one: every other entry point has flushed) rides on this
    // launch as its prologue -- the hyper-parameter iteration's "eta step, then factor the new -2 eta2 with its inverse" in ONE launch
    AGPCHK(run_deferred_safe());
    if (pendp.on) AGPCHK(flush());
    g.C_valid = false;
    ProHost<T> ph{};
    bool use_pro = false;
    if (pend.on) {
      use_pro = nl == 1 && pro_allowed() && chol_use_dag(ctx, mp / TILE, Bq / TILE + 1) &&
                pend.Bq >= TILE && mp / TILE > 1;
      if (!use_pro) AGPCHK(flush());
    }
    if (use_pro) {
      ph.kap = pend.kap;
      ph.ldk = mp;
      ph.Kdim = pend.Bq;
      ph.w = pend.w;
      ph.r = pend.r;
      ph.eta2 = g.eta2;
      ph.Kinv = pend.Kinv;
      ph.ldm = mp;
      ph.eta1 = g.eta1;
      ph.kinv_mu0 = pend.kinv_mu0;
      ph.lr = pend.lr;
      // the hyper-parameter iteration's launch (the inverse rides along): leave C = S + K^-1 / 4 for the gradient's G_K (hypergrad)
      if (with_x && (hy_k || hy_z) && gk_fused_on() && !g.stale_on && !g.on && bs_world == 1 && !mo &&
          lp.kind != AGP_LIK_HETEROSCEDASTIC) {
        if (!g.Cmat) AGPCHK(dmalloc(ctx, &g.Cmat, mp * mp));
        ph.Cout = g.Cmat;
      }
    } else if (g.la_state != 0) {  // La holds a factor: rebuild -2*eta2
      hipLaunchKernelGGL((k_copy2d<T>), grid2(mp, mp), blk2, 0, st(), (const T*)g.eta2, mp, mp, mp, g.La, mp, mp, mp,
                         T(1), T(-2));
      LAUNCHCHK(ctx);
    }
    AGPCHK(timing_begin());
    SafeSrc<T> src{};
    src.Bq = Bq;
    src.kappa[0] = g.kappa;  // every caller copies kappa into Wbuf first (or passes Bq = 0)
    src.eta1[0] = g.eta1;
    src.eta2[0] = g.eta2;
    // with_x: the inverse rides along (identity block rows of the task graph); its fallback forms X = L^-1 in the stream as well
    // (SafeSrc::want_x), so that no host check -- no stream synchronisation -- sits behind the launch (round 3: the hyper-parameter
    // iteration used to wait here once per step)
    src.want_x = with_x ? 1 : 0;
    // with_x: Sigma = Xa' Xa is what every caller forms next -- product workgroups at the end of the same task-graph launch do it
    bool sigma_done = false;
    bool defer = tail != nullptr;
    AGPCHK(potrf_fused<T>(ctx, g.La, mp, mp, g.Xa, mp, g.DgA, g.Wbuf, mp, Bq / TILE + 1, with_x, info_dev, m,
                          (const T*)g.eta1, false, &src, tail ? &defer : nullptr, nullptr, use_pro ? &ph : nullptr,
                          (const EpiArgs<T>*)nullptr, with_x ? g.Sigma : (T*)nullptr, &sigma_done));
    if (tail) {
      tail->on = defer;
      if (defer) {
        tail->bt = CholBatch<T>{};
        tail->bt.A[0] = g.La;
        tail->bt.X[0] = g.Xa;
        tail->bt.Dg[0] = g.DgA;
        tail->bt.E[0] = g.Wbuf;
        tail->src = src;
        tail->ne = Bq / TILE + 1;
        tail->nt = mp / TILE;
      }
    }
    if (use_pro) {  // taken by the launch (a refused launch leaves it pending for flush())
      pend.on = false;
      n_prologue += 1;
      if (ph.Cout) {
        g.C_valid = true;
        g.C_kap = ph.kap;
      }
    }
    AGPCHK(timing_end(chol_use_dag(ctx, mp / TILE, Bq / TILE + 1) ? 1 : chol_launch_count(mp / TILE, Bq / TILE + 1)));
    g.la_state = 1;
    g.xa_valid = with_x != 0;
    if (with_x) g.xa_epoch += 1;
    g.sigma_epoch = sigma_done ? g.xa_epoch : -1;
    return AGP_OK;
  }

  // after eta changed outside a step (set_state): La = -2*eta2 is (re)built lazily; validate it is SPD now
  agp_status refactor(Latent& g) {
    g.la_state = 1;  // force the rebuild from eta2
    g.post_valid = false;
    g.pred_valid = g.predvar_valid = false;
    AGPCHK(aug_factor(g, 0, 1));
    return AGP_OK;
  }

  agp_status step_global(bool fused) override {
    const T lr = (T)cur_lr();
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      if (!fused) {
        const T* sl = stats + l * stats_stride();
        const int64_t ntri = (mp / TILE) * (mp / TILE + 1) / 2;
        hipLaunchKernelGGL((k_eta2_from_packed<T>), dim3((unsigned)(4 * ntri + (mp + 255) / 256)), dim3(256), 0, st(), sl + mp, mp,
                           g.eta2, kinv_step(g), g.La, lr, ntri, sl, kinv_mu0_step(g), g.eta1, mp);
        LAUNCHCHK(ctx);
      }
      g.la_state = 0;  // La now holds the new -2*eta2 (unfactored); it is factored inside the next local phase
      g.xa_valid = false;
      g.C_valid = false;
      g.post_valid = false;
      g.pred_valid = g.predvar_valid = false;
    }
    n_opt += 1;
    if (fused) AGPCHK(kappa_released());
    return AGP_OK;
  }
  // marks the last use of this step's kappa buffers (they become the next prefetch target): after the fused step, or -- phase-split
  // path -- right after the packed statistics, so that the next look-ahead does not wait for the all-reduce and the eta step
  agp_status kappa_released() {
    if (pf_stream) {
      if (rel_pending) {  // (two releases without a step in between: cannot happen through the ABI's step sequence, but be safe)
        slot_kind[rel_slot] = 0;
        HIPCHK(ctx, hipEventRecord(step_done[rel_slot], st()));
        rel_pending = false;
      }
      step_parity ^= 1;
      if (sig_state == 1 && nl == 1) {  // decided when the next step is enqueued (step_local)
        rel_pending = true;
        rel_slot = step_parity ^ 1;
      } else {
        // (the step's deferred fallback, if it re-runs, rewrites the step's Wbuf and reads its pk: before the event releases them)
        AGPCHK(run_deferred_safe());
        slot_kind[step_parity ^ 1] = 0;
        HIPCHK(ctx, hipEventRecord(step_done[step_parity ^ 1], st()));
      }
    }
    return AGP_OK;
  }

  agp_status check_status() override {
    int32_t words[4] = {0, 0, 0, 0};  // info | infoK | flags | orderK (k_logdiag_sum: did K_ZZ fail before or after the others?)
    HIPCHK(ctx, hipMemcpyAsync(words, info_dev, sizeof(words), hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    const int32_t info = words[0], infoK = words[1], orderK = words[3];
    const int flags = (int)words[2];
    dag_retry_check(ctx);  // steps the in-stream fallback had to re-run: warn once, stop using the task graph
    if (info != 0 || flags != 0) {
      HIPCHK(ctx, hipMemsetAsync(info_dev, 0, sizeof(int32_t), st()));
      HIPCHK(ctx, hipMemsetAsync(flags_dev, 0, sizeof(int), st()));
    }
    if (infoK != 0) HIPCHK(ctx, hipMemsetAsync(infoK_dev, 0, sizeof(int32_t), st()));
    if (orderK != 0) HIPCHK(ctx, hipMemsetAsync(info_dev + 3, 0, sizeof(int32_t), st()));
    // a non-SPD K_ZZ latched by a refresh inside the training loop is the ROOT cause of whatever followed it on a garbage inverse
    // (negative K~, NaNs, a non-SPD -2 eta2): it is reported first, like the PosDefException the reference raises at that refresh --
    // unless an earlier step had ALREADY latched a failure of its own when K_ZZ failed (orderK == 2: e.g. the stale-K quirk drives
    // K~ negative, the NaNs that follow reach the kernel parameters, and K_ZZ is the consequence), which then comes first below
    if (infoK > 0 && orderK != 2) {
      for (auto& q : lat) q.K_stale = true;
      ctx->err = "PosDefException: K_ZZ + jitter*I is not positive definite; leading minor " + std::to_string(infoK);
      return AGP_ERR_NOT_POSDEF;
    }
    if (flags & FLAG_NEG_KTILDE) {
      ctx->err = "K~ has negative values";  // latentgp.jl:213
      return AGP_ERR_NEG_KTILDE;
    }
    if (flags & FLAG_BAD_LABEL) {
      ctx->err = "class label outside the likelihood's classes";  // multiclass.jl:81-83
      return AGP_ERR_LABELS;
    }
    if (info == -2) {
      ctx->err = "the look-ahead stream waited about a minute for a CAVI step that never started (k_wait_ge)";
      return AGP_ERR_HIP;
    }
    if (info == -3) {  // (grid_barrier, agp_chol.h: a bounded wait since round 6 -- this used to be a hang)
      if (ctx->safe_bar) HIPCHK(ctx, hipMemsetAsync(ctx->safe_bar, 0, 2 * sizeof(unsigned), st()));
      dag_pause(ctx);
      ctx->err = "the in-stream fallback of an aborted task-graph launch could not complete: one of its workgroups did not become "
                 "resident within the grid barrier's limit (is another process holding compute units of this GPU?); the step's "
                 "results are not valid";
      return AGP_ERR_HIP;
    }
    if (info == -4) {
      ctx->err = "AGP_SPLIT_OVERLAP: a column group of the all-reduced statistics did not arrive within the gate's limit";
      return AGP_ERR_HIP;
    }
    if (info < 0) {
      ctx->err = "task-graph factorisation aborted: a tile dependency never arrived (spin limit)";
      return AGP_ERR_HIP;
    }
    if (info != 0) {
      ctx->err = "PosDefException: -2*eta2 is not positive definite; leading minor " + std::to_string(info);
      return AGP_ERR_NOT_POSDEF;
    }
    if (infoK > 0) {
      for (auto& q : lat) q.K_stale = true;
      ctx->err = "PosDefException: K_ZZ + jitter*I is not positive definite; leading minor " + std::to_string(infoK);
      return AGP_ERR_NOT_POSDEF;
    }
    if (infoK < 0) {
      ctx->err = "task-graph factorisation of K_ZZ aborted: a tile dependency never arrived (spin limit)";
      return AGP_ERR_HIP;
    }
    return AGP_OK;
  }

  // Sigma = Xa' Xa ; mu = Xa' v   with Xa = chol(-2 eta2)^-1, v = Xa eta1     (inference.jl:25-28)
  agp_status materialize(Latent& g) {
    if (g.post_valid) return AGP_OK;
    SafeTail tail;
    if (!(g.la_state == 1 && g.xa_valid)) AGPCHK(aug_factor(g, 0, 1, &tail));
    if (tw2_kis_of == (int)(&g - lat.data())) tw2_kis_of = -1;  // K^-1 Sigma in the scratch belongs to the Sigma before this one
    if (g.sigma_epoch != g.xa_epoch) {  // (else: left by the launch itself)
      if (tail.on) {
        AGPCHK(launch_chol_safe<T>(ctx, tail.bt, tail.src, 1, mp, mp, mp, tail.ne, tail.nt, info_dev, m));
        tail.on = false;
      }
      AGPCHK(xtx_padded<T>(ctx, g.Xa, mp, mp, g.Sigma, mp));
    }
    // mu = Sigma eta1 as the reference writes it (global_update!, analyticVI.jl:229-246).  Until round 4 this was Xa' (Xa eta1) from
    // the factorisation's [eta1'] row: a copy and a 16-workgroup triangular mat-vec (4.8 + 15.6 us at m = 1024 against 7.0)
    if (tail.on) {  // the task graph's fallback rides on this launch
      AGPCHK(ensure_safe_words<T>(ctx));
      const int64_t most = std::max<int64_t>(tail.nt + tail.ne + tail.nt * (tail.nt + 1) / 2 + tail.ne * tail.nt, (2 * mp + 7) / 8);
      const unsigned g1 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(safe_grid_cap(ctx), most));
      hipLaunchKernelGGL((k_safe_symv_trmv<T>), dim3(g1), dim3(CHOL_THREADS), 0, st(), tail.bt, tail.src, mp, mp, mp, tail.ne, tail.nt,
                         info_dev, m, ctx->safe_bar, ctx->safe_retries, (const T*)g.Sigma, (const T*)g.Xa, mp, mp, (const T*)g.eta1,
                         g.mu, g.v);
    } else
      hipLaunchKernelGGL((k_symv_trmv<T>), grid1(2 * mp * 64), dim3(256), 0, st(), (const T*)g.Sigma, (const T*)g.Xa, mp, mp,
                         (const T*)g.eta1, g.mu, g.v);
    LAUNCHCHK(ctx);
    g.v_epoch = g.xa_epoch;
    g.post_valid = true;
    return AGP_OK;
  }

  agp_status elbo(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho, int fresh,
                  double* out) override {
    AGPCHK(check_batch(B));
    const bool lsm = lp.kind == AGP_LIK_LOGISTICSOFTMAX;
    const T* mf;
    const T* vf;
    if (mo_sharded) {
      if (fresh || fall_state != 2) {
        ctx->err = "latent-sharded multi-output ELBO: fresh_local = 0 only, after agp_svgp_mo_refresh_f + the all-reduce "
                   "of the exchange buffer";
        return fresh ? AGP_ERR_UNSUPPORTED : AGP_ERR_INVALID;
      }
      if (B != B_last) return AGP_ERR_INVALID;
      mf = fall;
      vf = fall + (int64_t)qtot * Bp;
    } else if (fresh) {
      // ELBO.jl:32-47 : recompute kernel matrices on (x, y), fresh local variables, one local update
      if (lsm) {
        HIPCHK(ctx, hipMemcpyAsync(alpha_save, alpha, sizeof(T) * Bp, hipMemcpyDeviceToDevice, st()));
        hipLaunchKernelGGL((k_fill<T>), grid1(Bp), dim3(256), 0, st(), alpha, Bp, (T)desc.lik.n_class);
      }
      if (mo) HIPCHK(ctx, hipMemcpyAsync(mo_th_save, mo_th, sizeof(T) * MO_MAXT * Bp, hipMemcpyDeviceToDevice, st()));
      // the Gaussian KL below needs Sigma, mu anyway: factor -2*eta2 WITH its inverse first, so that the local step on the
      // evaluation batch (and the next training step) take W, v from the inverse instead of two more factorisation chains
      for (auto& g : lat) AGPCHK(materialize(g));
      eval_cache_ok = true;  // (only this caller: the online model's fresh local steps see new data behind the same pointers)
      const agp_status sl = step_local(x, ldx, y, idx, B, rho, true);
      eval_cache_ok = false;
      AGPCHK(sl);
      if (lsm) {
        for (int it = 0; it < 2; ++it) {
          AGPCHK(lsm_gamma());
          AGPCHK(lsm_alpha());
        }
        AGPCHK(lsm_finish());
      }
      for (auto& g : lat) g.kappa_valid = false;  // (the training step's kappa is gone; the evaluation batch's stays: kappa_eval)
      mf = muf;
      vf = varf;
    } else {
      if (B != B_last) {
        ctx->err = "agp_svgp_elbo(fresh_local=0) must be called with the batch of the last cavi_step";
        return AGP_ERR_INVALID;
      }
      // mean_f / var_f with the UPDATED posterior, local variables from the step (analyticVI.jl:260-266)
      AGPCHK(posterior_f(B));
      mf = emuf;
      vf = evarf;
    }
    if (mo) {
      // multi-output ELBO (analyticVI.jl:277-297): per-task terms on the A-mixed mean_f / var_f
      if (!fresh)
        hipLaunchKernelGGL((k_mo_local<T>), grid1(B), dim3(256), 0, st(), B, Qa(), Bp, mocfg, (const T*)A_dev, (T)rho,
                           (const T*)y, ystride, idx, mf, vf, mo_mixm, mo_mixv, mo_th, mo_cc, rbuf, wbuf, 0, qlo, nl);
      for (int t = 0; t < nT; ++t) {
        LikParams<T> lt{mocfg.kind[t], mocfg.p0[t], mocfg.p1[t]};
        AGPCHK(launch_elbo_terms(B, 1, lt, 0, 0, (const T*)y + t, (const int32_t*)nullptr, idx,
                                 (const T*)(mo_mixm + (int64_t)t * Bp), (const T*)(mo_mixv + (int64_t)t * Bp),
                                 (const T*)(mo_cc + (int64_t)t * Bp), (const T*)(mo_th + (int64_t)t * Bp), (const T*)nullptr,
                                 (const T*)nullptr, (const T*)nullptr, scal_dev + 8 + 2 * t, (int64_t)nT));
      }
      LAUNCHCHK(ctx);
      std::vector<double> ht(2 * nT);
      HIPCHK(ctx, hipMemcpyAsync(ht.data(), scal_dev + 8, sizeof(double) * 2 * nT, hipMemcpyDeviceToHost, st()));
      HIPCHK(ctx, hipStreamSynchronize(st()));
      mo_e = mo_kl = 0.0;
      for (int t = 0; t < nT; ++t) {
        mo_e += ht[2 * t];
        mo_kl += ht[2 * t + 1];
      }
      if (fresh) HIPCHK(ctx, hipMemcpyAsync(mo_th, mo_th_save, sizeof(T) * MO_MAXT * Bp, hipMemcpyDeviceToDevice, st()));
    } else {
      AGPCHK(launch_elbo_terms(B, nl, lp, desc.latent_offset, (int)(desc.latent_offset == 0), (const T*)y, (const int32_t*)y, idx,
                               (const T*)mf, (const T*)vf, (const T*)cbuf, (const T*)theta, (const T*)gamma, (const T*)alpha,
                               (const T*)beta, scal_dev, (int64_t)1));
    }
    if (fresh && lsm)
      HIPCHK(ctx, hipMemcpyAsync(alpha, alpha_save, sizeof(T) * Bp, hipMemcpyDeviceToDevice, st()));
    // GaussianKL per latent (KLdivergences.jl:11-18)
    double kl_gauss = 0.0;
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      AGPCHK(materialize(g));
      hipLaunchKernelGGL((k_logdiag_sum<T>), dim3(1), dim3(1024), 0, st(), (const T*)g.DgA, m, scal_dev + 2);
      // objective(model, state, y) reads the state's kernel matrices -- the frozen ones under AGP_FLAG_STALE_K; the external
      // ELBO(model, X, y) recomputes them (ELBO.jl:32-47)
      const bool use_stale = g.stale_on && !fresh;
      AGPCHK(frob_dot((const T*)(use_stale ? g.sKinv : g.Kinv), (const T*)g.Sigma, mp, m, scal_dev + 3));
      hipLaunchKernelGGL((k_axpby<T>), grid1(mp), dim3(256), 0, st(), mp, T(1), (const T*)g.mu, T(-1), (const T*)g.mu0,
                         tmpv);
      hipLaunchKernelGGL((k_trmv_lower<T>), grid1(mp * 64), dim3(256), 0, st(), (const T*)(use_stale ? g.sXk : g.Xk), mp, mp,
                         (const T*)tmpv, pw0);
      hipLaunchKernelGGL((k_sumsq<T>), dim3(1), dim3(1024), 0, st(), (const T*)pw0, mp, scal_dev + 4);
      LAUNCHCHK(ctx);
      if (elbo_async >= 0 && nl == 1 && !mo && !g.on && !use_stale && logdetK_dev) {
        // agp_svgp_elbo_enqueue: the scalar is put together on the device and lands in mapped host memory behind an event
        hipLaunchKernelGGL(k_elbo_combine, dim3(1), dim3(1), 0, st(), (const double*)scal_dev, (const double*)logdetK_dev, (double)m,
                           rho, elbo_pin + elbo_async);
        LAUNCHCHK(ctx);
        HIPCHK(ctx, hipEventRecord(elbo_ev[elbo_async], st()));
        elbo_async_used = true;
        *out = 0.0;
        return AGP_OK;
      }
      double h[5];
      HIPCHK(ctx, hipMemcpyAsync(h, scal_dev, sizeof(double) * 5, hipMemcpyDeviceToHost, st()));
      HIPCHK(ctx, hipStreamSynchronize(st()));
      if (l == 0) {
        e_data = mo ? mo_e : h[0];
        kl_aug = mo ? mo_kl : h[1];
      }
      AGPCHK(resolve_logdet(g));
      const double logdetK = 2.0 * (use_stale ? g.s_half_logdetK : g.half_logdetK), logdetS = -2.0 * h[2];
      kl_gauss += 0.5 * (logdetK - logdetS + h[3] + h[4] - (double)m);
      if (g.on) {
        double ek = 0.0;
        AGPCHK(extra_kl(g, &ek));
        kl_gauss += ek;
      }
    }
    // sharded multi-output: the (replicated) data terms are counted by the rank that owns latent 0, the Gaussian KLs by their
    // owners; the driver sums the scalars over ranks
    if (mo_sharded && qlo != 0) e_data = kl_aug = 0.0;
    kl_gauss_last = kl_gauss;
    *out = rho * e_data - kl_gauss - rho * kl_aug;
    return AGP_OK;
  }
  double e_data = 0, kl_aug = 0, mo_e = 0, mo_kl = 0, kl_gauss_last = 0;
  bool shard_once = true;

  // ---- ELBO without a host round trip (round 4): agp_svgp_elbo_enqueue / agp_svgp_elbo_fetch --------------------------------------
  // Convergence monitoring evaluates the ELBO every few iterations; the synchronous call costs a stream synchronisation and a
  // read-back each time (C2: ~1 ms of idle GPU per check, a fifth of the time to the ELBO plateau).  enqueue() puts the same
  // evaluation into the stream and returns a ticket; the value arrives in mapped host memory behind an event, and fetch() reads it
  // when the caller wants it (wait = 0: only if it is there).  Models whose ELBO needs host-side pieces (several latents,
  // multi-output, streaming prior, the frozen matrices of AGP_FLAG_STALE_K) are evaluated synchronously inside enqueue().
  static constexpr int ELBO_RING = 8;
  double* elbo_pin = nullptr;
  hipEvent_t elbo_ev[ELBO_RING] = {};
  bool elbo_ready[ELBO_RING] = {};   // the value was computed synchronously (elbo_val)
  bool elbo_open[ELBO_RING] = {};
  double elbo_val[ELBO_RING] = {};
  int elbo_next = 0, elbo_async = -1;
  bool elbo_async_used = false;
  agp_status elbo_enqueue(const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho, int fresh,
                          int32_t* ticket) override {
    if (!ticket) return AGP_ERR_INVALID;
    if (!elbo_pin) {
      HIPCHK(ctx, hipHostMalloc((void**)&elbo_pin, sizeof(double) * ELBO_RING, hipHostMallocMapped));
      for (auto& e : elbo_ev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // any closed slot will do (tickets may be fetched out of order), searched from the one after the last ticket handed out
    int slot = -1;
    for (int q = 0; q < ELBO_RING && slot < 0; ++q)
      if (!elbo_open[(elbo_next + q) % ELBO_RING]) slot = (elbo_next + q) % ELBO_RING;
    if (slot < 0) {
      ctx->err = "agp_svgp_elbo_enqueue: 8 evaluations in flight, none fetched (agp_svgp_elbo_fetch closes a ticket)";
      return AGP_ERR_INVALID;
    }
    // a batch-sharded handle's ELBO is a sum over the ranks (agp_svgp_elbo_multi): its local terms are what agp_svgp_elbo_terms
    // reports after a SYNCHRONOUS evaluation -- the enqueued form would leave them stale
    const bool force_sync = bs_world > 1;
    elbo_async = force_sync ? -1 : slot;
    elbo_async_used = false;
    double v = 0.0;
    const agp_status st_ = elbo(x, ldx, y, idx, B, rho, fresh, &v);
    elbo_async = -1;
    AGPCHK(st_);
    if (elbo_async_used && fresh && pf_stream) {
      // The evaluation's local step ran on the CURRENT kernel-matrix buffers (Knm, Wbuf, K~ slices) after its step_local had
      // released them to the look-ahead (a fresh step between two CAVI steps releases the previous step's buffers where it starts).
      // With the synchronous call the host waits for the evaluation before it enqueues anything else; here the look-ahead after
      // next could overwrite them while the evaluation still reads them (seen as ELBO values off by tens of per cent): the release
      // it will wait for is recorded again, behind the evaluation.
      const int sl = step_parity ^ 1;
      slot_kind[sl] = 0;
      HIPCHK(ctx, hipEventRecord(step_done[sl], st()));
    }
    elbo_ready[slot] = !elbo_async_used;
    elbo_val[slot] = v;
    elbo_open[slot] = true;
    elbo_next = (slot + 1) % ELBO_RING;
    *ticket = slot;
    return AGP_OK;
  }
  agp_status elbo_fetch(int32_t ticket, int wait, double* out, int32_t* ready) override {
    if (ticket < 0 || ticket >= ELBO_RING || !elbo_open[ticket] || !out) return AGP_ERR_INVALID;
    if (!elbo_ready[ticket]) {
      if (wait) {
        HIPCHK(ctx, hipEventSynchronize(elbo_ev[ticket]));
      } else {
        const hipError_t q = hipEventQuery(elbo_ev[ticket]);
        if (q == hipErrorNotReady) {
          (void)hipGetLastError();
          if (ready) *ready = 0;
          return AGP_OK;
        }
        HIPCHK(ctx, q);
      }
      elbo_val[ticket] = elbo_pin[ticket];
      elbo_ready[ticket] = true;
    }
    *out = elbo_val[ticket];
    elbo_open[ticket] = false;
    if (ready) *ready = 1;
    return AGP_OK;
  }
  // this handle sees shard bs_rank of bs_world of every minibatch (batch-parallel).  Set by agp_svgp_set_batch_shard and -- so that
  // a host cannot forget it after a handle was re-created -- by every batch-mode *_multi call from its communicator.
  int bs_rank = 0, bs_world = 1;
  void adopt_batch_shard(const agp_comm* cm, int mode) {
    if (mode == AGP_SHARD_BATCH && cm && cm->world > 1) {
      bs_rank = cm->rank;
      bs_world = cm->world;
      shard_once = bs_rank == 0;
    }
  }
  agp_status set_batch_shard(int rank, int world) override {
    if (world < 1 || rank < 0 || rank >= world) return AGP_ERR_INVALID;
    shard_once = rank == 0;
    bs_rank = rank;
    bs_world = world;
    return AGP_OK;
  }
  agp_status elbo_terms(double* out3) override {
    if (!out3) return AGP_ERR_INVALID;
    out3[0] = e_data;
    out3[1] = kl_gauss_last;
    out3[2] = kl_aug;
    return AGP_OK;
  }

  agp_status get_state(int l, void* mu, void* sigma, void* eta1, void* eta2) override {
    if (l < 0 || l >= nl) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    if (mu || sigma) AGPCHK(materialize(g));
    if (mu) HIPCHK(ctx, hipMemcpyAsync(mu, g.mu, sizeof(T) * m, hipMemcpyDeviceToDevice, st()));
    if (eta1) HIPCHK(ctx, hipMemcpyAsync(eta1, g.eta1, sizeof(T) * m, hipMemcpyDeviceToDevice, st()));
    if (sigma)
      HIPCHK(ctx, hipMemcpy2DAsync(sigma, sizeof(T) * m, g.Sigma, sizeof(T) * mp, sizeof(T) * m, m,
                                   hipMemcpyDeviceToDevice, st()));
    if (eta2)
      HIPCHK(ctx, hipMemcpy2DAsync(eta2, sizeof(T) * m, g.eta2, sizeof(T) * mp, sizeof(T) * m, m,
                                   hipMemcpyDeviceToDevice, st()));
    return AGP_OK;
  }

  agp_status set_state(int l, const void* eta1, const void* eta2) override {
    if (l < 0 || l >= nl || !eta1 || !eta2) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    HIPCHK(ctx, hipMemsetAsync(g.eta1, 0, sizeof(T) * mp, st()));
    HIPCHK(ctx, hipMemcpyAsync(g.eta1, eta1, sizeof(T) * m, hipMemcpyDeviceToDevice, st()));
    hipLaunchKernelGGL((k_copy2d<T>), grid2(mp, mp), blk2, 0, st(), (const T*)eta2, m, m, m, g.eta2, mp, mp, mp,
                       T(-0.5), T(1));
    hipLaunchKernelGGL((k_copy2d<T>), grid2(mp, mp), blk2, 0, st(), (const T*)eta2, m, m, m, g.La, mp, mp, mp, T(1),
                       T(-2));
    LAUNCHCHK(ctx);
    return refactor(g);
  }

  int64_t last_batch() override { return B_last; }
  // init_state(model) (src/training/states.jl:1-9), what train! does when called without a state: fresh local variables
  // (LogisticSoftMax alpha = K, states.jl:11-31 -> logisticsoftmax.jl:43-53), RobbinsMonro counters back to 1 (states.jl:61-84,
  // optimisers.jl:12) and new hyper-optimiser states.  The posterior (eta1, eta2) belongs to the model, not the state: kept.
  agp_status init_state() override {
    n_opt = 1;
    if (noise_adam) HIPCHK(ctx, hipMemsetAsync(noise_adam, 0, sizeof(double) * 6, st()));  // init_local_vars: new state_sigma2
    const T kk = (T)(lp.kind == AGP_LIK_LOGISTICSOFTMAX ? desc.lik.n_class : 1);
    hipLaunchKernelGGL((k_fill<T>), grid1(Bp), dim3(256), 0, st(), alpha, Bp, kk);
    LAUNCHCHK(ctx);
    for (auto& g : lat) {
      g.k_m.clear();
      g.k_v.clear();
      g.k_step = 0;
      g.z_step = 0;
      if (g.kadam) HIPCHK(ctx, hipMemsetAsync(g.kadam, 0, sizeof(double) * 2 * (1 + D), st()));
      if (g.z_am) {
        HIPCHK(ctx, hipMemsetAsync(g.z_am, 0, sizeof(double) * m * D, st()));
        HIPCHK(ctx, hipMemsetAsync(g.z_av, 0, sizeof(double) * m * D, st()));
      }
    }
    return invalidate_data();
  }
  // the caller refilled / mutated X (or y) in place: nothing cached from it may be reused (AnalyticVI kappa cache, look-ahead)
  agp_status invalidate_data() override {
    for (auto& g : lat) g.kappa_valid = g.kappa_eval = false;
    pf_valid = false;
    x_last = nullptr;
    idx_last = nullptr;
    return AGP_OK;
  }

  agp_status get_matrix(int l, int which, void* out, int64_t ldo, int64_t cap) override {
    if (l < 0 || l >= nl || !out || cap <= 0) return AGP_ERR_INVALID;
    Latent& g = lat[l];
    const int64_t B = B_last;
    const bool m_sized = which == AGP_MAT_L || which == AGP_MAT_KINV;
    if (m_sized ? cap < m : (which != AGP_VEC_ALPHA && (B <= 0 || cap < B))) {
      ctx->err = "agp_svgp_get_matrix: output capacity " + std::to_string(cap) + " is smaller than what the last batch left (" +
                 std::to_string(m_sized ? m : B) + ")";
      return AGP_ERR_INVALID;
    }
    auto copy2 = [&](const T* src, int64_t lds, int64_t rows, int64_t cols) -> agp_status {
      if (ldo < cols) return AGP_ERR_INVALID;
      HIPCHK(ctx, hipMemcpy2DAsync(out, sizeof(T) * ldo, src, sizeof(T) * lds, sizeof(T) * cols, rows,
                                   hipMemcpyDeviceToDevice, st()));
      return AGP_OK;
    };
    auto copy1 = [&](const T* src, int64_t n) -> agp_status {
      HIPCHK(ctx, hipMemcpyAsync(out, src, sizeof(T) * n, hipMemcpyDeviceToDevice, st()));
      return AGP_OK;
    };
    switch (which) {
      case AGP_MAT_L: {
        AGPCHK(refresh_K());
        hipLaunchKernelGGL((k_publish_diag<T>), dim3((unsigned)(mp / TILE)), dim3(256), 0, st(), g.L, mp, (const T*)g.DgK);
        LAUNCHCHK(ctx);
        AGPCHK(copy2(g.L, mp, m, m));
        // strict upper part of the stored factor is not maintained outside the diagonal tiles: zero it in the copy
        hipLaunchKernelGGL((k_zero_strict_upper<T>), grid2(m, m), blk2, 0, st(), (T*)out, ldo, m);
        LAUNCHCHK(ctx);
        return AGP_OK;
      }
      case AGP_MAT_KINV:
        AGPCHK(refresh_K());
        return copy2(g.Kinv, mp, m, m);
      case AGP_MAT_KNM:
        return copy2(g.Knm, mp, B, m);
      case AGP_MAT_KAPPA:
        return copy2(g.kappa, mp, B, m);
      case AGP_VEC_KTILDE:
        return copy1(Kt + l * Bp, B);
      case AGP_VEC_MEAN_F:
        return copy1(muf + l * Bp, B);
      case AGP_VEC_VAR_F:
        return copy1(varf + l * Bp, B);
      case AGP_VEC_THETA:
        return copy1(theta + l * Bp, B);
      case AGP_VEC_C:
        return copy1(cbuf + l * Bp, B);
      case AGP_VEC_GAMMA:
        return copy1(gamma + l * Bp, B);
      case AGP_VEC_ALPHA:  // state carried across minibatches: exported by capacity, not by the last batch
        return copy1(alpha, std::min<int64_t>(cap, Bmax));
      default:
        return AGP_ERR_INVALID;
    }
  }

  // ---- prediction (predictions.jl:25-50) -------------------------------------------------------------------
  agp_status ensure_pred(Latent& g, bool need_var) {
    AGPCHK(refresh_K());
    AGPCHK(materialize(g));
    if (!g.apred) {
      AGPCHK(dmalloc(ctx, &g.apred, mp));
    }
    if (need_var && !g.Apred) AGPCHK(dmalloc(ctx, &g.Apred, mp * mp));
    if (!g.pred_valid) {
      // K \ mu
      hipLaunchKernelGGL((k_symv<T>), grid1(mp * 64), dim3(256), 0, st(), (const T*)g.Kinv, mp, mp, (const T*)g.mu,
                         g.apred);
      LAUNCHCHK(ctx);
      g.pred_valid = true;
    }
    if (need_var && !g.predvar_valid) {
      // A = K \ (I - Sigma / K) = Kinv - Kinv Sigma Kinv :  T2 = Kinv Sigma (NT, both symmetric) ; A = Kinv - Kinv T2'
      AGPCHK((gemm_nt<T, EPI_STORE>(ctx, g.Kinv, mp, g.Sigma, mp, mp, mp, mp, 0, Tw2, mp, nullptr, 0, nullptr, nullptr,
                                    nullptr, 0)));
      tw2_kis_of = (int)(&g - lat.data());  // the hyper-gradient reuses K^-1 Sigma (hypergrad, one_product)
      {  // (symmetric result: lower tiles only, mirrored -- half the flops of the full EPI_EMINUS product)
        const int64_t ntm = mp / TILE, tiles = ntm * (ntm + 1) / 2;
        if (tiles <= 160 && mp >= 8 * BK)  // fewer tiles than CUs: four k-groups per workgroup, like the symmetric product (syrk_tn)
          hipLaunchKernelGGL((k_gemm_nt_eminus_sym<T, 4>), dim3((unsigned)tiles), dim3(4 * NTHREADS), 0, st(), (const T*)g.Kinv, mp,
                             (const T*)Tw2, mp, mp, g.Apred, mp, (const T*)g.Kinv, mp);
        else if (tiles <= kg2_limit() && mp >= 4 * BK)
          hipLaunchKernelGGL((k_gemm_nt_eminus_sym<T, 2>), dim3((unsigned)tiles), dim3(2 * NTHREADS), 0, st(), (const T*)g.Kinv, mp,
                             (const T*)Tw2, mp, mp, g.Apred, mp, (const T*)g.Kinv, mp);
        else
          hipLaunchKernelGGL((k_gemm_nt_eminus_sym<T, 1>), dim3((unsigned)tiles), dim3(NTHREADS), 0, st(), (const T*)g.Kinv, mp,
                             (const T*)Tw2, mp, mp, g.Apred, mp, (const T*)g.Kinv, mp);
        LAUNCHCHK(ctx);
      }
      g.predvar_valid = true;
    }
    return AGP_OK;
  }

  agp_status ensure_pred_ws(int64_t nt, bool need_var) {
    const int64_t CH = 4096;
    if (pred_chunk == 0) {
      pred_chunk = CH;
      AGPCHK(dmalloc(ctx, &ppm, (mp / TILE) * CH));
      AGPCHK(dmalloc(ctx, &ppv, (2 * mp / TILE) * CH));
    }
    if (need_var && !Kstar) AGPCHK(dmalloc(ctx, &Kstar, CH * mp));
    if (nt > pred_nt_cap) {
      if (pmu) (void)hipFree(pmu);
      if (pvar) (void)hipFree(pvar);
      AGPCHK(dmalloc(ctx, &pmu, nl * nt));
      AGPCHK(dmalloc(ctx, &pvar, nl * nt));
      pred_nt_cap = nt;
    }
    return AGP_OK;
  }

  // multi-output: latent predictions mixed by A (predictions.jl:52-92) -> T[n_task][n_t]
  agp_status predict_f(const void* xt, int64_t ldx, int64_t nt, void* mu_out, void* var_out) override {
    if (!mo) return predict_f_latent(xt, ldx, nt, mu_out, var_out);
    AGPCHK(ensure_pred_ws(nt, var_out != nullptr));
    if (nt > mo_pred_cap) {
      if (mo_pmu) (void)hipFree(mo_pmu);
      if (mo_pvar) (void)hipFree(mo_pvar);
      AGPCHK(dmalloc(ctx, &mo_pmu, (int64_t)nl * nt));
      AGPCHK(dmalloc(ctx, &mo_pvar, (int64_t)nl * nt));
      mo_pred_cap = nt;
    }
    AGPCHK(predict_f_latent(xt, ldx, nt, mo_pmu, var_out ? mo_pvar : nullptr));
    // a latent-sharded handle returns its PARTIAL mix (own columns of A): summed over ranks it is the full one
    hipLaunchKernelGGL((k_mo_mix<T>), grid1(nt), dim3(256), 0, st(), nt, nl, nT, (const T*)(A_dev + qlo), Qa(),
                       (const T*)mo_pmu, nt, (T*)mu_out, nt, 0);
    if (var_out)
      hipLaunchKernelGGL((k_mo_mix<T>), grid1(nt), dim3(256), 0, st(), nt, nl, nT, (const T*)(A_dev + qlo), Qa(),
                         (const T*)mo_pvar, nt, (T*)var_out, nt, 1);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
  T *mo_pmu = nullptr, *mo_pvar = nullptr;
  int64_t mo_pred_cap = 0;

  agp_status predict_f_latent(const void* xt, int64_t ldx, int64_t nt, void* mu_out, void* var_out) {
    if (!xt || nt <= 0 || ldx < D || !mu_out) return AGP_ERR_INVALID;
    for (auto& g : lat) AGPCHK(params_to_host(g));  // k_predict_finish takes the variance by value
    const bool need_var = var_out != nullptr;
    AGPCHK(ensure_pred_ws(0, need_var));
    for (auto& g : lat) {
      AGPCHK(ensure_pred(g, need_var));
      AGPCHK(ensure_zsc(g));
    }
    const int64_t CH = pred_chunk;
    for (int l = 0; l < nl; ++l) {
      Latent& g = lat[l];
      if (!need_var && kmm_usable(D)) {
        // means only: ONE launch over all test points, every workgroup carries its 64 rows through all inducing-point tiles with
        // the row-dot against K^-1 mu in registers -- K*m and per-tile partial sums never reach memory (predictions.jl:33-34).
        // (The VALU kernel -- D beyond the MFMA kernel's limit -- leaves one slice per column tile, which the caller's output
        // has no room for: it takes the chunked form below.)
        const int slices = launch_kernelmatrix<T>(ctx, st(), (const T*)xt, ldx, (const int64_t*)nullptr, nt, (const T*)g.Z, D, m, D,
                                                  (const T*)g.scales, g.k.kind, kvar(g), (T*)nullptr, mp, nt, mp, 0, T(0),
                                                  (const T*)g.apred, (T*)mu_out + (int64_t)l * nt, nt, 1, (const T*)g.Zsc,
                                                  (const T*)g.zn);
        LAUNCHCHK(ctx);
        if (slices == 1) continue;
      }
      for (int64_t s = 0; s < nt; s += CH) {
        const int64_t nc = (nt - s) < CH ? (nt - s) : CH;
        const int64_t nq = rup64(nc);
        const T* xs = (const T*)xt + s * ldx;
        const int slices = launch_kernelmatrix<T>(ctx, st(), xs, ldx, (const int64_t*)nullptr, nc, (const T*)g.Z, D, m, D,
                                                  (const T*)g.scales, g.k.kind, kvar(g), need_var ? Kstar : (T*)nullptr, mp,
                                                  nq, mp, 0, T(0), (const T*)g.apred, ppm, CH, 0, (const T*)g.Zsc, (const T*)g.zn);
        LAUNCHCHK(ctx);
        if (need_var)
          AGPCHK((gemm_nt<T, EPI_ROWDOT>(ctx, Kstar, mp, g.Apred, mp, nq, mp, mp, 0, nullptr, 0, Kstar, mp, nullptr, ppv,
                                         nullptr, CH)));
        hipLaunchKernelGGL((k_predict_finish<T>), grid1(nc), dim3(256), 0, st(), nc, slices, (const T*)ppm,
                           (int)(2 * mp / TILE), (const T*)ppv, CH, (T)g.k.variance, (T)jitter,
                           (T*)mu_out + (int64_t)l * nt + s, need_var ? (T*)var_out + (int64_t)l * nt + s : (T*)nullptr);
        LAUNCHCHK(ctx);
      }
    }
    return AGP_OK;
  }

  agp_status predict_y(const void* xt, int64_t ldx, int64_t nt, void* out) override {
    if (!out) return AGP_ERR_INVALID;
    if (mo) {  // T[n_task][n_t]: regression tasks -> mean ; Bernoulli tasks -> 1.0 / 0.0 (mu_f > 0)
      if (mo_sharded) return sharded_predict_error();
      AGPCHK(predict_f(xt, ldx, nt, out, nullptr));
      return mo_predict_from_f(nt, 0, out, nullptr, nullptr, nullptr, 0);
    }
    if (lp.kind == AGP_LIK_GAUSSIAN || lp.kind == AGP_LIK_STUDENTT || lp.kind == AGP_LIK_LAPLACE)
      return predict_f(xt, ldx, nt, out, nullptr);
    AGPCHK(ensure_pred_ws(nt, false));
    AGPCHK(predict_f(xt, ldx, nt, pmu, nullptr));
    if (lp.kind == AGP_LIK_HETEROSCEDASTIC) {  // heteroscedastic.jl:137-141 : the mean of the first latent
      HIPCHK(ctx, hipMemcpyAsync(out, pmu, sizeof(T) * nt, hipMemcpyDeviceToDevice, st()));
      return AGP_OK;
    }
    if (lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_NEGBINOMIAL) {
      hipLaunchKernelGGL((k_predict_event<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu,
                         (int)(lp.kind == AGP_LIK_NEGBINOMIAL), desc.lik.p0,
                         lp.kind == AGP_LIK_POISSON ? (const T*)lam_dev : (const T*)nullptr, (T*)out);
      LAUNCHCHK(ctx);
      return AGP_OK;
    }
    hipLaunchKernelGGL((k_predict_label<T>), grid1(nt), dim3(256), 0, st(), nt, nl, nt, desc.latent_offset,
                       (const T*)pmu, (int32_t*)out, (int)(lp.kind == AGP_LIK_LOGISTIC || lp.kind == AGP_LIK_BAYESIANSVM));
    LAUNCHCHK(ctx);
    return AGP_OK;
  }

  agp_status upload_gh(const double* nodes, const double* weights, int nn) {
    if (nn > gh_cap) {
      if (gh_dev) (void)hipFree(gh_dev);
      AGPCHK(dmalloc(ctx, &gh_dev, 2 * nn));
      gh_cap = nn;
    }
    HIPCHK(ctx, hipMemcpyAsync(gh_dev, nodes, sizeof(double) * nn, hipMemcpyHostToDevice, st()));
    HIPCHK(ctx, hipMemcpyAsync(gh_dev + nn, weights, sizeof(double) * nn, hipMemcpyHostToDevice, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    gh_n = nn;
    return AGP_OK;
  }
  agp_status set_quadrature(const double* nodes, const double* weights, int nn) override {
    if (!nodes || !weights || nn <= 0) return AGP_ERR_INVALID;
    return upload_gh(nodes, weights, nn);
  }
  agp_status set_lsm_alpha(const void* a, int64_t n) override {
    if (lp.kind != AGP_LIK_LOGISTICSOFTMAX || !a || n <= 0 || n > Bp) return AGP_ERR_INVALID;
    HIPCHK(ctx, hipMemcpyAsync(alpha, a, sizeof(T) * n, hipMemcpyDeviceToDevice, st()));
    return AGP_OK;
  }
  agp_status get_lik_param(double* out) override {
    if (!out) return AGP_ERR_INVALID;
    if (lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_HETEROSCEDASTIC || (lp.kind == AGP_LIK_GAUSSIAN && lp.noise_dev)) {
      T v;
      HIPCHK(ctx, hipMemcpyAsync(&v, lam_dev, sizeof(T), hipMemcpyDeviceToHost, st()));
      HIPCHK(ctx, hipStreamSynchronize(st()));
      *out = (double)v;
    } else {
      *out = desc.lik.p0;
    }
    return AGP_OK;
  }
  agp_status set_lik_param(double v) override {
    if (!(v > 0) || !(lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_HETEROSCEDASTIC ||
                      (lp.kind == AGP_LIK_GAUSSIAN && lp.noise_dev)))
      return AGP_ERR_INVALID;
    hipLaunchKernelGGL((k_fill<T>), dim3(1), dim3(64), 0, st(), lam_dev, (int64_t)1, (T)v);
    LAUNCHCHK(ctx);
    return AGP_OK;
  }

  agp_status proba_y(const void* xt, int64_t ldx, int64_t nt, const double* nodes, const double* weights, int nn,
                     void* o0, void* o1) override {
    if (!o0) return AGP_ERR_INVALID;
    AGPCHK(ensure_pred_ws(nt, true));
    if (mo) {  // per task compute_proba on the mixed (mu_f, var_f): out0 / out1 are T[n_task][n_t]
      if (!o1) return AGP_ERR_INVALID;
      if (mo_sharded) return sharded_predict_error();
      AGPCHK(predict_f(xt, ldx, nt, o0, o1));
      return mo_predict_from_f(nt, 1, o0, o1, nodes, weights, nn);
    }
    return proba_y_single(xt, ldx, nt, nodes, weights, nn, o0, o1);
  }

  agp_status sharded_predict_error() {
    ctx->err = "latent-sharded multi-output model: predict_f returns the partial mix; all-reduce it and finish with "
               "agp_svgp_mo_predict_from_f";
    return AGP_ERR_UNSUPPORTED;
  }
  // the likelihood half of predict_y (mode 0: o0 = mixed mean_f, in place) / proba_y (mode 1: o0, o1 = mixed mean_f, var_f, in
  // place) of a multi-output model: what follows the mixing in predictions.jl:178-247
  agp_status mo_predict_from_f(int64_t nt, int mode, void* o0, void* o1, const double* nodes, const double* weights,
                               int nn) override {
    if (!mo || !o0 || nt <= 0 || (mode == 1 && !o1)) return AGP_ERR_INVALID;
    if (mode == 0) {
      for (int t = 0; t < nT; ++t) {
        if (mocfg.kind[t] == AGP_LIK_LOGISTIC || mocfg.kind[t] == AGP_LIK_BAYESIANSVM)
          hipLaunchKernelGGL((k_step01<T>), grid1(nt), dim3(256), 0, st(), (T*)o0 + (int64_t)t * nt, nt);
        else if (mocfg.kind[t] == AGP_LIK_NEGBINOMIAL)
          hipLaunchKernelGGL((k_predict_event<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)o0 + (int64_t)t * nt, 1,
                             (double)mocfg.p0[t], (const T*)nullptr, (T*)o0 + (int64_t)t * nt);
      }
      LAUNCHCHK(ctx);
      return AGP_OK;
    }
    {
      for (int t = 0; t < nT; ++t) {
        T* m0 = (T*)o0 + (int64_t)t * nt;
        T* v0 = (T*)o1 + (int64_t)t * nt;
        if (mocfg.kind[t] == AGP_LIK_GAUSSIAN) {
          hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)m0, (const T*)v0,
                             mocfg.p0[t], 0, m0, v0);
        } else if (mocfg.kind[t] == AGP_LIK_STUDENTT) {
          const double nu = (double)mocfg.p0[t], sg = (double)mocfg.p1[t];
          hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)m0, (const T*)v0,
                             (T)(nu * sg * sg / (2.0 * (nu / 2.0 - 1.0))), 1, m0, v0);
        } else if (mocfg.kind[t] == AGP_LIK_LAPLACE) {
          hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)m0, (const T*)v0,
                             (T)(2.0 * (double)mocfg.p0[t] * (double)mocfg.p0[t]), 1, m0, v0);
        } else {
          if (!nodes || !weights || nn <= 0) return AGP_ERR_INVALID;
          if (gh_n != nn) AGPCHK(upload_gh(nodes, weights, nn));
          if (mocfg.kind[t] == AGP_LIK_LOGISTIC)
            hipLaunchKernelGGL((k_proba_logistic<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)m0, (const T*)v0, nn,
                               (const double*)gh_dev, (const double*)(gh_dev + nn), m0, v0);
          else
            hipLaunchKernelGGL((k_proba_gh<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)m0, (const T*)v0, nn,
                               (const double*)gh_dev, (const double*)(gh_dev + nn),
                               mocfg.kind[t] == AGP_LIK_BAYESIANSVM ? 1 : 3, (double)mocfg.p0[t], (const T*)nullptr, m0,
                               v0);
        }
      }
      LAUNCHCHK(ctx);
      return AGP_OK;
    }
  }


  // ---- full predictive covariance (predict_f(...; cov=true, diag=false), predictions.jl:45-49) ------------------------
  //   cov = K** + jitt I - K*m (K^-1 - K^-1 Sigma K^-1) Km*      (K*m materialised: meant for small n_t)
  agp_status predict_f_cov(const void* xt, int64_t ldx, int64_t nt, void* mu_out, void* cov_out) override {
    if (!xt || nt <= 0 || nt > 8192 || ldx < D || !mu_out || !cov_out) return AGP_ERR_INVALID;
    if (mo_sharded) {
      ctx->err = "full predictive covariance of a latent-sharded multi-output model is not wired (partial mixes would have to be "
                 "all-reduced)";
      return AGP_ERR_UNSUPPORTED;
    }
    // multi-output (predictions.jl:52-92): mu_out T[n_task][n_t] = sum_q A[t][q] mu_q, cov_out T[n_task][n_t][n_t] = sum_q A[t][q]^2 Cov_q
    if (mo) AGPCHK(predict_f(xt, ldx, nt, mu_out, nullptr));
    else AGPCHK(predict_f_latent(xt, ldx, nt, mu_out, nullptr));
    const int64_t nq = rup64(nt);
    T *Ks = nullptr, *T1 = nullptr, *Kss = nullptr, *Cq = nullptr;
    AGPCHK(dmalloc(ctx, &Ks, nq * mp));
    AGPCHK(dmalloc(ctx, &T1, nq * mp));
    AGPCHK(dmalloc(ctx, &Kss, nq * nq));
    AGPCHK(dmalloc(ctx, &Cq, nq * nq));
    agp_status rc = AGP_OK;
    for (int l = 0; l < nl && rc == AGP_OK; ++l) {
      Latent& g = lat[l];
      rc = ensure_pred(g, true);
      if (rc != AGP_OK) break;
      dim3 gk((unsigned)(mp / TILE), (unsigned)(nq / TILE));
      (void)launch_kernelmatrix<T>(ctx, st(), (const T*)xt, ldx, (const int64_t*)nullptr, nt,
                         (const T*)g.Z, D, m, D, (const T*)g.scales, g.k.kind, kvar(g), Ks, mp, nq, mp, 0, T(0),
                         (const T*)nullptr, (T*)nullptr, (int64_t)0, 0, (const T*)g.Zsc, (const T*)g.zn);
      dim3 gs((unsigned)(nq / TILE), (unsigned)(nq / TILE));
      (void)launch_kernelmatrix<T>(ctx, st(), (const T*)xt, ldx, (const int64_t*)nullptr, nt,
                         (const T*)xt, ldx, nt, D, (const T*)g.scales, g.k.kind, kvar(g), Kss, nq, nq, nq, 1,
                         (T)jitter, (const T*)nullptr, (T*)nullptr, (int64_t)0);
      rc = gemm_nt<T, EPI_STORE>(ctx, Ks, mp, g.Apred, mp, nq, mp, mp, 0, T1, mp, nullptr, 0, nullptr, nullptr, nullptr, 0);
      if (rc != AGP_OK) break;
      rc = gemm_nt<T, EPI_EMINUS>(ctx, T1, mp, Ks, mp, nq, nq, mp, 0, Cq, nq, Kss, nq, nullptr, nullptr, nullptr, 0);
      if (rc != AGP_OK) break;
      if (mo) {
        hipLaunchKernelGGL((k_mo_cov_acc<T>), grid2(nt, nt), blk2, 0, st(), nt, nT, (const T*)(A_dev + l), (int64_t)Qa(),
                           (const T*)Cq, nq, (T*)cov_out, l == 0 ? 1 : 0);
      } else if (hipMemcpy2DAsync((T*)cov_out + (int64_t)l * nt * nt, sizeof(T) * nt, Cq, sizeof(T) * nq, sizeof(T) * nt, nt,
                                  hipMemcpyDeviceToDevice, st()) != hipSuccess)
        rc = AGP_ERR_HIP;
    }
    (void)hipStreamSynchronize(st());
    dfree(Ks);
    dfree(T1);
    dfree(Kss);
    dfree(Cq);
    return rc;
  }

  // ---- multi-GPU drivers behind the ABI (include/agp_hip.h "multi-GPU") -----------------------
  agp_status comm_sum(agp_comm* cm, void* buf, int64_t count) {
    return comm_sum_typed(cm, buf, count, sizeof(T) == 8 ? AGP_F64 : AGP_F32, force_split());
  }
  // AGP_FORCE_SPLIT=1 (diagnostic): take the phase-split batch-parallel path (statistics -> all-reduce -> eta step) with a one-rank
  // communicator too, and issue its collective, so that its cost next to the fused step can be measured on a single GPU
  static bool force_split() {
    static const bool on = []() {
      const char* e = getenv("AGP_FORCE_SPLIT");
      return e && e[0] == '1';
    }();
    return on;
  }
  agp_status comm_sum_typed(agp_comm* cm, void* buf, int64_t count, int dtype, bool force = false) {
    // (AGP_ALLOW_PARTIAL_SHARD=1: bench.py times ONE rank's share of the 16-latent model on a one-GPU box -- the other latents'
    //  rows of the exchange buffer stay zero, the numbers are meaningless, the work per rank is what is measured)
    static const bool allow_partial = []() {
      const char* e = getenv("AGP_ALLOW_PARTIAL_SHARD");
      return e && e[0] == '1';
    }();
    if (!cm && mo_sharded && nl != qtot && !allow_partial) {
      // a handle that owns a slice of the latents cannot finish a mix, an ELBO or a prediction on its own
      ctx->err = "latent-sharded multi-output handle: this call needs the communicator of the run (comm = NULL)";
      return AGP_ERR_INVALID;
    }
    if (!cm || (cm->world <= 1 && !force)) return AGP_OK;
    if (cm->ctx != ctx) {
      ctx->err = "agp_comm belongs to another ctx (its collectives would run on another stream)";
      return AGP_ERR_INVALID;
    }
    return agp_comm_allreduce(cm, buf, count, dtype);
  }

  // block-column groups of the packed statistics for AGP_SPLIT_OVERLAP: four groups of about equal tile count (8 groups with a
  // 112 us train measured the same step time); group 0 also carries t (it sits in front of block column 0 in `stats`)
  agp_status overlap_groups() {
    const int64_t nt = mp / TILE;
    const int want = (int)std::max<int64_t>(1, std::min<int64_t>(4, nt));
    if (!arrive_dev) {
      HIPCHK(ctx, hipMalloc((void**)&arrive_dev, 8 * ARRIVE_STRIDE * sizeof(int32_t)));
      HIPCHK(ctx, hipMemsetAsync(arrive_dev, 0, 8 * ARRIVE_STRIDE * sizeof(int32_t), st()));
    }
    if (ov_nt == nt && ov_ng == want) return AGP_OK;
    auto cum = [&](int64_t c) { return c * nt - c * (c - 1) / 2; };  // tiles of block columns [0, c)
    const int64_t total = cum(nt);
    int64_t bnd[9];
    bnd[0] = 0;
    for (int g = 1; g < want; ++g) {
      const double tgt = (double)total * g / want;
      int64_t c = bnd[g - 1] + 1;
      while (c < nt - (want - g) && std::fabs((double)cum(c + 1) - tgt) < std::fabs((double)cum(c) - tgt)) ++c;
      bnd[g] = c;
    }
    bnd[want] = nt;
    for (int g = 0; g < want; ++g) {
      ov_off[g] = g == 0 ? 0 : mp + cum(bnd[g]) * TILE * TILE;
      ov_cnt[g] = (g == 0 ? mp : 0) + (cum(bnd[g + 1]) - cum(bnd[g])) * TILE * TILE;
      for (int64_t c = bnd[g]; c < bnd[g + 1]; ++c) ov_grp[c] = (unsigned char)g;
    }
    ov_ng = want;
    ov_nt = nt;
    return AGP_OK;
  }

  agp_status cavi_step_multi(agp_comm* cm, int mode, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                             double rho) override {
    if (mode != AGP_SHARD_LATENT && mode != AGP_SHARD_BATCH) return AGP_ERR_INVALID;
    const bool multi = cm && (cm->world > 1 || (force_split() && mode == AGP_SHARD_BATCH));
    adopt_batch_shard(cm, mode);
    const bool lam_lik = lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_HETEROSCEDASTIC ||
                         (lp.kind == AGP_LIK_GAUSSIAN && lp.noise_dev);  // likelihood state re-estimated from whole-minibatch sums
    if (multi && lam_lik && mode == AGP_SHARD_LATENT) {
      // a Poisson model has one latent, and the two heteroscedastic latents are coupled point-wise and stay on one handle:
      // these models shard over the minibatch only
      ctx->err = "Poisson / Heteroscedastic likelihoods shard over the minibatch (AGP_SHARD_BATCH), not over latents";
      return AGP_ERR_UNSUPPORTED;
    }
    if (mode == AGP_SHARD_BATCH && mo_sharded) return AGP_ERR_INVALID;
    // batch-sharded Poisson / Heteroscedastic: lambda is re-estimated from sums over the WHOLE minibatch (poisson.jl:78,
    // heteroscedastic.jl:94): the local update stops after its partial sums, three doubles are all-reduced, then it finishes
    lam_deferred = multi && lam_lik;
    // batch-parallel over several ranks, one latent: the step rides on the task-graph launches like the one-GPU step does (round 3)
    // -- row statistics as the epilogue of this launch, the eta step from the REDUCED statistics as the prologue of the next one
    // (PendingPacked) -- so that between two factorisations only the packed product, the all-reduce and one 5 us kernel remain.
    // AGP_SPLIT_MERGED=0 keeps the separate kernels (k_safe_rowstats, k_eta2_from_packed).
    static const bool split_merged = []() {
      const char* e = getenv("AGP_SPLIT_MERGED");
      return !(e && e[0] == '0');
    }();
    const bool merge = split_merged && multi && mode == AGP_SHARD_BATCH && !lam_lik && nl == 1 && pro_allowed() &&
                       chol_use_dag(ctx, mp / TILE, rup64(B) / TILE + 1, 1);
    in_cavi_step = merge;
    const agp_status sl = step_local(x, ldx, y, idx, B, rho, false);
    in_cavi_step = false;
    lam_deferred = false;
    AGPCHK(sl);
    AGPCHK(run_deferred_safe());  // (the launch's fallback: the packed product below needs this step's r, w now)
    if (multi && lam_lik) {
      AGPCHK(comm_sum_typed(cm, scal_dev + 60, 3, AGP_F64, force_split()));
      AGPCHK(lambda_finish_reduced());
    }
    const bool lsm = lp.kind == AGP_LIK_LOGISTICSOFTMAX;
    if (mode == AGP_SHARD_LATENT) {
      if (lsm) {
        for (int it = 0; it < 2; ++it) {  // the (gamma, alpha) fixed point needs sum_k gamma_k over ALL latents
          AGPCHK(lsm_gamma());
          AGPCHK(comm_sum(cm, gsum, B_last));
          AGPCHK(lsm_alpha());
        }
      }
      if (mo_sharded) {
        AGPCHK(comm_sum(cm, fall, (int64_t)2 * qtot * Bp));
        AGPCHK(mo_mix());
      }
      AGPCHK(step_stats(true));  // each latent is whole on its rank: the fused natural-gradient step applies
      return step_global(true);
    }
    if (lsm) AGPCHK(lsm_local_all());  // all latents are local: the fixed point is per point, nothing to exchange
    if (!multi) {
      AGPCHK(step_stats(true));
      return step_global(true);
    }
    AGPCHK(step_stats(false));
    // AGP_SPLIT_OVERLAP=1 (default off; read at every call so that bench.py can A/B it in one process): the statistics travel in
    // block-column groups on the communicator's own stream and the next task-graph launch starts on the first group
    // (comm_allreduce_groups).  Only where the step rides on the task-graph launches (`merge`).
    bool overlap = false;
    if (merge) {
      const char* e = getenv("AGP_SPLIT_OVERLAP");
      overlap = e && e[0] == '1';
    }
    if (overlap) {
      if (cm->ctx != ctx) {
        ctx->err = "agp_comm belongs to another ctx (its collectives would run on another stream)";
        return AGP_ERR_INVALID;
      }
      AGPCHK(overlap_groups());
      arrive_epoch += 1;
      AGPCHK(comm_allreduce_groups(cm, stats, (int)sizeof(T), ov_off, ov_cnt, ov_ng, sizeof(T) == 8 ? AGP_F64 : AGP_F32, arrive_dev,
                                   arrive_epoch));
    } else
      AGPCHK(comm_sum(cm, stats, (int64_t)nl * stats_stride()));
    if (merge) {
      Latent& g = lat[0];
      pendp.on = true;
      pendp.overlap = overlap;
      pendp.epoch = arrive_epoch;
      pendp.cm = cm;
      pendp.lr = (T)cur_lr();
      pendp.Kinv = kinv_step(g);
      pendp.kinv_mu0 = kinv_mu0_step(g);
      g.la_state = 1;  // La does not hold -2 eta2: whoever wants it rebuilds it from eta2 (after flush())
      g.xa_valid = false;
      g.post_valid = false;
      g.pred_valid = g.predvar_valid = false;
      n_opt += 1;
      return AGP_OK;
    }
    return step_global(false);
  }

  agp_status elbo_multi(agp_comm* cm, int mode, double* out) override {
    if (!out || !x_last || B_last <= 0) return AGP_ERR_INVALID;
    adopt_batch_shard(cm, mode);
    if (mo_sharded) {
      AGPCHK(mo_refresh_f());
      AGPCHK(comm_sum(cm, fall, (int64_t)2 * qtot * Bp));
    }
    double mine = 0.0;
    AGPCHK(elbo(x_last, ldx_last, y_last, idx_last, B_last, rho_last, 0, &mine));
    if (!cm || cm->world <= 1) {
      *out = mine;
      return AGP_OK;
    }
    // scalars travel as doubles through a small device buffer (scal_dev[56..59])
    double h[2] = {mode == AGP_SHARD_LATENT ? mine : e_data, mode == AGP_SHARD_LATENT ? 0.0 : kl_aug};
    HIPCHK(ctx, hipMemcpyAsync(scal_dev + 56, h, sizeof(double) * 2, hipMemcpyHostToDevice, st()));
    AGPCHK(comm_sum_typed(cm, scal_dev + 56, 2, AGP_F64));
    HIPCHK(ctx, hipMemcpyAsync(h, scal_dev + 56, sizeof(double) * 2, hipMemcpyDeviceToHost, st()));
    HIPCHK(ctx, hipStreamSynchronize(st()));
    *out = mode == AGP_SHARD_LATENT ? h[0] : rho_last * h[0] - kl_gauss_last - rho_last * h[1];
    return AGP_OK;
  }

  double* hy_tied = nullptr;  // [1 + D + m*D] doubles: summed gradient of the tied-Z mode
  agp_status hyper_step_multi(agp_comm* cm, int tied) override {
    if (!hy_k && !hy_z) return AGP_OK;
    if (cm && cm->world > 1 && bs_world > 1 && bs_world != cm->world) {
      ctx->err = "agp_svgp_hyper_step_multi: the handle's batch shard does not belong to this communicator";
      return AGP_ERR_INVALID;
    }
    if (mo_sharded) {  // the mixed data term of the gradient reads every latent's mean_f under the updated posterior
      AGPCHK(mo_refresh_f());
      AGPCHK(comm_sum(cm, fall, (int64_t)2 * qtot * Bp));
    }
    const int64_t ng = 1 + D + m * D;
    const bool batch_sharded = bs_world > 1;
    if (!tied && !batch_sharded) {
      hyper_multi_ok = true;
      const agp_status hs_ = hyper_step();
      hyper_multi_ok = false;
      return hs_;
    }
    if (!hy_tied) AGPCHK(dmalloc(ctx, &hy_tied, ng));
    if (!tied) {
      // batch-sharded handle, every latent its own kernel and Z: the data part of each latent's gradient is a sum over the ranks'
      // shards (all-reduced, 1 + D + m D doubles per latent), its Gaussian-KL part is replicated and enters with weight
      // 1 / world on every rank (k_hyper_gK) -- every rank then takes the identical ADAM step and the replicas stay together
      if (!cm || cm->world != bs_world) {
        ctx->err = "hyper step of a batch-sharded handle needs the communicator of the run";
        return AGP_ERR_INVALID;
      }
      for (int l = 0; l < nl; ++l) {
        AGPCHK(hypergrad(l, nullptr, nullptr, nullptr));
        HIPCHK(ctx, hipMemsetAsync(hy_tied, 0, sizeof(double) * ng, st()));
        HIPCHK(ctx, hipMemcpyAsync(hy_tied, hy_last.data(), sizeof(double) * (1 + D), hipMemcpyHostToDevice, st()));
        hipLaunchKernelGGL((k_acc_to_double<T>), grid1(m * D), dim3(256), 0, st(), m * D, (const T*)hy_dZ, hy_tied + 1 + D);
        LAUNCHCHK(ctx);
        AGPCHK(comm_sum_typed(cm, hy_tied, ng, AGP_F64));
        std::vector<double> hg(1 + D, 0.0);
        HIPCHK(ctx, hipMemcpyAsync(hg.data(), hy_tied, sizeof(double) * (1 + D), hipMemcpyDeviceToHost, st()));
        hipLaunchKernelGGL((k_double_to<T>), grid1(m * D), dim3(256), 0, st(), m * D, (const double*)(hy_tied + 1 + D), hy_dZ);
        LAUNCHCHK(ctx);
        HIPCHK(ctx, hipStreamSynchronize(st()));
        AGPCHK(hyper_apply_one(l, &hg, (const T*)hy_dZ));
      }
      return hyper_finish();
    }
    if (batch_sharded && (!cm || cm->world != bs_world)) {  // (the Gaussian-KL part enters with weight 1 / world: without the
      ctx->err = "tied hyper step of a batch-sharded handle needs the communicator of the run";  // all-reduce the step is wrong)
      return AGP_ERR_INVALID;
    }
    HIPCHK(ctx, hipMemsetAsync(hy_tied, 0, sizeof(double) * ng, st()));
    std::vector<double> hs(1 + D, 0.0);
    for (int l = 0; l < nl; ++l) {
      AGPCHK(hypergrad(l, nullptr, nullptr, nullptr));
      for (int64_t i = 0; i < 1 + D; ++i) hs[i] += hy_last[i];
      hipLaunchKernelGGL((k_acc_to_double<T>), grid1(m * D), dim3(256), 0, st(), m * D, (const T*)hy_dZ, hy_tied + 1 + D);
      LAUNCHCHK(ctx);
    }
    HIPCHK(ctx, hipMemcpyAsync(hy_tied, hs.data(), sizeof(double) * (1 + D), hipMemcpyHostToDevice, st()));
    AGPCHK(comm_sum_typed(cm, hy_tied, ng, AGP_F64));
    HIPCHK(ctx, hipMemcpyAsync(hs.data(), hy_tied, sizeof(double) * (1 + D), hipMemcpyDeviceToHost, st()));
    hipLaunchKernelGGL((k_double_to<T>), grid1(m * D), dim3(256), 0, st(), m * D, (const double*)(hy_tied + 1 + D), hy_dZ);
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipStreamSynchronize(st()));
    for (int l = 0; l < nl; ++l) AGPCHK(hyper_apply_one(l, &hs, (const T*)hy_dZ));
    return hyper_finish();
  }

  agp_status predict_multi(agp_comm* cm, int what, const void* xt, int64_t ldx, int64_t nt, void* mu, void* var,
                           const double* nodes, const double* weights, int nn) override {
    if (what < 0 || what > 2 || !mu || (what == 2 && !var)) return AGP_ERR_INVALID;
    if (!mo_sharded) {  // nothing is sharded on the prediction side: the plain calls apply
      if (what == 0) return predict_f(xt, ldx, nt, mu, var);
      if (what == 1) return predict_y(xt, ldx, nt, mu);
      return proba_y(xt, ldx, nt, nodes, weights, nn, mu, var);
    }
    void* v = what == 1 ? nullptr : var;
    AGPCHK(predict_f(xt, ldx, nt, mu, v));  // partial mix over the owned columns of A
    AGPCHK(comm_sum(cm, mu, (int64_t)nT * nt));
    if (v) AGPCHK(comm_sum(cm, v, (int64_t)nT * nt));
    if (what == 0) return AGP_OK;
    return mo_predict_from_f(nt, what == 1 ? 0 : 1, mu, var, nodes, weights, nn);
  }

  agp_status proba_y_single(const void* xt, int64_t ldx, int64_t nt, const double* nodes, const double* weights, int nn,
                            void* o0, void* o1) {
    AGPCHK(predict_f(xt, ldx, nt, pmu, pvar));
    if (lp.kind == AGP_LIK_GAUSSIAN) {
      if (!o1) return AGP_ERR_INVALID;
      T s2 = lp.p0;
      if (lp.noise_dev) {  // the optimised noise lives on the device (compute_proba adds noise(l), gaussian.jl:41-45)
        double v = 0.0;
        AGPCHK(get_lik_param(&v));
        s2 = (T)v;
      }
      hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu, (const T*)pvar, s2, 0, (T*)o0,
                         (T*)o1);
    } else if (lp.kind == AGP_LIK_STUDENTT) {
      if (!o1) return AGP_ERR_INVALID;
      const double nu = desc.lik.p0, sg = desc.lik.p1;
      hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu, (const T*)pvar,
                         (T)(nu * sg * sg / (2.0 * (nu / 2.0 - 1.0))), 1, (T*)o0, (T*)o1);
    } else if (lp.kind == AGP_LIK_LOGISTIC) {
      if (!o1 || !nodes || !weights || nn <= 0) return AGP_ERR_INVALID;
      AGPCHK(upload_gh(nodes, weights, nn));
      hipLaunchKernelGGL((k_proba_logistic<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu, (const T*)pvar, nn,
                         (const double*)gh_dev, (const double*)(gh_dev + nn), (T*)o0, (T*)o1);
    } else if (lp.kind == AGP_LIK_LAPLACE) {  // laplace.jl:48-52
      if (!o1) return AGP_ERR_INVALID;
      hipLaunchKernelGGL((k_proba_regression<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu, (const T*)pvar,
                         (T)(2.0 * desc.lik.p0 * desc.lik.p0), 1, (T*)o0, (T*)o1);
    } else if (lp.kind == AGP_LIK_HETEROSCEDASTIC) {
      if (!o1) return AGP_ERR_INVALID;
      hipLaunchKernelGGL((k_proba_hetero<T>), grid1(nt), dim3(256), 0, st(), nt, nt, (const T*)pmu, (const T*)pvar,
                         (const T*)lam_dev, (T*)o0, (T*)o1);
    } else if (lp.kind == AGP_LIK_BAYESIANSVM || lp.kind == AGP_LIK_POISSON || lp.kind == AGP_LIK_NEGBINOMIAL) {
      if (!o1 || !nodes || !weights || nn <= 0) return AGP_ERR_INVALID;
      AGPCHK(upload_gh(nodes, weights, nn));
      const int link = lp.kind == AGP_LIK_BAYESIANSVM ? 1 : (lp.kind == AGP_LIK_POISSON ? 2 : 3);
      hipLaunchKernelGGL((k_proba_gh<T>), grid1(nt), dim3(256), 0, st(), nt, (const T*)pmu, (const T*)pvar, nn,
                         (const double*)gh_dev, (const double*)(gh_dev + nn), link, desc.lik.p0,
                         lp.kind == AGP_LIK_POISSON ? (const T*)lam_dev : (const T*)nullptr, (T*)o0, (T*)o1);
    } else {
      hipLaunchKernelGGL((k_proba_lsm<T>), grid1(nt), dim3(256), 0, st(), nt, nl, nt, (const T*)pmu, (T*)o0);
    }
    LAUNCHCHK(ctx);
    return AGP_OK;
  }
};

// ================================================================================================================
// C entry points
// ================================================================================================================
extern "C" {

int32_t agp_version(void) { return 200; }  // 2xx: round-2 ABI (agp_comm_*, *_multi, get_matrix capacity, kernel structure flags)

agp_status agp_ctx_create(int32_t device, void* hip_stream, agp_ctx** out) {
  if (!out) return AGP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return AGP_ERR_HIP;
  {
    DevGuard guard(device);  // touches the device once (creates its primary context) and puts the caller's back
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) return AGP_ERR_HIP;
  }
  agp_ctx* c = new agp_ctx();
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  *out = c;
  return AGP_OK;
}

agp_status agp_ctx_destroy(agp_ctx* ctx) {
  if (!ctx) return AGP_OK;
  DevGuard guard(ctx->device);
  if (ctx->tri_scratch || ctx->dag_flags || ctx->hset[0]) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->dag_flags) (void)hipFree(ctx->dag_flags);
    for (int q = 0; q < 2; ++q)
      if (ctx->hset[q]) (void)hipFree(ctx->hset[q]);
    if (ctx->tri_scratch) (void)hipFree(ctx->tri_scratch);
  }
  if (ctx->kmm_scratch) (void)hipFree(ctx->kmm_scratch);
  if (ctx->bal_ws) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->bal_ws);
  }
  if (ctx->chain_stream) {
    (void)hipStreamSynchronize(ctx->chain_stream);
    (void)hipStreamDestroy(ctx->chain_stream);
  }
  if (ctx->chain_go) (void)hipFree(ctx->chain_go);
  if (ctx->chain_ctr) (void)hipFree(ctx->chain_ctr);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->chol_li) (void)hipFree(ctx->chol_li);
  if (ctx->safe_bar) (void)hipFree(ctx->safe_bar);
  if (ctx->safe_retries) (void)hipFree(ctx->safe_retries);
  delete ctx;
  return AGP_OK;
}

agp_status agp_ctx_sync(agp_ctx* ctx) {
  if (!ctx) return AGP_ERR_INVALID;
  DevGuard guard(ctx->device);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

// task-graph launches of this context that lost a tile dependency and were re-run by the in-stream fallback (cumulative)
agp_status agp_ctx_task_graph_fallbacks(agp_ctx* ctx, int64_t* n_host) {
  if (!ctx || !n_host) return AGP_ERR_INVALID;
  DevGuard guard(ctx->device);
  *n_host = 0;
  if (!ctx->safe_retries) return AGP_OK;  // no launch of this context ever carried a fallback
  int32_t r = 0;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(&r, ctx->safe_retries, sizeof(r), hipMemcpyDeviceToHost));
  *n_host = r;
  return AGP_OK;
}

const char* agp_last_error(agp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

}  // extern "C"

// ---- building blocks ---------------------------------------------------------------------------------------------
template <typename T>
static agp_status bb_kernelmatrix(agp_ctx* ctx, const agp_kernel_desc* k, const void* x, int64_t n, int64_t ldx,
                                  const int64_t* idx, const void* y, int64_t p, int64_t ldy, int64_t D, void* out,
                                  int64_t ldo) {
  std::vector<T> hs(D);
  for (int64_t d = 0; d < D; ++d) hs[d] = (T)(k->ard ? k->ard_scales_host[d] : k->scale);
  T* ds = nullptr;
  AGPCHK(dmalloc(ctx, &ds, D));
  HIPCHK(ctx, hipMemcpyAsync(ds, hs.data(), sizeof(T) * D, hipMemcpyHostToDevice, ctx->stream));
  const bool sym = (y == nullptr);
  const void* yy = sym ? x : y;
  const int64_t pp = sym ? n : p, ldyy = sym ? ldx : ldy;
  dim3 g((unsigned)((pp + TILE - 1) / TILE), (unsigned)((n + TILE - 1) / TILE));
  (void)launch_kernelmatrix<T>(ctx, ctx->stream, (const T*)x, ldx, idx, n, (const T*)yy, ldyy,
                     pp, D, (const T*)ds, k->kind, (T)k->variance, (T*)out, ldo, n, pp, 0, T(0), (const T*)nullptr,
                     (T*)nullptr, (int64_t)0);
  hipError_t e = hipGetLastError();
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ds);
  if (e != hipSuccess) {
    ctx->err = hipGetErrorString(e);
    return AGP_ERR_HIP;
  }
  return AGP_OK;
}

// copy A (n x n, lda) into a padded np x np buffer (identity padding), add jitter on the diagonal
template <typename T>
static agp_status pad_spd(agp_ctx* ctx, const void* a, int64_t lda, int64_t n, double jitter, T** Ap, int64_t* np) {
  *np = rup64(n);
  AGPCHK(dmalloc(ctx, Ap, (*np) * (*np)));
  hipLaunchKernelGGL((k_copy2d<T>), grid2(*np, *np), blk2, 0, ctx->stream, (const T*)a, lda, n, n, *Ap, *np, *np, *np,
                     T(1), T(1));
  if (jitter != 0.0)
    hipLaunchKernelGGL((k_add_diag<T>), grid1(n), dim3(256), 0, ctx->stream, *Ap, *np, n, (T)jitter);
  LAUNCHCHK(ctx);
  return AGP_OK;
}

template <typename T>
static agp_status bb_potrf(agp_ctx* ctx, void* a, int64_t lda, int64_t n, double jitter, int32_t* info_host) {
  T *Ap = nullptr, *X = nullptr;
  int32_t* info = nullptr;
  int64_t np;
  AGPCHK(pad_spd<T>(ctx, a, lda, n, jitter, &Ap, &np));
  AGPCHK(dmalloc(ctx, &X, np * np));
  AGPCHK(dmalloc(ctx, &info, 1));
  HIPCHK(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
  T* Dg = nullptr;
  AGPCHK(dmalloc(ctx, &Dg, np * TILE));
  AGPCHK(potrf_fused<T>(ctx, Ap, np, np, X, np, Dg, (T*)nullptr, 0, 0, 0, info, n));
  if (chol_use_dag(ctx, np / TILE)) {
    bool lost = false;
    AGPCHK(dag_lost_dependency(ctx, info, &lost));
    if (lost) {  // the input is still intact in `a`: pad it again and factor with per-column launches
      hipLaunchKernelGGL((k_copy2d<T>), grid2(np, np), blk2, 0, ctx->stream, (const T*)a, lda, n, n, Ap, np, np, np, T(1), T(1));
      if (jitter != 0.0) hipLaunchKernelGGL((k_add_diag<T>), grid1(n), dim3(256), 0, ctx->stream, Ap, np, n, (T)jitter);
      AGPCHK(potrf_fused<T>(ctx, Ap, np, np, X, np, Dg, (T*)nullptr, 0, 0, 0, info, n));
    }
  }
  hipLaunchKernelGGL((k_publish_diag<T>), dim3((unsigned)(np / TILE)), dim3(256), 0, ctx->stream, Ap, np, (const T*)Dg);
  HIPCHK(ctx, hipMemcpy2DAsync(a, sizeof(T) * lda, Ap, sizeof(T) * np, sizeof(T) * n, n, hipMemcpyDeviceToDevice,
                               ctx->stream));
  hipLaunchKernelGGL((k_zero_strict_upper<T>), grid2(n, n), blk2, 0, ctx->stream, (T*)a, lda, n);
  int32_t hinfo = 0;
  HIPCHK(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (info_host) *info_host = hinfo;
  (void)hipFree(Ap);
  (void)hipFree(X);
  (void)hipFree(Dg);
  (void)hipFree(info);
  if (hinfo != 0) {
    ctx->err = "PosDefException: matrix is not positive definite; leading minor " + std::to_string(hinfo);
    return AGP_ERR_NOT_POSDEF;
  }
  return AGP_OK;
}

template <typename T>
static agp_status bb_spd_inverse(agp_ctx* ctx, const void* a, int64_t lda, int64_t n, void* ainv, int64_t ldi,
                                 double* logdet_host, int32_t* info_host, T** keep_inv_padded, int64_t* np_out) {
  T *Ap = nullptr, *X = nullptr, *Tw = nullptr, *Inv = nullptr;
  int32_t* info = nullptr;
  double* sc = nullptr;
  int64_t np;
  AGPCHK(pad_spd<T>(ctx, a, lda, n, 0.0, &Ap, &np));
  AGPCHK(dmalloc(ctx, &X, np * np));
  AGPCHK(dmalloc(ctx, &Tw, np * np));
  AGPCHK(dmalloc(ctx, &Inv, np * np));
  AGPCHK(dmalloc(ctx, &info, 1));
  AGPCHK(dmalloc(ctx, &sc, 1));
  HIPCHK(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
  AGPCHK(potrf_fused<T>(ctx, Ap, np, np, X, np, Tw, (T*)nullptr, 0, 0, 1, info, n));
  if (chol_use_dag(ctx, np / TILE)) {
    bool lost = false;
    AGPCHK(dag_lost_dependency(ctx, info, &lost));
    if (lost) {
      hipLaunchKernelGGL((k_copy2d<T>), grid2(np, np), blk2, 0, ctx->stream, (const T*)a, lda, n, n, Ap, np, np, np, T(1), T(1));
      AGPCHK(potrf_fused<T>(ctx, Ap, np, np, X, np, Tw, (T*)nullptr, 0, 0, 1, info, n));
    }
  }
  AGPCHK(xtx_padded<T>(ctx, X, np, np, Inv, np));
  hipLaunchKernelGGL((k_logdiag_sum<T>), dim3(1), dim3(1024), 0, ctx->stream, (const T*)Tw, n, sc);
  LAUNCHCHK(ctx);
  if (ainv)
    HIPCHK(ctx, hipMemcpy2DAsync(ainv, sizeof(T) * ldi, Inv, sizeof(T) * np, sizeof(T) * n, n, hipMemcpyDeviceToDevice,
                                 ctx->stream));
  int32_t hinfo = 0;
  double hl = 0;
  HIPCHK(ctx, hipMemcpyAsync(&hinfo, info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(&hl, sc, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (info_host) *info_host = hinfo;
  if (logdet_host) *logdet_host = 2.0 * hl;
  (void)hipFree(Ap);
  (void)hipFree(X);
  (void)hipFree(Tw);
  (void)hipFree(info);
  (void)hipFree(sc);
  if (keep_inv_padded) {
    *keep_inv_padded = Inv;
    *np_out = np;
  } else {
    (void)hipFree(Inv);
  }
  if (hinfo != 0) {
    ctx->err = "PosDefException: matrix is not positive definite; leading minor " + std::to_string(hinfo);
    return AGP_ERR_NOT_POSDEF;
  }
  return AGP_OK;
}

template <typename T>
static agp_status bb_solve_right(agp_ctx* ctx, const void* a, int64_t lda, int64_t n, const void* b, int64_t ldb,
                                 int64_t r, void* x, int64_t ldx, int32_t* info_host) {
  T* Inv = nullptr;
  int64_t np = 0;
  agp_status s = bb_spd_inverse<T>(ctx, a, lda, n, nullptr, 0, nullptr, info_host, &Inv, &np);
  if (s != AGP_OK) {
    if (Inv) (void)hipFree(Inv);
    return s;
  }
  const int64_t rp = rup64(r);
  T *Bp = nullptr, *Xp = nullptr;
  AGPCHK(dmalloc(ctx, &Bp, rp * np));
  AGPCHK(dmalloc(ctx, &Xp, rp * np));
  // zero the padded rows/cols of B (pad_diag = 0) ; padded block of Inv is the identity
  hipLaunchKernelGGL((k_copy2d_zero<T>), grid2(rp, np), blk2, 0, ctx->stream, (const T*)b, ldb, r, n, Bp, np, rp, np);
  LAUNCHCHK(ctx);
  AGPCHK((gemm_nt<T, EPI_STORE>(ctx, Bp, np, Inv, np, rp, np, np, 0, Xp, np, nullptr, 0, nullptr, nullptr, nullptr, 0)));
  HIPCHK(ctx, hipMemcpy2DAsync(x, sizeof(T) * ldx, Xp, sizeof(T) * np, sizeof(T) * n, r, hipMemcpyDeviceToDevice,
                               ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(Inv);
  (void)hipFree(Bp);
  (void)hipFree(Xp);
  return AGP_OK;
}

template <typename T>
static agp_status bb_mfma_peak(agp_ctx* ctx, double* tflops) {
  const int blocks = 256 * 8, iters = 4096;
  T* out = nullptr;
  AGPCHK(dmalloc(ctx, &out, (int64_t)blocks * NTHREADS));
  hipEvent_t e0, e1;
  HIPCHK(ctx, hipEventCreate(&e0));
  HIPCHK(ctx, hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mfma_peak<T>), dim3(blocks), dim3(NTHREADS), 0, ctx->stream, out, 64);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL((k_mfma_peak<T>), dim3(blocks), dim3(NTHREADS), 0, ctx->stream, out, iters);
  HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * (NTHREADS / 64) * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return AGP_OK;
}

template <typename T, int VAR>
static agp_status bb_diag_bench(agp_ctx* ctx, int blocks, int reps, double* us) {
  T *A = nullptr, *out = nullptr;
  int32_t* info = nullptr;
  AGPCHK(dmalloc(ctx, &A, TILE * TILE));
  AGPCHK(dmalloc(ctx, &out, (int64_t)blocks * 2 * TILE * TILE));
  AGPCHK(dmalloc(ctx, &info, 1));
  std::vector<T> h(TILE * TILE);
  for (int i = 0; i < TILE; ++i)
    for (int j = 0; j < TILE; ++j) h[i * TILE + j] = (T)((i == j ? 2.0 : 0.0) + 0.5 / (1.0 + std::abs(i - j)));
  HIPCHK(ctx, hipMemcpy(A, h.data(), sizeof(T) * TILE * TILE, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemset(info, 0, 4));
  hipEvent_t e0, e1;
  HIPCHK(ctx, hipEventCreate(&e0));
  HIPCHK(ctx, hipEventCreate(&e1));
  hipLaunchKernelGGL((k_diag_bench<T, VAR>), dim3(blocks), dim3(CHOL_THREADS), 0, ctx->stream, (const T*)A, out, 2, info);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
  hipLaunchKernelGGL((k_diag_bench<T, VAR>), dim3(blocks), dim3(CHOL_THREADS), 0, ctx->stream, (const T*)A, out, reps, info);
  HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
  *us = ms * 1e3 / reps;
  {  // residuals of block 0 (development aid): |L L' - A|, |X L - I|, strict-upper leakage
    std::vector<T> o(2 * TILE * TILE);
    HIPCHK(ctx, hipMemcpy(o.data(), out, sizeof(T) * 2 * TILE * TILE, hipMemcpyDeviceToHost));
    const T* Lh = o.data();
    const T* Xh = o.data() + TILE * TILE;
    double e1m = 0, e2m = 0, e3m = 0;
    for (int i = 0; i < TILE; ++i)
      for (int j = 0; j < TILE; ++j) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < TILE; ++k) {
          s1 += (double)Lh[i * TILE + k] * (double)Lh[j * TILE + k];
          s2 += (double)Xh[i * TILE + k] * (double)Lh[k * TILE + j];
        }
        e1m = std::max(e1m, std::abs(s1 - (double)h[i * TILE + j]));
        e2m = std::max(e2m, std::abs(s2 - (i == j ? 1.0 : 0.0)));
        if (j > i) e3m = std::max(e3m, std::abs((double)Lh[i * TILE + j]) + std::abs((double)Xh[i * TILE + j]));
      }
    const double tol = sizeof(T) == 8 ? 1e-12 : 1e-4;
    if (!(e1m < tol) || !(e2m < tol) || e3m != 0.0) {
      ctx->err = "diag_bench residual check failed";
      return AGP_ERR_NOT_POSDEF;
    }
  }
  (void)hipFree(A);
  (void)hipFree(out);
  (void)hipFree(info);
  return AGP_OK;
}


// ---- inducing-point selection: nearest-centre assignment and Lloyd iterations (agp_kmeans.h) --------------------------
template <typename T>
static agp_status km_assign(agp_ctx* ctx, const T* x, int64_t n, int64_t ldx, int64_t D, const T* c, int64_t ldc, int64_t m,
                            T* cn, int32_t* labels, T* mind) {
  const int64_t mp = rup64(m);
  const int Dp = (int)((D + 15) / 16 * 16);
  hipLaunchKernelGGL((k_km_cnorm<T>), grid1(mp), dim3(256), 0, ctx->stream, c, ldc, m, mp, D, cn);
  const size_t sh = sizeof(T) * (2 * TILE * (Dp + 2) + 4 * TILE) + sizeof(int) * 2 * TILE;
  if (sh > 64 * 1024)  // gfx950 has 160 KB of LDS per workgroup; more than 64 KB of dynamic LDS has to be requested
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_km_assign<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  hipLaunchKernelGGL((k_km_assign<T>), dim3((unsigned)((n + TILE - 1) / TILE)), dim3(NTHREADS), sh, ctx->stream, x, ldx, n, D,
                     Dp, c, ldc, m, mp, (const T*)cn, labels, mind);
  LAUNCHCHK(ctx);
  return AGP_OK;
}

template <typename T>
static agp_status km_objective(agp_ctx* ctx, const T* mind, int64_t n, double* part, double* host) {
  const int nb = 256;
  hipLaunchKernelGGL((k_km_sum_partial<T>), dim3(nb), dim3(256), 0, ctx->stream, mind, n, part);
  hipLaunchKernelGGL(k_km_sum_final, dim3(1), dim3(256), 0, ctx->stream, (const double*)part, nb, part + nb);
  LAUNCHCHK(ctx);
  HIPCHK(ctx, hipMemcpyAsync(host, part + nb, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

template <typename T>
static agp_status bb_nearest(agp_ctx* ctx, const void* x, int64_t n, int64_t ldx, int64_t D, const void* c, int64_t ldc,
                             int64_t m, int32_t* labels, void* mind) {
  if (!x || !c || n <= 0 || m <= 0 || D <= 0 || D > KM_MAXD || ldx < D || ldc < D || (!labels && !mind)) return AGP_ERR_INVALID;
  T* cn = nullptr;
  int32_t* lab = labels;
  T* md = (T*)mind;
  AGPCHK(dmalloc(ctx, &cn, rup64(m)));
  if (!lab) AGPCHK(dmalloc(ctx, &lab, n));
  if (!md) AGPCHK(dmalloc(ctx, &md, n));
  agp_status st = km_assign<T>(ctx, (const T*)x, n, ldx, D, (const T*)c, ldc, m, cn, lab, md);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(cn);
  if (!labels) (void)hipFree(lab);
  if (!mind) (void)hipFree(md);
  return st;
}

template <typename T>
static agp_status bb_kmeans(agp_ctx* ctx, const void* xv, int64_t n, int64_t ldx, int64_t D, void* cv, int64_t ldc,
                            int64_t m, int max_iter, double tol, int32_t* labels_out, int32_t* counts_out, int32_t* iters,
                            double* objective, int32_t* converged) {
  if (!xv || !cv || n <= 0 || m <= 0 || m > n || D <= 0 || D > KM_MAXD || ldx < D || ldc < D || max_iter < 0)
    return AGP_ERR_INVALID;
  const T* x = (const T*)xv;
  T* c = (T*)cv;
  const int64_t mp = rup64(m);
  const int Dp = (int)((D + 15) / 16 * 16);
  const int nchunks = (int)((n + KM_CHUNK - 1) / KM_CHUNK);
  T *cn = nullptr, *mind = nullptr, *part = nullptr;
  double* red = nullptr;
  int32_t *lab = labels_out, *blist = nullptr, *boff = nullptr;
  if (mp / TILE > KM_MAXTILES) {
    ctx->err = "agp_kmeans: more than 16384 centres";
    return AGP_ERR_UNSUPPORTED;
  }
  AGPCHK(dmalloc(ctx, &cn, mp));
  AGPCHK(dmalloc(ctx, &mind, n));
  AGPCHK(dmalloc(ctx, &part, (int64_t)nchunks * mp * (Dp + 16)));
  AGPCHK(dmalloc(ctx, &red, 512));
  AGPCHK(dmalloc(ctx, &blist, (int64_t)nchunks * KM_CHUNK));
  AGPCHK(dmalloc(ctx, &boff, (int64_t)nchunks * (KM_MAXTILES + 1)));
  if (!lab) AGPCHK(dmalloc(ctx, &lab, n));
  agp_status st = AGP_OK;
  double obj = 0.0, prev = 0.0;
  int it = 0, conv = 0;
  // Clustering.kmeans!: assignments for the seeds, then { centres <- cluster means ; assignments ; |change of cost| < tol }
  st = km_assign<T>(ctx, x, n, ldx, D, c, ldc, m, cn, lab, mind);
  if (st == AGP_OK) st = km_objective<T>(ctx, mind, n, red, &obj);
  while (st == AGP_OK && it < max_iter && !conv) {
    ++it;
    hipLaunchKernelGGL(k_km_bucket, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, (const int32_t*)lab, n,
                       (int)(mp / TILE), blist, boff);
    hipLaunchKernelGGL((k_km_sums<T>), dim3((unsigned)(mp / TILE), (unsigned)nchunks), dim3(NTHREADS), 0, ctx->stream, x, ldx, n,
                       D, Dp, (const int32_t*)lab, mp, (const int32_t*)blist, (const int32_t*)boff, part);
    hipLaunchKernelGGL((k_km_finish<T>), grid1(m * D), dim3(256), 0, ctx->stream, (const T*)part, nchunks, mp, Dp, m, D, c, ldc,
                       counts_out);
    if (hipGetLastError() != hipSuccess) {
      st = AGP_ERR_HIP;
      break;
    }
    st = km_assign<T>(ctx, x, n, ldx, D, c, ldc, m, cn, lab, mind);
    prev = obj;
    if (st == AGP_OK) st = km_objective<T>(ctx, mind, n, red, &obj);
    if (m == 1 || std::abs(obj - prev) < tol) conv = 1;
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(cn);
  (void)hipFree(mind);
  (void)hipFree(part);
  (void)hipFree(red);
  (void)hipFree(blist);
  (void)hipFree(boff);
  if (!labels_out) (void)hipFree(lab);
  if (iters) *iters = it;
  if (objective) *objective = obj;
  if (converged) *converged = conv;
  return st;
}

#define DISPATCH(dtype, call_f64, call_f32)        \
  do {                                             \
    DevGuard _dev_guard(ctx->device);              \
    if ((dtype) == AGP_F64) return call_f64;       \
    if ((dtype) == AGP_F32) return call_f32;       \
    return AGP_ERR_INVALID;                        \
  } while (0)

extern "C" {

agp_status agp_kernelmatrix(agp_ctx* ctx, int32_t dtype, const agp_kernel_desc* k, const void* x, int64_t n, int64_t ldx,
                            const int64_t* idx, const void* y, int64_t p, int64_t ldy, int64_t D, void* out,
                            int64_t ldo) {
  if (!ctx || !k || !x || !out || n <= 0 || D <= 0) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_kernelmatrix<double>(ctx, k, x, n, ldx, idx, y, p, ldy, D, out, ldo),
           bb_kernelmatrix<float>(ctx, k, x, n, ldx, idx, y, p, ldy, D, out, ldo));
}

agp_status agp_potrf_jitter(agp_ctx* ctx, int32_t dtype, void* a, int64_t lda, int64_t n, double jitter,
                            int32_t* info_host) {
  if (!ctx || !a || n <= 0 || lda < n) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_potrf<double>(ctx, a, lda, n, jitter, info_host), bb_potrf<float>(ctx, a, lda, n, jitter, info_host));
}

agp_status agp_spd_inverse(agp_ctx* ctx, int32_t dtype, const void* a, int64_t lda, int64_t n, void* ainv, int64_t ldi,
                           double* logdet_host, int32_t* info_host) {
  if (!ctx || !a || !ainv || n <= 0 || lda < n || ldi < n) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_spd_inverse<double>(ctx, a, lda, n, ainv, ldi, logdet_host, info_host, nullptr, nullptr),
           bb_spd_inverse<float>(ctx, a, lda, n, ainv, ldi, logdet_host, info_host, nullptr, nullptr));
}

agp_status agp_solve_right_spd(agp_ctx* ctx, int32_t dtype, const void* a, int64_t lda, int64_t n, const void* b,
                               int64_t ldb, int64_t r, void* x, int64_t ldx, int32_t* info_host) {
  if (!ctx || !a || !b || !x || n <= 0 || r <= 0) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_solve_right<double>(ctx, a, lda, n, b, ldb, r, x, ldx, info_host),
           bb_solve_right<float>(ctx, a, lda, n, b, ldb, r, x, ldx, info_host));
}

agp_status agp_mfma_peak(agp_ctx* ctx, int32_t dtype, double* tflops_host) {
  if (!ctx || !tflops_host) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_mfma_peak<double>(ctx, tflops_host), bb_mfma_peak<float>(ctx, tflops_host));
}

// development micro-benchmark (not part of include/agp_hip.h): microseconds per 64x64 diagonal-tile factorisation
agp_status agp_nearest_center(agp_ctx* ctx, int32_t dtype, const void* x, int64_t n, int64_t ldx, int64_t D,
                              const void* centers, int64_t ldc, int64_t m, int32_t* labels_out, void* mind_out) {
  if (!ctx) return AGP_ERR_INVALID;
  DISPATCH(dtype, bb_nearest<double>(ctx, x, n, ldx, D, centers, ldc, m, labels_out, mind_out),
           bb_nearest<float>(ctx, x, n, ldx, D, centers, ldc, m, labels_out, mind_out));
}

agp_status agp_kmeans(agp_ctx* ctx, int32_t dtype, const void* x, int64_t n, int64_t ldx, int64_t D, void* centers,
                      int64_t ldc, int64_t m, int32_t max_iter, double tol, int32_t* labels_out, int32_t* counts_out,
                      int32_t* iters_host, double* objective_host, int32_t* converged_host) {
  if (!ctx) return AGP_ERR_INVALID;
  DISPATCH(dtype,
           bb_kmeans<double>(ctx, x, n, ldx, D, centers, ldc, m, max_iter, tol, labels_out, counts_out, iters_host,
                             objective_host, converged_host),
           bb_kmeans<float>(ctx, x, n, ldx, D, centers, ldc, m, max_iter, tol, labels_out, counts_out, iters_host,
                            objective_host, converged_host));
}

// development / test hook (not part of include/agp_hip.h): how many task-graph factorisations the in-stream fallback re-ran
agp_status agp_dev_dag_retries(agp_ctx* ctx, int64_t* n) {
  if (!ctx || !n) return AGP_ERR_INVALID;
  DevGuard guard(ctx->device);
  *n = 0;
  if (ctx->safe_retries) {
    int32_t r = 0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(&r, ctx->safe_retries, sizeof(r), hipMemcpyDeviceToHost));
    *n = r;
  }
  return AGP_OK;
}

agp_status agp_dev_diag_bench(agp_ctx* ctx, int32_t dtype, int32_t variant, int32_t blocks, int32_t reps, double* us) {
  if (!ctx || !us) return AGP_ERR_INVALID;
  DevGuard guard(ctx->device);
  switch (variant * 2 + (dtype == AGP_F64 ? 0 : 1)) {
    case 0: return bb_diag_bench<double, 0>(ctx, blocks, reps, us);
    case 1: return bb_diag_bench<float, 0>(ctx, blocks, reps, us);
    case 2: return bb_diag_bench<double, 1>(ctx, blocks, reps, us);
    case 3: return bb_diag_bench<float, 1>(ctx, blocks, reps, us);
    case 4: return bb_diag_bench<double, 2>(ctx, blocks, reps, us);
    case 5: return bb_diag_bench<float, 2>(ctx, blocks, reps, us);
    case 6: return bb_diag_bench<double, 3>(ctx, blocks, reps, us);  // timing experiments (residual check fails by design)
    case 8: return bb_diag_bench<double, 4>(ctx, blocks, reps, us);
    case 10: return bb_diag_bench<double, 5>(ctx, blocks, reps, us);
    case 12: return bb_diag_bench<double, 6>(ctx, blocks, reps, us);
    case 14: return bb_diag_bench<double, 7>(ctx, blocks, reps, us);
    case 16: return bb_diag_bench<double, 8>(ctx, blocks, reps, us);
    case 17: return bb_diag_bench<float, 8>(ctx, blocks, reps, us);
    case 18: return bb_diag_bench<double, 9>(ctx, blocks, reps, us);
    case 20: return bb_diag_bench<double, 10>(ctx, blocks, reps, us);
    case 22: return bb_diag_bench<double, 11>(ctx, blocks, reps, us);
    case 24: return bb_diag_bench<double, 12>(ctx, blocks, reps, us);
    default: return AGP_ERR_INVALID;
  }
}

agp_status agp_svgp_create(agp_ctx* ctx, const agp_svgp_desc* desc, agp_svgp** out) {
  if (!ctx || !desc || !out) return AGP_ERR_INVALID;
  DevGuard guard(ctx->device);
  *out = nullptr;
  SvgpBase* impl = nullptr;
  if (desc->dtype == AGP_F64) impl = new Svgp<double>();
  else if (desc->dtype == AGP_F32) impl = new Svgp<float>();
  else return AGP_ERR_INVALID;
  impl->ctx = ctx;
  impl->desc = *desc;
  agp_status s = impl->init();
  if (s != AGP_OK) {
    delete impl;
    return s;
  }
  agp_svgp* h = new agp_svgp();
  h->impl = impl;
  *out = h;
  return AGP_OK;
}

agp_status agp_svgp_destroy(agp_svgp* h) {
  if (!h) return AGP_OK;
  DevGuard guard(h->impl->ctx->device);
  (void)hipStreamSynchronize(h->impl->ctx->stream);
  delete h->impl;
  delete h;
  return AGP_OK;
}

#define HCHK(h)                                    \
  if (!(h) || !(h)->impl) return AGP_ERR_INVALID;  \
  DevGuard _dev_guard((h)->impl->ctx->device)
// ... and, for every entry point outside the CAVI step / look-ahead pair: take a pending natural-gradient step first
#define HCHKF(h) \
  HCHK(h);       \
  AGPCHK((h)->impl->flush())

agp_status agp_svgp_set_kernel(agp_svgp* h, int32_t latent, const agp_kernel_desc* k) {
  HCHKF(h);
  return h->impl->set_kernel(latent, k);
}
agp_status agp_svgp_set_Z(agp_svgp* h, int32_t latent, const void* z, int64_t ldz) {
  HCHKF(h);
  return h->impl->set_Z(latent, z, ldz);
}
agp_status agp_svgp_get_Z(agp_svgp* h, int32_t latent, void* z, int64_t ldz) {
  HCHKF(h);
  return h->impl->get_Z(latent, z, ldz);
}
agp_status agp_svgp_set_prior_mean(agp_svgp* h, int32_t latent, const void* mu0) {
  HCHKF(h);
  return h->impl->set_mu0(latent, mu0);
}
agp_status agp_svgp_refresh_K(agp_svgp* h) {
  HCHKF(h);
  return h->impl->refresh_K_explicit();
}
agp_status agp_svgp_set_opt_state(agp_svgp* h, int64_t n) {
  HCHK(h);
  if (n < 1) return AGP_ERR_INVALID;
  h->impl->n_opt = n;
  return AGP_OK;
}
agp_status agp_svgp_get_opt_state(agp_svgp* h, int64_t* n_host) {
  HCHK(h);
  *n_host = h->impl->n_opt;
  return AGP_OK;
}

agp_status agp_svgp_cavi_step(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                              double rho) {
  HCHK(h);
  SvgpBase* s = h->impl;
  s->in_cavi_step = true;
  const agp_status sl_ = s->step_local(x, ldx, y, idx, B, rho, false);
  s->in_cavi_step = false;
  AGPCHK(sl_);
  AGPCHK(s->lsm_local_all());
  s->n_steps += 1;
  return s->step_finish();
}
agp_status agp_svgp_step_counters(agp_svgp* h, int64_t* n_steps_host, int64_t* n_prologue_host) {
  HCHK(h);
  if (n_steps_host) *n_steps_host = h->impl->n_steps;
  if (n_prologue_host) *n_prologue_host = h->impl->n_prologue;
  return AGP_OK;
}

agp_status agp_svgp_hyper_counters(agp_svgp* h, int64_t* n_grad_host, int64_t* n_gk_fused_host) {
  HCHK(h);
  if (n_grad_host) *n_grad_host = h->impl->n_hgrad;
  if (n_gk_fused_host) *n_gk_fused_host = h->impl->n_gk_fused;
  return AGP_OK;
}

agp_status agp_svgp_step_local(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                               double rho) {
  HCHKF(h);
  return h->impl->step_local(x, ldx, y, idx, B, rho, false);
}
agp_status agp_svgp_prefetch(agp_svgp* h, const void* x, int64_t ldx, const int64_t* idx, int64_t B) {
  HCHK(h);
  return h->impl->prefetch(x, ldx, idx, B);
}
agp_status agp_svgp_set_multioutput(agp_svgp* h, int32_t n_task, const agp_lik_desc* liks_host, const double* A_host,
                                    double adam_eta, double adam_b1, double adam_b2, double adam_eps) {
  HCHKF(h);
  return h->impl->set_multioutput(n_task, liks_host, A_host, adam_eta, adam_b1, adam_b2, adam_eps);
}
agp_status agp_svgp_get_A(agp_svgp* h, double* A_host) {
  HCHKF(h);
  return h->impl->get_A(A_host);
}
agp_status agp_svgp_elbo_terms(agp_svgp* h, double* terms_host) {
  HCHKF(h);
  return h->impl->elbo_terms(terms_host);
}
agp_status agp_svgp_set_batch_shard(agp_svgp* h, int32_t rank, int32_t world) {
  HCHKF(h);
  return h->impl->set_batch_shard(rank, world);
}
agp_status agp_svgp_mo_shard(agp_svgp* h, int32_t q_total) {
  HCHKF(h);
  return h->impl->mo_shard(q_total);
}
agp_status agp_svgp_mo_fbuf_ptr(agp_svgp* h, void** ptr, int64_t* count) {
  HCHKF(h);
  return h->impl->mo_fbuf_ptr(ptr, count);
}
agp_status agp_svgp_mo_mix(agp_svgp* h) {
  HCHKF(h);
  return h->impl->mo_mix();
}
agp_status agp_svgp_mo_refresh_f(agp_svgp* h) {
  HCHKF(h);
  return h->impl->mo_refresh_f();
}
agp_status agp_svgp_mo_predict_from_f(agp_svgp* h, int64_t n_t, int32_t mode, void* out0, void* out1,
                                      const double* gh_nodes_host, const double* gh_weights_host, int32_t n_nodes) {
  HCHKF(h);
  return h->impl->mo_predict_from_f(n_t, mode, out0, out1, gh_nodes_host, gh_weights_host, n_nodes);
}
agp_status agp_svgp_elbo_enqueue(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B, double rho,
                                 int32_t fresh_local, int32_t* ticket) {
  HCHKF(h);
  return h->impl->elbo_enqueue(x, ldx, y, idx, B, rho, fresh_local, ticket);
}
agp_status agp_svgp_elbo_fetch(agp_svgp* h, int32_t ticket, int32_t wait, double* out_host, int32_t* ready) {
  HCHK(h);
  return h->impl->elbo_fetch(ticket, wait, out_host, ready);
}

agp_status agp_svgp_hyper_rule(agp_svgp* h, int32_t kernel_rule, double kernel_rho, int32_t z_rule, double z_rho) {
  HCHK(h);
  return h->impl->hyper_rule(kernel_rule, kernel_rho, z_rule, z_rho);
}

agp_status agp_svgp_hyper_configure(agp_svgp* h, int32_t opt_kernel, double kernel_eta, int32_t opt_Z, double z_eta,
                                    double adam_b1, double adam_b2, double adam_eps) {
  HCHKF(h);
  return h->impl->hyper_configure(opt_kernel, kernel_eta, opt_Z, z_eta, adam_b1, adam_b2, adam_eps);
}
agp_status agp_svgp_hypergrad(agp_svgp* h, int32_t latent, double* dvariance_host, double* dscale_host, void* dZ) {
  HCHKF(h);
  return h->impl->hypergrad(latent, dvariance_host, dscale_host, dZ);
}
agp_status agp_svgp_hyper_step(agp_svgp* h) {
  HCHK(h);  // (no flush: a pending natural-gradient step becomes the prologue of the factorisation the gradient needs, aug_factor)
  return h->impl->hyper_step();
}
agp_status agp_svgp_get_kernel(agp_svgp* h, int32_t latent, double* variance_host, double* scales_host) {
  HCHKF(h);
  return h->impl->get_kernel(latent, variance_host, scales_host);
}
agp_status agp_svgp_lsm_gamma(agp_svgp* h) {
  HCHKF(h);
  return h->impl->lsm_gamma();
}
agp_status agp_svgp_lsm_alpha(agp_svgp* h) {
  HCHKF(h);
  return h->impl->lsm_alpha();
}
agp_status agp_svgp_lsm_gsum_ptr(agp_svgp* h, void** ptr, int64_t* count) {
  HCHKF(h);
  return h->impl->lsm_gsum_ptr(ptr, count);
}
agp_status agp_svgp_step_stats(agp_svgp* h) {
  HCHKF(h);
  return h->impl->step_stats(false);
}
#ifdef AGP_DEBUG_PTRS
void* agp_debug_ptr(agp_svgp* h, int what) { return (h && h->impl) ? h->impl->debug_ptr(what) : nullptr; }
// the words of agp_dag_diag (agp_chol.h) + [6] the context's flag block, [7] int32 words per flag, DAG_FS (so that [1] can be turned
// into a flag index: ([1] - [6]) / 4 / [7]); synchronises the device
int agp_debug_dag_diag(agp_ctx* c, unsigned long long* out8) {
  if (!out8) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(agp_dag_diag), 8 * sizeof(unsigned long long)) != hipSuccess) return 3;
  out8[6] = c ? (unsigned long long)c->dag_flags : 0ull;
  out8[7] = (unsigned long long)DAG_FS;
  return 0;
}
#endif
agp_status agp_svgp_stats_ptr(agp_svgp* h, void** ptr, int64_t* count) {
  HCHKF(h);
  return h->impl->stats_ptr(ptr, count);
}
agp_status agp_svgp_step_global(agp_svgp* h) {
  HCHKF(h);
  return h->impl->step_global(false);
}
agp_status agp_svgp_timing_enable(agp_svgp* h, int32_t on) {
  HCHK(h);
  h->impl->timing = on != 0;
  h->impl->timing_every = on > 1 ? on : 1;
  h->impl->timing_ctr = 0;
  return AGP_OK;
}
agp_status agp_svgp_timing_read(agp_svgp* h, int64_t* n_launches_host, double* total_ms_host) {
  HCHK(h);
  if (!n_launches_host || !total_ms_host) return AGP_ERR_INVALID;
  return h->impl->timing_read(n_launches_host, total_ms_host);
}
agp_status agp_svgp_check_status(agp_svgp* h) {
  HCHKF(h);
  return h->impl->check_status();
}
agp_status agp_svgp_elbo(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                         double rho, int32_t fresh_local, double* elbo_host) {
  HCHKF(h);
  if (!elbo_host) return AGP_ERR_INVALID;
  return h->impl->elbo(x, ldx, y, idx, B, rho, fresh_local, elbo_host);
}
agp_status agp_svgp_get_state(agp_svgp* h, int32_t latent, void* mu, void* sigma, void* eta1, void* eta2) {
  HCHKF(h);
  return h->impl->get_state(latent, mu, sigma, eta1, eta2);
}
agp_status agp_svgp_set_state(agp_svgp* h, int32_t latent, const void* eta1, const void* eta2) {
  HCHKF(h);
  return h->impl->set_state(latent, eta1, eta2);
}
agp_status agp_svgp_get_matrix(agp_svgp* h, int32_t latent, int32_t which, void* out, int64_t ldo, int64_t cap) {
  HCHKF(h);
  return h->impl->get_matrix(latent, which, out, ldo, cap);
}
agp_status agp_svgp_last_batch(agp_svgp* h, int64_t* B_host) {
  HCHK(h);
  if (!B_host) return AGP_ERR_INVALID;
  *B_host = h->impl->last_batch();
  return AGP_OK;
}
agp_status agp_svgp_invalidate_data(agp_svgp* h) {
  HCHKF(h);
  return h->impl->invalidate_data();
}
agp_status agp_svgp_init_state(agp_svgp* h) {
  HCHKF(h);
  return h->impl->init_state();
}
agp_status agp_svgp_predict_f_cov(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* mu_out, void* cov_out) {
  HCHKF(h);
  if (n_t == 0) return AGP_OK;  // no test points: nothing to write (the reference returns empty arrays)
  return h->impl->predict_f_cov(xt, ldx, n_t, mu_out, cov_out);
}
agp_status agp_svgp_predict_f(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* mu_out, void* var_out) {
  HCHKF(h);
  if (n_t == 0) return AGP_OK;  // no test points: nothing to write (the reference returns empty arrays)
  return h->impl->predict_f(xt, ldx, n_t, mu_out, var_out);
}
agp_status agp_svgp_predict_y(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* y_out) {
  HCHKF(h);
  if (n_t == 0) return AGP_OK;  // no test points: nothing to write (the reference returns empty arrays)
  return h->impl->predict_y(xt, ldx, n_t, y_out);
}
agp_status agp_svgp_set_online_prior(agp_svgp* h, int32_t latent, const void* za, int64_t ldza, int64_t ma, const void* invDa,
                                     int64_t ldi, const void* prev_eta1, double prevLa) {
  HCHKF(h);
  return h->impl->set_online_prior(latent, za, ldza, ma, invDa, ldi, prev_eta1, prevLa);
}
agp_status agp_svgp_online_snapshot(agp_svgp* h, int32_t latent, void* invDa_out, int64_t ldi, void* eta1_out,
                                    double* prevLa_host) {
  HCHKF(h);
  return h->impl->online_snapshot(latent, invDa_out, ldi, eta1_out, prevLa_host);
}
agp_status agp_svgp_adopt_local(agp_svgp* dst, agp_svgp* src) {
  HCHKF(dst);
  if (!src || !src->impl) return AGP_ERR_INVALID;
  AGPCHK(src->impl->flush());
  return dst->impl->adopt_local(src->impl);
}
agp_status agp_svgp_online_first_step(agp_svgp* h_new, agp_svgp* h_old, const void* x, int64_t ldx, const void* y,
                                      int64_t B) {
  HCHKF(h_new);
  if (!h_old || !h_old->impl) return AGP_ERR_INVALID;
  AGPCHK(h_old->impl->flush());
  SvgpBase *n = h_new->impl, *o = h_old->impl;
  // local update of the new batch under the OLD inducing points and posterior (compute_old_matrices, onlinetraining.jl:80-89)
  AGPCHK(o->step_local(x, ldx, y, nullptr, B, 1.0, true));
  if (o->desc.lik.kind == AGP_LIK_LOGISTICSOFTMAX) {
    for (int it = 0; it < 2; ++it) {
      AGPCHK(o->lsm_gamma());
      AGPCHK(o->lsm_alpha());
    }
  }
  // kernel matrices of the new inducing points, then the online natural gradient with those expectation gradients (:93-104)
  AGPCHK(n->step_local(x, ldx, y, nullptr, B, 1.0, true));
  AGPCHK(n->adopt_local(o));
  AGPCHK(n->step_stats(true));
  return n->step_global(true);
}

agp_status agp_svgp_hyper_apply(agp_svgp* h, int32_t latent, const double* dvariance_host, const double* dscale_host,
                                const void* dZ) {
  HCHKF(h);
  return h->impl->hyper_apply(latent, dvariance_host, dscale_host, dZ);
}
agp_status agp_svgp_hyper_opt_state(agp_svgp* h, int32_t latent, int32_t set, double* k_m_host, double* k_v_host,
                                    int32_t* k_step_host) {
  HCHKF(h);
  return h->impl->hyper_state(latent, set, k_m_host, k_v_host, k_step_host);
}
agp_status agp_svgp_set_quadrature(agp_svgp* h, const double* gh_nodes_host, const double* gh_weights_host,
                                   int32_t n_nodes) {
  HCHKF(h);
  return h->impl->set_quadrature(gh_nodes_host, gh_weights_host, n_nodes);
}
agp_status agp_svgp_set_lsm_alpha(agp_svgp* h, const void* alpha, int64_t n) {
  HCHKF(h);
  return h->impl->set_lsm_alpha(alpha, n);
}
agp_status agp_svgp_get_lik_param(agp_svgp* h, double* out) {
  HCHKF(h);
  return h->impl->get_lik_param(out);
}
agp_status agp_svgp_set_lik_param(agp_svgp* h, double value) {
  HCHKF(h);
  return h->impl->set_lik_param(value);
}


// ---- collectives (agp_comm.h) ------------------------------------------------------------------------------------
agp_status agp_comm_unique_id(uint8_t* id_host) {
  if (!id_host) return AGP_ERR_INVALID;
  std::string why;
  RcclApi* api = rccl_api(why);
  if (!api) return AGP_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return AGP_ERR_HIP;
  static_assert(sizeof(id) == AGP_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_host, &id, sizeof(id));
  return AGP_OK;
}

agp_status agp_comm_init(agp_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id_host, agp_comm** out) {
  if (!ctx || !out || !id_host || world < 1 || rank < 0 || rank >= world) return AGP_ERR_INVALID;
  *out = nullptr;
  DevGuard guard(ctx->device);
  std::string why;
  RcclApi* api = rccl_api(why);
  if (!api) {
    ctx->err = "agp_comm_init: " + why;
    return AGP_ERR_UNSUPPORTED;
  }
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t r = api->CommInitRank(&c, world, id, rank);  // binds to the current device = ctx->device
  if (r != ncclSuccess) {
    ctx->err = std::string("ncclCommInitRank : ") + api->GetErrorString(r) + " [" + api->where + "]";
    return AGP_ERR_HIP;
  }
  agp_comm* cm = new agp_comm();
  cm->ctx = ctx;
  cm->rank = rank;
  cm->world = world;
  cm->kind = 0;
  cm->nccl = c;
  *out = cm;
  return AGP_OK;
}

agp_status agp_comm_init_callback(agp_ctx* ctx, int32_t rank, int32_t world, agp_allreduce_fn fn, void* user,
                                  agp_comm** out) {
  if (!ctx || !out || !fn || world < 1 || rank < 0 || rank >= world) return AGP_ERR_INVALID;
  agp_comm* cm = new agp_comm();
  cm->ctx = ctx;
  cm->rank = rank;
  cm->world = world;
  cm->kind = 1;
  cm->fn = fn;
  cm->user = user;
  *out = cm;
  return AGP_OK;
}

agp_status agp_comm_destroy(agp_comm* cm) {
  if (!cm) return AGP_OK;
  DevGuard guard(cm->ctx->device);
  (void)hipStreamSynchronize(cm->ctx->stream);
  for (auto e : cm->ev) (void)hipEventDestroy(e);
  if (cm->side) {
    (void)hipStreamSynchronize(cm->side);
    (void)hipStreamDestroy(cm->side);
  }
  if (cm->ev_in) (void)hipEventDestroy(cm->ev_in);
  if (cm->ev_out) (void)hipEventDestroy(cm->ev_out);
  if (cm->kind == 0 && cm->nccl) {
    std::string why;
    RcclApi* api = rccl_api(why);
    if (api) (void)api->CommDestroy(cm->nccl);
  }
  delete cm;
  return AGP_OK;
}

agp_status agp_comm_info(agp_comm* cm, int32_t* rank_host, int32_t* world_host, int32_t* is_rccl_host) {
  if (!cm) return AGP_ERR_INVALID;
  if (rank_host) *rank_host = cm->rank;
  if (world_host) *world_host = cm->world;
  if (is_rccl_host) *is_rccl_host = cm->kind == 0;
  return AGP_OK;
}

}  // extern "C"

// one in-place sum on `stream` through whichever transport the communicator has
static agp_status comm_issue(agp_comm* cm, void* buf, int64_t count, int32_t dtype, hipStream_t stream) {
  agp_ctx* ctx = cm->ctx;
  if (cm->kind == 0) {
    std::string why;
    RcclApi* api = rccl_api(why);
    if (!api) return AGP_ERR_UNSUPPORTED;
    ncclResult_t r = api->AllReduce(buf, buf, (size_t)count, dtype == AGP_F64 ? ncclFloat64 : ncclFloat32, ncclSum, cm->nccl, stream);
    if (r != ncclSuccess) {
      ctx->err = std::string("ncclAllReduce : ") + api->GetErrorString(r);
      return AGP_ERR_HIP;
    }
  } else {
    const int32_t r = cm->fn(cm->user, buf, count, dtype, (void*)stream);
    if (r != 0) {
      ctx->err = "agp_comm: the host all-reduce callback failed with code " + std::to_string(r);
      return AGP_ERR_HIP;
    }
  }
  return AGP_OK;
}
static agp_status comm_timing_begin(agp_comm* cm, hipStream_t stream) {
  agp_ctx* ctx = cm->ctx;
  cm->timing_now = cm->timing && ((cm->n_calls - 1) % cm->timing_every == 0);
  if (cm->timing_now) {
    if (cm->ev_used + 2 > cm->ev.size())
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreate(&e));
        cm->ev.push_back(e);
      }
    HIPCHK(ctx, hipEventRecord(cm->ev[cm->ev_used], stream));
  }
  return AGP_OK;
}
static agp_status comm_timing_end(agp_comm* cm, hipStream_t stream) {
  agp_ctx* ctx = cm->ctx;
  if (cm->timing_now) {
    HIPCHK(ctx, hipEventRecord(cm->ev[cm->ev_used + 1], stream));
    cm->ev_used += 2;
    cm->n_timed += 1;
  }
  return AGP_OK;
}

// AGP_SPLIT_OVERLAP: the batch-parallel statistics as a train of `ng` ranges (block-column groups, agp_chol.h pack_index) on the
// communicator's OWN stream, ordered after what the ctx's stream holds now; after range g a one-thread kernel stores `epoch` into
// arrive[g * ARRIVE_STRIDE], which the tile workgroups of the NEXT task-graph launch poll (pro_arrival_gate) -- that launch is
// enqueued on the ctx's stream right away and starts on group 0 while the others are still travelling.  Nothing on the ctx's stream
// waits for the train (cm->ev_out is there for a flush that wants the whole statistic).  Counted as ONE collective call of the sum
// of the ranges for agp_comm_stats; with timing on, the events bracket the whole train on the side stream.
static agp_status comm_allreduce_groups(agp_comm* cm, void* base, int esz, const int64_t* off, const int64_t* cnt, int ng,
                                        int32_t dtype, int32_t* arrive, int32_t epoch) {
  agp_ctx* ctx = cm->ctx;
  DevGuard guard(ctx->device);
  if (!cm->side) {
    int lo = 0, hi = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(ctx, hipStreamCreateWithPriority(&cm->side, hipStreamNonBlocking, hi));
    HIPCHK(ctx, hipEventCreateWithFlags(&cm->ev_in, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&cm->ev_out, hipEventDisableTiming));
  }
  HIPCHK(ctx, hipEventRecord(cm->ev_in, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(cm->side, cm->ev_in, 0));
  cm->n_calls += 1;
  AGPCHK(comm_timing_begin(cm, cm->side));
  for (int g = 0; g < ng; ++g) {
    cm->bytes += cnt[g] * esz;
    AGPCHK(comm_issue(cm, (char*)base + off[g] * esz, cnt[g], dtype, cm->side));
    hipLaunchKernelGGL(k_set_arrive, dim3(1), dim3(1), 0, cm->side, arrive + g * ARRIVE_STRIDE, epoch);
    LAUNCHCHK(ctx);
  }
  AGPCHK(comm_timing_end(cm, cm->side));
  HIPCHK(ctx, hipEventRecord(cm->ev_out, cm->side));
  return AGP_OK;
}

// ---- stand-in for an xGMI ring all-reduce on a one-GPU box (diagnostic; agp_comm_standin_allreduce) -------------------------------
// RCCL's ring kernel is a handful of workgroups ("channels") that pass chunks to one another through flags: it only finishes when all
// of them are RESIDENT at the same time.  A sleep kernel in the collective's place does not exercise that; this one does: n_wg
// workgroups meet at a grid barrier (every one must have a CU), move the buffer through their registers chunk by chunk (sum over one
// rank = the values they found), meet again, and do not return before min_us have passed.  The barrier counter only grows (two
// arrivals per workgroup and launch): launches on one stream are ordered, which is how the callback transport uses it.  The spins
// are bounded (about a second): a stand-in that cannot become resident reports it through `stuck` instead of hanging the device.
__device__ unsigned long long g_standin_bar;
__global__ void k_standin_allreduce(unsigned long long* __restrict__ buf8, int64_t n8, unsigned long long base, double min_us,
                                    int32_t* __restrict__ stuck) {
  const unsigned long long t0 = wall_clock64();  // 100 MHz constant clock
  const unsigned long long n_wg = gridDim.x;
  __shared__ int ok;
  auto meet = [&](unsigned long long target) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&g_standin_bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      ok = 1;
      while (__hip_atomic_load(&g_standin_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1L << 22)) {
          ok = 0;
          if (stuck) atomicAdd(stuck, 1);
          break;
        }
      }
    }
    __syncthreads();
    return ok != 0;
  };
  const bool all_here = meet(base + n_wg);
  if (all_here)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)n_wg * blockDim.x) {
      const unsigned long long v = __hip_atomic_load(buf8 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(buf8 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  (void)meet(base + 2 * n_wg);
  if (threadIdx.x == 0)
    while ((double)(wall_clock64() - t0) * 0.01 < min_us) __builtin_amdgcn_s_sleep(16);
}
static unsigned long long g_standin_launches = 0;

extern "C" {

agp_status agp_comm_standin_allreduce(void* buf, int64_t count, int32_t dtype, void* stream, int32_t n_wg, int32_t n_threads,
                                      double min_us, int32_t* stuck_dev) {
  if (!buf || count <= 0 || (dtype != AGP_F64 && dtype != AGP_F32) || n_wg < 1 || n_wg > 64 || n_threads < 64 || n_threads > 1024)
    return AGP_ERR_INVALID;
  const int64_t n8 = count * (dtype == AGP_F64 ? 8 : 4) / 8;  // whole 8-byte words (a trailing float stays where it is)
  const unsigned long long base = g_standin_launches * 2ull * (unsigned long long)n_wg;
  // (the counter's base assumes one n_wg per process: the stand-in is a diagnostic of ONE configuration per run)
  g_standin_launches += 1;
  hipLaunchKernelGGL(k_standin_allreduce, dim3((unsigned)n_wg), dim3((unsigned)n_threads), 0, (hipStream_t)stream,
                     (unsigned long long*)buf, n8, base, min_us, stuck_dev);
  return hipGetLastError() == hipSuccess ? AGP_OK : AGP_ERR_HIP;
}

agp_status agp_comm_allreduce(agp_comm* cm, void* buf, int64_t count, int32_t dtype) {
  if (!cm || !buf || count <= 0 || (dtype != AGP_F64 && dtype != AGP_F32)) return AGP_ERR_INVALID;
  agp_ctx* ctx = cm->ctx;
  DevGuard guard(ctx->device);
  cm->n_calls += 1;
  cm->bytes += count * (dtype == AGP_F64 ? 8 : 4);
  AGPCHK(comm_timing_begin(cm, ctx->stream));
  AGPCHK(comm_issue(cm, buf, count, dtype, ctx->stream));
  return comm_timing_end(cm, ctx->stream);
}

agp_status agp_comm_timing(agp_comm* cm, int32_t on) {
  if (!cm) return AGP_ERR_INVALID;
  cm->timing = on != 0;
  cm->timing_every = on > 1 ? on : 1;
  return AGP_OK;
}

agp_status agp_comm_stats(agp_comm* cm, int64_t* n_calls_host, int64_t* bytes_host, double* ms_host) {
  if (!cm) return AGP_ERR_INVALID;
  agp_ctx* ctx = cm->ctx;
  DevGuard guard(ctx->device);
  if (n_calls_host) *n_calls_host = cm->n_calls;
  if (bytes_host) *bytes_host = cm->bytes;
  double ms = 0.0;
  if (cm->ev_used) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (cm->side) HIPCHK(ctx, hipStreamSynchronize(cm->side));
    for (size_t i = 0; i + 1 < cm->ev_used; i += 2) {
      float t = 0;
      HIPCHK(ctx, hipEventElapsedTime(&t, cm->ev[i], cm->ev[i + 1]));
      ms += t;
    }
  }
  // with every n-th collective bracketed the sum is scaled to all of them
  if (cm->n_timed > 0 && cm->n_timed < cm->n_calls) ms *= (double)cm->n_calls / (double)cm->n_timed;
  if (ms_host) *ms_host = ms;
  cm->n_calls = cm->bytes = 0;
  cm->n_timed = 0;
  cm->ev_used = 0;
  return AGP_OK;
}

agp_status agp_svgp_cavi_step_multi(agp_svgp* h, agp_comm* comm, int32_t mode, const void* x, int64_t ldx, const void* y,
                                    const int64_t* idx, int64_t B, double rho) {
  HCHK(h);  // (no flush: like agp_svgp_cavi_step, the step takes a pending natural-gradient step itself -- as its prologue)
  h->impl->n_steps += 1;
  return h->impl->cavi_step_multi(comm, mode, x, ldx, y, idx, B, rho);
}
agp_status agp_svgp_elbo_multi(agp_svgp* h, agp_comm* comm, int32_t mode, double* elbo_host) {
  HCHKF(h);
  return h->impl->elbo_multi(comm, mode, elbo_host);
}
agp_status agp_svgp_hyper_step_multi(agp_svgp* h, agp_comm* comm, int32_t tied) {
  HCHKF(h);
  return h->impl->hyper_step_multi(comm, tied);
}
agp_status agp_svgp_predict_multi(agp_svgp* h, agp_comm* comm, int32_t what, const void* xt, int64_t ldx, int64_t n_t,
                                  void* mu_out, void* var_out, const double* gh_nodes_host, const double* gh_weights_host,
                                  int32_t n_nodes) {
  HCHKF(h);
  if (n_t == 0) return AGP_OK;  // no test points: nothing to write (the reference returns empty arrays)
  return h->impl->predict_multi(comm, what, xt, ldx, n_t, mu_out, var_out, gh_nodes_host, gh_weights_host, n_nodes);
}

agp_status agp_svgp_proba_y(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, const double* gh_nodes_host,
                            const double* gh_weights_host, int32_t n_nodes, void* out0, void* out1) {
  HCHKF(h);
  if (n_t == 0) return AGP_OK;  // no test points: nothing to write (the reference returns empty arrays)
  return h->impl->proba_y(xt, ldx, n_t, gh_nodes_host, gh_weights_host, n_nodes, out0, out1);
}

}  // extern "C"
