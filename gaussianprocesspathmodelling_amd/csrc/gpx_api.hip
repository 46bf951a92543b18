// gpx_api.hip — host side of libgpx.so: the C ABI of include/gpx.h.
//
// Implements the fit()/predict() hot path of SURVEY.md §8 (rows a1-a6) on one MI355X:
//   fit:     K = sf2 k(X,X) + (sn2+jitter) I  ->  blocked Cholesky (in place, lower)
//            ->  alpha = L^-T L^-1 y  ->  logdet
//   predict: K* = sf2 k(Xs,X) -> mean = K* alpha -> V^T = K* L^-T -> var = sf2 - rowsumsq(V^T)
// The reference has no such path (GPmap.py has no fit/predict/linalg beyond
// np.linalg.norm at GPmap.py:120); the arithmetic follows oracle/gp_oracle.py
// (R&W Alg. 2.1).
//
// Data layout in HBM (element type T = double, or float for the fp32 / mixed handles; row-major):
//   K / L   [Npad+64][ld]  ld = Npad + one 128-byte line (skew against power-of-two strides; rows
//                          128-byte aligned); only the lower triangle is referenced; rows/cols >= N
//                          are identity; rows [Npad, Npad+64) carry the right-hand sides ("bordered")
//   Winv    [Npad/64][64][64]   inverses of the 64x64 diagonal blocks of L
//   Wblk    [Npad/nb][nb][nb]   explicit inverses of the nb x nb diagonal blocks (panel / block solves
//                               are dense products with them), built beside the factorisation
//   P       2 x [Npad+64][nb+skew]  compact copies of the current / next panel (SYRK operand)
//   VT      [MB][ld]            K* and, after the forward solve, V^T — one batch of MB <= 8192 query points
//   ZT      [Npad][ld]          L^-T, on demand, for gpx_lml_grad
//                               (gpx_fit_predict: Mpad more rows below the right-hand sides carry K* -> V^T)
// Streams: st (main: the throughput work — rows below the next diagonal block, REST, the rows of the panel
// solve nobody on the chain needs, bordered / query rows), st2 (look-ahead CHAIN, high priority: update of the
// next diagonal block, its factorisation, the panel solve of the rows of the diagonal block after it), st3
// (side: block-inverse extension, parked on the POTF2's device flag), st4 (copies of solved panels / blocks
// back into their matrices).  Schedule: chol_enqueue.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/gpx.h"
#include "gpx_internal.h"

using namespace gpx;

namespace {

thread_local std::string g_create_error;

// No exception crosses the C ABI (include/gpx.h): every extern "C" entry point that can
// allocate (std::string / std::vector / std::thread) is a function-try-block ending in this.
#define GPX_CATCH_ALL                                  \
  catch (const std::bad_alloc&) { return GPX_E_NOMEM; } \
  catch (...) { return GPX_E_HIP; }

constexpr int RHS_ROWS = 64;  // right-hand sides are padded to one 64-row MFMA slab

struct Phase {
  hipEvent_t a, b;
  double* target;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct Comm;   // gpx_shard.inc
struct Group;  // gpx_group.inc

}  // namespace

struct gpx_handle {
  gpx_config cfg{};
  int nb = 1024;       // Cholesky panel width
  int nb_solve = 256;  // block width of the alpha solves (one 64-row slab: latency-bound)
  int nb_pred = 1024;  // block width of the variance TRSM (follows the panel width when the library chooses that: auto_panel_width)
  bool nb_pred_env = false;  // GPX_NB_PRED given: never overridden
  hipStream_t st = nullptr;   // main stream
  hipStream_t st2 = nullptr;  // look-ahead (panel) stream, high priority
  hipStream_t st3 = nullptr;  // side stream of the diagonal chain: in-block SYRKs, block inverses
  hipStream_t st4 = nullptr;  // copy stream: solved panels / blocks back into their matrices
  hipStream_t st5 = nullptr;  // sharded factorisation: the owner's diagonal chain beside the all-gather of the previous panel (created by the first sharded fit)
  std::string err;
  gpx_timings tm{};
  // fitted state
  bool fitted = false;
  int64_t N = 0, Npad = 0, ld = 0, ldp = 0;
  int d = 0, k = 0, n_ls = 1;
  double sf2 = 0, sn2 = 0, jitter = 0;
  double logdet = 0;
  DevBuf X, Xs, ls, K, Winv, P, YT, Y, scalars, info;
  // predict state
  DevBuf Q, Qs, VT, MT, var, meanout;
  // row-block shard (world > 1)
  Comm* comm = nullptr;  // RCCL, in-process or host-callback transport (gpx_shard.inc)
  Group* group = nullptr;    // ndev > 1: this handle only fronts per-device member handles (gpx_group.inc)
  bool in_group = false;     // a member of a group: inputs may live on another device of the process
  bool discard_out = false;  // a member with rank > 0: rank 0 delivers the (identical) results
  void* zT = nullptr;      // z^T = (L^-1 y)^T (64 x ld): the bordered rows of the K buffer
  void* alphaT = nullptr;  // alpha^T (64 x ld): AT (computed on demand from z^T) or YT (shard)
  bool alpha_ready = false;
  DevBuf AT;
  DevBuf ZT, gpart;  // gpx_lml_grad: L^-T (Npad x ld) and the per-tile partial sums
  // GPX_MIXED: fp64 side of the mixed-precision mode (the fp32 engine uses the buffers above)
  DevBuf RTloc, P32out;  // GPX_MIXED on a distributed shard: local columns of the refinement's right-hand sides; fp32 predict outputs
  DevBuf X64, Y64, Xs64, A64, Aprev, R64, X32, Y32, RT32, Q64, Qs64, Q32, M64, rn, Zfew;
  int refine = 0;    // GPX_MIXED: 0 = adaptive, > 0 = fixed iteration count
  DevBuf Tsol;       // predict: compact solved blocks of V^T (2 x batch x (nb + skew))
  DevBuf MTpart;         // split-K partial tiles of the posterior-mean product
  DevBuf ZTloc, ZTpack;  // sharded gradient: own row blocks of L^-T (stacked), one packed block in flight
  DevBuf resv_ring;  // device counters of the self-reserving trailing updates ([1024][8] unsigned; GPX_CU_SELF_RESERVE)
  DevBuf Wblk, Ublk; // explicit inverses of the nb x nb diagonal blocks of L ([Npad/nb][nb][nb]) + scratch
  int nbw = 0;       // block width of Wblk (0: not built)
  int nb_shard = 512;  // distribution block = panel width of the sharded factorisation (chosen per fit)
  int nb_shard_env = 0;  // GPX_NB_SHARD override (0: choose from N and the number of ranks)
  int64_t nloc = 0, ldy = 0;
  DevBuf G, Dbuf, Sbuf, YTloc, Cneg, Sv;
  // replicated-factor mode of the shard: every rank keeps the whole L (the panels pass through
  // it anyway), so solves and predictions need no per-panel exchange (gpx_shard.inc)
  bool repl = false;
  DevBuf Lfull, GatherS, GatherR, outM, outV;
  const void* Lfac = nullptr;  // the factor the single-GPU solves read: K (unsharded) or Lfull
  // device-flag hand-overs between this handle's streams (diag_enqueue, fused strip): -1 not probed yet, 1 a kernel parked
  // on one stream sees the store of a kernel launched later on another (flag_handover_probe), 0 it does not: hipEvents
  int shard_snake = 1;  // the dealing of row blocks the last sharded fit used (gpx_internal.h: Deal)
  int64_t fq_rows = 0;  // gpx_fit_predict on a shard: padded rows of this rank's slice of the query points (bordered rows of its K buffer)
  int flag_ok = -1;
  int flag_retries = 0;  // fits of this handle that were run again with hipEvents after a parked stream timed out
  // event pool
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<Phase> phases;
};

namespace {

#define HIPCHK(h, call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      char buf_[512];                                                                      \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
               __FILE__, __LINE__);                                                        \
      (h)->err = buf_;                                                                     \
      return e_ == hipErrorOutOfMemory ? GPX_E_NOMEM : GPX_E_HIP;                          \
    }                                                                                      \
  } while (0)

int fail(gpx_handle* h, int code, const char* msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

#define LAUNCHCHK(h)                                                                                  \
  do {                                                                                                \
    if (take_launch_error())                                                                          \
      return fail(h, GPX_E_ARG, "internal error: a triangular-solve operand was not 128-byte aligned"); \
  } while (0)

int ensure(gpx_handle* h, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes && b.p) return GPX_OK;
  if (b.p) HIPCHK(h, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  HIPCHK(h, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return GPX_OK;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

hipEvent_t next_event(gpx_handle* h) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    h->ev_pool.push_back(e);
  }
  return h->ev_pool[h->ev_used++];
}

// roctx ranges around the phases (SURVEY.md §5 "tracing"): `rocprofv3 --marker-trace --kernel-trace` then shows which
// phase of a fit / predict enqueued which kernels.  librocprofiler-sdk-roctx.so is resolved at run time — the copy a
// profiler already mapped if there is one — and everything is a no-op when it is absent or GPX_ROCTX=0.  The ranges
// bracket the HOST-side enqueue of a phase (roctx marks host time; the per-phase GPU time is gpx_timings').
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    const char* e = getenv("GPX_ROCTX");
    if (e && atoi(e) == 0) return x;
    void* lib = nullptr;
    for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
      if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!lib)
      for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"})
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return x;
    x.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    x.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (!x.push || !x.pop) x.push = nullptr, x.pop = nullptr;
    return x;
  }();
  return r;
}

// the range name of a phase = the gpx_timings field its time goes to
const char* phase_name(const gpx_handle* h, const double* t) {
  const gpx_timings& tm = h->tm;
#define GPX_PHASE(f) \
  if (t == &tm.f) return "gpx:" #f;
  GPX_PHASE(h2d) GPX_PHASE(kbuild) GPX_PHASE(chol) GPX_PHASE(solve) GPX_PHASE(logdet) GPX_PHASE(fit_total)
  GPX_PHASE(kstar) GPX_PHASE(mean) GPX_PHASE(trsm) GPX_PHASE(var) GPX_PHASE(d2h) GPX_PHASE(predict_total)
  GPX_PHASE(comm) GPX_PHASE(chol_diag) GPX_PHASE(chol_trsm) GPX_PHASE(chol_strip) GPX_PHASE(chol_syrk)
  GPX_PHASE(grad_trtri) GPX_PHASE(grad_trace) GPX_PHASE(grad_total) GPX_PHASE(refine)
#undef GPX_PHASE
  return "gpx:phase";
}

struct PhaseScope {
  gpx_handle* h;
  Phase ph;
  bool on;
  hipStream_t s;
  bool ranged = false;
  PhaseScope(gpx_handle* h_, double* target, bool enable = true, hipStream_t stream = nullptr)
      : h(h_), on(enable), s(stream ? stream : h_->st) {
    if (roctx().push) ranged = roctx().push(phase_name(h, target)) >= 0;
    if (!on) return;
    ph.a = next_event(h);
    ph.b = next_event(h);
    ph.target = target;
    if (ph.a) (void)hipEventRecord(ph.a, s);
  }
  ~PhaseScope() {
    if (ranged) (void)roctx().pop();
    if (!on) return;
    if (ph.b) (void)hipEventRecord(ph.b, s);
    if (ph.a && ph.b) h->phases.push_back(ph);
  }
};

// after the stream is idle: fold event pairs into their targets
void collect_phases(gpx_handle* h) {
  for (auto& p : h->phases) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) *p.target += ms;
  }
  h->phases.clear();
  h->ev_used = 0;
}

// rocprofv3 --pmc (or a counter input file) exports ROCPROF_COUNTER_COLLECTION=1 into the profiled process.
// Counter collection serialises kernels across streams in an order of its own, which a stream parked on a
// device flag (wait_counter_kernel) does not survive — it would spin until its time-out — so the device-side
// hand-overs switch themselves off there and the schedule falls back to hipEvents (same kernels otherwise).
bool counters_are_being_collected() {
  const char* e = getenv("ROCPROF_COUNTER_COLLECTION");
  return e && atoi(e) != 0;
}

// Self-test of the device-flag hand-over, once per handle: a wait kernel on the side stream, then a
// one-thread store on the main stream.  With concurrent queues the waiter sees the flag within microseconds; where
// kernels are serialised across streams (AMD_SERIALIZE_KERNEL, counter collection, a debugger or tool funnelling the
// streams into one hardware queue in submission order) it runs out of its ~50 ms of polls first, and the handle hands
// over by hipEvents (same kernels, same arithmetic: bit-identical, tests/test_gp_parity_gpu.py) instead of meeting
// wait_counter_kernel's 15 s time-out inside a fit.  GPX_CHAIN_FLAG=0|1 overrides (and skips the probe).
int flag_handover_probe(gpx_handle* h) {
  if (h->flag_ok >= 0) return GPX_OK;
  if (getenv("GPX_CHAIN_FLAG")) return GPX_OK;  // explicit override (chain_flag_enabled reads it per call): nothing cached
  if (counters_are_being_collected() || !h->st3) {
    h->flag_ok = 0;
    return GPX_OK;
  }
  unsigned* d = nullptr;
  unsigned seen = 0;
  HIPCHK(h, hipMalloc(&d, 64));
  hipError_t e = hipMemsetAsync(d, 0, 64, h->st);
  if (e == hipSuccess) e = hipStreamSynchronize(h->st);
  if (e == hipSuccess) {
    launch_flag_probe_wait(d, d + 1, 1u << 15, h->st3);  // ~2 us per poll: gives up after ~50-70 ms
    launch_flag_probe_set(d, h->st);
    e = hipStreamSynchronize(h->st3);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->st);
  if (e == hipSuccess) e = hipMemcpy(&seen, d + 1, sizeof(unsigned), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIPCHK(h, e);
  h->flag_ok = seen ? 1 : 0;
  return GPX_OK;
}

// device-flag hand-overs for this handle's schedule?  (read per call: tests switch the environment)
bool chain_flag_enabled(const gpx_handle* h) {
  if (const char* e = getenv("GPX_CHAIN_FLAG")) return atoi(e) != 0;
  return !counters_are_being_collected() && h->flag_ok == 1;
}

struct LatencyGuard {  // set_latency_mode(0) on every exit path of a scheduler function
  ~LatencyGuard() { set_latency_mode(0); }
};
struct ReserveGuard {  // reserve mode off on every exit path of chol_enqueue
  ~ReserveGuard() { reserve_ring(nullptr, 0); }
};
constexpr int RESV_RING = 1024;  // self-reserving launches per factorisation (more: plain kernels)

// ---- blocked right-looking Cholesky, in place on the lower triangle -------------------
// A [n][ld]; Winv [n/64][64*64]; P = two compact panel buffers [n][ldp]; n multiple of
// 64 (128 when n > nb).  Two levels: panels of width nb; inside a panel's diagonal block,
// 64-wide sub-panels whose diagonal is factorised by the one-workgroup POTF2.
//
// One-panel look-ahead: the trailing update of step p is split into the STRIP (the nb
// columns that form panel p+1) and the REST.  As soon as the strip is done, the
// high-priority stream st2 factorises diagonal block p+1 and solves panel p+1 (into the
// other P buffer) while the main stream runs the rest of update p; the two streams
// touch disjoint columns of A.

// Explicit inverses W_p = L_pp^-1 of the nb x nb diagonal blocks, built BESIDE the factorisation
// of each block on the auxiliary stream: with them a panel solve X <- X L_pp^-T is ONE dense
// product X W_p^T on the MFMA tile engine instead of 64-row slabs each walking the nb/64 column
// blocks in sequence (a latency chain of ~0.7 ms per panel whatever the number of rows — the
// largest single item of a small-N fit: BASELINE.json configs[1]).
template <typename T>
struct InvWork {
  T* W = nullptr;   // [n / nbw][nbw][nbw], row-major, lower triangular (strictly upper part zero)
  T* U = nullptr;   // scratch: (L_pp^-1)^T of the block in flight, [nbw][ldu]
  int64_t ldu = 0;
  int nbw = 0;
  hipStream_t aux = nullptr;
  hipEvent_t ready = nullptr;      // recorded on aux when the inverse of the last enqueued block is complete
  hipEvent_t copied[2] = {};       // copy-back of the panel last written into P buffer 0 / 1 (copy stream)
  unsigned seq = 0;                // step flag's last value (info[9]): counts on from block to block of one factorisation
};

// diagonal block [o, o+nbp) on stream s.  iw != null: after the POTF2 of every 64-column step the
// auxiliary stream extends the block's inverse by one column block of U / row block of W
// (launch_inv_extend) — one event per step, nothing on the critical chain.
template <typename T>
int diag_enqueue(gpx_handle* h, T* A, int64_t ld, int64_t o, int nbp, T* Winv, int* info, int64_t gidx0,
                 hipStream_t s, InvWork<T>* iw = nullptr) {
  T* Wp = nullptr;
  // Hand-over to the side stream WITHOUT events on the chain (round 3): POTF2 of step i publishes i + 1 in
  // info[9] after a device-scope release; the side stream is parked on wait_counter_kernel in front of each
  // inverse extension.  The extension of column blocks [q, q + w) needs the rows of L of those blocks, which
  // are complete with the step's POTF2 (their off-diagonal part came from earlier steps' solves).  A hipEvent
  // per step cost the chain ~7 us of every 70 (record + the gap it opens).
  // The hipEvent per step comes back (GPX_CHAIN_FLAG=0, or by itself: flag_handover_probe) where kernels of different
  // streams do not run concurrently, e.g. under `rocprofv3 --pmc` (counter collection serialises kernels across queues
  // in an order of its own; the parked wait kernel then runs before the POTF2 it waits for and only its time-out ends
  // the stand-off — measured in round 3: the bench died after 15 s).
  unsigned* dflag = reinterpret_cast<unsigned*>(info + 9);  // `info` is a 64-byte buffer: [0] pivot, [8] strip counter, [9] step flag
  unsigned seq = iw ? iw->seq : 0;
  const bool use_flag = chain_flag_enabled(h);  // the handle's self-test (flag_handover_probe) or GPX_CHAIN_FLAG
  if (iw) {
    Wp = iw->W + (o / iw->nbw) * (int64_t)iw->nbw * iw->nbw;
    // the side stream starts behind everything already queued on s: the W block may live in a
    // buffer that earlier work on s (or, through its waits, on other ranks) is still reading
    hipEvent_t e0 = next_event(h);
    if (!e0) return fail(h, GPX_E_HIP, "hipEventCreate failed (block inverse)");
    // the step flag (info[9]) is reset in front of the FIRST block only; later blocks count on (one memset and
    // the gaps around it less on the chain per block; every stream that polls the flag starts behind e0)
    if (seq == 0) HIPCHK(h, hipMemsetAsync(dflag, 0, sizeof(unsigned), s));
    HIPCHK(h, hipEventRecord(e0, s));
    HIPCHK(h, hipStreamWaitEvent(iw->aux, e0, 0));
    HIPCHK(h, hipMemsetAsync(iw->U, 0, (size_t)iw->nbw * iw->ldu * sizeof(T), iw->aux));
    launch_set_diag_one_t<T>(iw->U, iw->ldu, nbp, iw->aux);
    HIPCHK(h, hipMemsetAsync(Wp, 0, (size_t)iw->nbw * iw->nbw * sizeof(T), iw->aux));
  }
  // The serial chain — POTF2, solve of the 64 columns below, SYRK inside the block — stays on ONE
  // stream: moving the SYRK to a side stream (so that it would run beside the next POTF2) made
  // the chain 40 % LONGER (measured at N = 8192: 49 -> 73 us per step), because every cross-stream
  // event wait costs more than the 6 us launch it hides.  The inverse extension only ever waits
  // FOR the chain (never the other way round), one event per step: batching the events four steps
  // at a time saved 0.3 ms of chain at N = 8192 but let the inverse lag behind the block's last
  // POTF2, and the panel solve waiting for it lost 0.7 ms.
  // Round 3: the chain advances 128 columns per step — ONE launch factors the 128 x 128 diagonal tile
  // (POTF2 of 64, the 64 x 64 block below it, its update, POTF2 of the second 64: potf2_128_kernel),
  // then one solve of the rows below, one update inside the block and one inverse-extension event
  // per 128 columns instead of per 64 (GPX_DIAG_STEP=64 selects the old stepping: A/B measurement).
  const int diag_step = [] {  // read per call (tests switch it)
    const char* e = getenv("GPX_DIAG_STEP");
    return (e && atoi(e) == 64) ? 64 : 128;
  }();
  const int nq = nbp / KB;
  for (int q = 0; q < nq;) {
    const int w = (diag_step == 128 && q + 1 < nq) ? 2 : 1;  // 64-blocks factored by this step
    const int64_t oq = o + (int64_t)q * KB;
    T* Aqq = A + oq * ld + oq;
    T* Wq = Winv + (oq / KB) * (KB * KB);
    seq += 2;  // POTF2 writes seq - 1 when it starts ("everything before it on the chain is complete") and seq when it is done
    if (w == 2)
      launch_potf2_128<T>(Aqq, ld, Wq, gidx0 + oq, info, s, iw && use_flag ? dflag : nullptr, seq);
    else
      launch_potf2_64<T>(Aqq, ld, Wq, gidx0 + oq, info, s, iw && use_flag ? dflag : nullptr, seq);
    const int64_t rem = o + nbp - (oq + w * KB);
    if (rem > 0) {
      T* panel = A + (oq + w * KB) * ld + oq;
      launch_trsm_rlt<T>(panel, ld, rem, Aqq, ld, Wq, w * KB, nullptr, 0, s);
      launch_gemm_nt<T>(64, A + (oq + w * KB) * ld + (oq + w * KB), ld, panel, ld, panel, ld, rem, rem, w * KB, 1, 0, s);
    }
    if (iw) {  // column blocks q .. q + w - 1 of the inverse, behind this step's POTF2
      if (use_flag) {
        // the part of the extension's walk that lies left of this step's columns needs nothing of this step:
        // it starts with the step's POTF2 and runs beside it; the rest follows the POTF2
        if (q > 0) {
          launch_wait_counter(dflag, seq - 1, info, iw->aux);
          launch_inv_extend<T>(iw->U, iw->ldu, A + o * ld + o, ld, Winv + (o / KB) * (KB * KB), q, q + w, Wp,
                               iw->nbw, iw->aux, 1);
        }
        launch_wait_counter(dflag, seq, info, iw->aux);
      } else {  // the same two launches (the same arithmetic, bit for bit), both behind the step's event
        hipEvent_t e = next_event(h);
        if (!e) return fail(h, GPX_E_HIP, "hipEventCreate failed (block inverse)");
        HIPCHK(h, hipEventRecord(e, s));
        HIPCHK(h, hipStreamWaitEvent(iw->aux, e, 0));
        launch_inv_extend<T>(iw->U, iw->ldu, A + o * ld + o, ld, Winv + (o / KB) * (KB * KB), q, q + w, Wp, iw->nbw,
                             iw->aux, 1);
      }
      launch_inv_extend<T>(iw->U, iw->ldu, A + o * ld + o, ld, Winv + (o / KB) * (KB * KB), q, q + w, Wp, iw->nbw,
                           iw->aux, 2);
    }
    q += w;
  }
  if (iw) {
    iw->seq = seq;
    iw->ready = next_event(h);
    if (!iw->ready) return fail(h, GPX_E_HIP, "hipEventCreate failed (block inverse)");
    HIPCHK(h, hipEventRecord(iw->ready, iw->aux));
  }
  return GPX_OK;
}

// Panel solve below diagonal block [o, o+nbp): P (rows x nbp, ldp) = A[t.., o..] * L_pp^-T, also
// stored back into A.  With the block inverse: two dense products (main rows / bordered rows) into
// P on stream s, and the copy back into A on the auxiliary stream (nobody reads the panel from A
// before the factorisation is over: the trailing updates read P).  Without: the slab kernel.
// row0 / prev / last (round 3): the rows [row0, row0 + rows_main + nx) of the panel only — the split schedule solves
// the rows of the next diagonal block on the chain and the rest on the main stream.  `prev` is the copy-back event
// of the panel that last lived in P[set] (every part waits for it before it writes); the part with `last` set
// records the new one (both parts' copies run in order on the copy stream).
template <typename T>
int panel_solve_enqueue(gpx_handle* h, T* A, int64_t ld, int64_t o, int nbp, int64_t rows_main, int64_t nx,
                        const T* Winv, T* P, int64_t ldp, hipStream_t s, InvWork<T>* iw, int set, int64_t row0 = 0,
                        hipEvent_t prev = nullptr, bool last = true) {
  T* Apanel = A + (o + nbp + row0) * ld + o;
  P += row0 * ldp;
  const int64_t rows = rows_main + nx;
  if (rows <= 0) {
    if (iw && last && row0 > 0) {  // nothing in this part: the first part's copy is the panel's
      iw->copied[set] = next_event(h);
      if (!iw->copied[set]) return fail(h, GPX_E_HIP, "hipEventCreate failed (panel copy)");
      HIPCHK(h, hipEventRecord(iw->copied[set], h->st4));
    }
    return GPX_OK;
  }
  if (!iw) {
    launch_trsm_rlt<T>(Apanel, ld, rows, A + o * ld + o, ld, Winv + (o / KB) * (KB * KB), nbp, P, ldp, s);
    return GPX_OK;
  }
  const T* Wp = iw->W + (o / iw->nbw) * (int64_t)iw->nbw * iw->nbw;
  HIPCHK(h, hipStreamWaitEvent(s, iw->ready, 0));
  if (row0 == 0 && last) prev = iw->copied[set];
  if (prev) HIPCHK(h, hipStreamWaitEvent(s, prev, 0));  // P[set] is about to be overwritten
  const int tile_main = gemm_nt_tile((rows_main % 128 == 0 && nbp % 128 == 0) ? 128 : 64, rows_main, nbp, 4);
  if (rows_main > 0 && nx > 0 && tile_main == 64) {
    // the bordered rows lie directly below the matrix rows, in A and in P: with 64-tiles on both, ONE launch
    // (as a launch of its own the 64-row product is a serial K = nb walk of ~37 us on the chain)
    launch_gemm_nt_fixed<T>(64, P, ldp, Apanel, ld, Wp, iw->nbw, rows, nbp, nbp, 4, 1, s);
  } else {
    if (rows_main > 0) launch_gemm_nt_fixed<T>(tile_main, P, ldp, Apanel, ld, Wp, iw->nbw, rows_main, nbp, nbp, 4, 1, s);
    if (nx > 0)
      launch_gemm_nt<T>(64, P + rows_main * ldp, ldp, Apanel + rows_main * ld, ld, Wp, iw->nbw, nx, nbp, nbp, 4, 1,
                        s);
  }
  hipEvent_t e = next_event(h);
  if (!e) return fail(h, GPX_E_HIP, "hipEventCreate failed (panel copy)");
  HIPCHK(h, hipEventRecord(e, s));
  HIPCHK(h, hipStreamWaitEvent(h->st4, e, 0));
  launch_copy2d<T>(Apanel, ld, P, ldp, rows, nbp, h->st4);
  if (last) {
    iw->copied[set] = next_event(h);
    if (!iw->copied[set]) return fail(h, GPX_E_HIP, "hipEventCreate failed (panel copy)");
    HIPCHK(h, hipEventRecord(iw->copied[set], h->st4));
  }
  return GPX_OK;
}

// nx extra rows (0 or a multiple of 64) directly below row n-1 ride along through every
// panel solve and trailing update without being factorised: with the right-hand sides
// stored there as rows ("bordered matrix"), they leave the factorisation as
// z^T = (L^-1 y)^T — the forward substitution costs no serial pass of its own.
// Round 3 — FUSED trailing update (GPX_FUSED_STRIP=1; the default stays the two-launch form: on the
// bench both take the same time — the boundary between STRIP and REST costs nothing measurable, the
// "gaps" of the round-2 trace were the strips' own work — and the fused form must not run under
// rocprofv3 --pmc, which serialises kernels and would starve the parked stream until its time-out):
// STRIP and REST are ONE launch (gemm_nt_fused_kernel) with the strip's tiles enumerated first.  Each strip slot bumps a device counter (info[8]) when its tile is released; the look-ahead
// stream is parked on wait_counter_kernel until the strip of this panel is complete and then runs
// the next diagonal block and panel solve beside the rest of the same launch.  The bordered rows'
// update of panel p+1 moves behind the panel solve on the look-ahead stream (it touches no row of
// the matrix proper), i.e. off the chain and inside the previous update.
template <typename T>
int chol_enqueue(gpx_handle* h, T* A, int64_t ld, int64_t n, int nb, T* Winv, T* P0, T* P1,
                 int64_t ldp, int* info, int64_t gidx0, bool profile, int64_t nx = 0,
                 InvWork<T>* iw = nullptr, int64_t nx_side = 0) {
  hipStream_t s0 = h->st, s1 = h->st2;
  T* Pbuf[2] = {P0, P1};
  int rc;
  const bool fuse_env = [] {  // read per call (tests switch it); default OFF: measured equal, see DESIGN.md §5.2
    const char* e = getenv("GPX_FUSED_STRIP");
    return e && atoi(e) != 0;
  }() && chain_flag_enabled(h);
  unsigned* ctr = reinterpret_cast<unsigned*>(info + 8);  // the info buffer is 64 bytes: [0] pivot, [8] strip counter
  unsigned target = 0;
  // is the trailing update of the panel at offset o one fused launch?  (128-tiles for strip and rest)
  auto is_fused = [&](int64_t o) -> bool {
    if (!fuse_env || o >= n) return false;
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t ntrail = n - (o + nbp);
    if (ntrail <= 0) return false;
    const int nbn = (int)std::min<int64_t>(nb, ntrail);
    const int64_t nrest = ntrail - nbn;
    if (nrest <= 0 || ntrail % 128 != 0 || nbn % 128 != 0) return false;
    return gemm_nt_tile(128, nrest, nrest, 1) == 128 && gemm_nt_tile(128, ntrail, nbn, 2) == 128;
  };
  // bordered rows [n, n+nx) x trailing columns of the panel at offset o (solved panel in Pc)
  auto bordered_update = [&](int64_t o, const T* Pc, hipStream_t s) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp, ntrail = n - t0;
    if (nx <= 0 || ntrail <= 0) return;
    launch_gemm_nt<T>(nx % 128 == 0 && ntrail % 128 == 0 ? 128 : 64, A + n * ld + t0, ld, Pc + ntrail * ldp, ldp, Pc,
                      ldp, nx, ntrail, nbp, 0, 0, s);
  };
  // bordered rows [n + r0, n + r0 + nr) x matrix columns [c0, c0 + nc), panel at offset o (split schedule)
  auto bordered_block = [&](int64_t o, const T* Pc, hipStream_t s, int64_t c0, int64_t nc, int64_t r0, int64_t nr) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp, ntrail = n - t0;
    if (nr <= 0 || nc <= 0) return;
    launch_gemm_nt<T>(nr % 128 == 0 && nc % 128 == 0 ? 128 : 64, A + (n + r0) * ld + c0, ld, Pc + (ntrail + r0) * ldp, ldp,
                      Pc + (c0 - t0) * ldp, ldp, nr, nc, nbp, 0, 0, s);
  };
  // does a 128-tile trailing update run beside the look-ahead chain of the panel behind offset o?  If not,
  // the chain's small launches are alone on the GPU and take the latency mode of the 64-tile engine.
  auto update_is_big = [&](int64_t o) -> bool {
    if (o >= n) return false;
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t ntrail = n - (o + nbp);
    if (ntrail <= 0) return false;
    const int nbn = (int)std::min<int64_t>(nb, ntrail);
    const int64_t nrest = ntrail - nbn;
    return nrest > 0 && ntrail % 128 == 0 && nbn % 128 == 0 && gemm_nt_tile(128, nrest, nrest, 1) == 128;
  };
  const bool split_env = [] {  // read per call (tests switch it)
    const char* e = getenv("GPX_SPLIT_STRIP");
    return !e || atoi(e) != 0;
  }() && !fuse_env;  // (the fused update keeps its own hand-over)
  const bool top_env = [] {  // GPX_SOLVE_TOP=0: the whole panel solve on the chain (A/B)
    const char* e = getenv("GPX_SOLVE_TOP");
    return !e || atoi(e) != 0;
  }();
  const int rest_split = [] {  // trailing rows (in panels) up to which the REST hands its first column block over early
    const char* e = getenv("GPX_REST_SPLIT");
    return e ? atoi(e) : 16;
  }();
  // The LAST nx_side bordered rows (the query points of gpx_fit_predict: a multiple of 128) are the main stream's
  // business alone in the split schedule: their panel solves, like their updates, are throughput work no step of the
  // chain waits for (tried: a stream of their own that trails the factorisation panel by panel, reading L from A —
  // slower, N = 8192: 12.4 against 11.6 ms; one more stream's worth of serial block solves).  The other bordered
  // rows (the right-hand sides) ride in the chain's panel solves as before.
  const int64_t nxs = split_env ? nx_side : 0, nxc = nx - nxs;
  auto side_solve = [&](int64_t o, T* Pp, hipStream_t s) -> int {  // side rows of the panel at offset o -> P and back into A
    if (nxs <= 0) return GPX_OK;
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    T* Aside = A + (n + nxc) * ld + o;
    T* Pside = Pp + ((n - o - nbp) + nxc) * ldp;
    if (!iw) {
      launch_trsm_rlt<T>(Aside, ld, nxs, A + o * ld + o, ld, Winv + (o / KB) * (KB * KB), nbp, Pside, ldp, s);
      return GPX_OK;
    }
    HIPCHK(h, hipStreamWaitEvent(s, iw->ready, 0));
    launch_gemm_nt<T>(nxs % 128 == 0 && nbp % 128 == 0 ? 128 : 64, Pside, ldp, Aside, ld,
                      iw->W + (o / iw->nbw) * (int64_t)iw->nbw * iw->nbw, iw->nbw, nxs, nbp, nbp, 4, 1, s);
    launch_copy2d<T>(Aside, ld, Pside, ldp, nxs, nbp, s);
    return GPX_OK;
  };
  LatencyGuard latency_guard;
  set_latency_mode(1);  // panel 0: nothing runs beside it
  // GPX_CU_SELF_RESERVE = k (round 4): where the chain sets the pace, the main stream's products leave k CUs per XCD to it
  // by themselves (gemm_nt_resv_kernel: persistent grids that exit on the reserved CUs).  Split schedule only.
  const int resv_k = [] {
    const char* e = getenv("GPX_CU_SELF_RESERVE");
    const int v = e ? atoi(e) : 0;
    return v > 0 && v <= 4 ? v : 0;
  }();
  ReserveGuard reserve_guard;
  if (resv_k > 0 && split_env) {
    if ((rc = ensure(h, h->resv_ring, (size_t)RESV_RING * 8 * sizeof(unsigned)))) return rc;
    HIPCHK(h, hipMemsetAsync(h->resv_ring.p, 0, (size_t)RESV_RING * 8 * sizeof(unsigned), s0));
    reserve_ring((unsigned*)h->resv_ring.p, RESV_RING);
  }
  const bool any_fused = is_fused(0);
  if (any_fused) HIPCHK(h, hipMemsetAsync(ctr, 0, sizeof(unsigned), s0));
  // prologue: panel 0 on the main stream
  {
    const int nb0 = (int)std::min<int64_t>(nb, n);
    {
      PhaseScope ps(h, &h->tm.chol_diag, profile);
      if ((rc = diag_enqueue(h, A, ld, 0, nb0, Winv, info, gidx0, s0, iw))) return rc;
    }
    {
      PhaseScope ps(h, &h->tm.chol_trsm, profile);
      if ((rc = panel_solve_enqueue(h, A, ld, 0, nb0, n - nb0, nxc, Winv, Pbuf[0], ldp, s0, iw, 0))) return rc;
    }
    if (any_fused) {
      bordered_update(0, Pbuf[0], s0);
      hipEvent_t e_init = next_event(h);
      if (!e_init) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
      HIPCHK(h, hipEventRecord(e_init, s0));
      HIPCHK(h, hipStreamWaitEvent(s1, e_init, 0));  // counter reset + panel 0 before the parked stream's first poll
    }
  }
  hipEvent_t e_main = nullptr;  // split strip: "the main stream's work on the trailing matrix so far is complete"
  if (split_env && n > nb) {
    e_main = next_event(h);
    if (!e_main) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
    HIPCHK(h, hipEventRecord(e_main, s0));  // panel 0 ran on the main stream
  }
  int step = 0;
  for (int64_t o = 0; o < n; o += nb, ++step) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp;          // first trailing row/col
    const int64_t ntrail = n - t0;
    if (ntrail <= 0) {  // the last panel: nothing to update, but the side rows still take their solve
      if ((rc = side_solve(o, Pbuf[step & 1], s0))) return rc;
      break;
    }
    T* Pc = Pbuf[step & 1];              // panel of this step, rows [t0, n + nx)
    T* Pn = Pbuf[(step + 1) & 1];
    const int nbn = (int)std::min<int64_t>(nb, ntrail);  // width of the next panel
    const int64_t nrest = ntrail - nbn;
    const int tile = (ntrail % 128 == 0 && nbn % 128 == 0) ? 128 : 64;
    hipEvent_t e_panel = next_event(h);
    if (!e_panel) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
    set_latency_mode(update_is_big(o) ? 0 : 1);  // for everything this iteration enqueues (strip, chain, solve, rest)
    if (is_fused(o)) {
      {  // main stream: the whole trailing update, strip first
        PhaseScope ps(h, &h->tm.chol_syrk, profile);
        target += launch_gemm_nt_fused<T>(A + t0 * ld + t0, ld, Pc, ldp, ntrail, nbn, nbp, ctr, s0);
        h->tm.syrk_flops += (double)ntrail * (double)(ntrail + 1) * (double)nbp;
        h->tm.syrk_launches += 1;
      }
      // look-ahead stream: parked until the strip is released, then diagonal block p+1, panel p+1
      launch_wait_counter(ctr, target, info, s1);
      {
        PhaseScope ps(h, &h->tm.chol_diag, profile, s1);
        if ((rc = diag_enqueue(h, A, ld, t0, nbn, Winv, info, gidx0, s1, iw))) return rc;
      }
      {
        PhaseScope ps(h, &h->tm.chol_trsm, profile, s1);
        if ((rc = panel_solve_enqueue(h, A, ld, t0, nbn, nrest, nx, Winv, Pn, ldp, s1, iw, (step + 1) & 1))) return rc;
      }
      HIPCHK(h, hipEventRecord(e_panel, s1));
      if (is_fused(t0)) bordered_update(t0, Pn, s1);  // for update p+1: off the chain, beside the rest of update p
      HIPCHK(h, hipStreamWaitEvent(s0, e_panel, 0));
      continue;
    }
    if (split_env) {
      hipStream_t sm = s0;
      // self-reserving main-stream products where the chain sets the pace (the criterion of the REST's early hand-over)
      static const bool resv_all = [] {  // GPX_RESV_ALL=1 (experiment): every panel's main-stream products, not only the chain-bound ones
        const char* e = getenv("GPX_RESV_ALL");
        return e && atoi(e) != 0;
      }();
      const int resv_it = (resv_k > 0 && (resv_all || ntrail <= (int64_t)(rest_split + 1) * nb)) ? resv_k : 0;
      // SPLIT STRIP (round 3, the default): of the strip only the next DIAGONAL block is on the chain.  It is
      // updated on the look-ahead stream itself, right behind the panel solve that produced its operand (no
      // hand-over between streams on the chain; 64-tiles in latency mode: a K = nb walk of a 64-tile is a
      // quarter of a 128-tile's), after the previous panel's REST (which also touched the block) is complete.
      // The rows BELOW it are needed only by the next panel solve, one diagonal chain later: they are updated
      // on the main stream, beside that chain, followed by the REST.
      if (e_main) HIPCHK(h, hipStreamWaitEvent(s1, e_main, 0));
      {
        PhaseScope ps(h, &h->tm.chol_strip, profile, s1);
        launch_gemm_nt<T>(tile, A + t0 * ld + t0, ld, Pc, ldp, Pc, ldp, nbn, nbn, nbp, 1, 0, s1);
      }
      // The host enqueues the main stream's whole iteration BEFORE the diagonal chain's ~50 small launches: the
      // GPU runs the chain about as fast as the host can enqueue it (N = 8192: the update used to reach the GPU
      // 0.5 ms after the panel it needs, and then collided with the NEXT panel solve and diagonal-block update).
      set_reserve_mode(resv_it);  // everything the main stream runs beside the next diagonal chain
      {  // rows below the diagonal block in the next panel's columns; the chain's bordered rows in a small launch
        PhaseScope ps(h, &h->tm.chol_strip, profile, sm);
        if (nrest > 0)
          launch_gemm_nt<T>(tile, A + (t0 + nbn) * ld + t0, ld, Pc + (int64_t)nbn * ldp, ldp, Pc, ldp, nrest, nbn, nbp, 0,
                            0, sm);
        bordered_block(o, Pc, sm, t0, nbn, 0, nxc);
      }
      hipEvent_t e_below = next_event(h);
      if (!e_below) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
      HIPCHK(h, hipEventRecord(e_below, sm));
      // REST.  Where the chain, not the update, sets the pace (the last `rest_split` panels' worth of trailing
      // rows: all of N = 8192, the tail of N = 65536), the column block of the panel AFTER next goes first, as a
      // launch of its own, and the look-ahead stream's next diagonal-block update waits for that launch only:
      // the chain then never waits for the bulk of an update (it used to, for every early panel of N = 8192,
      // where update and chain take about the same time).  The panel buffer this update reads is rewritten two
      // panel solves later, behind the next "rows below" event of this stream, i.e. behind all of this update.
      const int nbn2 = (int)std::min<int64_t>(nb, nrest);
      const bool ahead = nrest > nbn2 && nrest <= (int64_t)rest_split * nb;
      T* Cr = A + (t0 + nbn) * ld + (t0 + nbn);
      const T* Pr = Pc + (int64_t)nbn * ldp;
      if (ahead) {
        {
          PhaseScope ps(h, &h->tm.chol_strip, profile, sm);
          launch_gemm_nt<T>(tile, Cr, ld, Pr, ldp, Pr, ldp, nrest, nbn2, nbp, 2, 0, sm);
        }
        e_main = next_event(h);
        if (!e_main) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
        HIPCHK(h, hipEventRecord(e_main, sm));
      }
      const int64_t nr = ahead ? nrest - nbn2 : nrest, off = ahead ? nbn2 : 0;
      if (nr > 0) {
        const bool big = gemm_nt_tile(tile, nr, nr, 1) == 128;
        PhaseScope ps(h, big ? &h->tm.chol_syrk : &h->tm.chol_strip, profile, sm);
        launch_gemm_nt<T>(tile, Cr + off * ld + off, ld, Pr + off * ldp, ldp, Pr + off * ldp, ldp, nr, nr, nbp, 1, 0, sm);
        if (big) {
          h->tm.syrk_flops += (double)nr * (double)(nr + 1) * (double)nbp;
          h->tm.syrk_launches += 1;
        }
      }
      if (!ahead) {
        e_main = next_event(h);
        if (!e_main) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
        HIPCHK(h, hipEventRecord(e_main, sm));
      }
      {  // the bordered rows' share of the REST (columns beyond the next panel); the side rows' whole step — their
         // panel solve and the update of every column to the right — behind everything the chain waits for
        PhaseScope ps(h, &h->tm.chol_strip, profile, sm);
        bordered_block(o, Pc, sm, t0 + nbn, nrest, 0, nxc);
        if ((rc = side_solve(o, Pc, sm))) return rc;
        bordered_block(o, Pc, sm, t0, ntrail, nxc, nxs);
      }
      set_reserve_mode(0);
      {
        PhaseScope ps(h, &h->tm.chol_diag, profile, s1);
        static const int chain_env = [] {  // GPX_RESV_CHAIN: 0 chain untouched, 1 (default) POTF2 padded to 130 KB of LDS, 2 also its slab launches
          const char* e = getenv("GPX_RESV_CHAIN");
          return e ? atoi(e) : 1;
        }();
        set_reserve_chain(resv_it > 0 ? chain_env : 0);
        rc = diag_enqueue(h, A, ld, t0, nbn, Winv, info, gidx0, s1, iw);
        set_reserve_chain(0);
        if (rc) return rc;
      }
      HIPCHK(h, hipStreamWaitEvent(s1, e_below, 0));
      // Panel solve p+1: only the rows of the NEXT diagonal block (what STRIP_D(p+1) and the B operands need) on the
      // chain; the rows below them, and the chain's bordered rows, on the main stream in front of its next update
      // (their inputs are the main stream's own work plus W_{p+1}) — with the block inverse only.
      const int64_t top = (iw && top_env) ? std::min<int64_t>(nrest, nb) : nrest;
      hipEvent_t prev_copied = iw ? iw->copied[(step + 1) & 1] : nullptr;
      {
        PhaseScope ps(h, &h->tm.chol_trsm, profile, s1);
        if (top < nrest || (iw && top_env)) {
          if ((rc = panel_solve_enqueue(h, A, ld, t0, nbn, top, 0, Winv, Pn, ldp, s1, iw, (step + 1) & 1, 0, prev_copied,
                                        false)))
            return rc;
        } else if ((rc = panel_solve_enqueue(h, A, ld, t0, nbn, nrest, nxc, Winv, Pn, ldp, s1, iw, (step + 1) & 1))) {
          return rc;
        }
      }
      HIPCHK(h, hipEventRecord(e_panel, s1));
      HIPCHK(h, hipStreamWaitEvent(sm, e_panel, 0));
      if (top < nrest || (iw && top_env)) {
        PhaseScope ps(h, &h->tm.chol_trsm, profile, sm);
        set_reserve_mode(resv_it);  // runs beside the NEXT diagonal chain
        rc = panel_solve_enqueue(h, A, ld, t0, nbn, nrest - top, nxc, Winv, Pn, ldp, sm, iw, (step + 1) & 1, top, prev_copied,
                                 true);
        set_reserve_mode(0);
        if (rc) return rc;
      }
      continue;
    }
    {  // STRIP: rows [t0, n) x cols [t0, t0+nbn), tiles on/below the diagonal
      PhaseScope ps(h, &h->tm.chol_strip, profile);
      set_latency_mode(1);  // the strip runs alone (the look-ahead stream waits for it)
      launch_gemm_nt<T>(tile, A + t0 * ld + t0, ld, Pc, ldp, Pc, ldp, ntrail, nbn, nbp, 2, 0, s0);
      set_latency_mode(update_is_big(o) ? 0 : 1);
      // bordered rows [n, n+nx) x all trailing cols: their own small launch, so that the
      // SYRK grids (and their XCD balance) stay exactly those of the plain factorisation
      bordered_update(o, Pc, s0);
    }
    // the two streams touch the same matrix: an event that cannot be created / recorded /
    // waited on must fail the call, never let the streams run unordered
    hipEvent_t e_strip = next_event(h);
    if (!e_strip) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
    HIPCHK(h, hipEventRecord(e_strip, s0));
    HIPCHK(h, hipStreamWaitEvent(s1, e_strip, 0));
    {  // look-ahead stream: factor diagonal block p+1, solve panel p+1
      {
        PhaseScope ps(h, &h->tm.chol_diag, profile, s1);
        if ((rc = diag_enqueue(h, A, ld, t0, nbn, Winv, info, gidx0, s1, iw))) return rc;
      }
      {
        PhaseScope ps(h, &h->tm.chol_trsm, profile, s1);
        if ((rc = panel_solve_enqueue(h, A, ld, t0, nbn, nrest, nx, Winv, Pn, ldp, s1, iw, (step + 1) & 1))) return rc;
      }
    }
    HIPCHK(h, hipEventRecord(e_panel, s1));
    if (nrest > 0) {  // REST: lower triangle of the trailing matrix beyond the strip
      // the roofline bookkeeping (time, flops, launches) follows ONE kernel, the 128-tile triangular
      // update; the last few, under-filled updates run as 64-tiles and are booked with the strips
      const bool big = gemm_nt_tile(tile, nrest, nrest, 1) == 128;
      PhaseScope ps(h, big ? &h->tm.chol_syrk : &h->tm.chol_strip, profile);
      launch_gemm_nt<T>(tile, A + (t0 + nbn) * ld + (t0 + nbn), ld, Pc + (int64_t)nbn * ldp, ldp,
                        Pc + (int64_t)nbn * ldp, ldp, nrest, nrest, nbp, 1, 0, s0);
      if (big) {
        h->tm.syrk_flops += (double)nrest * (double)(nrest + 1) * (double)nbp;
        h->tm.syrk_launches += 1;
      }
    }
    HIPCHK(h, hipStreamWaitEvent(s0, e_panel, 0));
  }
  if (iw) {  // join the side and copy streams: the last inverse and every panel copy-back are in W / A
    for (hipStream_t sj : {iw->aux, h->st4}) {
      hipEvent_t e = next_event(h);
      if (!e) return fail(h, GPX_E_HIP, "hipEventCreate failed (join)");
      HIPCHK(h, hipEventRecord(e, sj));
      HIPCHK(h, hipStreamWaitEvent(s0, e, 0));
    }
  }
  return GPX_OK;
}

// XT (rows x n, ld) <- XT * L^-T   (i.e. X <- L^-1 X for X = XT^T), block forward substitution.
// With many right-hand sides (rows >= 128: the variance TRSM) the same one-panel
// look-ahead as the factorisation: the update of block p is split into the STRIP (the
// columns of block p+1) and the REST; the high-priority stream solves block p+1 while
// the main stream runs the rest of update p (disjoint columns of XT).
template <typename T>
struct SolveWork {       // optional: explicit block inverses + two compact result buffers (rows x ldt)
  const T* W = nullptr;
  int nbw = 0;
  T *T0 = nullptr, *T1 = nullptr;
  int64_t ldt = 0;
};

template <typename T>
int solve_fwd_enqueue(gpx_handle* h, T* XT, int64_t rows, const T* L, int64_t ld, int64_t n, int nb,
                      const T* Winv, const SolveWork<T>* wb = nullptr) {
  hipStream_t s0 = h->st, s1 = h->st2;
  const int tile = (rows % 128 == 0) ? 128 : 64;
  LatencyGuard latency_guard;
  set_latency_mode(1);  // one stream, nothing beside it; below: until a large update runs beside the block solves
  if (rows < 128 || !s1) {
    for (int64_t o = 0; o < n; o += nb) {
      const int nbp = (int)std::min<int64_t>(nb, n - o);
      launch_trsm_rlt<T>(XT + o, ld, rows, L + o * ld + o, ld, Winv + (o / KB) * (KB * KB), nbp, nullptr,
                         0, s0);
      const int64_t ntrail = n - (o + nbp);
      if (ntrail > 0)
        launch_gemm_nt<T>((ntrail % 128 == 0) ? tile : 64, XT + o + nbp, ld, XT + o, ld,
                          L + (o + nbp) * ld + o, ld, rows, ntrail, nbp, 0, 0, s0);
    }
    return GPX_OK;
  }
  // With the explicit block inverses of the factorisation (h->Wblk, same block width) a block
  // solve is one dense product into a compact buffer Ts (which then also serves as the A operand
  // of the updates) instead of the slab kernel's serial walk; the copy back into XT runs on the
  // auxiliary stream.
  const bool dense = wb && wb->nbw == nb && wb->T0 && h->st4;
  T* Ts[2] = {dense ? wb->T0 : nullptr, dense ? wb->T1 : nullptr};
  hipEvent_t e_copy[2] = {nullptr, nullptr};
  const int64_t ldt = dense ? wb->ldt : 0;
  auto block_solve = [&](int64_t o, int nbp, int set, hipStream_t s) -> int {
    if (!dense) {
      launch_trsm_rlt<T>(XT + o, ld, rows, L + o * ld + o, ld, Winv + (o / KB) * (KB * KB), nbp, nullptr, 0, s);
      return GPX_OK;
    }
    if (e_copy[set]) HIPCHK(h, hipStreamWaitEvent(s, e_copy[set], 0));  // Ts[set] is free again
    launch_gemm_nt<T>((rows % 128 == 0 && nbp % 128 == 0) ? 128 : 64, Ts[set], ldt, XT + o, ld,
                      wb->W + (o / nb) * (int64_t)nb * nb, nb, rows, nbp, nbp, 4, 1, s);
    hipEvent_t e = next_event(h);
    e_copy[set] = next_event(h);
    if (!e || !e_copy[set]) return fail(h, GPX_E_HIP, "hipEventCreate failed (block solve)");
    HIPCHK(h, hipEventRecord(e, s));
    HIPCHK(h, hipStreamWaitEvent(h->st4, e, 0));
    launch_copy2d<T>(XT + o, ld, Ts[set], ldt, rows, nbp, h->st4);
    HIPCHK(h, hipEventRecord(e_copy[set], h->st4));
    return GPX_OK;
  };
  int rc;
  if ((rc = block_solve(0, (int)std::min<int64_t>(nb, n), 0, s0))) return rc;
  int step = 0;
  for (int64_t o = 0; o < n; o += nb, ++step) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp, ntrail = n - t0;
    if (ntrail <= 0) break;
    const int nbn = (int)std::min<int64_t>(nb, ntrail);
    const int64_t nrest = ntrail - nbn;
    const int tl = (nbn % 128 == 0 && nrest % 128 == 0) ? tile : 64;
    const int cur = step & 1;
    const T* Ablk = dense ? Ts[cur] : XT + o;  // the solved block o (rows x nbp)
    const int64_t lda = dense ? ldt : ld;
    set_latency_mode(1);  // the STRIP runs alone: the look-ahead stream waits for it, the previous REST is over
    launch_gemm_nt<T>(tl, XT + t0, ld, Ablk, lda, L + t0 * ld + o, ld, rows, nbn, nbp, 0, 0, s0);  // STRIP
    set_latency_mode(nrest > 0 && gemm_nt_tile(tl, rows, nrest, 0) == 128 ? 0 : 1);  // a 128-tile REST runs beside block p+1's solve
    hipEvent_t e_strip = next_event(h), e_panel = next_event(h);
    if (!e_strip || !e_panel) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
    HIPCHK(h, hipEventRecord(e_strip, s0));
    HIPCHK(h, hipStreamWaitEvent(s1, e_strip, 0));
    if ((rc = block_solve(t0, nbn, cur ^ 1, s1))) return rc;
    HIPCHK(h, hipEventRecord(e_panel, s1));
    if (nrest > 0)  // REST
      launch_gemm_nt<T>(tl, XT + t0 + nbn, ld, Ablk, lda, L + (t0 + nbn) * ld + o, ld, rows, nrest, nbp, 0, 0,
                        s0);
    HIPCHK(h, hipStreamWaitEvent(s0, e_panel, 0));
  }
  if (dense) {  // every block is back in XT
    for (int set = 0; set < 2; ++set)
      if (e_copy[set]) HIPCHK(h, hipStreamWaitEvent(s0, e_copy[set], 0));
  }
  return GPX_OK;
}

// XT (rows x n, ld) <- XT * L^-1   (X <- L^-T X), block back substitution; rows multiple of 64
template <typename T>
void solve_bwd_enqueue(gpx_handle* h, T* XT, int64_t rows, const T* L, int64_t ld, int64_t n, int nb,
                       const T* Winv) {
  hipStream_t st = h->st;
  int64_t last = ((n - 1) / nb) * nb;
  for (int64_t o = last; o >= 0; o -= nb) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    launch_trsm_rln<T>(XT + o, ld, rows, L + o * ld + o, ld, Winv + (o / KB) * (KB * KB), nbp, st);
    if (o > 0) launch_gemm_nn<T>(XT, ld, XT + o, ld, L + o * ld, ld, rows, o, nbp, st);
  }
}

int copy_in(gpx_handle* h, void* dst, const void* src, size_t bytes, int mem_kind) {
  if (bytes == 0) return GPX_OK;
  // group members: a device pointer of the caller lives on ONE of the group's devices
  const hipMemcpyKind kind = mem_kind == GPX_MEM_HOST ? hipMemcpyHostToDevice
                             : h->in_group            ? hipMemcpyDefault
                                                      : hipMemcpyDeviceToDevice;
  HIPCHK(h, hipMemcpyAsync(dst, src, bytes, kind, h->st));
  return GPX_OK;
}

int copy_out(gpx_handle* h, void* dst, const void* src, size_t bytes, int mem_kind) {
  if (bytes == 0) return GPX_OK;
  HIPCHK(h, hipMemcpyAsync(dst, src, bytes,
                           mem_kind == GPX_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                           h->st));
  return GPX_OK;
}

// Rows of K* / V^T per batch of query points.  Query rows are independent, so predicting in
// batches is numerically identical to one pass; the V^T buffer is the only O(M N) allocation
// of predict (M = 65536 at N = 131072 would be 69 GB beside the 137 GB factor), so it is capped:
// GPX_PRED_BATCH rows (default 8192, a multiple of 128), and — where every rank decides for
// itself (no collective inside the batch loop) — shrunk to what the card has left.
int64_t pred_batch_rows(gpx_handle* h, int64_t Mpad, size_t row_bytes, bool may_shrink) {
  int64_t cap = 8192;
  if (const char* e = getenv("GPX_PRED_BATCH")) {
    const long v = atol(e);
    if (v >= 128 && v <= (1L << 22)) cap = v / 128 * 128;
  }
  if (may_shrink) {
    size_t freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
      const double avail = 0.8 * ((double)freeb + (double)h->VT.cap);
      const int64_t fit = (int64_t)(avail / (double)row_bytes) / 128 * 128;
      cap = std::max<int64_t>(128, std::min(cap, fit));
    }
  }
  return std::min(cap, Mpad);
}

// Panel width the library picks when gpx_config.block == 0 (round 4).  Rounds 1-3 swept 256 ... 2048 and found a plateau
// from 1024 — with the diagonal chain partly exposed.  On the round-3/4 schedule nothing of the chain is exposed at large
// N, and a 2048-wide panel halves the number of trailing updates (K = 2048 per tile: the per-tile prologue + epilogue
// weigh half as much, 72.4 against 71.7 TF) and of block steps of the variance solve: C3 1602-1605 -> 1579-1584 ms, C5 fp32
// 881 -> 859 ms; equal at N = 16384 ... 32768, slower at N = 8192 where the chain sets the pace (12.5 vs 13.2 ms):
// tools/r04_nb_sweep.sh, profiles/r04_panel_width.txt.  GPX_NB_WIDE_FROM = rows from which 2048 is used (0: never).
int auto_panel_width(int64_t Npad) {
  static const int64_t from = [] {
    const char* e = getenv("GPX_NB_WIDE_FROM");
    return e ? (int64_t)atoll(e) : (int64_t)40960;
  }();
  return (from > 0 && Npad >= from) ? 2048 : 1024;
}

template <typename T>
int predict_core(gpx_handle* h, const void* Xq, int64_t M, bool want_var, int32_t mem_kind);
bool few_solver_applies(const gpx_handle* h);
template <typename T>
int solve_few(gpx_handle* h, T* RT, int k, const T* L, int64_t ld, int64_t n, const T* W, int nb, bool forward = true,
              bool backward = true);

}  // namespace

#include "gpx_shard.inc"
#include "gpx_group.inc"

namespace {

// Xq != null (gpx_fit_predict): the M query points' cross-kernel rows K* ride through the factorisation as
// bordered rows too — below the right-hand sides — and leave it as
// V^T = (L^-1 K*^T)^T: the variance solve of predict costs no pass of its own (its M N^2 flops are rows of
// the trailing updates, which at small N fill the CUs the serial diagonal chain leaves idle).
template <typename T>
int fit_impl(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
             const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter,
             int32_t mem_kind, int64_t* info, const void* Xq = nullptr, int64_t M = 0, bool retried = false) {
  const int64_t Npad = round_up(N, TILE);
  const int64_t Mpad = Xq ? round_up(M, TILE) : 0;
  const int64_t ld = Npad + ld_skew<T>();
  if (h->cfg.block == 0) {  // the library's choice; predict's block solves use the fit's block inverses: same width
    h->nb = auto_panel_width(Npad);
    if (!h->nb_pred_env) h->nb_pred = h->nb;
  }
  const int64_t ldp = h->nb + ld_skew<T>();
  h->N = N; h->Npad = Npad; h->ld = ld; h->ldp = ldp; h->d = d; h->k = k; h->n_ls = n_ls;
  h->sf2 = sf2; h->sn2 = sn2; h->jitter = jitter;
  const bool profile = (h->cfg.flags & GPX_FLAG_PROFILE) != 0;
  gpx_timings& tm = h->tm;
  tm.h2d = tm.kbuild = tm.chol = tm.solve = tm.logdet = tm.fit_total = 0;
  tm.chol_diag = tm.chol_trsm = tm.chol_strip = tm.chol_syrk = tm.syrk_flops = tm.comm = 0;
  tm.syrk_launches = 0;
  tm.kbuild_bytes = (double)sizeof(T) * ((double)N * (double)(N + 1) / 2.0 + (double)N * d);

  int rc;
  if ((rc = flag_handover_probe(h))) return rc;  // once per handle: device flags or hipEvents between the streams
  if ((rc = ensure(h, h->X, (size_t)N * d * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->Y, (size_t)N * k * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->Xs, (size_t)Npad * d * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->ls, 32 * 8))) return rc;
  const int64_t NX = RHS_ROWS + Mpad;  // bordered rows: [K* rows of a fused predict] + the right-hand sides
  if (Xq) {
    if ((rc = ensure(h, h->Q, (size_t)M * d * sizeof(T)))) return rc;
    if ((rc = ensure(h, h->Qs, (size_t)Mpad * d * sizeof(T)))) return rc;
  }
  if ((rc = ensure(h, h->K, (size_t)(Npad + NX) * ld * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->Winv, (size_t)(Npad / KB) * KB * KB * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->P, (size_t)2 * (Npad + NX) * ldp * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->scalars, 64))) return rc;
  if ((rc = ensure(h, h->info, 64))) return rc;
  const int64_t nblk = (Npad + h->nb - 1) / h->nb;
  const int64_t ldu = h->nb + ld_skew<T>();
  if ((rc = ensure(h, h->Wblk, (size_t)nblk * h->nb * h->nb * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->Ublk, (size_t)h->nb * ldu * sizeof(T)))) return rc;
  InvWork<T> iw;
  iw.W = (T*)h->Wblk.p;
  iw.U = (T*)h->Ublk.p;
  iw.ldu = ldu;
  iw.nbw = h->nb;
  iw.aux = h->st3;
  h->nbw = 0;  // valid again once this fit has finished

  T* dK = (T*)h->K.p;
  h->Lfac = dK;
  h->repl = false;
  T* dYT = dK + Npad * ld;  // rows [Npad, Npad + RHS_ROWS) of the K buffer: y^T, then z^T = (L^-1 y)^T
  h->zT = dYT;
  h->alphaT = nullptr;
  h->alpha_ready = false;
  int* dInfo = (int*)h->info.p;
  {
    PhaseScope total(h, &tm.fit_total);
    {
      PhaseScope ps(h, &tm.h2d);
      if ((rc = copy_in(h, h->X.p, X, (size_t)N * d * sizeof(T), mem_kind))) return rc;
      if ((rc = copy_in(h, h->Y.p, y, (size_t)N * k * sizeof(T), mem_kind))) return rc;
      if ((rc = copy_in(h, h->ls.p, lengthscale, (size_t)n_ls * 8, GPX_MEM_HOST))) return rc;
      const int init = INT_MAX;
      HIPCHK(h, hipMemcpyAsync(dInfo, &init, sizeof(int), hipMemcpyHostToDevice, h->st));
    }
    {
      PhaseScope ps(h, &tm.kbuild);
      launch_scale_points<T>((const T*)h->X.p, N, Npad, d, (const double*)h->ls.p, n_ls,
                          (T*)h->Xs.p, h->st);
      launch_kbuild_sym<T>(h->cfg.kernel, (const T*)h->Xs.p, N, Npad, d, sf2, sn2 + jitter, dK, ld,
                        h->st);
      if (Xq) {  // rows [Npad + RHS_ROWS, ... + Mpad): K(Xq, X)
        if ((rc = copy_in(h, h->Q.p, Xq, (size_t)M * d * sizeof(T), mem_kind))) return rc;
        launch_scale_points<T>((const T*)h->Q.p, M, Mpad, d, (const double*)h->ls.p, n_ls, (T*)h->Qs.p, h->st);
        launch_kbuild_cross<T>(h->cfg.kernel, (const T*)h->Qs.p, M, Mpad, (const T*)h->Xs.p, N, Npad, d, sf2,
                            dK + (Npad + RHS_ROWS) * ld, ld, h->st);
      }
    }
    {
      PhaseScope ps(h, &tm.chol);
      launch_pack_rhs<T>((const T*)h->Y.p, N, k, dYT, ld, Npad, RHS_ROWS, h->st);
      if ((rc = chol_enqueue<T>(h, dK, ld, Npad, h->nb, (T*)h->Winv.p, (T*)h->P.p,
                                (T*)h->P.p + (Npad + NX) * ldp, ldp, dInfo, 0, profile, NX, &iw, Mpad)))
        return rc;
    }
    // The forward substitution happened inside the factorisation (bordered rows).  The
    // back substitution alpha = L^-T z is NOT on the fit+predict path: the posterior mean is
    // K* K^-1 y = V^T z with V = L^-1 K*^T, which the variance path computes anyway;
    // alpha is produced on demand (gpx_get_alpha, mean-only predict): ensure_alpha().
    {
      PhaseScope ps(h, &tm.logdet);
      launch_logdet<T>(dK, ld, Npad, (double*)h->scalars.p, h->st);
    }
  }
  int hinfo = 0;
  HIPCHK(h, hipMemcpyAsync(&hinfo, dInfo, sizeof(int), hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(&h->logdet, h->scalars.p, sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  if (hinfo < 0) {
    // wait_counter_kernel gave up although the self-test passed (a tool attached later, or one that reorders rather than
    // serialises): this handle hands over by hipEvents from now on and the fit runs once more — same kernels, same bits.
    // Only a forced GPX_CHAIN_FLAG=1, or a second failure, is an error.
    if (!retried && !getenv("GPX_CHAIN_FLAG") && h->flag_ok != 0) {
      h->flag_ok = 0;
      h->flag_retries += 1;
      h->phases.clear();
      h->ev_used = 0;
      return fit_impl<T>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info, Xq, M, true);
    }
    return fail(h, GPX_E_HIP,
                "a stream parked on a device flag timed out: kernels are being serialised across streams — "
                "set GPX_CHAIN_FLAG=0 (and leave GPX_FUSED_STRIP unset) to hand over by hipEvents instead");
  }
  tm.handover_flags = chain_flag_enabled(h) ? 1.0 : 0.0;
  tm.handover_retries = (double)h->flag_retries;
  *info = (hinfo == INT_MAX) ? 0 : (int64_t)hinfo;
  h->fitted = (*info == 0);
  if (h->fitted) h->nbw = h->nb;
  return GPX_OK;
}

// RT (k <= 8 rows of ld) <- (L L^T)^-1 RT (or only L^-T RT) with the fit's explicit block inverses: a forward and a
// backward block substitution as streams over the factor (gpx_mixed.hip: rowdot / coldot kernels), one stream, in order.
template <typename T>
int solve_few(gpx_handle* h, T* RT, int k, const T* L, int64_t ld, int64_t n, const T* W, int nbw, bool forward, bool backward) {
  // W: [n / nbw][nbw][nbw] explicit inverses of the nbw-wide diagonal blocks.  The sweeps run in blocks of nb = min(nbw,
  // 1024): the diagonal nb-sub-blocks of a lower-triangular inverse ARE the inverses of the diagonal nb-sub-blocks of
  // the factor, so a 2048-wide W serves 1024-blocks through a pointer and its leading dimension (round 4).
  const int nb = std::min(nbw, 1024);
  auto Wsub = [&](int64_t o) -> const T* {
    return W + (o / nbw) * (int64_t)nbw * nbw + ((o % nbw) / nb) * ((int64_t)nb * nbw + nb);
  };
  int rc;
  if ((rc = ensure(h, h->Zfew, (size_t)8 * nb * sizeof(T)))) return rc;
  T* Zs = (T*)h->Zfew.p;
  hipStream_t st = h->st;
  for (int64_t o = 0; forward && o < n; o += nb) {  // L z = r
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp;
    launch_few_product<T>(false, true, Zs, nb, Wsub(o), nbw, nbp, nbp, RT + o, ld, k, st);
    HIPCHK(h, hipMemcpy2DAsync(RT + o, (size_t)ld * sizeof(T), Zs, (size_t)nb * sizeof(T), (size_t)nbp * sizeof(T), k,
                               hipMemcpyDeviceToDevice, st));
    launch_few_product<T>(false, false, RT + t0, ld, L + t0 * ld + o, ld, n - t0, nbp, Zs, nb, k, st);
  }
  for (int64_t o = ((n - 1) / nb) * nb; backward && o >= 0; o -= nb) {  // L^T x = z
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    launch_few_product<T>(true, true, Zs, nb, Wsub(o), nbw, nbp, nbp, RT + o, ld, k, st);
    HIPCHK(h, hipMemcpy2DAsync(RT + o, (size_t)ld * sizeof(T), Zs, (size_t)nb * sizeof(T), (size_t)nbp * sizeof(T), k,
                               hipMemcpyDeviceToDevice, st));
    launch_few_product<T>(true, false, RT, ld, L + o * ld, ld, o, nbp, Zs, nb, k, st);
  }
  return GPX_OK;
}

// the streaming solver needs the fit's explicit block inverses (h->Wblk, blocks of width h->nbw <= 1024: every block of
// an unsharded fit, or of a shard that keeps the whole factor), at most 8 right-hand sides (GPX_FEW_SOLVE=0: always the
// slab path)
bool few_solver_applies(const gpx_handle* h) {
  const char* e = getenv("GPX_FEW_SOLVE");
  return (!e || atoi(e) != 0) && h->nbw > 0 && (h->nbw <= 1024 ? h->nbw % 128 == 0 : h->nbw % 1024 == 0) && h->k <= 8 &&
         (!h->comm || h->repl) && h->Wblk.p;
}

// alpha^T = z^T L^-1 on a copy of the bordered rows (z stays available for V^T z)
template <typename T>
int ensure_alpha(gpx_handle* h) {
  if (h->alpha_ready) return GPX_OK;
  int rc;
  if ((rc = ensure(h, h->AT, (size_t)RHS_ROWS * h->ld * sizeof(T)))) return rc;
  PhaseScope ps(h, &h->tm.solve);
  HIPCHK(h, hipMemcpyAsync(h->AT.p, h->zT, (size_t)RHS_ROWS * h->ld * sizeof(T), hipMemcpyDeviceToDevice,
                           h->st));
  if (few_solver_applies(h)) {  // k <= 8 targets: the back substitution as a stream over the factor (round 3)
    if ((rc = solve_few<T>(h, (T*)h->AT.p, h->k, (const T*)h->Lfac, h->ld, h->Npad, (const T*)h->Wblk.p, h->nbw, false)))
      return rc;
  } else
    solve_bwd_enqueue<T>(h, (T*)h->AT.p, RHS_ROWS, (const T*)h->Lfac, h->ld, h->Npad, h->nb_solve,
                         (const T*)h->Winv.p);
  h->alphaT = h->AT.p;
  h->alpha_ready = true;
  return GPX_OK;
}

// K*, variance solve, mean and variance of M query points; results stay on the device in
// h->meanout (M x k) and h->var (M).  Reads the factor through h->Lfac (whole L).
template <typename T>
int predict_core(gpx_handle* h, const void* Xq, int64_t M, bool want_var, int32_t mem_kind) {
  if (M <= 0) return GPX_OK;  // a rank whose slice of the query points is empty
  const int64_t N = h->N, Npad = h->Npad, ld = h->ld;
  const int d = h->d, k = h->k;
  const int64_t Mpad = round_up(M, TILE);
  const int64_t ldm = Mpad + ld_skew<T>();
  const int64_t MB = pred_batch_rows(h, Mpad, (size_t)ld * sizeof(T), true);
  gpx_timings& tm = h->tm;
  int rc;
  if ((rc = ensure(h, h->Q, (size_t)M * d * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->Qs, (size_t)Mpad * d * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->VT, (size_t)MB * ld * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->MT, (size_t)RHS_ROWS * ldm * sizeof(T)))) return rc;
  // the mean is a 64-row product with K = N: split the contraction so that it fills the chip
  const int ksplit = splitk_splits(Npad);
  const int64_t ldpm = std::min(MB, Mpad) + ld_skew<T>();  // partial tiles of ONE batch
  if (ksplit > 1 && (rc = ensure(h, h->MTpart, (size_t)ksplit * RHS_ROWS * ldpm * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->meanout, (size_t)M * k * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->var, (size_t)Mpad * sizeof(T)))) return rc;
  SolveWork<T> sw;
  if (want_var && h->nbw == h->nb_pred && (!h->comm || h->repl)) {  // block inverses of this fit (a shard: of the factor it keeps whole), same block width
    sw.ldt = h->nbw + ld_skew<T>();
    if ((rc = ensure(h, h->Tsol, (size_t)2 * MB * sw.ldt * sizeof(T)))) return rc;
    sw.W = (const T*)h->Wblk.p;
    sw.nbw = h->nbw;
    sw.T0 = (T*)h->Tsol.p;
    sw.T1 = sw.T0 + MB * sw.ldt;
  }
  T* dVT = (T*)h->VT.p;
  const T* dK = (const T*)h->Lfac;
  const T* dWinv = (const T*)h->Winv.p;
  if (!want_var && (rc = ensure_alpha<T>(h))) return rc;  // mean only: K* alpha with the cached alpha
  const T* rhsT = (const T*)(want_var ? h->zT : h->alphaT);
  {
    PhaseScope ps(h, &tm.kstar);
    if ((rc = copy_in(h, h->Q.p, Xq, (size_t)M * d * sizeof(T), mem_kind))) return rc;
    launch_scale_points<T>((const T*)h->Q.p, M, Mpad, d, (const double*)h->ls.p, h->n_ls,
                        (T*)h->Qs.p, h->st);
  }
  for (int64_t m0 = 0; m0 < Mpad; m0 += MB) {  // batches of query points through one V^T buffer
    const int64_t mp = std::min(MB, Mpad - m0);       // padded rows of this batch
    const int64_t mv = std::min<int64_t>(mp, M - m0);  // valid rows
    {
      PhaseScope ps(h, &tm.kstar);
      launch_kbuild_cross<T>(h->cfg.kernel, (const T*)h->Qs.p + m0 * d, mv, mp, (const T*)h->Xs.p, N,
                          Npad, d, h->sf2, dVT, ld, h->st);
    }
    if (want_var) {
      PhaseScope ps(h, &tm.trsm);
      if ((rc = solve_fwd_enqueue<T>(h, dVT, mp, dK, ld, Npad, h->nb_pred, dWinv, sw.W ? &sw : nullptr))) return rc;
    }
    {  // with the variance: mean^T (64 x mp) = z^T (64 x Npad) * V   (mu = K* K^-1 y = V^T z);
       // mean only: alpha^T * K*^T
      PhaseScope ps(h, &tm.mean);
      if (ksplit > 1)
        launch_gemm_nt_splitk<T>((T*)h->MT.p + m0, ldm, rhsT, ld, dVT, ld, RHS_ROWS, mp, Npad, ksplit,
                                 (T*)h->MTpart.p, ldpm, h->st);
      else
        launch_gemm_nt<T>(64, (T*)h->MT.p + m0, ldm, rhsT, ld, dVT, ld, RHS_ROWS, mp, Npad, 0, 1, h->st);
    }
    if (want_var) {
      PhaseScope ps(h, &tm.var);
      launch_var_rows<T>(dVT, ld, mv, Npad, h->sf2, (T*)h->var.p + m0, h->st);
    }
  }
  {
    PhaseScope ps(h, &tm.mean);
    launch_unpack_rhs<T>((const T*)h->MT.p, ldm, M, k, 1.0, (T*)h->meanout.p, h->st);
  }
  return GPX_OK;
}

template <typename T>
int predict_impl(gpx_handle* h, const void* Xq, int64_t M, void* mean, void* var, int32_t mem_kind) {
  gpx_timings& tm = h->tm;
  tm.kstar = tm.mean = tm.trsm = tm.var = tm.d2h = tm.predict_total = 0;
  int rc;
  {
    PhaseScope total(h, &tm.predict_total);
    if ((rc = predict_core<T>(h, Xq, M, var != nullptr, mem_kind))) return rc;
    PhaseScope ps(h, &tm.d2h);
    if ((rc = copy_out(h, mean, h->meanout.p, (size_t)M * h->k * sizeof(T), mem_kind))) return rc;
    if (var && (rc = copy_out(h, var, h->var.p, (size_t)M * sizeof(T), mem_kind))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  return GPX_OK;
}

// The predict half of gpx_fit_predict: V^T already sits in the Mpad rows below the right-hand sides in the factor's buffer
// (fit_impl with query points); what is left is the mean V^T z and the row norms.
template <typename T>
int fused_predict_tail(gpx_handle* h, int64_t M, void* mean, void* var, int32_t mem_kind) {
  const int64_t Npad = h->Npad, ld = h->ld;
  const int64_t Mpad = round_up(M, TILE), ldm = Mpad + ld_skew<T>();
  gpx_timings& tm = h->tm;
  tm.kstar = tm.mean = tm.trsm = tm.var = tm.d2h = tm.predict_total = 0;
  int rc;
  if ((rc = ensure(h, h->MT, (size_t)RHS_ROWS * ldm * sizeof(T)))) return rc;
  const int ksplit = splitk_splits(Npad);
  const int64_t ldpm = Mpad + ld_skew<T>();
  if (ksplit > 1 && (rc = ensure(h, h->MTpart, (size_t)ksplit * RHS_ROWS * ldpm * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->meanout, (size_t)M * h->k * sizeof(T)))) return rc;
  if ((rc = ensure(h, h->var, (size_t)Mpad * sizeof(T)))) return rc;
  const T* dVT = (const T*)h->Lfac + (Npad + RHS_ROWS) * ld;
  {
    PhaseScope total(h, &tm.predict_total);
    {
      PhaseScope ps(h, &tm.mean);
      if (ksplit > 1)
        launch_gemm_nt_splitk<T>((T*)h->MT.p, ldm, (const T*)h->zT, ld, dVT, ld, RHS_ROWS, Mpad, Npad, ksplit,
                                 (T*)h->MTpart.p, ldpm, h->st);
      else
        launch_gemm_nt<T>(64, (T*)h->MT.p, ldm, (const T*)h->zT, ld, dVT, ld, RHS_ROWS, Mpad, Npad, 0, 1, h->st);
      launch_unpack_rhs<T>((const T*)h->MT.p, ldm, M, h->k, 1.0, (T*)h->meanout.p, h->st);
    }
    if (var) {
      PhaseScope ps(h, &tm.var);
      launch_var_rows<T>(dVT, ld, M, Npad, h->sf2, (T*)h->var.p, h->st);
    }
    PhaseScope ps(h, &tm.d2h);
    if ((rc = copy_out(h, mean, h->meanout.p, (size_t)M * h->k * sizeof(T), mem_kind))) return rc;
    if (var && (rc = copy_out(h, var, h->var.p, (size_t)M * sizeof(T), mem_kind))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  return GPX_OK;
}

// ZT <- ZT * L^-T = L^-T restricted to the rows this rank owns: the forward substitution of
// solve_fwd_enqueue on the rows that are not structurally zero.  Row r of L^-T is zero left of
// column r and the rows are independent (a right-hand multiplication), so the row blocks of
// height nb are dealt block-cyclically over P ranks with no exchange: rank `rank` keeps its blocks
// b = rank, rank + P, ... stacked in ZT (local block j = global block j P + rank, the identity at
// columns [b nb, b nb + nb) on entry), and column block [o, o + nb) only involves the local rows of
// the blocks below o / nb — a prefix of the stack.  P = 1: the whole L^-T in place, N^3/3 flops;
// same STRIP / REST look-ahead as the factorisation.
int trtri_enqueue(gpx_handle* h, double* ZT, const double* L, int64_t ld, int64_t n, int nb,
                  const double* Winv, int P, int rank) {
  hipStream_t s0 = h->st, s1 = h->st2;
  auto rows_below = [&](int64_t t) -> int64_t {  // local rows of the own blocks with index < t / nb
    const int64_t q = t / nb;
    return q > rank ? (q - rank + P - 1) / P * nb : 0;
  };
  auto own = [&](int64_t t) { return (int)((t / nb) % P) == rank; };
  if (own(0))
    launch_trsm_rlt<double>(ZT, ld, std::min<int64_t>(nb, n), L, ld, Winv, (int)std::min<int64_t>(nb, n), nullptr, 0, s0);
  for (int64_t o = 0; o < n; o += nb) {
    const int nbp = (int)std::min<int64_t>(nb, n - o);
    const int64_t t0 = o + nbp, ntrail = n - t0;
    if (ntrail <= 0) break;
    const int nbn = (int)std::min<int64_t>(nb, ntrail);
    const int64_t nrest = ntrail - nbn;
    const int64_t R = rows_below(t0);                    // local rows that are non-zero in column block o
    const int64_t Rn = R + (own(t0) ? nbn : 0);          // ... plus the identity rows of block t0 / nb
    const int tl = (R % 128 == 0 && nbn % 128 == 0 && nrest % 128 == 0) ? 128 : 64;
    if (R > 0) launch_gemm_nt<double>(tl, ZT + t0, ld, ZT + o, ld, L + t0 * ld + o, ld, R, nbn, nbp, 0, 0, s0);  // STRIP
    hipEvent_t e_strip = next_event(h), e_blk = next_event(h);
    if (!e_strip || !e_blk) return fail(h, GPX_E_HIP, "hipEventCreate failed (look-ahead)");
    HIPCHK(h, hipEventRecord(e_strip, s0));
    HIPCHK(h, hipStreamWaitEvent(s1, e_strip, 0));
    if (Rn > 0)
      launch_trsm_rlt<double>(ZT + t0, ld, Rn, L + t0 * ld + t0, ld, Winv + (t0 / KB) * (KB * KB), nbn, nullptr, 0, s1);
    HIPCHK(h, hipEventRecord(e_blk, s1));
    if (nrest > 0 && R > 0)  // REST
      launch_gemm_nt<double>(tl, ZT + t0 + nbn, ld, ZT + o, ld, L + (t0 + nbn) * ld + o, ld, R, nrest, nbp, 0, 0,
                             s0);
    HIPCHK(h, hipStreamWaitEvent(s0, e_blk, 0));
  }
  return GPX_OK;
}

// Gradient of the log marginal likelihood.  Sharded handles (replicated factor: every rank holds
// the whole L): L^-T in row blocks dealt over the ranks (no exchange), one all-gather of its
// non-zero part (block b: rows x columns >= b nb, packed), the fused K^-1 trace pass with the
// tile groups dealt over the ranks, one all-reduce of ntheta sums.  alpha and its quadratic
// form are replicated work (O(N^2)), identical on every rank.
int lml_grad_impl(gpx_handle* h, double* lml, double* grad) {
  const int64_t N = h->N, Npad = h->Npad, ld = h->ld;
  const int d = h->d, k = h->k, ntheta = h->n_ls + 2, ard = h->n_ls > 1;
  Comm* cm = h->comm;
  const int P = cm ? cm->world : 1, rank = cm ? cm->rank : 0;
  // the gradient's sweeps keep their measured 1024-blocks whatever the fit's panel width (GPX_NB_GRAD overrides: A/B)
  const int nb = [&] {
    const char* e = getenv("GPX_NB_GRAD");
    const int v = e ? atoi(e) : 0;
    return (v >= 128 && v <= 4096 && v % 128 == 0) ? v : std::min(h->nb_pred, 1024);
  }();
  gpx_timings& tm = h->tm;
  tm.grad_trtri = tm.grad_trace = tm.grad_total = 0;
  const int64_t s1n = kinv_trace_slots(Npad), s2n = alpha_quad_slots(Npad);
  const int64_t nblk = (Npad + nb - 1) / nb;
  int64_t nloc = 0;  // rows of the own blocks (the last block may be ragged)
  for (int64_t b = rank; b < nblk; b += P) nloc += std::min<int64_t>(nb, Npad - b * nb);
  int rc;
  if ((rc = ensure_alpha<double>(h))) return rc;
  if ((rc = ensure(h, h->ZT, (size_t)Npad * ld * 8))) return rc;
  if (P > 1) {
    if ((rc = ensure(h, h->ZTloc, (size_t)std::max<int64_t>(nloc, 1) * ld * 8))) return rc;
    if ((rc = ensure(h, h->ZTpack, (size_t)nb * Npad * 8))) return rc;
  }
  if ((rc = ensure(h, h->gpart, (size_t)((s1n + s2n) * ntheta + 2 * ntheta + 1) * 8))) return rc;
  double* ZT = (double*)h->ZT.p;
  double* Zl = P > 1 ? (double*)h->ZTloc.p : ZT;
  double* part1 = (double*)h->gpart.p;
  double* part2 = part1 + s1n * ntheta;
  double* outv = part2 + s2n * ntheta;  // [ntheta] K^-1 sums, [ntheta] alpha sums, [1] y . alpha
  hipStream_t st = h->st;
  {
    PhaseScope total(h, &tm.grad_total);
    {
      PhaseScope ps(h, &tm.grad_trtri);
      HIPCHK(h, hipMemsetAsync(Zl, 0, (size_t)(P > 1 ? nloc : Npad) * ld * 8, st));
      if (P == 1) {
        launch_set_diag_one(ZT, ld, Npad, st);
      } else {
        int64_t j = 0;
        for (int64_t b = rank; b < nblk; b += P, ++j)
          launch_set_diag_one(Zl + j * nb * ld + b * nb, ld, std::min<int64_t>(nb, Npad - b * nb), st);
      }
      if ((rc = trtri_enqueue(h, Zl, (const double*)h->Lfac, ld, Npad, nb, (const double*)h->Winv.p, P, rank)))
        return rc;
      if (P > 1) {  // all-gather: block b from rank b % P, only the columns right of its zeros
        double* pack = (double*)h->ZTpack.p;
        HIPCHK(h, hipMemsetAsync(ZT, 0, (size_t)Npad * ld * 8, st));
        for (int64_t b = 0; b < nblk; ++b) {
          const int64_t r0 = b * nb, hb = std::min<int64_t>(nb, Npad - r0), w = Npad - r0;
          const int root = (int)(b % P);
          if (root == rank) launch_copy2d<double>(pack, w, Zl + (b / P) * nb * ld + r0, ld, hb, w, st);
          if ((rc = cm->bcast(h, pack, (size_t)(hb * w), root, st))) return rc;
          launch_copy2d<double>(ZT + r0 * ld + r0, ld, pack, w, hb, w, st);
        }
      }
    }
    {
      PhaseScope ps(h, &tm.grad_trace);
      HIPCHK(h, hipMemsetAsync(part1, 0, (size_t)(s1n + s2n) * ntheta * 8, st));  // ragged-edge / other ranks' slots write nothing
      launch_kinv_trace(h->cfg.kernel, ZT, ld, Npad, N, (const double*)h->Xs.p, d, ard, h->sf2, h->sn2, part1,
                        ntheta, P, rank, st);
    }
    launch_alpha_quad(h->cfg.kernel, (const double*)h->alphaT, ld, k, Npad, N, (const double*)h->Xs.p, d, ard,
                      h->sf2, h->sn2, part2, ntheta, st);
    launch_reduce_partials(part1, s1n, ntheta, 1.0, outv, st);
    launch_reduce_partials(part2, s2n, ntheta, 1.0, outv + ntheta, st);
    launch_dot_rhs((const double*)h->Y.p, (const double*)h->alphaT, ld, N, k, outv + 2 * ntheta, st);
    if (P > 1 && (rc = cm->allreduce(h, outv, (size_t)ntheta, COMM_SUM))) return rc;
  }
  double host[2 * 34 + 1];
  HIPCHK(h, hipMemcpyAsync(host, outv, (size_t)(2 * ntheta + 1) * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  for (int t = 0; t < ntheta; ++t) grad[t] = 0.5 * (host[ntheta + t] - (double)k * host[t]);
  *lml = -0.5 * host[2 * ntheta] - 0.5 * (double)k * h->logdet -
         0.5 * (double)N * (double)k * 1.8378770664093454835606594728112;  // log(2 pi)
  return GPX_OK;
}

// ---- GPX_MIXED: fp32 factorisation, fp64 refinement of alpha, fp64 mean --------------------------
constexpr int MIXED_KMAX = 8;
constexpr int MIXED_MAX_ITERS = 12;   // refine == 0: cap of the adaptive refinement
// relative residual at which the adaptive refinement stops.  At N = 65536 (C5) the largest elementwise
// relative error of the posterior mean is ~2000x the relative residual (means near zero count with a
// floor of 1e-6: profiles/r02_c5_mixed_precision_study.json), so north_star's 1e-6 needs <= 4e-10.
constexpr double MIXED_TOL = 1e-10;

void launch_rows_sumsq_y(gpx_handle* h, double* out) {  // ||y||^2: y (N x k) is one row of N k values
  launch_rows_sumsq((const double*)h->Y64.p, 0, 1, h->N * h->k, out, h->st);
}

int mixed_fit(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
              const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter, int32_t mem_kind,
              int64_t* info) {
  if (k > MIXED_KMAX) return fail(h, GPX_E_ARG, "gpx_fit: the mixed-precision mode takes at most 8 target columns");
  int rc;
  if ((rc = ensure(h, h->X64, (size_t)N * d * 8))) return rc;
  if ((rc = ensure(h, h->Y64, (size_t)N * k * 8))) return rc;
  if ((rc = ensure(h, h->X32, (size_t)N * d * 4))) return rc;
  if ((rc = ensure(h, h->Y32, (size_t)N * k * 4))) return rc;
  if ((rc = ensure(h, h->ls, 32 * 8))) return rc;
  hipStream_t st = h->st;
  if ((rc = copy_in(h, h->X64.p, X, (size_t)N * d * 8, mem_kind))) return rc;
  if ((rc = copy_in(h, h->Y64.p, y, (size_t)N * k * 8, mem_kind))) return rc;
  launch_f64_to_f32((const double*)h->X64.p, (float*)h->X32.p, N * d, st);
  launch_f64_to_f32((const double*)h->Y64.p, (float*)h->Y32.p, N * k, st);
  // the whole fp32 fit (kernel build, blocked Cholesky, z = L^-1 y) on the device copies — on a shard (round 4) the
  // sharded fp32 fit: every rank holds the same fp64 inputs and runs the same refinement (replicated fp64 work)
  if (h->comm)
    rc = shard_fit<float>(h, h->X32.p, h->Y32.p, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, GPX_MEM_DEVICE, info);
  else
    rc = fit_impl<float>(h, h->X32.p, h->Y32.p, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, GPX_MEM_DEVICE, info);
  if (rc) return rc;
  gpx_timings& tm = h->tm;
  tm.refine = tm.refine_resid0 = tm.refine_resid = tm.refine_iters = 0;
  if (*info != 0) return GPX_OK;
  const int64_t Npad = h->Npad;  // the fp32 side's padding (a shard pads to its block height): the fp64 side follows it
  if ((rc = ensure(h, h->Xs64, (size_t)Npad * d * 8))) return rc;
  if ((rc = ensure(h, h->A64, (size_t)MIXED_KMAX * Npad * 8))) return rc;
  if ((rc = ensure(h, h->R64, (size_t)MIXED_KMAX * Npad * 8))) return rc;
  if ((rc = ensure(h, h->Aprev, (size_t)MIXED_KMAX * Npad * 8))) return rc;
  if ((rc = ensure(h, h->rn, 64))) return rc;
  const bool dist = h->comm && !h->repl;  // factor only held distributed: the refinement's solves are collective
  if (dist && (rc = ensure(h, h->RTloc, (size_t)RHS_ROWS * h->ldy * 4))) return rc;
  const int64_t ld32 = h->ld;
  if ((rc = ensure(h, h->RT32, (size_t)RHS_ROWS * ld32 * 4))) return rc;
  const bool few = !dist && few_solver_applies(h);
  if (!dist && (rc = ensure_alpha<float>(h))) return rc;  // (distributed: the fit left alpha^T replicated in YT)
  double* A64 = (double*)h->A64.p;
  double* R64 = (double*)h->R64.p;
  double* rn = (double*)h->rn.p;
  const double* Xs64 = (const double*)h->Xs64.p;
  const float* L32 = (const float*)h->Lfac;
  double hn[3] = {0, 0, 0};
  {
    PhaseScope ps(h, &tm.refine);
    launch_scale_points<double>((const double*)h->X64.p, N, Npad, d, (const double*)h->ls.p, n_ls, (double*)h->Xs64.p,
                                st);
    HIPCHK(h, hipMemsetAsync(A64, 0, (size_t)MIXED_KMAX * Npad * 8, st));
    launch_rows_add_f32_to_f64((const float*)h->alphaT, ld32, A64, Npad, k, N, Npad, 0, st);
    launch_rows_sumsq_y(h, rn + 2);  // ||y||^2
  }
  // refine == 0: iterate until the fp64 residual ||y - K alpha|| falls below MIXED_TOL ||y|| or stops
  // contracting (the fp32 factor is no preconditioner for this matrix), at most MIXED_MAX_ITERS
  // times — the residual norm is read back once per iteration (one 31 ms pass at N = 65536 each).
  // A fixed count runs without reading anything back until the end.
  const bool adaptive = h->refine <= 0;
  const int max_it = adaptive ? MIXED_MAX_ITERS : h->refine;
  double prev = 0, restored_res = -1.0;
  int iters = 0;
  for (int it = 0;; ++it) {
    PhaseScope ps(h, &tm.refine);
    // r = y - (K + diag I) alpha  (fp64, matrix-free)
    launch_kmatvec(h->cfg.kernel, Xs64, N, Npad, Xs64, Npad, d, sf2, sn2 + jitter, (const double*)h->Y64.p, A64,
                   Npad, k, -1.0, R64, Npad, st);
    bool last = it == max_it;
    if (it == 0 || last || adaptive) launch_rows_sumsq(R64, Npad, k, N, rn + (it == 0 ? 0 : 1), st);
    if (adaptive) {
      HIPCHK(h, hipMemcpyAsync(hn, rn, 24, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      const double res = hn[2] > 0 ? std::sqrt(hn[it == 0 ? 0 : 1] / hn[2]) : 0.0;
      if (it > 0 && !(res <= prev)) {
        // the last correction made the residual WORSE (or not finite): the fp32 factor is no contraction for this
        // matrix.  alpha goes back to the iterate before it, and that iterate's residual and count are what is reported.
        HIPCHK(h, hipMemcpyAsync(A64, h->Aprev.p, (size_t)MIXED_KMAX * Npad * 8, hipMemcpyDeviceToDevice, st));
        restored_res = prev;
        iters = it - 1;
        break;
      }
      if (res <= MIXED_TOL || (it > 0 && res > 0.5 * prev)) last = true;
      prev = res;
    }
    if (last) {
      if (it == 0) HIPCHK(h, hipMemcpyAsync(rn + 1, rn, 8, hipMemcpyDeviceToDevice, st));
      iters = it;
      break;
    }
    // delta = (L L^T)^-1 r in fp32; alpha += delta in fp64 (adaptive: the iterate before the correction is kept)
    if (adaptive) HIPCHK(h, hipMemcpyAsync(h->Aprev.p, A64, (size_t)MIXED_KMAX * Npad * 8, hipMemcpyDeviceToDevice, st));
    launch_rows_f64_to_f32(R64, Npad, (float*)h->RT32.p, ld32, k, RHS_ROWS, N, Npad, st);
    if (dist) {  // own column blocks of the residual rows -> the distributed solve -> delta replicated in RT32 again
      const int nb = h->nb_shard;
      const Shard sh{h->comm->world, h->comm->rank, nb, Npad / nb, h->shard_snake};
      float* Rl = (float*)h->RTloc.p;
      for (int64_t lb = 0; lb < sh.nlb(sh.r); ++lb)
        HIPCHK(h, hipMemcpy2DAsync(Rl + lb * nb, (size_t)h->ldy * 4, (const float*)h->RT32.p + sh.global(sh.r, lb) * nb,
                                   (size_t)ld32 * 4, (size_t)nb * 4, RHS_ROWS, hipMemcpyDeviceToDevice, st));
      if ((rc = shard_solve_dist<float>(h, sh, Rl, (float*)h->RT32.p))) return rc;
    } else if (few) {
      if ((rc = solve_few<float>(h, (float*)h->RT32.p, k, L32, ld32, Npad, (const float*)h->Wblk.p, h->nbw))) return rc;
    } else {
      if ((rc = solve_fwd_enqueue<float>(h, (float*)h->RT32.p, RHS_ROWS, L32, ld32, Npad, h->nb_solve,
                                         (const float*)h->Winv.p)))
        return rc;
      solve_bwd_enqueue<float>(h, (float*)h->RT32.p, RHS_ROWS, L32, ld32, Npad, h->nb_solve, (const float*)h->Winv.p);
    }
    launch_rows_add_f32_to_f64((const float*)h->RT32.p, ld32, A64, Npad, k, N, Npad, 1, st);
  }
  HIPCHK(h, hipMemcpyAsync(hn, rn, 24, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  tm.fit_total += tm.refine;
  tm.refine_iters = iters;
  if (hn[2] > 0) {
    tm.refine_resid0 = std::sqrt(hn[0] / hn[2]);
    tm.refine_resid = restored_res >= 0 ? restored_res : std::sqrt(hn[1] / hn[2]);
  }
  return GPX_OK;
}

int mixed_predict(gpx_handle* h, const void* Xq, int64_t M, void* mean, void* var, int32_t mem_kind) {
  const int64_t N = h->N, Npad = h->Npad;
  const int d = h->d, k = h->k;
  const int64_t Mpad = round_up(M, TILE);
  gpx_timings& tm = h->tm;
  tm.kstar = tm.mean = tm.trsm = tm.var = tm.d2h = tm.predict_total = 0;
  int rc;
  if ((rc = ensure(h, h->Q64, (size_t)M * d * 8))) return rc;
  if ((rc = ensure(h, h->Qs64, (size_t)Mpad * d * 8))) return rc;
  if ((rc = ensure(h, h->Q32, (size_t)M * d * 4))) return rc;
  if ((rc = ensure(h, h->M64, (size_t)(MIXED_KMAX * Mpad + (size_t)M * k + Mpad) * 8))) return rc;
  double* MT64 = (double*)h->M64.p;            // [8][Mpad] transposed means
  double* mout = MT64 + MIXED_KMAX * Mpad;     // (M x k)
  double* vout = mout + (size_t)M * k;         // (M)
  hipStream_t st = h->st;
  double shard_ms[6] = {0, 0, 0, 0, 0, 0};  // a sharded variance predict is a call of its own: its clocks are added below
  if (var && h->comm) {
    // Variance through the sharded fp32 factor (round 4): the whole sharded predict — collective, every rank — into
    // device scratch, before this call's own phases (it synchronises and folds its own clocks).  Its fp32 mean is not
    // used: the mean comes from the refined fp64 alpha below, replicated work on every rank.
    if ((rc = ensure(h, h->P32out, (size_t)M * (k + 1) * 4))) return rc;
    if ((rc = copy_in(h, h->Q64.p, Xq, (size_t)M * d * 8, mem_kind))) return rc;
    launch_f64_to_f32((const double*)h->Q64.p, (float*)h->Q32.p, M * d, st);
    float* m32 = (float*)h->P32out.p;
    float* v32 = m32 + (size_t)M * k;
    const bool discard = h->discard_out;
    h->discard_out = false;  // the scratch outputs are wanted on every rank
    rc = shard_predict<float>(h, h->Q32.p, M, m32, v32, GPX_MEM_DEVICE);
    h->discard_out = discard;
    if (rc) return rc;
    const double keep[6] = {tm.kstar, tm.trsm, tm.var, tm.mean, tm.predict_total, tm.comm};
    for (int i = 0; i < 6; ++i) shard_ms[i] = keep[i];
    tm.kstar = tm.mean = tm.trsm = tm.var = tm.d2h = tm.predict_total = 0;
    launch_f32_to_f64(v32, vout, M, st);
  }
  {
    PhaseScope total(h, &tm.predict_total);
    if ((rc = copy_in(h, h->Q64.p, Xq, (size_t)M * d * 8, mem_kind))) return rc;
    if (var && !h->comm) {  // variance through the fp32 factor (its own K*, V^T = K* L^-T, row sums)
      launch_f64_to_f32((const double*)h->Q64.p, (float*)h->Q32.p, M * d, st);
      if ((rc = predict_core<float>(h, h->Q32.p, M, true, GPX_MEM_DEVICE))) return rc;
      launch_f32_to_f64((const float*)h->var.p, vout, M, st);
    }
    {
      PhaseScope ps(h, &tm.mean);
      launch_scale_points<double>((const double*)h->Q64.p, M, Mpad, d, (const double*)h->ls.p, h->n_ls,
                                  (double*)h->Qs64.p, st);
      launch_kmatvec(h->cfg.kernel, (const double*)h->Qs64.p, M, Mpad, (const double*)h->Xs64.p, Npad, d, h->sf2,
                     0.0, nullptr, (const double*)h->A64.p, Npad, k, 1.0, MT64, Mpad, st);
      launch_unpack_rhs<double>(MT64, Mpad, M, k, 1.0, mout, st);
    }
    PhaseScope ps(h, &tm.d2h);
    if (!h->discard_out) {  // a group's ranks > 0 hold the same result; rank 0 delivers it
      if ((rc = copy_out(h, mean, mout, (size_t)M * k * 8, mem_kind))) return rc;
      if (var && (rc = copy_out(h, var, vout, (size_t)M * 8, mem_kind))) return rc;
    }
  }
  HIPCHK(h, hipStreamSynchronize(st));
  HIPCHK(h, hipGetLastError());
  LAUNCHCHK(h);
  collect_phases(h);
  tm.kstar += shard_ms[0];
  tm.trsm += shard_ms[1];
  tm.var += shard_ms[2];
  tm.predict_total += shard_ms[4];
  (void)N;
  return GPX_OK;
}

int mixed_alpha(gpx_handle* h, void* out) {
  int rc;
  if ((rc = ensure(h, h->M64, (size_t)h->N * h->k * 8))) return rc;
  launch_unpack_rhs<double>((const double*)h->A64.p, h->Npad, h->N, h->k, 1.0, (double*)h->M64.p, h->st);
  HIPCHK(h, hipMemcpyAsync(out, h->M64.p, (size_t)h->N * h->k * 8, hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  return GPX_OK;
}

template <typename T>
int alpha_impl(gpx_handle* h, void* out) {
  int rc;
  if ((!h->comm || h->repl) && (rc = ensure_alpha<T>(h))) return rc;  // the distributed shard solves alpha in fit
  if ((rc = ensure(h, h->meanout, (size_t)h->N * h->k * sizeof(T)))) return rc;
  launch_unpack_rhs<T>((const T*)h->alphaT, h->ld, h->N, h->k, 1.0, (T*)h->meanout.p, h->st);
  HIPCHK(h, hipMemcpyAsync(out, h->meanout.p, (size_t)h->N * h->k * sizeof(T), hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  return GPX_OK;
}

}  // namespace

extern "C" {

int gpx_abi_version(void) { return GPX_ABI_VERSION; }

int gpx_device_count(int* count) try {
  if (!count) return GPX_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return GPX_E_HIP;
  }
  *count = n;
  return GPX_OK;
}
GPX_CATCH_ALL

const char* gpx_last_error(gpx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gpx_create(gpx_handle** out, const gpx_config* cfg) try {
  if (!out || !cfg) return fail(nullptr, GPX_E_ARG, "gpx_create: null argument");
  *out = nullptr;
  if (cfg->kernel != GPX_KERNEL_RBF && cfg->kernel != GPX_KERNEL_MATERN52)
    return fail(nullptr, GPX_E_ARG, "gpx_create: unknown kernel id");
  if (cfg->dtype != GPX_F64 && cfg->dtype != GPX_F32 && cfg->dtype != GPX_MIXED)
    return fail(nullptr, GPX_E_ARG, "gpx_create: unknown dtype id");
  if (cfg->refine < 0 || cfg->refine > 50) return fail(nullptr, GPX_E_ARG, "gpx_create: need 0 <= refine <= 50");
  if (cfg->world < 1 || cfg->world > 64 || cfg->rank < 0 || cfg->rank >= cfg->world)
    return fail(nullptr, GPX_E_ARG, "gpx_create: need 1 <= world <= 64 and 0 <= rank < world");
  if (cfg->ndev < 0 || cfg->ndev > GPX_MAX_GROUP)
    return fail(nullptr, GPX_E_ARG, "gpx_create: need 0 <= ndev <= GPX_MAX_GROUP");
  const int nb = cfg->block == 0 ? 1024 : cfg->block;
  if (nb < 128 || nb > 4096 || nb % 128 != 0)
    return fail(nullptr, GPX_E_ARG, "gpx_create: block must be a multiple of 128 in [128, 4096]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, GPX_E_HIP, "gpx_create: no HIP device visible (libgpx has no CPU fallback)");
  // ndev == 1 with an EXPLICIT transport is a one-rank group: the group code (rank threads' entry,
  // ncclCommInitAll / LocalComm, the sharded schedule) on one device — how a one-GPU box exercises
  // the RCCL device-group path at all; ndev == 1 with GPX_TRANSPORT_AUTO stays the plain handle
  if (cfg->ndev > 1 || (cfg->ndev == 1 && cfg->transport != GPX_TRANSPORT_AUTO)) {
    for (int i = 0; i < cfg->ndev; ++i)
      if (cfg->devices[i] < 0 || cfg->devices[i] >= ndev)
        return fail(nullptr, GPX_E_ARG, "gpx_create: group device ordinal out of range");
    return create_group(out, cfg);
  }
  const int device = cfg->ndev == 1 ? cfg->devices[0] : cfg->device;
  if (device < 0 || device >= ndev)
    return fail(nullptr, GPX_E_ARG, "gpx_create: device ordinal out of range");
  gpx_handle* h = new (std::nothrow) gpx_handle();
  if (!h) return fail(nullptr, GPX_E_NOMEM, "gpx_create: out of host memory");
  h->cfg = *cfg;
  h->cfg.device = device;
  h->nb = nb;
  h->refine = cfg->refine;  // 0: adaptive
  if (const char* e = getenv("GPX_NB_SHARD")) h->nb_shard_env = atoi(e);
  if (h->nb_shard_env < 128 || h->nb_shard_env > 2048 || h->nb_shard_env % 128 != 0) h->nb_shard_env = 0;
  // tuning overrides; anything that is not a multiple of 128 in [128, 2048] is ignored
  auto block_env = [](const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return (v >= 128 && v <= 4096 && v % 128 == 0) ? v : dflt;
  };
  h->nb_solve = block_env("GPX_NB_SOLVE", h->nb_solve);
  h->nb_pred = block_env("GPX_NB_PRED", h->nb_pred);
  h->nb_pred_env = getenv("GPX_NB_PRED") != nullptr;
  int prio_lo = 0, prio_hi = 0;
  if (hipSetDevice(device) != hipSuccess ||
      hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
      hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithPriority(&h->st2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&h->st3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithFlags(&h->st4, hipStreamNonBlocking) != hipSuccess) {
    for (hipStream_t sx : {h->st, h->st2, h->st3})
      if (sx) (void)hipStreamDestroy(sx);
    delete h;
    return fail(nullptr, GPX_E_HIP, "gpx_create: hipSetDevice/hipStreamCreate failed");
  }
  *out = h;
  return GPX_OK;
}
GPX_CATCH_ALL

void gpx_destroy(gpx_handle* h) {
  if (!h) return;
  if (h->group) {
    destroy_group(h);
    delete h;
    return;
  }
  (void)hipSetDevice(h->cfg.device);
  if (h->st) (void)hipStreamSynchronize(h->st);
  if (h->st2) (void)hipStreamSynchronize(h->st2);
  if (h->st3) (void)hipStreamSynchronize(h->st3);
  if (h->st4) (void)hipStreamSynchronize(h->st4);
  if (h->st5) (void)hipStreamSynchronize(h->st5);
  for (DevBuf* b : {&h->X, &h->Xs, &h->ls, &h->K, &h->Winv, &h->P, &h->YT, &h->Y, &h->scalars,
                    &h->info, &h->Q, &h->Qs, &h->VT, &h->MT, &h->MTpart, &h->var, &h->meanout, &h->G,
                    &h->Dbuf, &h->Sbuf, &h->YTloc, &h->Cneg, &h->Sv, &h->AT, &h->Lfull, &h->GatherS,
                    &h->GatherR, &h->outM, &h->outV, &h->ZT, &h->ZTloc, &h->ZTpack, &h->gpart, &h->Wblk, &h->Ublk, &h->Tsol, &h->X64, &h->Y64, &h->Xs64,
                    &h->A64, &h->Aprev, &h->R64, &h->resv_ring, &h->RTloc, &h->P32out, &h->X32, &h->Y32, &h->RT32, &h->Q64, &h->Qs64, &h->Q32, &h->M64, &h->rn, &h->Zfew})
    release(*b);
  destroy_comm(h);
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->st) (void)hipStreamDestroy(h->st);
  if (h->st2) (void)hipStreamDestroy(h->st2);
  if (h->st3) (void)hipStreamDestroy(h->st3);
  if (h->st4) (void)hipStreamDestroy(h->st4);
  if (h->st5) (void)hipStreamDestroy(h->st5);
  delete h;
}

int gpx_fit(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
            const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter,
            int32_t mem_kind, int64_t* info) try {
  if (!h) return GPX_E_ARG;
  if (!X || !y || !lengthscale || !info) return fail(h, GPX_E_ARG, "gpx_fit: null argument");
  if (N <= 0 || d <= 0 || d > 32) return fail(h, GPX_E_ARG, "gpx_fit: need N > 0 and 1 <= d <= 32");
  if (k <= 0 || k > RHS_ROWS) return fail(h, GPX_E_ARG, "gpx_fit: need 1 <= k <= 64 target columns");
  if (n_ls != 1 && n_ls != d) return fail(h, GPX_E_ARG, "gpx_fit: n_ls must be 1 or d");
  if (mem_kind != GPX_MEM_HOST && mem_kind != GPX_MEM_DEVICE)
    return fail(h, GPX_E_ARG, "gpx_fit: bad mem_kind");
  if (!(sf2 > 0.0) || sn2 < 0.0 || jitter < 0.0)
    return fail(h, GPX_E_ARG, "gpx_fit: need sf2 > 0, sn2 >= 0, jitter >= 0");
  for (int i = 0; i < n_ls; ++i)
    if (!(lengthscale[i] > 0.0)) return fail(h, GPX_E_ARG, "gpx_fit: lengthscale must be > 0");
  if (N > (int64_t)INT_MAX - 4096) return fail(h, GPX_E_ARG, "gpx_fit: N too large");
  if (h->group) return group_fit(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->fitted = false;
  h->err.clear();
  h->phases.clear();  // an earlier call that failed mid-way must not leak its event pairs
  h->ev_used = 0;
  if (h->cfg.world > 1 || h->comm) {  // a 1-rank communicator also takes the sharded schedule
    if (h->cfg.dtype == GPX_MIXED) return mixed_fit(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
    if (h->cfg.dtype == GPX_F32) return shard_fit<float>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
    return shard_fit<double>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
  }

  if (h->cfg.dtype == GPX_MIXED)
    return mixed_fit(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
  if (h->cfg.dtype == GPX_F32)
    return fit_impl<float>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
  return fit_impl<double>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
}
GPX_CATCH_ALL

int gpx_fit_predict(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
                    const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter, const void* Xq,
                    int64_t M, void* mean, void* var, int32_t mem_kind, int64_t* info) try {
  if (!h) return GPX_E_ARG;
  if (!X || !y || !lengthscale || !info || !Xq || !mean) return fail(h, GPX_E_ARG, "gpx_fit_predict: null argument");
  if (N <= 0 || d <= 0 || d > 32 || M <= 0) return fail(h, GPX_E_ARG, "gpx_fit_predict: need N, M > 0 and 1 <= d <= 32");
  if (k <= 0 || k > RHS_ROWS) return fail(h, GPX_E_ARG, "gpx_fit_predict: need 1 <= k <= 64 target columns");
  if (n_ls != 1 && n_ls != d) return fail(h, GPX_E_ARG, "gpx_fit_predict: n_ls must be 1 or d");
  if (mem_kind != GPX_MEM_HOST && mem_kind != GPX_MEM_DEVICE) return fail(h, GPX_E_ARG, "gpx_fit_predict: bad mem_kind");
  if (!(sf2 > 0.0) || sn2 < 0.0 || jitter < 0.0)
    return fail(h, GPX_E_ARG, "gpx_fit_predict: need sf2 > 0, sn2 >= 0, jitter >= 0");
  for (int i = 0; i < n_ls; ++i)
    if (!(lengthscale[i] > 0.0)) return fail(h, GPX_E_ARG, "gpx_fit_predict: lengthscale must be > 0");
  if (N > (int64_t)INT_MAX - 4096) return fail(h, GPX_E_ARG, "gpx_fit_predict: N too large");
  if (h->group) return group_fit_predict(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, Xq, M, mean, var, mem_kind, info);
  // Shards (round 4): every rank's slice of the query points rides through the sharded factorisation as bordered rows of
  // its local row set (shard_fit with query points + shard_fused_tail) — in the split schedule, up to 8192 rows per rank.
  // Otherwise, and in the mixed mode (its refinement needs the factor first), the call IS the two calls (same results;
  // ABI v5 — it was GPX_E_UNSUPPORTED).  *info > 0: not positive definite, nothing predicted.
  if (h->cfg.world > 1 || h->comm || h->cfg.dtype == GPX_MIXED) {
    const char* se = getenv("GPX_SPLIT_STRIP");
    const char* fe = getenv("GPX_SHARD_FUSED");
    const int P = h->comm ? h->comm->world : 1;
    const bool ride = h->comm && h->cfg.dtype != GPX_MIXED && (!se || atoi(se) != 0) && (!fe || atoi(fe) != 0) &&
                      round_up((M + P - 1) / P, TILE) <= 8192;
    if (!ride) {
      const int rc2 = gpx_fit(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info);
      if (rc2 != GPX_OK || *info != 0) return rc2;
      return gpx_predict(h, Xq, M, mean, var, mem_kind);
    }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->fitted = false;
    h->err.clear();
    h->phases.clear();
    h->ev_used = 0;
    int rc2 = h->cfg.dtype == GPX_F32
                  ? shard_fit<float>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info, false, Xq, M)
                  : shard_fit<double>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info, false, Xq, M);
    if (rc2 != GPX_OK || !h->fitted) return rc2;
    h->phases.clear();
    h->ev_used = 0;
    return h->cfg.dtype == GPX_F32 ? shard_fused_tail<float>(h, M, mean, var, mem_kind)
                                   : shard_fused_tail<double>(h, M, mean, var, mem_kind);
  }
  // one batch of query points rides through the factorisation; more than that (ABI v5: it was GPX_E_UNSUPPORTED): the
  // first batch rides, the others go through the ordinary predict against the factor the pass leaves behind — query
  // rows are independent, so the split changes no bit of either part
  const int64_t Mb = std::min<int64_t>(M, pred_batch_rows(h, round_up(M, TILE), 0, false));
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->fitted = false;
  h->err.clear();
  h->phases.clear();
  h->ev_used = 0;
  const size_t es = h->cfg.dtype == GPX_F32 ? sizeof(float) : sizeof(double);
  int rc = h->cfg.dtype == GPX_F32
               ? fit_impl<float>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info, Xq, Mb)
               : fit_impl<double>(h, X, y, N, d, k, lengthscale, n_ls, sf2, sn2, jitter, mem_kind, info, Xq, Mb);
  if (rc != GPX_OK || !h->fitted) return rc;  // *info > 0: not positive definite, nothing predicted
  rc = h->cfg.dtype == GPX_F32 ? fused_predict_tail<float>(h, Mb, mean, var, mem_kind)
                               : fused_predict_tail<double>(h, Mb, mean, var, mem_kind);
  if (rc != GPX_OK || Mb == M) return rc;
  const gpx_timings first = h->tm;  // the rest through predict; its clocks are added to the pass's
  h->phases.clear();
  h->ev_used = 0;
  const void* Xr = (const char*)Xq + (size_t)Mb * d * es;
  void* mr = (char*)mean + (size_t)Mb * k * es;
  void* vr = var ? (char*)var + (size_t)Mb * es : nullptr;
  rc = h->cfg.dtype == GPX_F32 ? predict_impl<float>(h, Xr, M - Mb, mr, vr, mem_kind)
                               : predict_impl<double>(h, Xr, M - Mb, mr, vr, mem_kind);
  h->tm.kstar += first.kstar;
  h->tm.mean += first.mean;
  h->tm.trsm += first.trsm;
  h->tm.var += first.var;
  h->tm.d2h += first.d2h;
  h->tm.predict_total += first.predict_total;
  return rc;
}
GPX_CATCH_ALL

int gpx_predict(gpx_handle* h, const void* Xq, int64_t M, void* mean, void* var, int32_t mem_kind) try {
  if (!h) return GPX_E_ARG;
  if (!h->fitted) return fail(h, GPX_E_ARG, "gpx_predict: handle has no successful fit");
  if (!Xq || !mean || M <= 0) return fail(h, GPX_E_ARG, "gpx_predict: bad argument");
  if (mem_kind != GPX_MEM_HOST && mem_kind != GPX_MEM_DEVICE)
    return fail(h, GPX_E_ARG, "gpx_predict: bad mem_kind");
  if (h->group) return group_predict(h, Xq, M, mean, var, mem_kind);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->err.clear();
  h->phases.clear();
  h->ev_used = 0;
  if (h->cfg.world > 1 || h->comm) {
    if (h->cfg.dtype == GPX_MIXED) return mixed_predict(h, Xq, M, mean, var, mem_kind);
    if (h->cfg.dtype == GPX_F32) return shard_predict<float>(h, Xq, M, mean, var, mem_kind);
    return shard_predict<double>(h, Xq, M, mean, var, mem_kind);
  }
  if (h->cfg.dtype == GPX_MIXED) return mixed_predict(h, Xq, M, mean, var, mem_kind);
  if (h->cfg.dtype == GPX_F32) return predict_impl<float>(h, Xq, M, mean, var, mem_kind);
  return predict_impl<double>(h, Xq, M, mean, var, mem_kind);
}
GPX_CATCH_ALL

int gpx_get_alpha(gpx_handle* h, void* out) try {
  if (!h) return GPX_E_ARG;
  if (!h->fitted || !out) return fail(h, GPX_E_ARG, "gpx_get_alpha: no fit or null output");
  if (h->group) {  // alpha is replicated (or solved on demand from the replicated factor): rank 0 has it
    gpx_handle* m0 = h->group->members[0];
    const int rc = gpx_get_alpha(m0, out);
    if (rc != GPX_OK) h->err = m0->err;
    return rc;
  }
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (h->cfg.dtype == GPX_MIXED) return mixed_alpha(h, out);
  if (h->cfg.dtype == GPX_F32) return alpha_impl<float>(h, out);
  return alpha_impl<double>(h, out);
}
GPX_CATCH_ALL

int gpx_lml_grad(gpx_handle* h, double* lml, double* grad) try {
  if (!h) return GPX_E_ARG;
  if (!h->fitted || !lml || !grad) return fail(h, GPX_E_ARG, "gpx_lml_grad: no fit or null output");
  if (h->cfg.dtype != GPX_F64) return fail(h, GPX_E_UNSUPPORTED, "gpx_lml_grad: fp64 handles only");
  if (h->group) return group_lml_grad(h, lml, grad);
  if (h->cfg.world > 1 && !h->comm) return fail(h, GPX_E_ARG, "gpx_lml_grad: sharded handle without a communicator");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->err.clear();
  h->phases.clear();
  h->ev_used = 0;
  if (h->comm && !h->repl) return shard_lml_grad_dist(h, lml, grad);  // factor only held distributed
  return lml_grad_impl(h, lml, grad);
}
GPX_CATCH_ALL

int gpx_release_scratch(gpx_handle* h) try {
  if (!h) return GPX_E_ARG;
  std::vector<gpx_handle*> hs;
  if (h->group)
    hs = h->group->members;
  else
    hs.push_back(h);
  for (gpx_handle* m : hs) {
    HIPCHK(h, hipSetDevice(m->cfg.device));
    for (hipStream_t sx : {m->st, m->st2, m->st3, m->st4, m->st5})
      if (sx) HIPCHK(h, hipStreamSynchronize(sx));
    for (DevBuf* b : {&m->ZT, &m->ZTloc, &m->ZTpack, &m->gpart, &m->MTpart, &m->VT, &m->Tsol, &m->Q, &m->Qs, &m->MT, &m->Sv, &m->Q64, &m->Qs64, &m->Q32,
                      &m->M64, &m->GatherS, &m->GatherR, &m->outM, &m->outV})
      release(*b);
  }
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_logdet(gpx_handle* h, double* out) try {
  if (!h) return GPX_E_ARG;
  if (!h->fitted || !out) return fail(h, GPX_E_ARG, "gpx_logdet: no fit or null output");
  *out = h->group ? h->group->members[0]->logdet : h->logdet;
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_set_flags(gpx_handle* h, int32_t flags) try {
  if (!h || (flags & ~GPX_FLAG_PROFILE)) return GPX_E_ARG;
  h->cfg.flags = flags;
  if (h->group)
    for (gpx_handle* m : h->group->members) m->cfg.flags = flags;
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_get_timings(gpx_handle* h, gpx_timings* out) try {
  if (!h || !out) return GPX_E_ARG;
  *out = h->group ? h->group->members[0]->tm : h->tm;  // a group reports rank 0's clocks
  return GPX_OK;
}
GPX_CATCH_ALL

// ---- kernel unit-test entry points -------------------------------------------------------
namespace {
struct Scratch {  // a throw-away handle-like context for the host-buffer entry points
  gpx_handle h;
  bool ok = false;
  Scratch() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    h.cfg.device = dev;
    ok = hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&h.st2, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&h.st3, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&h.st4, hipStreamNonBlocking) == hipSuccess;
  }
  ~Scratch() {
    for (hipStream_t s : {h.st, h.st2, h.st3, h.st4})
      if (s) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
      }
    for (auto e : h.ev_pool) (void)hipEventDestroy(e);
    release(h.resv_ring);
  }
};
#define TCHK(call)                                   \
  do {                                               \
    if ((call) != hipSuccess) {                      \
      g_create_error = #call " failed";             \
      rc = GPX_E_HIP;                                \
      goto done;                                     \
    }                                                \
  } while (0)
}  // namespace

int gpx_kernel_matrix(int32_t kernel, const double* A, int64_t na, const double* B, int64_t nb_,
                      int32_t d, const double* lengthscale, int32_t n_ls, double sf2,
                      double diag_add, double* K) try {
  if (!A || !K || !lengthscale || na <= 0 || d <= 0 || d > 32 || (n_ls != 1 && n_ls != d))
    return GPX_E_ARG;
  if (kernel != GPX_KERNEL_RBF && kernel != GPX_KERNEL_MATERN52) return GPX_E_ARG;
  const bool sym = (B == nullptr);
  const int64_t nbb = sym ? na : nb_;
  if (nbb <= 0) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const int64_t napad = round_up(na, 64), nbpad = round_up(nbb, 64), ld = nbpad;
  double *dA = nullptr, *dB = nullptr, *dAs = nullptr, *dBs = nullptr, *dls = nullptr, *dK = nullptr;
  std::vector<double> host((size_t)napad * ld);
  int rc = GPX_OK;
  TCHK(hipMalloc(&dA, (size_t)na * d * 8));
  TCHK(hipMalloc(&dAs, (size_t)napad * d * 8));
  TCHK(hipMalloc(&dls, 32 * 8));
  TCHK(hipMalloc(&dK, (size_t)napad * ld * 8));
  TCHK(hipMemcpyAsync(dA, A, (size_t)na * d * 8, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(dls, lengthscale, (size_t)n_ls * 8, hipMemcpyHostToDevice, st));
  TCHK(hipMemsetAsync(dK, 0, (size_t)napad * ld * 8, st));
  launch_scale_points(dA, na, napad, d, dls, n_ls, dAs, st);
  if (sym) {
    launch_kbuild_sym(kernel, dAs, na, napad, d, sf2, diag_add, dK, ld, st);
  } else {
    TCHK(hipMalloc(&dB, (size_t)nbb * d * 8));
    TCHK(hipMalloc(&dBs, (size_t)nbpad * d * 8));
    TCHK(hipMemcpyAsync(dB, B, (size_t)nbb * d * 8, hipMemcpyHostToDevice, st));
    launch_scale_points(dB, nbb, nbpad, d, dls, n_ls, dBs, st);
    launch_kbuild_cross(kernel, dAs, na, napad, dBs, nbb, nbpad, d, sf2, dK, ld, st);
  }
  TCHK(hipMemcpyAsync(host.data(), dK, (size_t)napad * ld * 8, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
  for (int64_t i = 0; i < na; ++i)
    memcpy(K + i * nbb, host.data() + i * ld, (size_t)nbb * 8);  // sym: upper part is zero (not built)
done:
  for (double* p : {dA, dB, dAs, dBs, dls, dK})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

int gpx_potrf(double* A, int64_t n, int32_t block, int64_t* info) try {
  if (!A || !info || n <= 0 || n % 64 != 0) return GPX_E_ARG;
  const int nb = block == 0 ? 1024 : block;
  if (nb % 128 != 0 || nb < 128) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const int64_t ld = n + LD_SKEW, ldp = nb + LD_SKEW;
  double *dA = nullptr, *dW = nullptr, *dP = nullptr, *dWb = nullptr, *dUb = nullptr;
  int* dInfo = nullptr;
  int hinfo = INT_MAX;
  int rc = GPX_OK;
  InvWork<double> iw;
  TCHK(hipMalloc(&dA, (size_t)n * ld * 8));
  TCHK(hipMalloc(&dW, (size_t)(n / 64) * 4096 * 8));
  TCHK(hipMalloc(&dP, (size_t)2 * n * ldp * 8));
  TCHK(hipMalloc(&dInfo, 64));
  TCHK(hipMalloc(&dWb, (size_t)((n + nb - 1) / nb) * nb * nb * 8));
  TCHK(hipMalloc(&dUb, (size_t)nb * ldp * 8));
  iw.W = dWb;
  iw.U = dUb;
  iw.ldu = ldp;
  iw.nbw = nb;
  iw.aux = sc.h.st3;
  if ((rc = flag_handover_probe(&sc.h))) {
    g_create_error = sc.h.err;
    goto done;
  }
  TCHK(hipMemcpy2DAsync(dA, (size_t)ld * 8, A, (size_t)n * 8, (size_t)n * 8, (size_t)n, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(dInfo, &hinfo, sizeof(int), hipMemcpyHostToDevice, st));
  if ((rc = chol_enqueue(&sc.h, dA, ld, n, nb, dW, dP, dP + n * ldp, ldp, dInfo, 0, false, 0, &iw))) goto done;
  TCHK(hipMemcpy2DAsync(A, (size_t)n * 8, dA, (size_t)ld * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost, st));
  TCHK(hipMemcpyAsync(&hinfo, dInfo, sizeof(int), hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
  if (hinfo < 0) {
    g_create_error = "a stream parked on a device flag timed out (kernels serialised across streams? set GPX_CHAIN_FLAG=0)";
    rc = GPX_E_HIP;
    goto done;
  }
  *info = (hinfo == INT_MAX) ? 0 : hinfo;
done:
  for (void* p : {(void*)dA, (void*)dW, (void*)dP, (void*)dInfo, (void*)dWb, (void*)dUb})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

int gpx_trsm(double* X, int64_t m, const double* L, int64_t nb) try {
  if (!X || !L || m <= 0 || nb <= 0 || m % 64 != 0 || nb % 64 != 0) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const int64_t ldx = nb + LD_SKEW, ldl = nb + LD_SKEW;
  double *dX = nullptr, *dL = nullptr, *dL2 = nullptr, *dW = nullptr;
  int* dInfo = nullptr;
  int hinfo = INT_MAX;
  int rc = GPX_OK;
  TCHK(hipMalloc(&dX, (size_t)m * ldx * 8));
  TCHK(hipMalloc(&dL, (size_t)nb * ldl * 8));
  TCHK(hipMalloc(&dL2, (size_t)nb * ldl * 8));
  TCHK(hipMalloc(&dW, (size_t)(nb / 64) * 4096 * 8));
  TCHK(hipMalloc(&dInfo, 64));
  TCHK(hipMemcpy2DAsync(dX, (size_t)ldx * 8, X, (size_t)nb * 8, (size_t)nb * 8, (size_t)m, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpy2DAsync(dL, (size_t)ldl * 8, L, (size_t)nb * 8, (size_t)nb * 8, (size_t)nb, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(dInfo, &hinfo, sizeof(int), hipMemcpyHostToDevice, st));
  // inverse diagonal blocks of the GIVEN factor: potf2 of (L_qq L_qq^T) reproduces L_qq up to
  // rounding, so build them from a scratch copy of the products instead.
  {
    std::vector<double> G((size_t)nb / 64 * 4096);
    for (int64_t q = 0; q < nb / 64; ++q)
      for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
          double s = 0.0;
          for (int t = 0; t <= std::min(i, j); ++t)
            s += L[(q * 64 + i) * nb + q * 64 + t] * L[(q * 64 + j) * nb + q * 64 + t];
          G[(size_t)q * 4096 + i * 64 + j] = s;
        }
    TCHK(hipMemcpyAsync(dL2, G.data(), G.size() * 8, hipMemcpyHostToDevice, st));
    TCHK(hipStreamSynchronize(st));
  }
  for (int64_t q = 0; q < nb / 64; ++q)
    launch_potf2_64(dL2 + q * 4096, 64, dW + q * 4096, q * 64, dInfo, st);
  launch_trsm_rlt<double>(dX, ldx, m, dL, ldl, dW, (int)nb, nullptr, 0, st);
  if (take_launch_error()) {
    rc = GPX_E_ARG;
    goto done;
  }
  TCHK(hipMemcpy2DAsync(X, (size_t)nb * 8, dX, (size_t)ldx * 8, (size_t)nb * 8, (size_t)m, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
done:
  for (void* p : {(void*)dX, (void*)dL, (void*)dL2, (void*)dW, (void*)dInfo})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

int gpx_gemm_nt(double* C, int64_t m, int64_t n, const double* A, const double* B, int64_t k,
                int32_t lower) try {
  if (!C || !A || !B || m <= 0 || n <= 0 || k <= 0 || m % 64 || n % 64 || k % 16) return GPX_E_ARG;
  if (lower && m != n) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const int64_t ldc = n + LD_SKEW, lda = k + LD_SKEW;
  const int tile = (m % 128 == 0 && n % 128 == 0) ? 128 : 64;
  double *dC = nullptr, *dA = nullptr, *dB = nullptr;
  int rc = GPX_OK;
  TCHK(hipMalloc(&dC, (size_t)m * ldc * 8));
  TCHK(hipMalloc(&dA, (size_t)m * lda * 8));
  TCHK(hipMalloc(&dB, (size_t)n * lda * 8));
  TCHK(hipMemcpy2DAsync(dC, (size_t)ldc * 8, C, (size_t)n * 8, (size_t)n * 8, (size_t)m, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpy2DAsync(dA, (size_t)lda * 8, A, (size_t)k * 8, (size_t)k * 8, (size_t)m, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpy2DAsync(dB, (size_t)lda * 8, B, (size_t)k * 8, (size_t)k * 8, (size_t)n, hipMemcpyHostToDevice, st));
  launch_gemm_nt_fixed(tile, dC, ldc, dA, lda, dB, lda, m, n, k, lower, 0, st);
  TCHK(hipMemcpy2DAsync(C, (size_t)n * 8, dC, (size_t)ldc * 8, (size_t)n * 8, (size_t)m, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
done:
  for (double* p : {dC, dA, dB})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

int gpx_mfma_probe(const double* A, const double* B, double* D) try {
  if (!A || !B || !D) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  double* d = nullptr;
  int rc = GPX_OK;
  TCHK(hipMalloc(&d, (64 + 64 + 256) * 8));
  TCHK(hipMemcpyAsync(d, A, 64 * 8, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(d + 64, B, 64 * 8, hipMemcpyHostToDevice, st));
  launch_mfma_probe(d, d + 64, d + 128, st);
  TCHK(hipMemcpyAsync(D, d + 128, 256 * 8, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
done:
  if (d) (void)hipFree(d);
  return rc;
}
GPX_CATCH_ALL

int gpx_path_distance(const double* paths, int64_t P, const double* cents, int64_t C, int32_t L,
                      double* D, int32_t mem_kind) try {
  if (!paths || !cents || !D || P <= 0 || C <= 0 || L <= 0 || L > 64) return GPX_E_ARG;
  if (mem_kind != GPX_MEM_HOST && mem_kind != GPX_MEM_DEVICE) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const bool host = mem_kind == GPX_MEM_HOST;
  double *dp = nullptr, *dc = nullptr, *dD = nullptr;
  int rc = GPX_OK;
  TCHK(hipMalloc(&dc, (size_t)C * L * 16));
  TCHK(hipMemcpyAsync(dc, cents, (size_t)C * L * 16, host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
  if (host) {
    TCHK(hipMalloc(&dp, (size_t)P * L * 16));
    TCHK(hipMalloc(&dD, (size_t)P * C * 8));
    TCHK(hipMemcpyAsync(dp, paths, (size_t)P * L * 16, hipMemcpyHostToDevice, st));
  }
  for (int64_t c0 = 0; c0 < C; c0 += 64)
    launch_path_distance(host ? dp : paths, P, dc + c0 * L * 2, (int)std::min<int64_t>(64, C - c0), L,
                         (host ? dD : D) + c0, C, st);
  if (host) TCHK(hipMemcpyAsync(D, dD, (size_t)P * C * 8, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
done:
  for (double* p : {dp, dc, dD})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

int gpx_mfma_probe_f32(const float* A, const float* B, float* D) try {
  if (!A || !B || !D) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  float* d = nullptr;
  int rc = GPX_OK;
  TCHK(hipMalloc(&d, (64 + 64 + 256) * 4));
  TCHK(hipMemcpyAsync(d, A, 64 * 4, hipMemcpyHostToDevice, st));
  TCHK(hipMemcpyAsync(d + 64, B, 64 * 4, hipMemcpyHostToDevice, st));
  launch_mfma_probe_f32(d, d + 64, d + 128, st);
  TCHK(hipMemcpyAsync(D, d + 128, 256 * 4, hipMemcpyDeviceToHost, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
done:
  if (d) (void)hipFree(d);
  return rc;
}
GPX_CATCH_ALL

// The tile engine alone, operands resident: C (n x n) op= A (n x k) A'(n x k)^T on zero-filled device
// buffers, `iters` back-to-back launches after one warm-up; *ms = mean launch time.
extern "C++" {
template <typename T>
int gemm_bench_t(int64_t n, int64_t k, int32_t lower, int32_t mode, int32_t iters, double* ms) {
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  const int64_t ldc = n + ld_skew<T>(), lda = k + ld_skew<T>();
  T *dC = nullptr, *dA = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float t = 0.f;
  int rc = GPX_OK;
  TCHK(hipMalloc(&dC, (size_t)n * ldc * sizeof(T)));
  TCHK(hipMalloc(&dA, (size_t)2 * n * lda * sizeof(T)));
  TCHK(hipMemsetAsync(dC, 0, (size_t)n * ldc * sizeof(T), st));
  TCHK(hipMemsetAsync(dA, 0, (size_t)2 * n * lda * sizeof(T), st));
  TCHK(hipEventCreate(&e0));
  TCHK(hipEventCreate(&e1));
  launch_gemm_nt_fixed<T>(128, dC, ldc, dA, lda, dA + n * lda, lda, n, n, k, lower, mode, st);
  TCHK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i)
    launch_gemm_nt_fixed<T>(128, dC, ldc, dA, lda, dA + n * lda, lda, n, n, k, lower, mode, st);
  TCHK(hipEventRecord(e1, st));
  TCHK(hipStreamSynchronize(st));
  TCHK(hipGetLastError());
  TCHK(hipEventElapsedTime(&t, e0, e1));
  *ms = (double)t / iters;
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (dC) (void)hipFree(dC);
  if (dA) (void)hipFree(dA);
  return rc;
}
}  // extern "C++"

int gpx_debug_gemm_bench(int32_t dtype, int64_t n, int64_t k, int32_t lower, int32_t mode, int32_t iters,
                         double* ms_per_launch) try {
  if (!ms_per_launch || n <= 0 || n % 128 || k <= 0 || k % 32 || iters <= 0 || (lower != 0 && lower != 1) ||
      (mode != 0 && mode != 1) || n > 131072 || k > 8192)
    return GPX_E_ARG;
  if (dtype == GPX_F64) return gemm_bench_t<double>(n, k, lower, mode, iters, ms_per_launch);
  if (dtype == GPX_F32) return gemm_bench_t<float>(n, k, lower, mode, iters, ms_per_launch);
  return GPX_E_ARG;
}
GPX_CATCH_ALL

int gpx_debug_set_delay(uint64_t seed) {
  debug_set_delay(seed);
  return GPX_OK;
}

int gpx_debug_tile_map(int32_t kind, int64_t tm, int64_t tn, int32_t P, int32_t tpb, int32_t c, int32_t* out,
                       int64_t cap, int64_t* count) try {
  if (!out || !count || tm <= 0 || cap <= 0 || kind < 0 || kind > 2) return GPX_E_ARG;
  if (kind == 1 && (tn <= 0 || P <= 0 || tpb <= 0 || c < 0)) return GPX_E_ARG;
  BcMask bc{P, tpb};  // cyclic dealing in its old shorthand: row block of local block j = j P + c
  bc.gc0 = -c;
  const int64_t n = debug_tile_map(kind, tm, tn, bc, out, cap);
  if (n < 0) return GPX_E_ARG;
  *count = n;
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_debug_stair_map(int64_t tm, int64_t tn, int32_t P, int32_t tpb, int32_t r, int32_t lbf, int32_t gc0,
                        int32_t snake, int32_t* out, int64_t cap, int64_t* count) try {
  if (!out || !count || tm <= 0 || tn <= 0 || cap <= 0 || P <= 0 || tpb <= 0 || r < 0 || r >= P || lbf < 0 || gc0 < 0)
    return GPX_E_ARG;
  BcMask bc{P, tpb};
  bc.r = r;
  bc.lbf = lbf;
  bc.gc0 = gc0;
  bc.snake = snake ? 1 : 0;
  if (bc.row_tile(0) < 0) return GPX_E_ARG;  // the first local row block must not lie left of the first column block
  const int64_t n = debug_tile_map(1, tm, tn, bc, out, cap);
  if (n < 0) return GPX_E_ARG;
  *count = n;
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_debug_deal(int32_t P, int32_t snake, int64_t nblk, int32_t* owner, int64_t* local, int64_t* upto) try {
  if (P <= 0 || nblk <= 0 || !owner || !local || !upto) return GPX_E_ARG;
  const Deal dl{P, snake ? 1 : 0};
  for (int64_t g = 0; g < nblk; ++g) {
    owner[g] = dl.owner(g);
    local[g] = dl.local(g);
    if (dl.global(owner[g], local[g]) != g) return GPX_E_ARG;  // the two directions must agree
    for (int r = 0; r < P; ++r) upto[g * P + r] = dl.upto(g, r);
  }
  return GPX_OK;
}
GPX_CATCH_ALL

int gpx_microbench(double* mfma_tflops, double* copy_gbs) try {
  if (!mfma_tflops || !copy_gbs) return GPX_E_ARG;
  Scratch sc;
  if (!sc.ok) return GPX_E_HIP;
  hipStream_t st = sc.h.st;
  double *sink = nullptr, *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const int64_t count = (int64_t)1 << 28;  // 2 GiB per buffer
  const int64_t skew = 0;  // (round 3: a skewed destination measured slower than an aligned one: tools/copy_bw.hip)
  int iters = 65536;  // ~0.23 s: long enough to be past the clock ramp (a 14 ms loop read 75 TF,
                      // the sustained rate is 77.8); GPX_MICROBENCH_ITERS overrides
  if (const char* e = getenv("GPX_MICROBENCH_ITERS")) {
    const long v = atol(e);
    if (v >= 64 && v <= (1L << 20)) iters = (int)v;
  }
  const int blocks = 256 * 8;
  float ms = 0.f;
  int rc = GPX_OK;
  TCHK(hipMalloc(&sink, 64));
  TCHK(hipMalloc(&src, (size_t)count * 8));
  TCHK(hipMalloc(&dst, (size_t)(count + skew) * 8));
  TCHK(hipEventCreate(&e0));
  TCHK(hipEventCreate(&e1));
  TCHK(hipMemsetAsync(src, 0x11, (size_t)count * 8, st));
  launch_mfma_loop(sink, 64, blocks, st);  // warm-up
  TCHK(hipEventRecord(e0, st));
  launch_mfma_loop(sink, iters, blocks, st);
  TCHK(hipEventRecord(e1, st));
  TCHK(hipEventSynchronize(e1));
  TCHK(hipEventElapsedTime(&ms, e0, e1));
  *mfma_tflops = (double)blocks * 4 * (double)iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
  for (int r = 0; r < 4; ++r) launch_copy(src, dst + skew, count, st);  // warm-up
  TCHK(hipEventRecord(e0, st));
  for (int r = 0; r < 20; ++r) launch_copy(src, dst + skew, count, st);
  TCHK(hipEventRecord(e1, st));
  TCHK(hipEventSynchronize(e1));
  TCHK(hipEventElapsedTime(&ms, e0, e1));
  *copy_gbs = 20.0 * 2.0 * (double)count * 8.0 / (ms * 1e-3) / 1e9;
  TCHK(hipGetLastError());
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  for (double* p : {sink, src, dst})
    if (p) (void)hipFree(p);
  return rc;
}
GPX_CATCH_ALL

}  // extern "C"
