/* gpx.h — C ABI of libgpx.so: exact Gaussian-process regression on MI355X (gfx950).
 *
 * Drop-in boundary for the GP fit/predict hot path.  The upstream
 * reference (/root/reference/GPmap.py) has NO fit/predict, no kernel matrix and no
 * Cholesky (its only linalg call is np.linalg.norm, GPmap.py:120), so there is no
 * reference FFI to mirror: each entry point below cites the SURVEY.md §8 row that
 * defines it and, where one exists, the nearest reference code.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ types, no exceptions, no abort().
 *   - return 0 = ok, <0 = API misuse / HIP / RCCL error (text via gpx_last_error).
 *   - numerical failure is NOT an error code: *info > 0 is the 1-based index of the
 *     first non-positive pivot (LAPACK potrf convention) and the call returns 0.
 *   - all matrices are row-major, C-contiguous unless a leading dimension is given.
 *   - the caller owns every pointer it passes; the library owns every device
 *     allocation inside a handle and frees it in gpx_destroy.
 *   - a handle is not thread-safe; distinct handles may be used from distinct threads.
 *   - calls are synchronous at return.
 */
#ifndef GPX_H_
#define GPX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_ABI_VERSION 5 /* v5: gpx_timings.handover_*, fp32 / mixed shards; v4: + gpx_fit_predict (v3: gpx_set_flags, refine, one-rank groups) */

/* kernel family — SURVEY.md §8 row a1 (nearest reference code: the pairwise
 * distance loop trajectories.calc_distance, GPmap.py:114-121, and the unused
 * scipy.spatial.distance import, GPmap.py:10). */
#define GPX_KERNEL_RBF 0      /* sf2 * exp(-r^2/2)                            */
#define GPX_KERNEL_MATERN52 1 /* sf2 * (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r)  */

#define GPX_F64 0
#define GPX_F32 1
#define GPX_MIXED 2 /* inputs / outputs double; factorisation in fp32, alpha refined in fp64 against the  \
                       matrix-free fp64 kernel until ||y - K alpha|| <= 1e-10 ||y||: fp64-GRADE posterior \
                       MEAN, fp32-GRADE VARIANCE (sf2 - ||L^-1 k*||^2 through the fp32 factor, never      \
                       refined: errors of ~1e-6 sf2 absolute, i.e. percents of a small variance) (configs[4]) */

#define GPX_MEM_HOST 0   /* pointers are host memory; library copies H2D/D2H   */
#define GPX_MEM_DEVICE 1 /* pointers are device memory on the handle's device  */

#define GPX_FLAG_PROFILE 1 /* per-launch hipEvent timing of the Cholesky sub-phases */

/* error codes */
#define GPX_OK 0
#define GPX_E_ARG (-1)     /* bad argument / state                               */
#define GPX_E_HIP (-2)     /* HIP runtime error                                  */
#define GPX_E_COMM (-3)    /* RCCL / communicator error                          */
#define GPX_E_UNSUPPORTED (-4)
#define GPX_E_NOMEM (-5)

typedef struct gpx_handle gpx_handle;

/* transport of a single-process device group (gpx_config.ndev > 1) */
#define GPX_TRANSPORT_AUTO 0  /* RCCL when the listed devices are distinct, else LOCAL          */
#define GPX_TRANSPORT_RCCL 1  /* ncclCommInitAll inside the process, one communicator per device */
#define GPX_TRANSPORT_LOCAL 2 /* stream-ordered peer copies between the ranks' device buffers,   \
                                 hipEvents between their streams; no library besides HIP.  Also   \
                                 valid with one device listed several times (ranks share it).    */
#define GPX_MAX_GROUP 8

typedef struct gpx_config {
  int32_t kernel;  /* GPX_KERNEL_*                                   */
  int32_t dtype;   /* GPX_F64 | GPX_F32 | GPX_MIXED                  */
  int32_t device;  /* HIP device ordinal this handle computes on (ndev <= 1)                 */
  int32_t block;   /* Cholesky panel width nb (multiple of 128, <= 4096), 0 = the library's choice: 1024; 2048 from N = 40960 on */
  int32_t rank;    /* process-per-GPU shard: this process' rank (0 if world==1)              */
  int32_t world;   /* process-per-GPU shard: number of processes sharing the Gram matrix     */
  int32_t flags;   /* GPX_FLAG_*                                     */
  /* Single-process multi-GPU (SURVEY.md §8b "Threading"): ndev > 1 makes the handle a GROUP —
   * one rank per entry of devices[], one worker thread per rank inside the calling process,
   * the same row-block-cyclic schedule as the process-per-GPU shard.  gpx_fit / gpx_predict /
   * gpx_get_alpha / ... on the group handle are ordinary blocking calls of a plain caller (no
   * launcher, no torch.distributed); rank/world must be 0/1 then. */
  int32_t ndev;                     /* 0 or 1: one device (`device`);  2..GPX_MAX_GROUP: group; 1 with an explicit \
                                       transport (not AUTO): a ONE-rank group on devices[0] (the group / RCCL   \
                                       code path on a single GPU: tests)                                        */
  int32_t devices[GPX_MAX_GROUP];   /* HIP ordinals of the group's ranks                       */
  int32_t transport;                /* GPX_TRANSPORT_*                                         */
  int32_t refine;                   /* GPX_MIXED: 0 = refine until the relative residual is <= 1e-10 or stops  \
                                       contracting (at most 12 iterations); > 0 = exactly that many      */
  int32_t reserved[2];
} gpx_config;

/* per-phase wall times (ms, hipEvent on the handle's stream) of the LAST
 * gpx_fit / gpx_predict — SURVEY.md §8(d). */
typedef struct gpx_timings {
  double h2d, kbuild, chol, solve, logdet, fit_total;          /* gpx_fit     */
  double kstar, mean, trsm, var, d2h, predict_total;           /* gpx_predict */
  double comm;                                                 /* RCCL time inside chol (sharded) */
  /* Cholesky sub-phases, filled only with GPX_FLAG_PROFILE: */
  double chol_diag, chol_trsm, chol_strip, chol_syrk; /* summed ms (diag/trsm run on the look-ahead stream) */
  double syrk_flops;                       /* algorithmic flops n(n+1) nb of the 128-tile trailing-update launches \
                                              (chol_syrk / syrk_launches cover the same launches; the last few,     \
                                              under-filled updates run as 64-tiles and are booked under chol_strip) */
  int64_t syrk_launches;
  double kbuild_bytes;                     /* algorithmic bytes of the kernel build */
  double grad_trtri, grad_trace, grad_total; /* gpx_lml_grad: L^-T build, fused K^-1 trace pass, whole call */
  double refine;                           /* GPX_MIXED: ms spent refining alpha in fp64 */
  double refine_resid0, refine_resid;      /* ||y - K alpha|| / ||y|| before / after the refinement */
  double refine_iters;                     /* GPX_MIXED: refinement iterations the last fit ran */
  /* ABI v5: how the streams of the last fit handed over inside the diagonal chain.  1 = device flags (a kernel parked  \
     on the side stream polls a word the POTF2 publishes), 0 = hipEvents — chosen by a ~100 us self-test at the handle's \
     first fit (does a parked kernel see a store launched later on another stream?), by GPX_CHAIN_FLAG=0|1, or because    \
     rocprofv3 counter collection is on.  Same kernels and bit-identical results either way. */
  double handover_flags;
  double handover_retries;                 /* fits of this handle re-run with hipEvents after a parked stream timed out */
} gpx_timings;

/* ---- lifecycle ------------------------------------------------------------- */
int gpx_abi_version(void);
int gpx_device_count(int* count);
int gpx_create(gpx_handle** out, const gpx_config* cfg);
void gpx_destroy(gpx_handle* h);
const char* gpx_last_error(gpx_handle* h); /* h may be NULL: last error of gpx_create */

/* ---- hot path (SURVEY.md §8 rows a1,a3,a4 = fit; a2,a5,a6 = predict) --------- */
/* K = sf2 k(X,X) + (sn2+jitter) I;  L = chol(K);  alpha = L^-T L^-1 y.
 * X (N,d), y (N,k) row-major, in the dtype of the handle (GPX_F64 and GPX_MIXED: double,
 * GPX_F32: float — everything including the factorisation then runs in fp32: config 5, the
 * precision study; GPX_MIXED: fp32 factorisation + fp64 refinement, k <= 8).  ABI v5: every dtype also on
 * a shard / device group (the row-block shard runs in the handle's element type; the mixed mode's fp64
 * refinement is replicated work on every rank, its fp32 solves local on a replicated factor and collective
 * on a factor that is only held distributed).  lengthscale: n_ls = 1 or d. */
int gpx_fit(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
            const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter,
            int32_t mem_kind, int64_t* info);

/* mean (M,k) = K* alpha;  var (M) = sf2 - colsumsq(L^-1 K*^T)  (latent variance,
 * raw — not clamped).  var may be NULL (mean only). */
int gpx_predict(gpx_handle* h, const void* Xs, int64_t M, void* mean, void* var,
                int32_t mem_kind);

/* gpx_fit and gpx_predict (with variance) of ONE batch of query points as one call (ABI v4): the M cross-kernel
 * rows K(Xs, X) ride through the blocked factorisation as bordered rows — the way the right-hand sides already
 * do — and leave it as V^T = (L^-1 K*^T)^T, so the variance solve costs no pass of its own: its M N^2 flops are
 * rows of the trailing updates (at small N they run on the CUs the serial diagonal chain leaves idle; N = 8192,
 * M = 4096: 12.6 -> 11.5 ms per step).  Same results as the two calls up to the rounding of a different
 * summation order; the handle is fitted afterwards exactly as after gpx_fit (gpx_predict, gpx_get_alpha,
 * gpx_lml_grad ... work on it).  *info > 0: not positive definite, nothing was predicted.  ABI v5: every handle
 * accepts the call (v4: GPX_E_UNSUPPORTED beyond single-device fp64 / fp32 handles and one batch).  Single device,
 * GPX_F64 / GPX_F32: one predict batch (8192 rows) rides, further batches go through the ordinary predict against the
 * factor the pass leaves behind.  Shards and device groups (GPX_F64 / GPX_F32): rank r's slice of the query points —
 * ceil(M / world) rounded up to 128 — rides through ITS part of the sharded factorisation as bordered rows of its local
 * row set (up to 8192 rows per rank; a collective call: every rank of a shard makes it with the same arguments), in
 * both solve modes; GPX_SHARD_FUSED=0 runs the two calls instead.  GPX_MIXED handles run the call as gpx_fit +
 * gpx_predict (the refinement needs the factor first).  Mirrors the reference-side usage `gp.fit(X, y);
 * gp.predict(Xs)`. */
int gpx_fit_predict(gpx_handle* h, const void* X, const void* y, int64_t N, int32_t d, int32_t k,
                    const double* lengthscale, int32_t n_ls, double sf2, double sn2, double jitter, const void* Xs,
                    int64_t M, void* mean, void* var /* may be NULL */, int32_t mem_kind, int64_t* info);

int gpx_get_alpha(gpx_handle* h, void* out /* (N,k) host */);
/* Log marginal likelihood of the last fit and its gradient w.r.t. the LOG hyper-parameters —
 * SURVEY.md §8(f) row 1 ("log marginal likelihood + hyper-parameter gradient hooks"; no anchor
 * in GPmap.py):  *lml = -1/2 sum_c y_c^T alpha_c - k/2 log|K| - N k/2 log 2 pi,
 * grad[0..n_ls) = d lml / d log lengthscale, grad[n_ls] = d / d log sf2, grad[n_ls+1] = d / d log sn2
 * (n_ls as passed to gpx_fit).  Costs about two more factorisations' worth of MFMA work
 * (L^-T, then K^-1 = L^-T L^-1 consumed tile by tile as it is formed) and one extra N x N
 * buffer; K^-1 itself is never stored.  fp64 handles.  Sharded handles and groups: collective
 * (every rank calls it, every rank gets the same numbers).  Replicated-factor mode: both passes are
 * split over the ranks with one all-gather of L^-T between them.  Factor only held distributed
 * (C4-sized problems): L^-T is built by the distributed forward substitution of the variance path,
 * each rank keeps the columns that belong to its own row blocks (N^2 / P numbers), contracts the
 * trace over them, and ntheta numbers are all-reduced (round 3; was GPX_E_UNSUPPORTED). */
int gpx_lml_grad(gpx_handle* h, double* lml, double* grad);
int gpx_logdet(gpx_handle* h, double* out);
/* Frees what only the NEXT predict / gradient call would use (the V^T batch, the L^-T buffer of
 * gpx_lml_grad, per-tile partials, compact block buffers); the fit itself (factor, alpha, block
 * inverses) stays valid.  Buffers grow on demand and are otherwise kept for reuse: call this
 * between a gradient and a large predict when N is close to what the card holds (at N = 131072 the
 * factor and L^-T are 137 GB each). */
int gpx_release_scratch(gpx_handle* h);
int gpx_get_timings(gpx_handle* h, gpx_timings* out);
/* Replaces gpx_config.flags of an existing handle (GPX_FLAG_PROFILE on / off between calls: bench.py
 * prices the flag on ONE handle, same buffers).  Groups: applied to every member. */
int gpx_set_flags(gpx_handle* h, int32_t flags);

/* ---- row-block sharding over RCCL ------------------------------ */
/* Two process models run the same schedule: the single-process device group above
 * (gpx_config.ndev) and one process per GPU (below).
 * One process per GPU.  Rank 0 calls gpx_comm_unique_id and ships the 128 bytes to
 * the other ranks by any means (the Python host uses torch.distributed); every rank
 * then calls gpx_comm_init on its handle (created with the same world, own rank).  A
 * handle that owns a communicator — even a 1-rank one — runs the sharded schedule. */
int gpx_comm_unique_id(void* id128);
int gpx_comm_init(gpx_handle* h, const void* id128);

/* Portable transport for the same shard schedule: collectives on HOST buffers supplied
 * by the caller (e.g. torch.distributed/gloo); the library stages device<->host around
 * each call.  Used by the multi-process tests that share one GPU (RCCL refuses duplicate
 * devices) and for fabrics without RCCL.  op: 0 = sum, 1 = min.  Return 0 on success. */
typedef struct gpx_host_comm {
  void* ctx;
  int (*bcast)(void* ctx, void* buf, int64_t bytes, int32_t root);
  int (*allgather)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank);
  int (*reduce)(void* ctx, const double* send, double* recv, int64_t count, int32_t root, int32_t op);
  int (*allreduce)(void* ctx, double* buf, int64_t count, int32_t op);
} gpx_host_comm;
int gpx_comm_init_host(gpx_handle* h, const gpx_host_comm* vt);

/* ---- batched path distance (reference: trajectories.calc_distance,
 * GPmap.py:114-121, the inner loop of kmeansclustering GPmap.py:72-80) ---------------- */
/* D (P,C)[p][c] = sum_{i<L} || paths[p][i] - cents[c][i] ||_2 with paths (P,L,2) and cents
 * (C,L,2) arrays of (x,y), fp64, L <= 64.  mem_kind as in gpx_fit (device pointers must
 * live on the current HIP device). */
int gpx_path_distance(const double* paths, int64_t P, const double* cents, int64_t C, int32_t L,
                      double* D, int32_t mem_kind);

/* ---- kernel unit-test entry points (host buffers, fp64) ------------------------- */
/* K (na,nb) = sf2 k(A,B) (+ diag_add on the diagonal when B == NULL, i.e. B = A). */
int gpx_kernel_matrix(int32_t kernel, const double* A, int64_t na, const double* B, int64_t nb,
                      int32_t d, const double* lengthscale, int32_t n_ls, double sf2,
                      double diag_add, double* K /* (na, nb or na) */);
/* in-place lower Cholesky of A (n,n), lda = n; n multiple of 64.  The strictly upper
 * triangle is never read and is scratch on return (diagonal tiles are updated whole).
 * block = panel width (0 = default). */
int gpx_potrf(double* A, int64_t n, int32_t block, int64_t* info);
/* X (m,nb) <- X L^-T, L (nb,nb) lower; m, nb multiples of 64. */
int gpx_trsm(double* X, int64_t m, const double* L, int64_t nb);
/* C (m,n) -= A (m,k) B(n,k)^T.  lower != 0: only tiles on/below the diagonal
 * (m == n required).  m,n multiples of 128, k multiple of 16. */
int gpx_gemm_nt(double* C, int64_t m, int64_t n, const double* A, const double* B, int64_t k,
                int32_t lower);
/* fp64 MFMA layout probe: D (16,16) = A (16,4) B (4,16) through one
 * v_mfma_f64_16x16x4_f64. */
int gpx_mfma_probe(const double* A, const double* B, double* D);
/* the same through one v_mfma_f32_16x16x4_f32 (different accumulator row map). */
int gpx_mfma_probe_f32(const float* A, const float* B, float* D);
/* microbenchmarks quoted beside the rooflines: sustained fp64 MFMA
 * TFLOP/s of a register-resident loop and HBM GB/s of a streaming copy. */
int gpx_microbench(double* mfma_tflops, double* copy_gbs);
/* host-only replay of the tile maps of the trailing-update kernels (no GPU needed): kind 0 =
 * lower triangle of a tm x tm grid of 128x128 tiles (unsharded SYRK), kind 1 = the
 * block-cyclic staircase of a rank's tm x tn grid (P ranks, tpb tiles per row block, offset
 * c: local tile row ti owns tj <= ((ti/tpb)*P + c)*tpb + ti%tpb), kind 2 = the fused trailing
 * update (strip of tn tile columns first, then the triangle beyond it; tm x tm lower).  Writes (ti, tj) int32
 * pairs in launch order into out[2*cap]; returns their count through *count.  Test hook:
 * every owned tile must appear exactly once. */
int gpx_debug_tile_map(int32_t kind, int64_t tm, int64_t tn, int32_t P, int32_t tpb, int32_t c,
                       int32_t* out, int64_t cap, int64_t* count);
/* the staircase of a rank under either dealing of the row blocks (round 4; cyclic: block g on rank g mod P; snake: rounds
 * of 2 P blocks dealt 0 .. P-1, P-1 .. 0 — balanced row work): local tile row ti belongs to block number lbf + ti / tpb of
 * rank r, tile column 0 to block gc0 of the matrix.  Same output as gpx_debug_tile_map kind 1. */
int gpx_debug_stair_map(int64_t tm, int64_t tn, int32_t P, int32_t tpb, int32_t r, int32_t lbf, int32_t gc0,
                        int32_t snake, int32_t* out, int64_t cap, int64_t* count);
/* the dealing itself: owner[g], local[g] (index of block g among its owner's blocks) for g < nblk, and
 * upto[g * P + r] = number of blocks of rank r with index <= g.  Test hook (tests/test_tile_maps.py). */
int gpx_debug_deal(int32_t P, int32_t snake, int64_t nblk, int32_t* owner, int64_t* local, int64_t* upto);
/* host-only exercise of the rendezvous the LOCAL transport's rank threads use (no GPU needed):
 * P threads run `rounds` barrier rounds, each checking that every rank published the round
 * number; if abort_rank >= 0 that rank leaves at round abort_round and aborts the hub instead
 * of arriving.  *completed = rounds every surviving rank finished; returns 0 when all threads
 * came back (no deadlock) and saw consistent data, GPX_E_COMM otherwise. */
int gpx_debug_local_hub(int32_t P, int32_t rounds, int32_t abort_rank, int32_t abort_round,
                        int32_t* completed);

/* Test hook (process-wide): seed != 0 puts a short bounded spin kernel in front of a random third
 * of the library's launches, on the stream of that launch, so that its streams race each other
 * differently per seed; 0 turns it off.  Results must not change by a bit — every cross-stream
 * dependency is an event (tests/test_delay_gpu.py). */
int gpx_debug_set_delay(uint64_t seed);
/* Diagnostics: the dense tile engine alone, operands resident in HBM (zero-filled): C (n,n) -= / =
 * A (n,k) B(n,k)^T, lower != 0: the triangular (SYRK-shaped) launch of the trailing update; mode 0:
 * C -= (atomic epilogue), 1: C = (plain stores).  dtype GPX_F64 / GPX_F32; n multiple of 128,
 * k of 32.  One warm-up launch, then the mean of `iters` back-to-back launches in ms. */
int gpx_debug_gemm_bench(int32_t dtype, int64_t n, int64_t k, int32_t lower, int32_t mode, int32_t iters,
                         double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H_ */
