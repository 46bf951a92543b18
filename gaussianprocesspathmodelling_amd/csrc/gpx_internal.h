// gpx_internal.h — launcher prototypes shared by the kernel TUs and the C-ABI TU.
// All launchers enqueue on `st` and never synchronise.  Every launcher is a template on
// the element type T, explicitly instantiated for double and float in its .hip file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpx {

constexpr int KB = 64;      // innermost Cholesky block (one-workgroup POTF2 + inverse)
constexpr int TILE = 128;   // trailing-update tile; every padded dimension is a multiple
// Leading-dimension skew against power-of-two strides: ONE 128-byte line, so that every row of
// every matrix starts on a 128-byte boundary (hipMalloc bases are 256-byte aligned; all padded
// dimensions are multiples of 64 elements).  trsm_rlt_kernel DEPENDS on this: it round-trips a
// 64x64 tile through global memory inside one workgroup — L2-side atomics, then LDS-DMA loads
// through the CU's L1 — which is only coherent because no 128-byte line of the tile can already
// sit in that L1: lines never straddle a tile edge (this alignment) and the earlier loads of the
// walk touch only columns left of the tile.  Keep skew * sizeof(T) a multiple of 128.
template <typename T>
constexpr int ld_skew() {
  return 128 / (int)sizeof(T);
}
constexpr int LD_SKEW = ld_skew<double>();  // fp64 buffers (16 elements)
static_assert(ld_skew<double>() * sizeof(double) % 128 == 0 && ld_skew<float>() * sizeof(float) % 128 == 0,
              "row starts must stay 128-byte aligned (trsm_rlt_kernel's L1 invariant)");

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Dealing of the row blocks of the Gram matrix over the P ranks of a shard (csrc/gpx_shard.inc).
//   cyclic (rounds 1-3): block g on rank g mod P — rank P-1 holds the lowest block of every round of P, and a row block's
//     share of every trailing update grows with its index: the last rank carries 8.4 % more than the mean at P = 8 /
//     nb = 512 (N = 65536), 17 % at nb = 1024, in EVERY panel (the ranks meet at each panel, so that is the fit's time).
//   snake (round 4, default): rounds of 2 P blocks dealt 0, 1, ..., P-1, P-1, ..., 1, 0 — every rank gets one block of the
//     ascending and one of the descending half, the sums agree: 0.65 % / 2.6 % in the same cases (tools/scaling_model.py).
// Everything that depends on the dealing asks this struct (host and device).
struct Deal {
  int P, snake;
  __host__ __device__ __forceinline__ int owner(int64_t g) const {
    if (!snake) return (int)(g % P);
    const int pos = (int)(g % (2 * P));
    return pos < P ? pos : 2 * P - 1 - pos;
  }
  __host__ __device__ __forceinline__ int64_t local(int64_t g) const {  // index of block g among its owner's blocks
    if (!snake) return g / P;
    return 2 * (g / (2 * P)) + ((g % (2 * P)) >= P ? 1 : 0);
  }
  __host__ __device__ __forceinline__ int64_t global(int r, int64_t lb) const {  // block number lb of rank r
    if (!snake) return lb * P + r;
    return (lb >> 1) * (2 * P) + ((lb & 1) ? 2 * P - 1 - r : r);
  }
  __host__ __device__ __forceinline__ int64_t upto(int64_t p, int r) const {  // blocks of rank r with index <= p (p >= -1)
    if (!snake) return p >= r ? (p - r) / P + 1 : 0;
    const int64_t n = p + 1, c = n / (2 * P), rem = n % (2 * P);
    return 2 * c + (rem > r ? 1 : 0) + (rem > 2 * P - 1 - r ? 1 : 0);
  }
};

struct BcMask {   // row map of the sharded trailing update (P == 0: unused)
  int P, tpb;     // ranks, tiles per block
  int r = 0, lbf = 0, gc0 = 0, snake = 0;  // local tile row ti belongs to row block Deal.global(r, lbf + ti / tpb) of the matrix,
                                           // tile column 0 to block gc0: row tile (that block - gc0) * tpb + ti % tpb of the launch
  // B operand read straight from the ALL-GATHERED panel (round 4; piece == 0: B is contiguous): rank-major pieces of
  // `piece` rows, every rank's own blocks beyond panel p in order.  Tile column tj is block g = gc0 + tj / tpb of the
  // matrix: block number local(g) - upto(p, owner(g)) of the piece of rank owner(g).
  int p = 0;
  int64_t piece = 0;
  __host__ __device__ __forceinline__ Deal deal() const { return Deal{P, snake}; }
  __host__ __device__ __forceinline__ int row_tile(int ti) const {  // launch-relative tile row of local tile row ti
    return (int)(deal().global(r, lbf + ti / tpb) - gc0) * tpb + ti % tpb;
  }
};

// first row of B's tile column tj (BT rows per tile; tpb counts BT-tiles per block)
template <int BT>
__host__ __device__ __forceinline__ int64_t bc_brow(const BcMask& bc, int tj) {
  if (bc.piece == 0) return (int64_t)tj * BT;
  const Deal dl = bc.deal();
  const int64_t g = bc.gc0 + tj / bc.tpb;
  const int rr = dl.owner(g);
  return (int64_t)rr * bc.piece + ((dl.local(g) - dl.upto(bc.p, rr)) * bc.tpb + tj % bc.tpb) * BT;
}


// ---- kernel-matrix build (gpx_kbuild.hip) ---------------------------------------
// Xs = X / lengthscale (per-dimension), rows >= n zero-filled up to npad.
template <typename T>
void launch_scale_points(const T* X, int64_t n, int64_t npad, int d, const double* ls, int n_ls, T* Xs,
                         hipStream_t st);
// Lower tiles (64x64) of K[npad, npad] (ld): sf2*k(xi,xj), + diag_add on the diagonal;
// padded rows/cols (>= n) become identity rows.
template <typename T>
void launch_kbuild_sym(int kernel, const T* Xs, int64_t n, int64_t npad, int d, double sf2,
                       double diag_add, T* K, int64_t ld, hipStream_t st);
// Rectangular K*[mpad, npad] (ld) = sf2*k(As_i, Bs_j); rows >= m or cols >= n are zero.
template <typename T>
void launch_kbuild_cross(int kernel, const T* As, int64_t m, int64_t mpad, const T* Bs, int64_t n,
                         int64_t npad, int d, double sf2, T* K, int64_t ld, hipStream_t st);

// ---- dense blocks (gpx_blas.hip) ---------------------------------------------------
// In-place Cholesky of one 64x64 diagonal block (lower) + its inverse Winv (64x64,
// row-major, upper part zero).  gidx0 = global index of the block's first row (info).
template <typename T>
void launch_potf2_64(T* A, int64_t lda, T* Winv, int64_t gidx0, int* info, hipStream_t st, unsigned* flag = nullptr,
                     unsigned flag_val = 0);
// The same for a 128x128 diagonal tile in one launch: L11, L21, L22 in place and the TWO 64x64 inverses
// of its diagonal blocks (Winv[0..4096), Winv[4096..8192)); the 64 x 64 block above the diagonal is
// not touched.
// flag != null (both kernels): *flag = flag_val after a device-scope release of the results — what a stream parked
// on launch_wait_counter(flag, flag_val) waits for.
template <typename T>
void launch_potf2_128(T* A, int64_t lda, T* Winv, int64_t gidx0, int* info, hipStream_t st, unsigned* flag = nullptr,
                      unsigned flag_val = 0);
// X (rows x nb, ldx) <- X * L^-T; L (nb x nb, ldl) lower, Winv = nb/64 inverse diag
// blocks.  rows, nb multiples of 64.  If P != null the result is also written to the
// compact panel P (rows x nb, ldp).
template <typename T>
void launch_trsm_rlt(T* X, int64_t ldx, int64_t rows, const T* L, int64_t ldl, const T* Winv, int nb,
                     T* P, int64_t ldp, hipStream_t st);
// Building the explicit inverse of a diagonal block L (nbw x nbw, ldl) next to its
// factorisation: U (ldu) holds (L^-1)^T column blocks [0, q0) already and the identity elsewhere;
// this solves column blocks [q0, q1) of U for row blocks 0..q1-1 and writes them transposed into
// row blocks [q0, q1) of W = L^-1 (ldw).  Needs the 64-block inverses and rows of L up to q1.
template <typename T>
void launch_inv_extend(T* U, int64_t ldu, const T* L, int64_t ldl, const T* Winv, int q0, int q1, T* W,
                       int64_t ldw, hipStream_t st, int phase = 0);  // 0 whole, 1 "pre" (left of block q0 only), 2 "post"
// Latency mode of the calling host thread (default off): the 64-tile launches (launch_gemm_nt with an
// under-filled 128-grid, launch_trsm_rlt / launch_inv_extend with few slabs) then stage several k-steps per
// barrier — 64 or 128 KB of LDS per workgroup instead of 32 — which shortens every latency-bound K walk
// when the GPU is otherwise idle and must NOT be used beside a large trailing update (see gpx_blas.hip).
void set_latency_mode(int on);
// Reserve mode of the calling host thread (round 4; gemm_nt_resv_kernel in gpx_blas.hip).
// reserve_ring(ring, cap): start of a factorisation — `ring` = [cap][8] device counters the caller has zeroed (null: off).
// set_reserve_mode(k): k > 0 — the products launch_gemm_nt enqueues from now on go out as persistent grids whose
//   workgroups leave k CUs per XCD alone (they exit at once there) and take their tiles from the next 8 counters of
//   the ring; launches beyond cap, and k = 0, are the plain kernels.
// set_reserve_chain(1): the chain's narrow slab launches (launch_trsm_rlt, <= 16 slabs) ask for a whole CU's LDS, so
//   that they land on the CUs the update leaves alone.
void reserve_ring(unsigned* ring, int cap);
void set_reserve_mode(int k);
void set_reserve_chain(int on);
// 1 if a launcher CALLED BY THIS HOST THREAD since the last call refused misaligned operands (and launched nothing);
// clears the flag.  Thread-local: an API call enqueues and checks on one thread, so a handle only ever sees its own
// launches' errors (distinct handles on distinct threads: include/gpx.h)
int take_launch_error();
// X (rows x nb, ldx) <- X * L^-1 (right, lower, no-transpose; descending blocks).
template <typename T>
void launch_trsm_rln(T* X, int64_t ldx, int64_t rows, const T* L, int64_t ldl, const T* Winv, int nb,
                     hipStream_t st);
// C (m x n, ldc) op= A (m x k, lda) * B (n x k, ldb)^T.
//   mode 0: C -= A B^T      mode 1: C = A B^T
//   lower 0: every tile;  1: lower triangle (m == n), triangular super-tile order;
//         2: rectangle masked to tile_col <= tile_row (look-ahead strip);
//         4: every tile, B lower triangular (k limited to the tile's own column range).
//   tile = 128 (m,n multiples of 128) or 64 (multiples of 64); k multiple of 64.
template <typename T>
void launch_gemm_nt(int tile, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
                    int64_t m, int64_t n, int64_t k, int lower, int mode, hipStream_t st);
// the tile size launch_gemm_nt really uses for (tile, m, n, lower): 64 when the 128-grid is under-filled
int gemm_nt_tile(int tile, int64_t m, int64_t n, int lower);
// the same with the tile size taken literally (unit-test entry point, engine benchmark)
template <typename T>
void launch_gemm_nt_fixed(int tile, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
                    int64_t m, int64_t n, int64_t k, int lower, int mode, hipStream_t st);
// The whole trailing update of a panel in one launch: C (n x n, 128-tiles, lower) -= P P^T (P: n x k, ldp)
// with the first ns columns (the strip that becomes the next panel) enumerated first; every strip slot
// adds 1 to *ctr once its tile is released at device scope.  Returns the number of strip slots.
template <typename T>
unsigned launch_gemm_nt_fused(T* C, int64_t ldc, const T* P, int64_t ldp, int64_t n, int64_t ns, int64_t k,
                              unsigned* ctr, hipStream_t st);
// One wave that returns once *ctr >= target (polled at device scope, ~2 us period).  Bounded: after
// ~15 s it gives up, sets *info = INT_MIN (the caller reports it) and returns — never a hung queue.
void launch_wait_counter(const unsigned* ctr, unsigned target, int* info, hipStream_t st);
// the two halves of the hand-over self-test: *seen = 1 iff the waiting kernel saw *flag != 0 within max_polls x ~2 us
void launch_flag_probe_wait(const unsigned* flag, unsigned* seen, unsigned max_polls, hipStream_t st);
void launch_flag_probe_set(unsigned* flag, hipStream_t st);
// Sharded trailing update: C (m local rows x n, 128-tiles) -= A B^T restricted to tiles with
//   tile_col <= bc.row_tile(tile_row)   (the row blocks' places in the dealing: Deal / BcMask above);
//   bc.piece > 0: B is the all-gathered panel bc.p in rank-major pieces (bc_brow).
// test hook: random spin kernels in front of launches (gpx_debug_set_delay; gpx_misc.hip)
void debug_set_delay(uint64_t seed);
void debug_delay(hipStream_t st);
// C (m x n) = A (m x k) B(n x k)^T for skinny C with a long contraction (m, n multiples of 64): S
// splits of k into partial tiles part[s] (m x ldp each, back to back), summed in split order
int splitk_splits(int64_t k);  // depends on k only: batching the columns must not change the bits
template <typename T>
void launch_gemm_nt_splitk(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                           int64_t n, int64_t k, int S, T* part, int64_t ldp, hipStream_t st);
template <typename T>
void launch_gemm_nt_bc(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                       int64_t n, int64_t k, const BcMask& bc, hipStream_t st);
// C (m x n, ldc) -= A (m x k, lda) * B (k x n, ldb); 64x64 tiles.
template <typename T>
void launch_gemm_nn(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                    int64_t n, int64_t k, hipStream_t st);

// ---- small helpers (gpx_misc.hip) ------------------------------------------------------
// YT (R x npad, ld) <- transpose of y (n x k) into rows [0,k), everything else zero.
template <typename T>
void launch_pack_rhs(const T* y, int64_t n, int k, T* YT, int64_t ld, int64_t npad, int R,
                     hipStream_t st);
// out (n x k) <- rows [0,k) of YT (R x *, ld), scaled by `scale`.
template <typename T>
void launch_unpack_rhs(const T* YT, int64_t ld, int64_t n, int k, double scale, T* out, hipStream_t st);
// var[i] = sf2 - sum_j VT[i, j]^2, i < m  (accumulated in fp64).
template <typename T>
void launch_var_rows(const T* VT, int64_t ld, int64_t m, int64_t ncols, double sf2, T* var,
                     hipStream_t st);
// dst (rows x cols, ldd) = src (rows x cols, lds), streaming; cols * sizeof(T) multiple of 16, 16-byte aligned rows
template <typename T>
void launch_copy2d(T* dst, int64_t ldd, const T* src, int64_t lds, int64_t rows, int64_t cols, hipStream_t st);
// A[i][i] = 1 for i < n
template <typename T>
void launch_set_diag_one_t(T* A, int64_t lda, int64_t n, hipStream_t st);
// out[0] = 2 * sum_{i<n} log(A[i, i])  (fp64 accumulation and result).
template <typename T>
void launch_logdet(const T* A, int64_t lda, int64_t n, double* out, hipStream_t st);
void launch_mfma_probe(const double* A, const double* B, double* D, hipStream_t st);
void launch_mfma_probe_f32(const float* A, const float* B, float* D, hipStream_t st);
void launch_mfma_loop(double* sink, int iters, int blocks, hipStream_t st);
int64_t debug_tile_map(int kind, int64_t tm, int64_t tn, const BcMask& bc, int32_t* out,
                       int64_t cap);
void launch_copy(const double* src, double* dst, int64_t count, hipStream_t st);

// ---- log-marginal-likelihood gradient (gpx_grad.hip, fp64) -------------------------------------
// theta = (lengthscale[0..n_ls), sf2, sn2), ntheta = n_ls + 2, ard = n_ls > 1.  Both passes write
// one partial per (tile slot, theta); reduce_partials sums them in a fixed order.
int64_t kinv_trace_slots(int64_t npad);  // slots of the 128-tile triangular map
int64_t alpha_quad_slots(int64_t npad);  // 64-tiles of the lower triangle
void launch_set_diag_one(double* A, int64_t lda, int64_t n, hipStream_t st);
// part[slot][t] = sum over the tile of (ZT ZT^T)_ij (dK/dlog theta_t)_ij, ZT = L^-T (npad x npad, ld);
// rank `rank` of P covers every P-th octet of 64-slot groups (P = 1: all of them)
void launch_kinv_trace(int kernel, const double* ZT, int64_t ld, int64_t npad, int64_t n, const double* Xs, int d,
                       int ard, double sf2, double sn2, double* part, int ntheta, int P, int rank, hipStream_t st);
// the same for a factor that is only held distributed: ZTc (npad rows x ncols, ldc) = the columns of L^-T
// that belong to this rank's row blocks of L (height nb, block-cyclic over P ranks, local order); every
// rank visits every tile and contracts over its own columns: the ranks' partial sums add up
void launch_kinv_trace_cols(int kernel, const double* ZTc, int64_t ldc, int64_t npad, int64_t n, const double* Xs,
                            int d, int ard, double sf2, double sn2, double* part, int ntheta, int nb, int P, int rank,
                            int64_t ncols, int snake, hipStream_t st);
// part[slot][t] = sum over the tile of (sum_c alpha_ic alpha_jc) (dK/dlog theta_t)_ij, alphaT (k x npad, ld)
void launch_alpha_quad(int kernel, const double* alphaT, int64_t ld, int k, int64_t npad, int64_t n,
                       const double* Xs, int d, int ard, double sf2, double sn2, double* part, int ntheta,
                       hipStream_t st);
void launch_reduce_partials(const double* part, int64_t ntile, int ntheta, double scale, double* out,
                            hipStream_t st);
// out[0] = sum_i sum_c y[i*k + c] * alphaT[c*ld + i]
void launch_dot_rhs(const double* y, const double* alphaT, int64_t ld, int64_t n, int k, double* out,
                    hipStream_t st);

// ---- mixed precision (gpx_mixed.hip) ------------------------------------------------------------
// outT[c][i] = (y ? y[i*k+c] : 0) + sign * (sum_j sf2 k(a_i, b_j) alphaT[c][j] + diag * alphaT[c][i]), i < m;
// fp64, matrix-free (K regenerated from the scaled points As (mpad x d) / Bs (npad x d)); k <= 8;
// alphaT (k x lda) zero beyond the valid columns.
void launch_kmatvec(int kernel, const double* As, int64_t m, int64_t mpad, const double* Bs, int64_t npad, int d,
                    double sf2, double diag, const double* y, const double* alphaT, int64_t lda, int k,
                    double sign, double* outT, int64_t ldo, hipStream_t st);
void launch_f64_to_f32(const double* in, float* out, int64_t count, hipStream_t st);
void launch_f32_to_f64(const float* in, double* out, int64_t count, hipStream_t st);
// dst (R x ldd, float) = rows [0, rows) x [0, n) of src (double), zero elsewhere up to npad
void launch_rows_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int rows, int R, int64_t n,
                            int64_t npad, hipStream_t st);
// dst (rows x ldd, double) (+)= src (float) on [0, n), (acc ? unchanged : 0) beyond
void launch_rows_add_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int rows, int64_t n,
                                int64_t npad, int acc, hipStream_t st);
void launch_rows_sumsq(const double* src, int64_t lds, int rows, int64_t n, double* out, hipStream_t st);
// few right-hand sides (k <= 8): out[c][i] (= | -=) sum_j M[i][j] v[c][j]  (cols: M[j][i]); nj <= 1024, cols: ni % 128 == 0
template <typename T>
void launch_few_product(bool cols, bool assign, T* out, int64_t ldo, const T* M, int64_t ldm, int64_t ni, int nj,
                        const T* v, int64_t ldv, int k, hipStream_t st);

// ---- path distance (gpx_paths.hip) -----------------------------------------------------------
// D (P, ldd)[p][c] = sum_i ||paths[p][i] - cents[c][i]||, paths (P, L, 2), cents (C <= 64, L <= 64, 2)
void launch_path_distance(const double* paths, int64_t P, const double* cents, int C, int L, double* D,
                          int64_t ldd, hipStream_t st);

// ---- row-block-cyclic shard helpers (gpx_misc.hip; T = double | float) ------------------------
// A[i][i] = i < nvalid ? A[i][i] + add : 1   for i < n (diagonal of one local row block)
template <typename T>
void launch_fix_diag(T* A, int64_t lda, int n, int nvalid, double add, hipStream_t st);
// Gathered panel G [P][maxcnt][ldp] (rank-major, each rank's trailing rows in local order)
// -> dst [(nblk-p-1)*nb][ldd] in global row order (the replicated factor's column block p: the update kernels read
// the gathered panel in place, bc_brow in gpx_tile.h).
template <typename T>
void launch_unpermute_panel(const T* G, int64_t ldp, T* dst, int64_t ldd, int nb, Deal dl, int p, int nblk, int64_t maxcnt,
                            hipStream_t st);
// YTloc[r][lb*nb + i] = y[(g*nb+i)*k + r] for the blocks g = dl.global(rank, lb) owned by `rank`.
template <typename T>
void launch_pack_rhs_local(const T* y, int64_t n, int k, T* YTloc, int64_t ldy, int nb, int nlb, Deal dl, int rank, int R,
                           hipStream_t st);
// Full[r][g*nb + i] = Loc[r][lb*nb + i] (own blocks), rest untouched.
template <typename T>
void launch_scatter_local(const T* Loc, int64_t ldl, T* Full, int64_t ldf, int nb, int nlb, Deal dl, int rank, int R,
                          hipStream_t st);
// dst (rows x cols, ldd) += sign * src (rows x cols, lds)
template <typename T>
void launch_add_block(T* dst, int64_t ldd, const T* src, int64_t lds, int rows, int cols, double sign, hipStream_t st);
template <typename T>
void launch_add_scalar(T* p, int64_t count, double v, hipStream_t st);
// dst[i] = op over q < P of src[q*count + i] in rank order (op 0 sum, 1 min); dst may not alias src
template <typename T>
void launch_reduce_ranks(const T* src, T* dst, int P, int64_t count, int op, hipStream_t st);
// out[0] += 2 * sum_i log(A[i][i]), i < n (one diagonal block)
template <typename T>
void launch_logdet_acc(const T* A, int64_t lda, int n, double* out, hipStream_t st);

}  // namespace gpx
