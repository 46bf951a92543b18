// gpx_blas.hip — dense building blocks of the blocked Cholesky / triangular solves
// (SURVEY.md §8 rows a3 "K2", a4 "K3", a5 "K3'"), templated on the element type:
// double (configs 1-4) and float (config 5, the mixed-precision study).  Absent in the
// reference (its only linalg call is np.linalg.norm, GPmap.py:120); algorithm =
// right-looking blocked Cholesky (R&W Alg. 2.1 line 2) as restated in oracle/gp_oracle.py.
//
// Everything is expressed with MFMA tile engines on 16x16 tiles of
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, 4 waves (2x2) per 256-thread
// workgroup, each wave owning (BM/2)x(BN/2):
//   gemm_tile_g<T,BM,BN>   NT products, LDS-DMA staging, swizzled LDS image (hot path)
//   gemm_tile_nn<T,BM,BN>  B read as [k][n] ("NN", back substitution), register staging
//
// MFMA layouts (cdna_hip_programming.md §3), verified on hardware by gpx_mfma_probe:
//   A lane l = A[l&15][l>>4],  B lane l = B[l>>4][l&15]   (both types)
//   D reg r of lane l = D[(l>>4) + 4r][l&15]  for f64,   D[4(l>>4) + r][l&15]  for f32.
#include <atomic>
#include <cstdlib>

#include "gpx_internal.h"
#include "gpx_tile.h"
#include <algorithm>

namespace gpx {
namespace {

// ---- NN tile engine (register staging) -------------------------------------------------------
// acc += A(BM x K) * B(K x BN), A row-major [row][k], B row-major [k][n].  Global ->
// registers (16-byte chunks, issued before the MFMAs of the current k-step) -> LDS
// (written after them), two LDS buffers, one barrier per step.  Only the back
// substitution uses it (gemm_nn_kernel, trsm_rln_kernel): not on the hot path.
template <typename T, int BM, int BN>
struct TileShapeNN {
  static constexpr int BK = Num<T>::BK;
  static constexpr int LDS_K = BK + 16 / (int)sizeof(T);  // A rows padded by 16 B
  static constexpr int LDS_BN = BN + 16;                  // B rows padded by 16 elements
  static constexpr int A_STAGE = BM * LDS_K;
  static constexpr int B_STAGE = BK * LDS_BN;
  static constexpr int SMEM_ELEMS = 2 * (A_STAGE + B_STAGE);
};

template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_tile_nn(const T* A, int64_t lda, const T* B, int64_t ldb, int K,
                                             typename Num<T>::v4 (&acc)[BM / 32][BN / 32], T* smem) {
  using S = TileShapeNN<T, BM, BN>;
  using slot_t = typename Num<T>::slot;
  constexpr int BK = S::BK, SL = Num<T>::SLOT;
  constexpr int MT = BM / 32, NT = BN / 32, WM = BM / 2, WN = BN / 2;
  constexpr int ACH = BK / SL;        // 16-B chunks per A row per k-step (= 8)
  constexpr int BCH = BN / SL;        // chunks per B row
  constexpr int CA = BM * ACH / 256;  // chunks per thread
  constexpr int CB = BK * BCH / 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  T* As = smem;
  T* Bs = smem + 2 * S::A_STAGE;

  slot_t ra[CA], rb[CB];
  const T* ga[CA];
  const T* gb[CB];
  int sa[CA], sb[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * 256;
    const int row = c / ACH, cc = c % ACH;
    ga[i] = A + (int64_t)row * lda + cc * SL;
    sa[i] = row * S::LDS_K + cc * SL;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int c = tid + i * 256;
    const int kr = c / BCH, cc = c % BCH;
    gb[i] = B + (int64_t)kr * ldb + cc * SL;
    sb[i] = kr * S::LDS_BN + cc * SL;
  }
  const int a_off = (wr * WM + (lane & 15)) * S::LDS_K + (lane >> 4);
  const int b_off = (lane >> 4) * S::LDS_BN + wc * WN + (lane & 15);

#pragma unroll
  for (int i = 0; i < CA; ++i) ra[i] = *reinterpret_cast<const slot_t*>(ga[i]);
#pragma unroll
  for (int i = 0; i < CB; ++i) rb[i] = *reinterpret_cast<const slot_t*>(gb[i]);
#pragma unroll
  for (int i = 0; i < CA; ++i) *reinterpret_cast<slot_t*>(As + sa[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < CB; ++i) *reinterpret_cast<slot_t*>(Bs + sb[i]) = rb[i];
  __syncthreads();

  const int KT = K / BK;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    // unconditional prefetch (the last step re-reads its own lines): keeps the staging
    // registers out of scratch
    const int kn = (kt + 1 < KT) ? kt + 1 : kt;
#pragma unroll
    for (int i = 0; i < CA; ++i) ra[i] = *reinterpret_cast<const slot_t*>(ga[i] + (int64_t)kn * BK);
#pragma unroll
    for (int i = 0; i < CB; ++i)
      rb[i] = *reinterpret_cast<const slot_t*>(gb[i] + (int64_t)kn * BK * ldb);
    const T* Ab = As + buf * S::A_STAGE + a_off;
    const T* Bb = Bs + buf * S::B_STAGE + b_off;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      T a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = Ab[m * 16 * S::LDS_K + ks * 4];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = Bb[ks * 4 * S::LDS_BN + n * 16];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = Num<T>::mfma(a[m], b[n], acc[m][n]);
    }
    T* Aw = As + (buf ^ 1) * S::A_STAGE;
    T* Bw = Bs + (buf ^ 1) * S::B_STAGE;
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<slot_t*>(Aw + sa[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<slot_t*>(Bw + sb[i]) = rb[i];
    __syncthreads();
  }
}

// Diagnostic build only (-DGPX_STAMPS, tools/syrk_clock.py): one workgroup per launch stamps the
// shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its
// tile — their ratio is the clock the chip really holds inside the MFMA loop (sysfs can read up to
// 10 % high; MI355X_MICROARCH.md "DVFS give-back" item 6).  Never in the shipped library.
#ifdef GPX_STAMPS
__device__ long long gpx_syrk_clock_buf[8];
#endif

// ---- C op= A * B^T --------------------------------------------------------------
// KSUB k-steps per barrier (round 3).  The 64-tile instantiations — what the library launches when a
// 128-tile grid would not fill the chip: strips, trailing updates and block products of small problems
// (BASELINE.json configs[1]), every step of the diagonal chain — are LATENCY-bound: a k-step of a 64 x 64
// tile is 1024 cycles of MFMA per wave against ~4000 cycles until its LDS-DMA lands, so a K = 1024 walk
// took 64 round trips (110 us).  With KS64 lines per barrier (64 elements of K: 4 lines fp64, 2 fp32) the
// loads of a super-step are in flight together: 16 round trips.
template <typename T, int BT, bool TRI, int MODE, int KSUB = 1>
__global__ __launch_bounds__(256, (BT == 64 && KSUB >= 4) ? 1 : 2) void gemm_nt_kernel(
    T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
    int64_t ldb, int tiles_m, int tiles_n, int sh, int mask_lower, BcMask bc, int K) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT, KSUB>::SMEM_ELEMS];
#ifdef GPX_STAMPS
  const long long cE = __builtin_amdgcn_s_memtime();
#endif
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  int ti, tj;
  if (!tile_coords<TRI>(lin, tiles_m, tiles_n, sh, mask_lower, bc, ti, tj)) return;
  typename Num<T>::v4 acc[BT / 32][BT / 32];
  zero_acc(acc);
  T* Ct = C + (int64_t)ti * BT * ldc + (int64_t)tj * BT;
#ifdef GPX_STAMPS
  const bool big = TRI && BT == 128 && K >= 512 && tiles_m >= 300 && threadIdx.x == 0;
  const bool stamp = big && blockIdx.x == gridDim.x / 2;
  const long long cA = big ? __builtin_amdgcn_s_memtime() : 0;
  long long c0 = 0, r0 = 0;
  if (stamp) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  gemm_tile_g<T, BT, BT, 2, KSUB>(A + (int64_t)ti * BT * lda, lda, B + bc_brow<BT>(bc, tj) * ldb, ldb, K, acc,
                                  smem);
#ifdef GPX_STAMPS
  long long cL = 0;
  if (big) {  // every workgroup of the big launches: sums of prologue / loop cycles and count
    const long long now = __builtin_amdgcn_s_memtime();
    atomicAdd(reinterpret_cast<unsigned long long*>(&gpx_syrk_clock_buf[5]), (unsigned long long)(now - cA));
    atomicAdd(reinterpret_cast<unsigned long long*>(&gpx_syrk_clock_buf[6]), 1ull);
    atomicAdd(reinterpret_cast<unsigned long long*>(&gpx_syrk_clock_buf[7]), (unsigned long long)(cA - cE));
    cL = now;
  }
  if (stamp) {
    gpx_syrk_clock_buf[0] = cL - c0;
    gpx_syrk_clock_buf[1] = __builtin_amdgcn_s_memrealtime() - r0;
    gpx_syrk_clock_buf[2] = K;
    gpx_syrk_clock_buf[3] = c0 - cE;  // kernel entry -> first DMA issue
  }
#endif
  // (a software-pipelined epilogue — strips of C prefetched / kept in flight — measured
  //  0.5-1 % slower than this plain strip-by-strip one: it pushes the kernel to 256 VGPRs)
  store_tile<T, BT, BT, MODE>(Ct, ldc, acc);
#ifdef GPX_STAMPS
  if (big)  // epilogue issue (not completion), summed over the workgroups
    atomicAdd(reinterpret_cast<unsigned long long*>(&gpx_syrk_clock_buf[4]),
              (unsigned long long)(__builtin_amdgcn_s_memtime() - cL));
#endif
}

// block id -> tile of the fused launch (also replayed on the host: gpx_debug_tile_map kind 2)
__host__ __device__ __forceinline__ bool fused_coords(int64_t bid, int64_t grid, int S, int tiles_m, int tiles_s, int sh,
                                                     int& ti, int& tj) {
  if (bid < S) return tile_coords<false>(xcd_chunk_id(bid, S), tiles_m, tiles_s, sh, 1, BcMask{0, 1, 0}, ti, tj);
  const int tr = tiles_m - tiles_s;
  const bool ok = tile_coords<true>(xcd_chunk_id(bid - S, grid - S), tr, tr, 8, 0, BcMask{0, 1, 0}, ti, tj);
  ti += tiles_s;
  tj += tiles_s;
  return ok;
}

// ---- the whole trailing update of a panel in ONE launch (round 3) ------------------------------
// C (n x n, lower) -= P P^T with the STRIP — the first ts tile columns, which become the next panel —
// enumerated FIRST: blocks [0, S) are the strip's rectangular super-tile grid (masked to tj <= ti),
// blocks [S, gridDim) the triangle beyond it, each part with its own XCD chunking (S is a multiple of
// 64, so a block's XCD is the same in both numberings).  The hardware dispatches blocks in index
// order, so the strip retires within the first rounds of slots; every strip slot then bumps *ctr
// (after a device-scope release of its tile), and the look-ahead stream — parked on
// wait_counter_kernel — starts the next diagonal block while the same launch carries on with the
// rest: no kernel boundary, no drain and no refill between "strip" and "rest" (they were two
// launches with an event between them: ~0.2 ms of every panel).
template <typename T, int BT>
__global__ __launch_bounds__(256, 2) void gemm_nt_fused_kernel(T* __restrict__ C, int64_t ldc,
                                                               const T* __restrict__ P, int64_t ldp, int tiles_m,
                                                               int tiles_s, int S, int sh, int K,
                                                               unsigned* __restrict__ ctr) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT>::SMEM_ELEMS];
  const bool strip = (int)blockIdx.x < S;
  int ti, tj;
  const bool valid = fused_coords(blockIdx.x, gridDim.x, S, tiles_m, tiles_s, sh, ti, tj);
  if (valid) {
    typename Num<T>::v4 acc[BT / 32][BT / 32];
    zero_acc(acc);
    gemm_tile_g<T, BT, BT>(P + (int64_t)ti * BT * ldp, ldp, P + (int64_t)tj * BT * ldp, ldp, K, acc, smem);
    store_tile<T, BT, BT, 0>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
  }
  if (strip) {  // release this tile (its atomics have reached memory-side coherence), then count the slot
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(ctr, 1u);
  }
}

// ---- C = A * W^T with W lower triangular (an explicit block inverse): panel / block solves -----
// Tile column tj only needs k < (tj + 1) BT.  One workgroup takes tile columns tj AND tn - 1 - tj of
// its tile row, so every workgroup walks the same total k ((tn + 1) BT): with one tile per
// workgroup the long columns set the time of every round of slots and the launch ran at 56 % of
// the engine's rate (39 TF at N = 65536).
template <typename T, int BT, int KSUB = 1>
__global__ __launch_bounds__(256, (BT == 64 && KSUB >= 4) ? 1 : 2) void gemm_nt_ltri_kernel(T* __restrict__ C, int64_t ldc,
                                                              const T* __restrict__ A, int64_t lda,
                                                              const T* __restrict__ W, int64_t ldw, int tiles_m,
                                                              int tiles_n, int sh, int K) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT, KSUB>::SMEM_ELEMS];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int half = (tiles_n + 1) >> 1;
  int ti, tp;
  if (!tile_coords<false>(lin, tiles_m, half, sh, 0, BcMask{0, 1, 0}, ti, tp)) return;
  const T* At = A + (int64_t)ti * BT * lda;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int tj = pass == 0 ? tiles_n - 1 - tp : tp;  // the long column first
    if (pass == 1 && tj == tiles_n - 1 - tp) break;    // odd tile count: the middle column once
    typename Num<T>::v4 acc[BT / 32][BT / 32];
    zero_acc(acc);
    gemm_tile_g<T, BT, BT, 2, KSUB>(At, lda, W + (int64_t)tj * BT * ldw, ldw, min(K, (tj + 1) * BT), acc, smem);
    store_tile<T, BT, BT, 1>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
  }
}

// ---- SELF-RESERVING trailing update (round 4) ---------------------------
// Where the diagonal chain sets the pace (N = 8192: every panel; the tail of N = 65536) its one-workgroup kernels
// share CUs with MFMA-saturated update workgroups and run 4-6x slower in every phase — the POTF2's fp64 vector
// FMAs share the SIMD's double-precision pipe with the co-resident MFMAs (profiles/r03_cu_reserve_experiments.txt).
// A CU-masked queue gives the chain CUs of its own but costs the update 13 % by itself (same record).  So the
// update leaves CUs alone BY ITSELF: the launch is a persistent grid — about as many workgroups as the chip has
// slots — and every workgroup first reads which CU it landed on (HW_REG_HW_ID): on a RESERVED CU (CU 0 of shader
// engines 0..k-1 of its XCD: k CUs per XCD, 8 k of 256) it exits at once; elsewhere it loops, taking block ids of
// the static launch's numbering from a device counter — its own XCD's ids first (b = x, x + 8, ...: the same
// contiguous chunk of the logical tile order the static launch would give that XCD, so the L2 sharing survives),
// then the other XCDs' leftovers.  The POTF2 (79 KB of LDS) then only fits on a reserved CU: the others hold two
// 64 KB update workgroups.  Arithmetic per tile is that of the static launch: bit-identical.
__device__ __forceinline__ bool on_reserved_cu(int resv) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  return ((hw >> 8) & 0xfu) == 0u && (int)((hw >> 13) & 0x7u) < resv;  // CU_ID [11:8], SE_ID [15:13]
}

// next block id for the calling workgroup (thread 0 only): -1 when every counter has run past `total`
__device__ __forceinline__ int resv_take(unsigned* ctr, unsigned total, unsigned x0) {
#pragma unroll 1
  for (unsigned s = 0; s < 8u; ++s) {
    const unsigned x = (x0 + s) & 7u;
    if (x >= total) continue;
    const unsigned t = atomicAdd(&ctr[x], 1u);
    const unsigned long long b = 8ull * t + x;
    if (b < total) return (int)b;
  }
  return -1;
}

template <typename T, int BT, bool TRI, int MODE, int KSUB = 1>
__global__ __launch_bounds__(256, (BT == 64 && KSUB >= 4) ? 1 : 2) void gemm_nt_resv_kernel(
    T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
    int64_t ldb, int tiles_m, int tiles_n, int sh, int mask_lower, int K, unsigned total, unsigned* __restrict__ ctr,
    int resv, unsigned sweep0) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT, KSUB>::SMEM_ELEMS];
  __shared__ int next_bid;
  // the LAST workgroup of the grid never leaves for a reservation: whatever the placement, somebody finishes the tiles
  if (blockIdx.x + 1 != gridDim.x && on_reserved_cu(resv)) return;
  const bool sweeper = blockIdx.x >= sweep0;  // loops until the counters run out; the others take ONE tile and give the slot back
  bool done_one = false;
#pragma unroll 1
  for (;;) {
    if (done_one && !sweeper) return;
    if (threadIdx.x == 0) next_bid = resv_take(ctr, total, blockIdx.x & 7u);
    __syncthreads();
    const int bid = next_bid;
    __syncthreads();  // everybody has read it before thread 0 writes the next one
    if (bid < 0) return;
    int ti, tj;
    if (!tile_coords<TRI>(xcd_chunk_id(bid, total), tiles_m, tiles_n, sh, mask_lower, BcMask{0, 1, 0}, ti, tj)) continue;
    typename Num<T>::v4 acc[BT / 32][BT / 32];
    zero_acc(acc);
    gemm_tile_g<T, BT, BT, 2, KSUB>(A + (int64_t)ti * BT * lda, lda, B + (int64_t)tj * BT * ldb, ldb, K, acc, smem);
    store_tile<T, BT, BT, MODE>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
    done_one = true;
  }
}

// the same for C = A W^T with W lower triangular (gemm_nt_ltri_kernel: the rows of a panel solve the main stream takes)
template <typename T, int BT, int KSUB = 1>
__global__ __launch_bounds__(256, (BT == 64 && KSUB >= 4) ? 1 : 2) void gemm_nt_ltri_resv_kernel(
    T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
    int tiles_m, int tiles_n, int sh, int K, unsigned total, unsigned* __restrict__ ctr, int resv, unsigned sweep0) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT, KSUB>::SMEM_ELEMS];
  __shared__ int next_bid;
  if (blockIdx.x + 1 != gridDim.x && on_reserved_cu(resv)) return;
  const bool sweeper = blockIdx.x >= sweep0;
  bool done_one = false;
  const int half = (tiles_n + 1) >> 1;
#pragma unroll 1
  for (;;) {
    if (done_one && !sweeper) return;
    if (threadIdx.x == 0) next_bid = resv_take(ctr, total, blockIdx.x & 7u);
    __syncthreads();
    const int bid = next_bid;
    __syncthreads();
    if (bid < 0) return;
    int ti, tp;
    if (!tile_coords<false>(xcd_chunk_id(bid, total), tiles_m, half, sh, 0, BcMask{0, 1, 0}, ti, tp)) continue;
    const T* At = A + (int64_t)ti * BT * lda;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int tj = pass == 0 ? tiles_n - 1 - tp : tp;
      if (pass == 1 && tj == tiles_n - 1 - tp) break;
      typename Num<T>::v4 acc[BT / 32][BT / 32];
      zero_acc(acc);
      gemm_tile_g<T, BT, BT, 2, KSUB>(At, lda, W + (int64_t)tj * BT * ldw, ldw, min(K, (tj + 1) * BT), acc, smem);
      store_tile<T, BT, BT, 1>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
    }
    done_one = true;
  }
}

// ---- the sharded trailing update: C -= A * B^T under the block-cyclic row map ----------
// Local tile row ti stands for tile row lim(ti) = BcMask::row_tile(ti) of the trailing matrix (its row block's
// place in the dealing, gpx_tile.h: Deal) and owns the tiles tj <= lim(ti): a staircase.  Launching the bounding
// rectangle leaves half of the XCD chunks empty (measured: 37.7 TF where the triangular
// enumeration of the unsharded path gets 68), so the host counts, per super-row of 8 tile rows, the
// 8x8 super-tiles all its rows own completely plus its ragged remainder in single tiles
// (pre[]: exclusive prefix sums of tile slots, passed by value) and a workgroup finds its
// super-row by binary search: exactly the owned tiles are launched — no masked slot that
// would exit at once and scramble the k-phase of its neighbours (cf. tile_coords<TRI>) —
// and the full part stays 8x8 for the L2.
constexpr int STAIR_MAX = 256;
struct StairMap {
  int nsr;
  unsigned pre[STAIR_MAX + 1];
};

// columns owned by local tile row ti (clipped to the launch): tj <= stair_lim
__host__ __device__ __forceinline__ int stair_lim(const BcMask& bc, int ti, int tiles_n) {
  const int lim = bc.row_tile(ti);
  return lim < tiles_n - 1 ? lim : tiles_n - 1;
}

// tile slot lin of the staircase launch -> (ti, tj); pure index arithmetic, also run on the
// host by gpx_debug_tile_map (tests/test_tile_maps.py checks every owned tile appears once)
__host__ __device__ __forceinline__ void stair_coords(const StairMap& map, const BcMask& bc, unsigned lin,
                                                      int tiles_m, int tiles_n, int& ti, int& tj) {
  int lo = 0, hi = map.nsr;
  while (hi - lo > 1) {  // largest super-row with pre[] <= lin (pre[] counts tile slots)
    const int mid = (lo + hi) >> 1;
    if (map.pre[mid] <= lin) lo = mid; else hi = mid;
  }
  // super-row lo: the super-columns every one of its 8 rows owns completely come first
  // (8x8 super-tiles, 64 slots each), then the ragged remainder row by row — no masked slot
  const int row0 = lo * 8;
  const int nfull = (stair_lim(bc, row0, tiles_n) + 1) >> 3;
  unsigned off = lin - map.pre[lo];
  if (off < (unsigned)nfull * 64u) {
    ti = row0 + (int)((off & 63u) >> 3);
    tj = (int)(off >> 6) * 8 + (int)(off & 7u);
  } else {
    off -= (unsigned)nfull * 64u;
    int r = 0;
    for (; r < 8; ++r) {
      const int row = row0 + r;
      const unsigned extra = row < tiles_m ? (unsigned)(stair_lim(bc, row, tiles_n) + 1 - nfull * 8) : 0u;
      if (off < extra) break;
      off -= extra;
    }
    ti = row0 + r;
    tj = nfull * 8 + (int)off;
  }
}

// host: prefix sums of tile slots per super-row; returns the grid size
inline unsigned build_stair_map(const BcMask& bc, int64_t tm, int64_t tn, StairMap& map) {
  const int64_t nsr = (tm + 7) / 8;
  map.nsr = (int)nsr;
  unsigned total = 0;  // exactly the owned tiles (plus the ragged bottom edge of full super-tiles)
  for (int64_t sr = 0; sr < nsr; ++sr) {
    map.pre[sr] = total;
    const int row0 = (int)(8 * sr);
    const int nfull = (stair_lim(bc, row0, (int)tn) + 1) >> 3;
    total += (unsigned)nfull * 64u;
    for (int r = 0; r < 8 && row0 + r < tm; ++r)
      total += (unsigned)(stair_lim(bc, row0 + r, (int)tn) + 1 - nfull * 8);
  }
  map.pre[nsr] = total;
  return total;
}

template <typename T, int BT, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_stair_kernel(
    T* __restrict__ C, int64_t ldc, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
    int64_t ldb, int tiles_m, int tiles_n, BcMask bc, int K, StairMap map) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, BT, BT>::SMEM_ELEMS];
  const unsigned lin = (unsigned)__builtin_amdgcn_readfirstlane((int)xcd_chunk_id(blockIdx.x, gridDim.x));
  int ti, tj;
  stair_coords(map, bc, lin, tiles_m, tiles_n, ti, tj);
  if (ti >= tiles_m || tj >= tiles_n) return;  // ragged bottom edge only
  typename Num<T>::v4 acc[BT / 32][BT / 32];
  zero_acc(acc);
  gemm_tile_g<T, BT, BT>(A + (int64_t)ti * BT * lda, lda, B + bc_brow<BT>(bc, tj) * ldb, ldb, K, acc,
                         smem);
  store_tile<T, BT, BT, MODE>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
}

// ---- skinny products with a long contraction: split-K into partial tiles ---------------------
// C (m x n) = A (m x K) B(n x K)^T with m = 64 and K = N (the posterior mean: z^T V or alpha^T K*^T)
// has n / 64 tiles, each walking the whole K — 64 workgroups on 256 CUs for a 3.4 ms latency chain
// at N = 65536.  blockIdx.y = split s contracts k in [s Ks, min(K, (s+1) Ks)) into
// part[s] (m x ldc); splitk_reduce_kernel adds the S partial tiles in split order, so the result
// is deterministic (no atomics).
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_splitk_kernel(T* __restrict__ part, int64_t ldp, int64_t pstride,
                                                               const T* __restrict__ A, int64_t lda,
                                                               const T* __restrict__ B, int64_t ldb, int tiles_n,
                                                               int K, int Ks) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, 64, 64, 2>::SMEM_ELEMS];
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n, sidx = blockIdx.y;
  const int k0 = sidx * Ks, kn = min(Ks, K - k0);  // Ks and K are multiples of 64
  typename Num<T>::v4 acc[2][2];
  zero_acc(acc);
  gemm_tile_g<T, 64, 64, 2, 2>(A + (int64_t)ti * 64 * lda + k0, lda, B + (int64_t)tj * 64 * ldb + k0, ldb, kn, acc, smem);
  store_tile<T, 64, 64, 1>(part + sidx * pstride + (int64_t)ti * 64 * ldp + (int64_t)tj * 64, ldp, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(T* __restrict__ C, int64_t ldc, const T* __restrict__ part,
                                                           int64_t ldp, int64_t pstride, int S, int64_t n) {
  const int64_t r = blockIdx.y;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
    T v = part[r * ldp + c];
    for (int q = 1; q < S; ++q) v += part[q * pstride + r * ldp + c];
    C[r * ldc + c] = v;
  }
}

// ---- C -= A * B, B stored [k][n]; 64x64 tiles --------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(T* __restrict__ C, int64_t ldc,
                                                         const T* __restrict__ A, int64_t lda,
                                                         const T* __restrict__ B, int64_t ldb,
                                                         int tiles_m, int tiles_n, int sh, int K) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeNN<T, 64, 64>::SMEM_ELEMS];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  int ti, tj;
  if (!tile_coords<false>(lin, tiles_m, tiles_n, sh, 0, BcMask{0, 1, 0}, ti, tj)) return;
  typename Num<T>::v4 acc[2][2];
  zero_acc(acc);
  gemm_tile_nn<T, 64, 64>(A + (int64_t)ti * 64 * lda, lda, B + (int64_t)tj * 64, ldb, K, acc, smem);
  store_tile<T, 64, 64, 0>(C + (int64_t)ti * 64 * ldc + (int64_t)tj * 64, ldc, acc);
}

// ---- POTF2 of one 64x64 block + explicit inverse -------------------------------------
// One workgroup, everything in LDS.  Factorisation in 8 steps of PW = 8 columns (4-column
// steps: 18.9 us of the kernel's 28 in 16 x two barriers + LDS round trips):
//   phase A  every thread factors the 8x8 diagonal block redundantly in registers
//            (rsqrt-based, no divisions); thread i < 64 solves its row of the 8-column
//            panel and drops it in PB[64][8];
//   phase B  the rank-8 trailing update C -= PB PB^T is TWO 16x16x4 MFMAs per 16x16 tile
//            (<= 10 lower tiles over 4 waves); columns already final are masked through a
//            zero B operand and predicated stores.
// Inverse W = L^-1 by recursive blocking: four 16x16 diagonal inverses (one column per
// lane, reciprocal pivots reused from the factorisation), then W21 = -W22 (L21 W11) at
// 32 and at 64 with MFMA products through a small LDS scratch tile.
// LDS (fp64): 2 x 64x66 + scratch = 77 KB — fits the slot of one retiring SYRK workgroup
// (64 KB + spare per CU), which lets the look-ahead stream run beside the trailing
// update; s_setprio(3) keeps its waves ahead of the co-resident SYRK waves.
// A non-positive / NaN pivot records (global index + 1) in *info by atomicMin and lets
// NaN propagate (LAPACK potrf info convention).
constexpr int PLD = 66;
constexpr int TLD = 34;
constexpr int PW = 8;  // columns per factorisation step (a multiple of 4: one MFMA per 4)

// Diagnostic build only (tools/potf2_stamps.py compiles its own copy with -DGPX_STAMPS): shader-clock
// stamps of thread 0 at the phase boundaries of the POTF2 kernel.  Never in the shipped library.
#ifdef GPX_STAMPS
__device__ long long gpx_stamp_buf[64];
#define GPX_STAMP(i)                                                  \
  do {                                                                \
    if (threadIdx.x == 0) gpx_stamp_buf[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define GPX_STAMP(i) \
  do {               \
  } while (0)
#endif

// Factorisation + inverse of the 64x64 block held in LDS (the body shared by potf2_64_kernel and
// potf2_128_kernel).  On entry Wk holds the block (lower triangle; whatever sits above the diagonal
// is never used) and the caller has passed a barrier; Wi is scratch.  On return Wk = L (lower),
// Wi = L^-1 (lower, strictly upper part zero), and a barrier has been passed.  stamp0 < 0: no stamps.
template <typename T>
__device__ __forceinline__ void potf2_lds(T* __restrict__ Wk, T* __restrict__ Wi, T* __restrict__ PB,
                                          T* __restrict__ Tm, T* __restrict__ Rinv, int64_t gidx0,
                                          int* __restrict__ info) {
  using v4 = typename Num<T>::v4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const T zero = (T)0, one = (T)1;
  // Round 3, second half — the two phases of a step run on DIFFERENT waves, and phase B of step j beside phase A
  // of step j + 1:  wave 0 does every phase A (the PW x PW factor, the 64 panel rows: one row per lane; nobody
  // else ever needed it), waves 1..3 hold the trailing matrix — the ten lower 16-tiles in accumulator layout,
  //     wave 1: (0,0) (3,0) (3,1)      wave 2: (1,0) (1,1) (2,2) (3,3)      wave 3: (2,0) (2,1) (3,2)
  // (at most two tiles of any tile column per wave) — and do every phase B.  Per step two barriers:
  //     wave 0: A(j) -> PB[j & 1], L columns into Wk      | waves 1-3: rest of B(j - PW)
  //     ---- X(j): the panel is in PB ----
  //     waves 1-3: update + export of the tile column that holds the NEXT PW columns      | wave 0 idle
  //     ---- Y(j): the next panel's columns are in Wk ----
  // so a step costs phase A + the export part (<= 2 tiles) instead of phase A + all of phase B; the panel buffer
  // is double-buffered (the rest of B(j) reads PB[j & 1] while A(j + PW) writes the other one).
  const unsigned tcode = wave == 1 ? 0xFF0D0C00u : wave == 2 ? 0x0F0A0504u : wave == 3 ? 0xFF0E0908u : 0xFFFFFFFFu;
  v4 R[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const unsigned code = (tcode >> (8 * sl)) & 0xFFu;
    const int tr = (int)(code >> 2) & 3, tc = (int)code & 3;
    R[sl] = (v4){0, 0, 0, 0};
    if (code != 0xFFu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) R[sl][r] = Wk[(tr * 16 + Num<T>::drow(l4, r)) * PLD + tc * 16 + l15];
    }
  }
  __syncthreads();  // every tile is in registers before phase A of step 0 starts to overwrite columns of Wk
  // the rank-PW update of this wave's tile in slot sl with the panel in PBc; `exp`: also write the NEXT panel's
  // columns [jn, jn + PW) of the tile back to Wk for phase A to read
  auto update_tile = [&](int sl, const T* PBc, int jn, bool exp) {
    const unsigned code = (tcode >> (8 * sl)) & 0xFFu;
    const int tr = (int)(code >> 2) & 3, tc = (int)code & 3;
    const int colg = tc * 16 + l15;
    v4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int hh = 0; hh < PW / 4; ++hh) {
      const T av = PBc[(tr * 16 + l15) * PW + 4 * hh + l4];
      const T bv = (colg >= jn) ? PBc[colg * PW + 4 * hh + l4] : zero;
      acc = Num<T>::mfma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) R[sl][r] -= acc[r];
    if (exp && (l15 >> 3) == ((jn >> 3) & 1)) {
      T* Cp = Wk + (tr * 16) * PLD + colg;
#pragma unroll
      for (int r = 0; r < 4; ++r) Cp[Num<T>::drow(l4, r) * PLD] = R[sl][r];
    }
  };
#ifndef GPX_POTF2_UNROLL
#pragma unroll 1
#endif
  for (int j = 0; j < 64; j += PW) {
    T* PBc = PB + ((j / PW) & 1) * (64 * PW);
    if (wave == 0) {
      // ---- phase A: PW x PW diagonal factor (right-looking in registers, the same on every lane) + this
      //      lane's panel row
      T a[PW][PW], rs[PW];
      bool okp[PW];
#pragma unroll
      for (int c = 0; c < PW; ++c)
#pragma unroll
        for (int r = c; r < PW; ++r) a[r][c] = Wk[(j + r) * PLD + j + c];
#pragma unroll
      for (int k = 0; k < PW; ++k) {
        okp[k] = a[k][k] > zero;
        rs[k] = Num<T>::rsq(a[k][k]);
#pragma unroll
        for (int r = k + 1; r < PW; ++r) a[r][k] *= rs[k];
#pragma unroll
        for (int c = k + 1; c < PW; ++c)
#pragma unroll
          for (int r = c; r < PW; ++r) a[r][c] -= a[r][k] * a[c][k];
      }
      if (lane == 0) {
        int bad = 0;
#pragma unroll
        for (int k = PW - 1; k >= 0; --k)
          if (!okp[k]) bad = k + 1;  // first non-positive (or NaN) pivot of this step
        if (bad) atomicMin(info, (int)(gidx0 + j + bad));
#pragma unroll
        for (int k = 0; k < PW; ++k) Rinv[j + k] = rs[k];
      }
      const int i = lane;
      T x[PW];
#pragma unroll
      for (int k = 0; k < PW; ++k) x[k] = zero;
      if (i >= j) {
        T* row = Wk + i * PLD + j;
        const int c = i - j;  // rows of the diagonal block: strictly-upper part is zero
#pragma unroll
        for (int k = 0; k < PW; ++k) {
          T v = row[k];
#pragma unroll
          for (int m = 0; m < k; ++m) v -= x[m] * a[k][m];
          x[k] = (c < k) ? zero : v * rs[k];
        }
#pragma unroll
        for (int k = 0; k < PW; ++k) row[k] = x[k];  // commit the panel: these columns of L are final
      }
#pragma unroll
      for (int k = 0; k < PW; ++k) PBc[i * PW + k] = x[k];
    }
    __syncthreads();  // X(j)
    GPX_STAMP(2 + 2 * (j / PW));
    const int jn = j + PW;
    const int t0 = jn >> 4;
    if (wave != 0) {  // the tile column the next panel lives in: update, export
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const unsigned code = (tcode >> (8 * sl)) & 0xFFu;
        if (code != 0xFFu && (int)(code & 3) == t0) update_tile(sl, PBc, jn, jn < 64);
      }
    }
    __syncthreads();  // Y(j)
    GPX_STAMP(3 + 2 * (j / PW));
    if (wave != 0) {  // the tile columns to the right of it: beside phase A of the next step
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const unsigned code = (tcode >> (8 * sl)) & 0xFFu;
        if (code != 0xFFu && (int)(code & 3) > t0) update_tile(sl, PBc, jn, false);
      }
    }
  }
  __syncthreads();  // (the last step leaves nothing to update; everybody meets before the inverse)

  // ---- inverse, level 0: wave w inverts diagonal block w; lane c < 16 owns column c.  The six
  //      16x16 blocks above the block diagonal are zeroed here (level 2 reads two of them, and the
  //      caller stores the whole 64x64 inverse): Wi needs no initialisation by the caller.
  {
    for (int e = tid; e < 6 * 256; e += 256) {
      const int bidx = e >> 8, w = e & 255;          // blocks (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
      const int br = bidx < 3 ? 0 : bidx < 5 ? 1 : 2;
      const int bc = bidx < 3 ? bidx + 1 : bidx < 5 ? bidx - 1 : 3;
      Wi[(br * 16 + (w >> 4)) * PLD + bc * 16 + (w & 15)] = zero;
    }
    const int b0 = wave * 16;
    if (lane < 16) {
      T w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        T sacc = (i == lane) ? one : zero;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= Wk[(b0 + i) * PLD + b0 + k] * w[k];
        w[i] = sacc * Rinv[b0 + i];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Wi[(b0 + i) * PLD + b0 + lane] = w[i];
    }
  }
  __syncthreads();
  // ---- level 1: blocks (1,0) on wave 0 and (3,2) on wave 1:  W10 = -W11 (L10 W00)
  if (wave < 2) {
    const int c0 = wave * 32, r1 = c0 + 16;
    v4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      acc = Num<T>::mfma(Wk[(r1 + l15) * PLD + c0 + ks * 4 + l4], Wi[(c0 + ks * 4 + l4) * PLD + c0 + l15],
                         acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Tm[(wave * 16 + Num<T>::drow(l4, r)) * TLD + l15] = acc[r];
  }
  __syncthreads();
  if (wave < 2) {
    const int c0 = wave * 32, r1 = c0 + 16;
    v4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      acc = Num<T>::mfma(Wi[(r1 + l15) * PLD + r1 + ks * 4 + l4], Tm[(wave * 16 + ks * 4 + l4) * TLD + l15],
                         acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wi[(r1 + Num<T>::drow(l4, r)) * PLD + c0 + l15] = -acc[r];
  }
  __syncthreads();
  // ---- level 2: W_BA = -W_BB (L_BA W_AA), 32x32 blocks, one 16x16 tile per wave
  {
    const int tr = wave >> 1, tc = wave & 1;
    v4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc = Num<T>::mfma(Wk[(32 + tr * 16 + l15) * PLD + ks * 4 + l4], Wi[(ks * 4 + l4) * PLD + tc * 16 + l15],
                         acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) Tm[(tr * 16 + Num<T>::drow(l4, r)) * TLD + tc * 16 + l15] = acc[r];
    __syncthreads();
    v4 acc2 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc2 = Num<T>::mfma(Wi[(32 + tr * 16 + l15) * PLD + 32 + ks * 4 + l4],
                          Tm[(ks * 4 + l4) * TLD + tc * 16 + l15], acc2);
    __syncthreads();  // every read of W_BB / Tm done before W_BA lands next to them
#pragma unroll
    for (int r = 0; r < 4; ++r)
      Wi[(32 + tr * 16 + Num<T>::drow(l4, r)) * PLD + tc * 16 + l15] = -acc2[r];
  }
  __syncthreads();
}

// flag != null: after the last store of L and the inverse, a device-scope release and *flag = flag_val —
// the side stream that extends the panel's block inverse is parked on wait_counter_kernel(flag, flag_val)
// instead of a hipEvent recorded on the chain's stream (round 3: ~7 us per step of the serial chain).
template <typename T>
__global__ __launch_bounds__(256) void potf2_64_kernel(T* __restrict__ A, int64_t lda,
                                                       T* __restrict__ Winv, int64_t gidx0,
                                                       int* __restrict__ info, unsigned* __restrict__ flag,
                                                       unsigned flag_val) {
  __shared__ __attribute__((aligned(16))) T Wk[64 * PLD];  // working matrix -> L (lower)
  __shared__ __attribute__((aligned(16))) T Wi[64 * PLD];  // inverse
  __shared__ __attribute__((aligned(16))) T PB[2 * 64 * PW];  // the current and the next PW-column panel
  __shared__ __attribute__((aligned(16))) T Tm[32 * TLD];  // product scratch
  __shared__ T Rinv[64];                                    // 1 / L[i][i]
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const T zero = (T)0;
  // "begin" mark (flag_val - 1): everything queued before this kernel on its stream is complete — what the
  // "pre" part of the inverse extension waits for; no fence needed (the kernel boundary was the release)
  if (flag && tid == 0) __hip_atomic_store(flag, flag_val - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  GPX_STAMP(0);
  {
    // all 16 loads of a thread in flight before the first LDS write: the tile comes from HBM /
    // Infinity Cache (the trailing update's atomics leave nothing in L2), and one dependent
    // load -> ds_write pair per iteration serialised 16 misses (18 k of the kernel's 62 k cycles)
    T v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u, i = e >> 6, k = e & 63;
      v[u] = (k <= i) ? A[(int64_t)i * lda + k] : zero;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = tid + 256 * u, i = e >> 6, k = e & 63;
      Wk[i * PLD + k] = v[u];
    }
  }
  __syncthreads();
  GPX_STAMP(1);
  potf2_lds<T>(Wk, Wi, PB, Tm, Rinv, gidx0, info);
  GPX_STAMP(20);
  for (int e = tid; e < 4096; e += 256) {
    const int ii = e >> 6, k = e & 63;
    if (k <= ii) A[(int64_t)ii * lda + k] = Wk[ii * PLD + k];
    Winv[e] = Wi[ii * PLD + k];
  }
  GPX_STAMP(21);
  if (flag) {
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, flag_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- POTF2 of a 128x128 diagonal tile in ONE launch (round 3) ------------------------------------
//   [A11     ]      L11 = chol(A11), W11 = L11^-1             (potf2_lds)
//   [A21  A22]      L21 = A21 W11^T                            (MFMA, k limited to W11's triangle)
//                   A22 <- A22 - L21 L21^T (lower 16-tiles)    (MFMA, A22 held in the accumulators)
//                   L22 = chol(A22), W22 = L22^-1              (potf2_lds, buffers swapped)
// The serial diagonal chain of a panel then runs ONE POTF2, ONE solve of the rows below, ONE in-block
// update and ONE inverse-extension event per 128 columns instead of per 64 (the
// chain, not the flops, is the time of a small problem — BASELINE.json configs[1]).  Same two
// 64x66 LDS buffers as the 64-wide kernel, used in turn: U = A11 -> L11 -> A21 -> L21 -> W22,
// V = W11 -> A22' -> L22.  All three input tiles are requested before the first factorisation, so the
// second one never waits for memory.  Winv: the two 64x64 inverses, back to back.
template <typename T>
__global__ __launch_bounds__(256) void potf2_128_kernel(T* __restrict__ A, int64_t lda,
                                                        T* __restrict__ Winv, int64_t gidx0,
                                                        int* __restrict__ info, unsigned* __restrict__ flag,
                                                        unsigned flag_val) {
  using v4 = typename Num<T>::v4;
  __shared__ __attribute__((aligned(16))) T U[64 * PLD];
  __shared__ __attribute__((aligned(16))) T V[64 * PLD];
  __shared__ __attribute__((aligned(16))) T PB[2 * 64 * PW];
  __shared__ __attribute__((aligned(16))) T Tm[32 * TLD];
  __shared__ T Rinv[64];
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const T zero = (T)0;
  if (flag && tid == 0) __hip_atomic_store(flag, flag_val - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // "begin": see potf2_64_kernel
  T* A21 = A + (int64_t)64 * lda;
  T* A22 = A21 + 64;
  // lower 16-tiles (tr, tc), tc <= tr, of the 64x64 block A22 in row-major order idx = tr (tr + 1) / 2 + tc;
  // wave w holds tiles w, w + 4, w + 8 (< 10) in accumulator layout
  T v11[16], v21[16];
  v4 c22[3];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, i = e >> 6, k = e & 63;
    v11[u] = (k <= i) ? A[(int64_t)i * lda + k] : zero;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, i = e >> 6, k = e & 63;
    v21[u] = A21[(int64_t)i * lda + k];
  }
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {  // this wave's tiles: idx = wave + 4 sl < 10
    const int idx = wave + 4 * sl;
    const int tr = idx >= 6 ? 3 : idx >= 3 ? 2 : idx >= 1 ? 1 : 0, tc = idx - tr * (tr + 1) / 2;
    c22[sl] = (v4){0, 0, 0, 0};
    if (idx < 10) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tr * 16 + Num<T>::drow(l4, r), col = tc * 16 + l15;
        c22[sl][r] = (col <= row) ? A22[(int64_t)row * lda + col] : zero;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, i = e >> 6, k = e & 63;
    U[i * PLD + k] = v11[u];
  }
  __syncthreads();
  potf2_lds<T>(U, V, PB, Tm, Rinv, gidx0, info);  // U = L11, V = W11
  for (int e = tid; e < 4096; e += 256) {
    const int ii = e >> 6, k = e & 63;
    if (k <= ii) A[(int64_t)ii * lda + k] = U[ii * PLD + k];
    Winv[e] = V[ii * PLD + k];
  }
  __syncthreads();  // every read of L11 done: U becomes the staging buffer of A21
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, i = e >> 6, k = e & 63;
    U[i * PLD + k] = v21[u];
  }
  __syncthreads();
  {
    // L21 = A21 W11^T: wave w owns the 16-row strip w (tiles (w, 0..3)); W11[c][k] = 0 for k > c
    v4 acc[4];
#pragma unroll
    for (int tc = 0; tc < 4; ++tc) {
      acc[tc] = (v4){0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        if (ks < 4 * (tc + 1))
          acc[tc] = Num<T>::mfma(U[(wave * 16 + l15) * PLD + ks * 4 + l4], V[(tc * 16 + l15) * PLD + ks * 4 + l4],
                                 acc[tc]);
    }
    __syncthreads();  // every read of A21 (U) and W11 (V) done
#pragma unroll
    for (int tc = 0; tc < 4; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + Num<T>::drow(l4, r), col = tc * 16 + l15;
        U[row * PLD + col] = acc[tc][r];
        A21[(int64_t)row * lda + col] = acc[tc][r];
      }
  }
  __syncthreads();
  {
    // A22 <- A22 - L21 L21^T on the lower 16-tiles; the result becomes the working matrix in V
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int idx = wave + 4 * sl;
      const int tr = idx >= 6 ? 3 : idx >= 3 ? 2 : idx >= 1 ? 1 : 0, tc = idx - tr * (tr + 1) / 2;
      if (idx < 10) {
        v4 acc = c22[sl];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          acc = Num<T>::mfma(-U[(tr * 16 + l15) * PLD + ks * 4 + l4], U[(tc * 16 + l15) * PLD + ks * 4 + l4], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) V[(tr * 16 + Num<T>::drow(l4, r)) * PLD + tc * 16 + l15] = acc[r];
      }
    }
  }
  __syncthreads();
  potf2_lds<T>(V, U, PB, Tm, Rinv, gidx0 + 64, info);  // V = L22, U = W22
  for (int e = tid; e < 4096; e += 256) {
    const int ii = e >> 6, k = e & 63;
    if (k <= ii) A22[(int64_t)ii * lda + k] = V[ii * PLD + k];
    Winv[4096 + e] = U[ii * PLD + k];
  }
  if (flag) {  // see potf2_64_kernel
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, flag_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- X <- X * L^-T (right, lower, transposed): ascending 64-column blocks --------------
// One workgroup owns a 64-row slab of X and walks the nb/64 column blocks in order:
//   T    = X[:, jb] - sum_{kb<jb} X[:, kb] L[jb, kb]^T      (MFMA, K = 64*jb)
//   X_jb = T * Winv_jb^T                                       (MFMA, K = 64)
// T makes a round trip through the slab's own global tile (L2-resident) so both
// products run on the same tile engine; a slab is private to its workgroup, so the
// only ordering needed is the workgroup barrier.
// MEMORY-ORDERING INVARIANT: store_tile<MODE 0> updates T with no-return atomics that execute at
// L2, and the LDS-DMA loads that read it back go through this CU's L1, which those atomics do
// not refresh.  The read is fresh only because no 128-byte line of the tile can be in the L1
// already: (i) every row of X starts on a 128-byte boundary and the tile's columns start at a
// multiple of 64 elements, so no line straddles the tile's left edge (ld_skew, gpx_internal.h;
// checked by the launcher), and (ii) the loads issued earlier in the walk touch only columns
// < 64 jb of this slab, and nothing else of X.  The final result of block jb is written with
// plain stores AFTER the read and never loaded by this workgroup again.
// C^T -> Wt: element (r, c) of the wave-tiled 64x64 accumulator goes to Wt[c * ldw + r]
template <typename T>
__device__ __forceinline__ void store_tile_transposed(T* Wt, int64_t ldw, const typename Num<T>::v4 (&acc)[2][2],
                                                      T sign) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1, l4 = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 32 + m * 16 + Num<T>::drow(l4, r), col = wc * 32 + n * 16 + l15;
        Wt[(int64_t)col * ldw + row] = sign * acc[m][n][r];
      }
}

// Column blocks [jb_lo, jb_hi) only (the earlier ones must already be solved).  tri != 0: X is
// block upper triangular (slab s is zero left of column block s — the inverse being built from
// the identity), so slab s starts its contraction at column 64 s.  Wt != null: the solved block
// is also written transposed, Wt[(64 jb + c) * ldw + 64 slab + r] = X[r][64 jb + c].
// phase 0: the whole walk.  phase 1 ("pre"): only the part of every block's contraction that lies LEFT of
// column block jb_lo — X_jb -= sum_{kb < jb_lo} X_kb L[jb, kb]^T for jb in [jb_lo, jb_hi), no inverse — which
// needs nothing the current diagonal step produces; phase 2 ("post"): the rest (the blocks inside
// [jb_lo, jb), then the inverse).  The extension of a panel's block inverse is split that way (round 3): the
// long part of its walk runs BESIDE the step's POTF2 instead of behind it.
template <typename T, int KSUB>
__global__ __launch_bounds__(256, KSUB >= 4 ? 1 : 2) void trsm_rlt_kernel(T* X, int64_t ldx, const T* L, int64_t ldl,
                                                          const T* Winv, int jb_lo, int jb_hi, T* P, int64_t ldp,
                                                          int tri, T* Wt, int64_t ldw, int phase) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeG<T, 64, 64, KSUB>::SMEM_ELEMS];
  __builtin_amdgcn_s_setprio(2);  // panel solve is on the critical path of the look-ahead
  T* Xs = X + (int64_t)blockIdx.x * 64 * ldx;
  T* Ps = P ? P + (int64_t)blockIdx.x * 64 * ldp : nullptr;
  const int k0 = tri ? (int)blockIdx.x * 64 : 0;
  typename Num<T>::v4 acc[2][2];
  if (phase == 1) {
    const int kend = jb_lo * 64;
    if (kend <= k0) return;
    const int jb = jb_lo + (int)blockIdx.y;  // the blocks of a "pre" walk do not depend on each other: one per workgroup
    zero_acc(acc);
    gemm_tile_g<T, 64, 64, 2, KSUB>(Xs + k0, ldx, L + (int64_t)jb * 64 * ldl + k0, ldl, kend - k0, acc, smem);
    store_tile<T, 64, 64, 0>(Xs + jb * 64, ldx, acc);
    return;
  }
  for (int jb = jb_lo; jb < jb_hi; ++jb) {
    T* Xj = Xs + jb * 64;
    const int ks = phase == 2 ? max(k0, jb_lo * 64) : k0;  // phase 2: what lies left of block jb_lo is in already
    if (jb * 64 > ks) {
      zero_acc(acc);
      gemm_tile_g<T, 64, 64, 2, KSUB>(Xs + ks, ldx, L + (int64_t)jb * 64 * ldl + ks, ldl, jb * 64 - ks, acc, smem);
      store_tile<T, 64, 64, 0>(Xj, ldx, acc);
      __syncthreads();
    }
    zero_acc(acc);
    gemm_tile_g<T, 64, 64, 2, KSUB>(Xj, ldx, Winv + (int64_t)jb * 4096, 64, 64, acc, smem);
    // gemm_tile ends with a barrier: every read of T is complete
    store_tile<T, 64, 64, 1>(Xj, ldx, acc);
    if (Ps) store_tile<T, 64, 64, 1>(Ps + jb * 64, ldp, acc);
    if (Wt) store_tile_transposed<T>(Wt + (int64_t)jb * 64 * ldw + (int64_t)blockIdx.x * 64, ldw, acc, (T)1);
    __syncthreads();
  }
}

// ---- X <- X * L^-1 (right, lower, no transpose): descending blocks ----------------------
//   T   = X[:, q] - sum_{k>q} X[:, k] L[k, q]        (B operand is L stored [k][n])
//   X_q = T * Winv_q
template <typename T>
__global__ __launch_bounds__(256, 2) void trsm_rln_kernel(T* X, int64_t ldx, const T* L, int64_t ldl,
                                                          const T* Winv, int nbq) {
  __shared__ __attribute__((aligned(16))) T smem[TileShapeNN<T, 64, 64>::SMEM_ELEMS];
  T* Xs = X + (int64_t)blockIdx.x * 64 * ldx;
  typename Num<T>::v4 acc[2][2];
  for (int q = nbq - 1; q >= 0; --q) {
    T* Xq = Xs + q * 64;
    if (q < nbq - 1) {
      zero_acc(acc);
      gemm_tile_nn<T, 64, 64>(Xs + (q + 1) * 64, ldx, L + (int64_t)(q + 1) * 64 * ldl + q * 64, ldl,
                              (nbq - 1 - q) * 64, acc, smem);
      store_tile<T, 64, 64, 0>(Xq, ldx, acc);
      __syncthreads();
    }
    zero_acc(acc);
    gemm_tile_nn<T, 64, 64>(Xq, ldx, Winv + (int64_t)q * 4096, 64, 64, acc, smem);
    store_tile<T, 64, 64, 1>(Xq, ldx, acc);
    __syncthreads();
  }
}

int64_t rect_grid(int64_t tm, int64_t tn, int& sh) {
  sh = tm >= 8 ? 8 : 1;
  const int sw = 64 / sh;
  return ((tm + sh - 1) / sh) * ((tn + sw - 1) / sw) * 64;
}


// Small grids: when a launch has no more tiles than the chip has CUs, the dispatcher still packs
// two workgroups onto one CU (64 KB of LDS each) and leaves half the CUs idle — each tile then
// walks its K at the half-CU rate.  Asking for enough extra (unused) dynamic LDS that only one
// workgroup fits per CU spreads them out: a lone workgroup runs 1.75x as fast as one of a
// co-resident pair.  Matters for the panel / block solves and the strips of small
// problems (BASELINE.json configs[1]); large launches are unaffected.
// see launch_gemm_nt_t: several k-steps per barrier for 64-tile launches, only when the caller says that
// nothing large runs beside them (per host thread: a handle is driven by one thread at a time)
thread_local int g_latency_mode = 0;
// reserve mode of the calling host thread (set_reserve_mode): k > 0 — the launches of launch_gemm_nt go out in their
// self-reserving form, each with 8 zeroed device counters of its own out of the caller's ring
struct ReserveState {
  int k = 0;
  unsigned* ring = nullptr;  // [cap][8] counters, zeroed by the caller before the first launch of a fit
  int cap = 0, used = 0;
  int chain = 0;             // narrow slab launches of the chain ask for a whole CU's worth of LDS (see launch_trsm_rlt)
};
thread_local ReserveState g_resv;

inline int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

template <typename T, int BT>
void launch_gemm_nt_t(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                      int64_t n, int64_t k, int lower, int mode, BcMask bc, hipStream_t st) {
  const int64_t tm = m / BT, tn = n / BT;
  dim3 block(256);
  const int64_t live = lower == 1   ? tm * (tm + 1) / 2
                       : lower == 4 ? tm * ((tn + 1) / 2)
                       : lower == 2 ? tm * tn - std::min(tm, tn) * (std::min(tm, tn) - 1) / 2 - (tn > tm ? (tn - tm) * tm : 0)
                                    : tm * tn;
  // 160 KB per CU: static staging (64 KB at BT = 128) + this > 80 KB  =>  one workgroup per CU
  const unsigned spread = (BT == 128 && live <= cu_count()) ? 32 * 1024 : 0;
  // 64-tiles (latency-bound small launches): several k-steps per barrier, as many as the number of live
  // tiles lets the LDS afford — up to one tile per CU: 4 (fp64; 128 KB, one workgroup per CU anyway), up to
  // four per CU: 2 (64 KB, two workgroups per CU), beyond that the plain engine at four workgroups per CU,
  // whose occupancy hides the latency.  k is a multiple of 64 inside the library.
  // ONLY in latency mode (set_latency_mode: the caller knows that no large update runs beside this launch):
  // under a 128-tile trailing update a 64-tile workgroup must fit BESIDE the two resident update workgroups
  // (2 x 64 KB of the CU's 160 KB: 32 KB is all that is left) — with 64 or 128 KB it waits for a CU to drain,
  // and at N = 65536 the chain's kernels then took 6x as long (measured: 1242 ms of chain instead of 214).
  int ks = 1;
  if (BT == 64 && g_latency_mode) {
    if (live <= 4 * cu_count() && k % (2 * Num<T>::BK) == 0) ks = 2;
    if (Num<T>::KS64 == 4 && live <= cu_count() && k % (4 * Num<T>::BK) == 0) ks = 4;
  }
  // self-reserving form?  (block-cyclic masks never: the sharded update has its own kernel)
  unsigned* rctr = nullptr;
  if (g_resv.k > 0 && bc.P == 0 && lower != 3 && g_resv.used < g_resv.cap) rctr = g_resv.ring + 8 * (size_t)g_resv.used++;
  // persistent grid: the chip's slots for this kernel (two 64 KB workgroups per CU; 64-tiles: up to four) plus what
  // the reserved CUs burn (their workgroups exit at once and the dispatcher refills those slots)
  // Two forms (GPX_RESV_FORM).  1, "turnover" (default): one tile per workgroup as in the static launch — slots keep
  // turning over, so the chain's wide launches (strip of the next diagonal block, panel product: hundreds of
  // workgroups) find room everywhere as before — plus a margin for the workgroups the reserved CUs burn (they exit at
  // once there) and a tail of persistent "sweepers" that finish whatever the burnt ones left.  0, "persistent": about
  // as many workgroups as the chip has slots, all looping (measured: the chain's wide launches then only get the
  // reserved CUs until the update retires — C2 13.3 against 12.5 ms).
  static const int form = [] {
    const char* e = getenv("GPX_RESV_FORM");
    return e ? atoi(e) : 1;
  }();
  unsigned sweep0 = 0;
  auto resv_grid = [&](int64_t total) {
    const int per_cu = BT == 128 ? 2 : (ks == 4 ? 1 : ks == 2 ? 2 : 4);
    const int64_t slots = (int64_t)cu_count() * per_cu;
    if (form == 0) {
      sweep0 = 0;
      return dim3((unsigned)(std::min<int64_t>(total, slots) + 64 * g_resv.k));
    }
    const int64_t sweepers = std::min<int64_t>(total, slots / 2);
    const int64_t ones = total - sweepers + 256 * g_resv.k;  // one-tile workgroups incl. what the reserved CUs will burn
    sweep0 = (unsigned)ones;
    return dim3((unsigned)(ones + sweepers));
  };
#define GPX_NT_LAUNCH(TRI_, MODE_, SH_, MASK_)                                                                                  \
  do {                                                                                                                         \
    if (rctr) {                                                                                                                \
      const unsigned total_ = grid.x;                                                                                          \
      const dim3 pg_ = resv_grid(total_);                                                                                      \
      if (ks == 4)                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_resv_kernel<T, BT, TRI_, MODE_, (BT == 64 ? 4 : 1)>), pg_, block, 0, st, C, ldc, A, lda,   \
                           B, ldb, (int)tm, (int)tn, SH_, MASK_, (int)k, total_, rctr, g_resv.k, sweep0);                              \
      else if (ks == 2)                                                                                                        \
        hipLaunchKernelGGL((gemm_nt_resv_kernel<T, BT, TRI_, MODE_, (BT == 64 ? 2 : 1)>), pg_, block, 0, st, C, ldc, A, lda,   \
                           B, ldb, (int)tm, (int)tn, SH_, MASK_, (int)k, total_, rctr, g_resv.k, sweep0);                              \
      else                                                                                                                     \
        hipLaunchKernelGGL((gemm_nt_resv_kernel<T, BT, TRI_, MODE_, 1>), pg_, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm,   \
                           (int)tn, SH_, MASK_, (int)k, total_, rctr, g_resv.k, sweep0);                                               \
    } else if (ks == 4)                                                                                                        \
      hipLaunchKernelGGL((gemm_nt_kernel<T, BT, TRI_, MODE_, (BT == 64 ? 4 : 1)>), grid, block, spread, st, C, ldc, A, lda, B, \
                         ldb, (int)tm, (int)tn, SH_, MASK_, bc, (int)k);                                                       \
    else if (ks == 2)                                                                                                          \
      hipLaunchKernelGGL((gemm_nt_kernel<T, BT, TRI_, MODE_, (BT == 64 ? 2 : 1)>), grid, block, spread, st, C, ldc, A, lda, B, \
                         ldb, (int)tm, (int)tn, SH_, MASK_, bc, (int)k);                                                       \
    else                                                                                                                       \
      hipLaunchKernelGGL((gemm_nt_kernel<T, BT, TRI_, MODE_, 1>), grid, block, spread, st, C, ldc, A, lda, B, ldb, (int)tm,    \
                         (int)tn, SH_, MASK_, bc, (int)k);                                                                     \
  } while (0)
  if (lower == 1) {  // full lower triangle, triangular super-tile enumeration
    const int64_t ts = (tm + 7) / 8;
    dim3 grid((unsigned)(ts * (ts - 1) / 2 * 64 + ts * 36));  // tile_coords<TRI>: no masked slots
    if (mode == 0)
      GPX_NT_LAUNCH(true, 0, 8, 0);
    else
      GPX_NT_LAUNCH(true, 1, 8, 0);
  } else if (lower == 4) {  // C = A W^T, W lower triangular: paired tile columns (mode 1 only)
    int sh;
    dim3 grid((unsigned)rect_grid(tm, (tn + 1) / 2, sh));
    if (rctr) {
      const unsigned total_ = grid.x;
      const dim3 pg_ = resv_grid(total_);
      if (ks == 4)
        hipLaunchKernelGGL((gemm_nt_ltri_resv_kernel<T, BT, (BT == 64 ? 4 : 1)>), pg_, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k, total_, rctr, g_resv.k, sweep0);
      else if (ks == 2)
        hipLaunchKernelGGL((gemm_nt_ltri_resv_kernel<T, BT, (BT == 64 ? 2 : 1)>), pg_, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k, total_, rctr, g_resv.k, sweep0);
      else
        hipLaunchKernelGGL((gemm_nt_ltri_resv_kernel<T, BT, 1>), pg_, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k, total_, rctr, g_resv.k, sweep0);
    } else if (ks == 4)
      hipLaunchKernelGGL((gemm_nt_ltri_kernel<T, BT, (BT == 64 ? 4 : 1)>), grid, block, spread, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k);
    else if (ks == 2)
      hipLaunchKernelGGL((gemm_nt_ltri_kernel<T, BT, (BT == 64 ? 2 : 1)>), grid, block, spread, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k);
    else
      hipLaunchKernelGGL((gemm_nt_ltri_kernel<T, BT, 1>), grid, block, spread, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, (int)k);
  } else {  // rectangle; lower == 2: masked to tj <= ti; lower == 3: block-cyclic mask
    int sh;
    dim3 grid((unsigned)rect_grid(tm, tn, sh));
    const int mask = lower == 2 ? 1 : lower == 3 ? 2 : 0;
    if (mode == 0)
      GPX_NT_LAUNCH(false, 0, sh, mask);
    else
      GPX_NT_LAUNCH(false, 1, sh, mask);
  }
#undef GPX_NT_LAUNCH
}

}  // namespace

// A launcher that refuses its operands (the 128-byte row alignment trsm_rlt_kernel depends on)
// launches nothing and raises this flag; every API entry point turns it into an error return.
void set_latency_mode(int on) { g_latency_mode = on; }
void reserve_ring(unsigned* ring, int cap) {
  g_resv.ring = ring;
  g_resv.cap = ring ? cap : 0;
  g_resv.used = 0;
  g_resv.k = 0;
  g_resv.chain = 0;
}
void set_reserve_mode(int k) { g_resv.k = g_resv.ring ? k : 0; }
void set_reserve_chain(int on) { g_resv.chain = g_resv.ring ? on : 0; }

static thread_local int g_launch_error = 0;  // per host thread = per API call in flight (a handle is used by one thread at a time)
int take_launch_error() {
  const int e = g_launch_error;
  g_launch_error = 0;
  return e;
}

template <typename T>
void launch_potf2_64(T* A, int64_t lda, T* Winv, int64_t gidx0, int* info, hipStream_t st, unsigned* flag,
                     unsigned flag_val) {
  debug_delay(st);
  hipLaunchKernelGGL(potf2_64_kernel<T>, dim3(1), dim3(256), 0, st, A, lda, Winv, gidx0, info, flag, flag_val);
}

template <typename T>
void launch_potf2_128(T* A, int64_t lda, T* Winv, int64_t gidx0, int* info, hipStream_t st, unsigned* flag,
                      unsigned flag_val) {
  debug_delay(st);
  // Reserve mode, chain side: the POTF2 asks for so much LDS (130 KB in all) that no CU holding even ONE update workgroup
  // (32 KB at least) can take it — it lands on a CU the self-reserving update leaves alone, where its fp64 vector work has
  // the double-precision pipe to itself.  (Small-LDS kernels — the side stream's slabs — may still share that CU.)
  unsigned pad = 0;
  if (g_resv.chain) {
    static const unsigned pad130 = [] {
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(potf2_128_kernel<T>)) != hipSuccess) return 0u;
      const unsigned want = 130u * 1024u;
      if (fa.sharedSizeBytes >= want) return 0u;
      const unsigned p = want - (unsigned)fa.sharedSizeBytes;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_128_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)p) != hipSuccess)
        return 0u;
      return p;
    }();
    pad = pad130;
  }
  hipLaunchKernelGGL(potf2_128_kernel<T>, dim3(1), dim3(256), pad, st, A, lda, Winv, gidx0, info, flag, flag_val);
}

template <typename T>
void launch_trsm_rlt(T* X, int64_t ldx, int64_t rows, const T* L, int64_t ldl, const T* Winv, int nb,
                     T* P, int64_t ldp, hipStream_t st) {
  debug_delay(st);
  // the kernel's L1 invariant (see trsm_rlt_kernel): 128-byte aligned rows of X.  Every caller
  // builds ldx from ld_skew<T>(); a violation is a programming error in this library, caught
  // here before a silently stale read can happen.
  if (((uintptr_t)X | (uintptr_t)(ldx * (int64_t)sizeof(T))) % 128 != 0) {
    g_launch_error = 1;  // reported by the API call in flight (take_launch_error)
    return;
  }
  // few slabs (the diagonal chain, the alpha solves): every product is one latency-bound walk -> KS64 lines
  // per barrier; many slabs: the plain engine, whose occupancy hides the latency
  // (reserve mode, chain side: a narrow slab launch takes the KS64 form too — 128 KB of LDS per workgroup, which only an
  // empty CU holds, i.e. one the self-reserving update leaves alone)
  if ((g_latency_mode && rows / 64 <= cu_count()) || (g_resv.chain > 1 && rows / 64 <= 16))
    hipLaunchKernelGGL((trsm_rlt_kernel<T, Num<T>::KS64>), dim3((unsigned)(rows / 64)), dim3(256), 0, st, X, ldx, L, ldl,
                       Winv, 0, nb / 64, P, ldp, 0, (T*)nullptr, (int64_t)0, 0);
  else
    hipLaunchKernelGGL((trsm_rlt_kernel<T, 1>), dim3((unsigned)(rows / 64)), dim3(256), 0, st, X, ldx, L, ldl,
                       Winv, 0, nb / 64, P, ldp, 0, (T*)nullptr, (int64_t)0, 0);
}

template <typename T>
void launch_inv_extend(T* U, int64_t ldu, const T* L, int64_t ldl, const T* Winv, int q0, int q1, T* W,
                       int64_t ldw, hipStream_t st, int phase) {
  debug_delay(st);
  if (((uintptr_t)U | (uintptr_t)(ldu * (int64_t)sizeof(T))) % 128 != 0) {
    g_launch_error = 1;
    return;
  }
  if (phase == 1 && q0 == 0) return;  // nothing lies left of the first block
  // "pre": only the slabs that start left of block q0 have work, one workgroup per (slab, block)
  const dim3 grid = phase == 1 ? dim3((unsigned)q0, (unsigned)(q1 - q0)) : dim3((unsigned)q1);
  if (g_latency_mode)
    hipLaunchKernelGGL((trsm_rlt_kernel<T, Num<T>::KS64>), grid, dim3(256), 0, st, U, ldu, L, ldl, Winv, q0, q1,
                       (T*)nullptr, (int64_t)0, 1, W, ldw, phase);
  else
    hipLaunchKernelGGL((trsm_rlt_kernel<T, 1>), grid, dim3(256), 0, st, U, ldu, L, ldl, Winv, q0, q1,
                       (T*)nullptr, (int64_t)0, 1, W, ldw, phase);
}

template <typename T>
void launch_trsm_rln(T* X, int64_t ldx, int64_t rows, const T* L, int64_t ldl, const T* Winv, int nb,
                     hipStream_t st) {
  debug_delay(st);
  hipLaunchKernelGGL(trsm_rln_kernel<T>, dim3((unsigned)(rows / 64)), dim3(256), 0, st, X, ldx, L, ldl,
                     Winv, nb / 64);
}

// splits for a skinny product (see gemm_nt_splitk_kernel): a function of the contraction length
// ONLY — the summation order of an output element must not depend on how many columns happen to be
// computed beside it (predicting query points in batches is bit-identical to one pass) — splits of
// at least 1024, at most 16 of them (N = 65536, M = 4096: 64 tiles x 16 = 1024 workgroups)
int splitk_splits(int64_t k) { return (int)std::max<int64_t>(1, std::min<int64_t>(16, k / 1024)); }

template <typename T>
void launch_gemm_nt_splitk(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                           int64_t n, int64_t k, int S, T* part, int64_t ldp, hipStream_t st) {
  debug_delay(st);
  const int64_t Ks = ((k + S - 1) / S + 127) / 128 * 128;
  const int Sx = (int)((k + Ks - 1) / Ks);  // splits that are not empty
  const int64_t pstride = m * ldp;          // part: [S][m][ldp]
  hipLaunchKernelGGL(gemm_nt_splitk_kernel<T>, dim3((unsigned)((m / 64) * (n / 64)), (unsigned)Sx), dim3(256), 0, st, part,
                     ldp, pstride, A, lda, B, ldb, (int)(n / 64), (int)k, (int)Ks);
  const int64_t bx = (n + 255) / 256;
  hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)std::min<int64_t>(bx, 64), (unsigned)m), dim3(256), 0, st, C,
                     ldc, part, ldp, pstride, Sx, n);
}

template <typename T>
void launch_gemm_nt_fixed(int tile, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
                          int64_t m, int64_t n, int64_t k, int lower, int mode, hipStream_t st) {
  debug_delay(st);
  if (m <= 0 || n <= 0) return;
  const BcMask bc{0, 1, 0};
  if (tile == 128)
    launch_gemm_nt_t<T, 128>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, bc, st);
  else
    launch_gemm_nt_t<T, 64>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, bc, st);
}

// The library's own products: `tile` is the largest tile the shape allows; a 128-tile grid that
// does not even fill the 512 workgroup slots once (strips and trailing updates of the last panels,
// everything at N <= 8192) leaves CUs idle or half occupied, and the same product in 64-tiles has
// four times the workgroups.  N = 8192: Cholesky 10.7 -> 10.35 ms, and 10.05 ms with the products
// against the block inverses (lower == 4) included, variance solve 5.45 -> 5.15 ms; no effect at
// N = 65536 (measured: switching at 320 / 448 / 512 / 768 live tiles all within noise).  With a
// triangular mask the 64-tile grid leaves the upper 64-blocks of the diagonal 128-tiles untouched:
// nothing reads them (only the lower triangle of the Gram matrix is ever read).
int gemm_nt_tile(int tile, int64_t m, int64_t n, int lower) {
  constexpr int64_t thr = 448;
  if (tile != 128) return tile;
  const int64_t tm = m / 128, tn = n / 128;
  const int64_t live = lower == 1   ? tm * (tm + 1) / 2
                       : lower == 4 ? tm * ((tn + 1) / 2)
                       : lower == 2 ? tm * tn - std::min(tm, tn) * (std::min(tm, tn) - 1) / 2 - (tn > tm ? (tn - tm) * tm : 0)
                                    : tm * tn;
  return live <= thr ? 64 : 128;
}

template <typename T>
void launch_gemm_nt(int tile, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb,
                    int64_t m, int64_t n, int64_t k, int lower, int mode, hipStream_t st) {
  launch_gemm_nt_fixed<T>(gemm_nt_tile(tile, m, n, lower), C, ldc, A, lda, B, ldb, m, n, k, lower, mode, st);
}

// The fused trailing update (gemm_nt_fused_kernel): C (n x n) -= P P^T, lower part, strip of `ns`
// columns first; returns the number of strip slots (each adds 1 to *ctr when its tile is released).
template <typename T>
unsigned launch_gemm_nt_fused(T* C, int64_t ldc, const T* P, int64_t ldp, int64_t n, int64_t ns, int64_t k,
                              unsigned* ctr, hipStream_t st) {
  debug_delay(st);
  const int64_t tm = n / 128, tsn = ns / 128, tr = tm - tsn;
  int sh;
  const int64_t S = rect_grid(tm, tsn, sh);
  const int64_t ts = (tr + 7) / 8;
  const int64_t R = ts * (ts - 1) / 2 * 64 + ts * 36;
  hipLaunchKernelGGL((gemm_nt_fused_kernel<T, 128>), dim3((unsigned)(S + R)), dim3(256), 0, st, C, ldc, P, ldp, (int)tm,
                     (int)tsn, (int)S, sh, (int)k, ctr);
  return (unsigned)S;
}

template <typename T>
void launch_gemm_nt_bc(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                       int64_t n, int64_t k, const BcMask& bc, hipStream_t st) {
  debug_delay(st);
  if (m <= 0 || n <= 0) return;
  const int64_t tm = m / 128, tn = n / 128;
  const int64_t nsr = (tm + 7) / 8;
  if (nsr > STAIR_MAX) {  // beyond the by-value map: bounding rectangle with the mask
    launch_gemm_nt_t<T, 128>(C, ldc, A, lda, B, ldb, m, n, k, 3, 0, bc, st);
    return;
  }
  StairMap map;
  const unsigned total = build_stair_map(bc, tm, tn, map);
  hipLaunchKernelGGL((gemm_nt_stair_kernel<T, 128, 0>), dim3(total), dim3(256), 0, st, C, ldc, A, lda,
                     B, ldb, (int)tm, (int)tn, bc, (int)k, map);
}

template <typename T>
void launch_gemm_nn(T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t m,
                    int64_t n, int64_t k, hipStream_t st) {
  debug_delay(st);
  if (m <= 0 || n <= 0) return;
  const int64_t tm = m / 64, tn = n / 64;
  int sh;
  const int64_t nblk = rect_grid(tm, tn, sh);
  hipLaunchKernelGGL(gemm_nn_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, st, C, ldc, A, lda, B, ldb,
                     (int)tm, (int)tn, sh, (int)k);
}

// Host-side replay of the two hole-free tile maps, exactly as the kernels index them
// (xcd_chunk_id over the launched grid, then tile_coords<TRI> / stair_coords).  kind 0: lower
// triangle of a tm x tm tile grid; kind 1: block-cyclic staircase (bc) of a tm x tn grid.
// Writes (ti, tj) pairs of the valid slots; returns their number, or -1 if cap is too small
// or the staircase needs more than STAIR_MAX super-rows.
int64_t debug_tile_map(int kind, int64_t tm, int64_t tn, const BcMask& bc, int32_t* out,
                       int64_t cap) {
  int64_t n = 0;
  if (kind == 0) {
    const int64_t ts = (tm + 7) / 8;
    const int64_t grid = ts * (ts - 1) / 2 * 64 + ts * 36;
    for (int64_t b = 0; b < grid; ++b) {
      int ti, tj;
      if (!tile_coords<true>(xcd_chunk_id(b, grid), (int)tm, (int)tm, 8, 0, BcMask{0, 1, 0}, ti, tj)) continue;
      if (n >= cap) return -1;
      out[2 * n] = ti;
      out[2 * n + 1] = tj;
      ++n;
    }
  } else if (kind == 2) {  // fused trailing update: tm tile rows, strip of tn tile columns first
    if (tn <= 0 || tn >= tm) return -1;
    int sh;
    const int64_t S = rect_grid(tm, tn, sh);
    const int64_t tr = tm - tn, ts = (tr + 7) / 8;
    const int64_t grid = S + ts * (ts - 1) / 2 * 64 + ts * 36;
    for (int64_t b = 0; b < grid; ++b) {
      int ti, tj;
      if (!fused_coords(b, grid, (int)S, (int)tm, (int)tn, sh, ti, tj)) continue;
      if (n >= cap) return -1;
      out[2 * n] = ti;
      out[2 * n + 1] = tj;
      ++n;
    }
  } else {
    if ((tm + 7) / 8 > STAIR_MAX) return -1;
    StairMap map;
    const unsigned grid = build_stair_map(bc, tm, tn, map);
    for (unsigned b = 0; b < grid; ++b) {
      int ti, tj;
      stair_coords(map, bc, (unsigned)xcd_chunk_id(b, grid), (int)tm, (int)tn, ti, tj);
      if (ti >= tm || tj >= tn) continue;
      if (n >= cap) return -1;
      out[2 * n] = ti;
      out[2 * n + 1] = tj;
      ++n;
    }
  }
  return n;
}

#ifdef GPX_STAMPS
extern "C" int gpx_debug_read_stamps(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gpx_stamp_buf), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
extern "C" int gpx_debug_read_syrk_clock(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gpx_syrk_clock_buf), 64) == hipSuccess ? 0 : -2;
}
#endif

#define GPX_INSTANTIATE_BLAS(T)                                                                         \
  template void launch_potf2_64<T>(T*, int64_t, T*, int64_t, int*, hipStream_t, unsigned*, unsigned);   \
  template void launch_potf2_128<T>(T*, int64_t, T*, int64_t, int*, hipStream_t, unsigned*, unsigned);  \
  template void launch_trsm_rlt<T>(T*, int64_t, int64_t, const T*, int64_t, const T*, int, T*, int64_t, \
                                   hipStream_t);                                                        \
  template void launch_trsm_rln<T>(T*, int64_t, int64_t, const T*, int64_t, const T*, int, hipStream_t); \
  template void launch_inv_extend<T>(T*, int64_t, const T*, int64_t, const T*, int, int, T*, int64_t,   \
                                     hipStream_t, int);                                                 \
  template void launch_gemm_nt<T>(int, T*, int64_t, const T*, int64_t, const T*, int64_t, int64_t,      \
                                  int64_t, int64_t, int, int, hipStream_t);                             \
  template void launch_gemm_nt_fixed<T>(int, T*, int64_t, const T*, int64_t, const T*, int64_t, int64_t, \
                                        int64_t, int64_t, int, int, hipStream_t);                       \
  template unsigned launch_gemm_nt_fused<T>(T*, int64_t, const T*, int64_t, int64_t, int64_t, int64_t,  \
                                            unsigned*, hipStream_t);                                    \
  template void launch_gemm_nt_bc<T>(T*, int64_t, const T*, int64_t, const T*, int64_t, int64_t,        \
                                     int64_t, int64_t, const BcMask&, hipStream_t);                     \
  template void launch_gemm_nn<T>(T*, int64_t, const T*, int64_t, const T*, int64_t, int64_t, int64_t,  \
                                  int64_t, hipStream_t);                                                \
  template void launch_gemm_nt_splitk<T>(T*, int64_t, const T*, int64_t, const T*, int64_t, int64_t,    \
                                         int64_t, int64_t, int, T*, int64_t, hipStream_t);
GPX_INSTANTIATE_BLAS(double)
GPX_INSTANTIATE_BLAS(float)

}  // namespace gpx
