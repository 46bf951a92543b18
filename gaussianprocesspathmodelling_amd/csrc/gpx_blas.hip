// gpx_blas.hip — fp64 dense building blocks of the blocked Cholesky / triangular
// solves (SURVEY.md §8 rows a3 "K2", a4 "K3", a5 "K3'").  Absent in the reference
// (its only linalg call is np.linalg.norm, GPmap.py:120); algorithm = right-looking
// blocked Cholesky (R&W Alg. 2.1 line 2) as restated in oracle/gp_oracle.py.
//
// Everything is expressed with ONE MFMA tile engine:
//   gemm_tile<BM,BN,BKN>:  acc(BM x BN) += A(BM x K) * op(B)
//     A row-major, k contiguous;  B either [n][k] (BKN=false, "NT") or [k][n] ("NN").
//   4 waves (2x2) per 256-thread workgroup, each wave owns (BM/2)x(BN/2) as
//   16x16 tiles of v_mfma_f64_16x16x4_f64 (4 f64 accumulators per lane per tile).
//   K is walked in steps of 16 doubles = one 128-byte line per row, staged
//   global -> registers -> LDS with two LDS buffers (loads for step t+1 are issued
//   before the MFMAs of step t and written to LDS after them: one barrier per step).
//   LDS rows are padded by 16 B ([rows][16+2] doubles, stride 144 B) so that the
//   ds_read_b64 fragment reads (lane -> row l&15, k l>>4) hit 64 distinct banks per
//   32-lane half; the [k][n] image pads rows by 128 B for the same reason.
//
// fp64 MFMA layouts (cdna_hip_programming.md §3): A lane l = A[l&15][l>>4],
// B lane l = B[l>>4][l&15], D reg r of lane l = D[(l>>4) + 4r][l&15].
#include <cstdlib>

#include "gpx_internal.h"

namespace gpx {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BK = 16;          // k-step (doubles) = one 128-B line per row
constexpr int LDS_K = BK + 2;   // padded row (doubles) of a [row][k] LDS image

template <int BM, int BN, bool BKN>
struct TileShape {
  static constexpr int MT = BM / 32, NT = BN / 32;   // 16x16 tiles per wave
  static constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
  static constexpr int A_STAGE = BM * LDS_K;
  static constexpr int LDS_BN = BN + 16;             // padded row of the [k][n] image
  static constexpr int B_STAGE = BKN ? BK * LDS_BN : BN * LDS_K;
  static constexpr int SMEM_DOUBLES = 2 * (A_STAGE + B_STAGE);
};

// acc += A * op(B) for one BM x BN tile; all 256 threads participate.
// A -> first row of the tile (BM rows, lda);  NT: B -> first row of the BN rows (ldb);
// NN: B -> &B[0][n0] (K rows, ldb).  K multiple of 16, all pointers 16-B aligned.
template <int BM, int BN, bool BKN>
__device__ __forceinline__ void gemm_tile(const double* A, int64_t lda, const double* B,
                                          int64_t ldb, int K,
                                          v4d (&acc)[BM / 32][BN / 32], double* smem) {
  using S = TileShape<BM, BN, BKN>;
  constexpr int CA = BM / 32;  // 16-B chunks per thread per k-step (A)
  constexpr int CB = BN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  double* As = smem;
  double* Bs = smem + 2 * S::A_STAGE;

  double2 ra[CA], rb[CB];
  // global addresses of this thread's chunks
  const double* ga[CA];
  const double* gb[CB];
  int sa[CA], sb[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * 256;
    const int row = c >> 3, cc = c & 7;
    ga[i] = A + (int64_t)row * lda + cc * 2;
    sa[i] = row * LDS_K + cc * 2;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int c = tid + i * 256;
    if (BKN) {
      const int kr = c / (BN / 2), cc = c % (BN / 2);
      gb[i] = B + (int64_t)kr * ldb + cc * 2;
      sb[i] = kr * S::LDS_BN + cc * 2;
    } else {
      const int row = c >> 3, cc = c & 7;
      gb[i] = B + (int64_t)row * ldb + cc * 2;
      sb[i] = row * LDS_K + cc * 2;
    }
  }
  const int64_t bstep = BKN ? (int64_t)BK * ldb : BK;

  // fragment read offsets
  const int a_off = (wr * S::WM + (lane & 15)) * LDS_K + (lane >> 4);
  const int b_off = BKN ? ((lane >> 4) * S::LDS_BN + wc * S::WN + (lane & 15))
                        : ((wc * S::WN + (lane & 15)) * LDS_K + (lane >> 4));

#pragma unroll
  for (int i = 0; i < CA; ++i) ra[i] = *reinterpret_cast<const double2*>(ga[i]);
#pragma unroll
  for (int i = 0; i < CB; ++i) rb[i] = *reinterpret_cast<const double2*>(gb[i]);
#pragma unroll
  for (int i = 0; i < CA; ++i) *reinterpret_cast<double2*>(As + sa[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < CB; ++i) *reinterpret_cast<double2*>(Bs + sb[i]) = rb[i];
  __syncthreads();

  const int KT = K / BK;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    // Prefetch the next k-step into registers (unconditional: the last step re-reads
    // its own lines, which keeps the staging registers out of scratch).
    const int kn = (kt + 1 < KT) ? kt + 1 : kt;
#pragma unroll
    for (int i = 0; i < CA; ++i)
      ra[i] = *reinterpret_cast<const double2*>(ga[i] + (int64_t)kn * BK);
#pragma unroll
    for (int i = 0; i < CB; ++i)
      rb[i] = *reinterpret_cast<const double2*>(gb[i] + (int64_t)kn * bstep);
    const double* Ab = As + buf * S::A_STAGE + a_off;
    const double* Bb = Bs + buf * S::B_STAGE + b_off;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double a[S::MT], b[S::NT];
#pragma unroll
      for (int m = 0; m < S::MT; ++m) a[m] = Ab[m * 16 * LDS_K + ks * 4];
#pragma unroll
      for (int n = 0; n < S::NT; ++n)
        b[n] = BKN ? Bb[ks * 4 * S::LDS_BN + n * 16] : Bb[n * 16 * LDS_K + ks * 4];
#pragma unroll
      for (int m = 0; m < S::MT; ++m)
#pragma unroll
        for (int n = 0; n < S::NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    double* Aw = As + (buf ^ 1) * S::A_STAGE;
    double* Bw = Bs + (buf ^ 1) * S::B_STAGE;
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<double2*>(Aw + sa[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<double2*>(Bw + sb[i]) = rb[i];
    __syncthreads();
  }
}

// ---- NT tile engine with LDS-DMA staging ------------------------------------------------
// acc += A(BM x K) * B(BN x K)^T, both row-major with k contiguous.  Per 16-double
// k-step every wave issues BM/32 + BN/32 `global_load_lds_dwordx4` (1 KiB = 8 rows x
// 128 B each, straight into LDS: no staging VGPRs, no ds_write) for step t+1 before the
// MFMAs of step t; `__syncthreads()` drains them (vmcnt(0)) once per step.
// LDS image: [row][8 slots of 16 B], physical slot = logical slot ^ swz(row).  LDS-DMA
// writes lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
// fragment reads (cdna_hip_programming.md rule 21).  swz() is chosen so that a
// ds_read_b128 lane group (rows {0-3,12-15} at k-group g with rows {4-11} at g+1, and the
// three analogous groups) hits 16 distinct 16-byte bank slots: conflict-free.
// k permutation: lane group g = l>>4 consumes k = 4g+s at MFMA step s (instead of 4s+g) —
// the same for A and B, so one ds_read_b128 pair per fragment feeds all four steps.
__device__ __forceinline__ int swz(int row) {
  const int t = ((row >> 1) + 2) & 7;
  return ((t & 3) << 1) | (t >> 2);
}

template <int BM, int BN>
struct TileShapeG {
  static constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK;
  static constexpr int SMEM_DOUBLES = 2 * (A_STAGE + B_STAGE);
};

typedef double v2d __attribute__((ext_vector_type(2)));

#define GPX_GLDS16(gptr, lptr)                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),    \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int BM, int BN>
__device__ __forceinline__ void gemm_tile_g(const double* A, int64_t lda, const double* B,
                                            int64_t ldb, int K, v4d (&acc)[BM / 32][BN / 32],
                                            double* smem) {
  using S = TileShapeG<BM, BN>;
  constexpr int MT = BM / 32, NT = BN / 32, WM = BM / 2, WN = BN / 2;
  constexpr int IA = BM / 32, IB = BN / 32;  // DMA instructions per wave per k-step
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  double* As = smem;
  double* Bs = smem + 2 * S::A_STAGE;

  // DMA: instruction q of this wave covers rows 8*(wave*I + q) .. +7; lane -> (row, slot)
  const int drow = lane >> 3, dslot = lane & 7;
  const double* ga[IA];
  const double* gb[IB];
#pragma unroll
  for (int q = 0; q < IA; ++q) {
    const int row = (wave * IA + q) * 8 + drow;
    ga[q] = A + (int64_t)row * lda + ((dslot ^ swz(row)) << 1);
  }
#pragma unroll
  for (int q = 0; q < IB; ++q) {
    const int row = (wave * IB + q) * 8 + drow;
    gb[q] = B + (int64_t)row * ldb + ((dslot ^ swz(row)) << 1);
  }
  double* const la = As + wave * IA * 128;  // wave-uniform LDS destinations (doubles)
  double* const lb = Bs + wave * IB * 128;

  // fragment reads: row = w*W + t*16 + l15, logical slots 2*l4 and 2*l4+1
  const int sw = swz(l15);
  const int a_off0 = (wr * WM + l15) * BK + (((2 * l4) ^ sw) << 1);
  const int a_off1 = (wr * WM + l15) * BK + (((2 * l4 + 1) ^ sw) << 1);
  const int b_off0 = (wc * WN + l15) * BK + (((2 * l4) ^ sw) << 1);
  const int b_off1 = (wc * WN + l15) * BK + (((2 * l4 + 1) ^ sw) << 1);

#pragma unroll
  for (int q = 0; q < IA; ++q) GPX_GLDS16(ga[q], la + q * 128);
#pragma unroll
  for (int q = 0; q < IB; ++q) GPX_GLDS16(gb[q], lb + q * 128);
  __syncthreads();

  const int KT = K / BK;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) {
      const int64_t ko = (int64_t)(kt + 1) * BK;
#pragma unroll
      for (int q = 0; q < IA; ++q) GPX_GLDS16(ga[q] + ko, la + (buf ^ 1) * S::A_STAGE + q * 128);
#pragma unroll
      for (int q = 0; q < IB; ++q) GPX_GLDS16(gb[q] + ko, lb + (buf ^ 1) * S::B_STAGE + q * 128);
    }
    const double* Ab = As + buf * S::A_STAGE;
    const double* Bb = Bs + buf * S::B_STAGE;
    v2d a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a0[m] = *reinterpret_cast<const v2d*>(Ab + a_off0 + m * 16 * BK);
#pragma unroll
    for (int n = 0; n < NT; ++n) b0[n] = *reinterpret_cast<const v2d*>(Bb + b_off0 + n * 16 * BK);
#pragma unroll
    for (int m = 0; m < MT; ++m) a1[m] = *reinterpret_cast<const v2d*>(Ab + a_off1 + m * 16 * BK);
#pragma unroll
    for (int n = 0; n < NT; ++n) b1[n] = *reinterpret_cast<const v2d*>(Bb + b_off1 + n * 16 * BK);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[m].x, b0[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[m].y, b0[n].y, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m].x, b1[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[m].y, b1[n].y, acc[m][n], 0, 0, 0);
    // keep the MFMAs ABOVE the barrier: hipcc otherwise sinks them below the vmcnt(0)
    // drain of __syncthreads() and the DMA latency is exposed on every k-step
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(v4d (&acc)[MT][NT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// MODE 0: C -= acc;  MODE 1: C = acc;  C -> tile origin.
// MODE 0 loads one 16-row strip of C (NT*4 values per lane) before storing it, so the
// loads of a strip are in flight together instead of one round trip per element.
template <int BM, int BN, int MODE>
__device__ __forceinline__ void store_tile(double* C, int64_t ldc,
                                           const v4d (&acc)[BM / 32][BN / 32]) {
  constexpr int NT = BN / 32;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  double* Cw = C + (int64_t)(wr * (BM / 2) + (lane >> 4)) * ldc + wc * (BN / 2) + (lane & 15);
#pragma unroll
  for (int m = 0; m < BM / 32; ++m) {
    double* Cm = Cw + (int64_t)(m * 16) * ldc;
    if (MODE == 0) {
      double c[NT][4];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[n][r] = Cm[(int64_t)(4 * r) * ldc + n * 16];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cm[(int64_t)(4 * r) * ldc + n * 16] = c[n][r] - acc[m][n][r];
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cm[(int64_t)(4 * r) * ldc + n * 16] = acc[m][n][r];
    }
  }
}

// blocks are dealt round-robin over the 8 XCDs; give each XCD one contiguous chunk of
// the logical tile order so that neighbouring tiles share an L2 (speed only).
__device__ __forceinline__ int64_t xcd_chunk_id(int64_t bid, int64_t nblk) {
  const int64_t q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ void tri_coords(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > t) --i;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

// Logical tile order: 8x8 super-tiles (64 tiles = what one XCD runs concurrently at 2
// workgroups per CU), so the tiles in flight on an XCD share 8 A row-slices and 8 B
// row-slices through its L2 instead of re-fetching the panel per tile.
//   TRI : super-tiles enumerate the lower triangle of the super-tile grid (m == n);
//         slots above the diagonal exit at once (<= 3 % of the grid at T >= 64).
//   !TRI: rectangular super-tile grid (sh x 64/sh tiles each); mask_lower != 0 also
//         drops tiles with tj > ti (look-ahead strip of the SYRK).
struct BcMask {   // block-cyclic row map of the sharded trailing update (P == 0: unused)
  int P, tpb, c;  // global row tile (relative to the trailing start) of local row tile ti:
                  //   ((ti / tpb) * P + c) * tpb + ti % tpb
};

template <bool TRI>
__device__ __forceinline__ bool tile_coords(int64_t lin, int tiles_m, int tiles_n, int sh,
                                            int mask_lower, const BcMask& bc, int& ti, int& tj) {
  const int64_t st = lin >> 6;
  const int inner = (int)(lin & 63);
  if (TRI) {
    int sr, sc;
    tri_coords(st, sr, sc);
    ti = sr * 8 + (inner >> 3);
    tj = sc * 8 + (inner & 7);
    return ti < tiles_m && tj <= ti;
  } else {
    const int sw = 64 / sh;                       // sh in {1, 8}
    const int sn = (tiles_n + sw - 1) / sw;
    const int sr = (int)(st / sn), sc = (int)(st - (int64_t)sr * sn);
    ti = sr * sh + inner / sw;
    tj = sc * sw + inner % sw;
    if (ti >= tiles_m || tj >= tiles_n) return false;
    if (mask_lower == 1) return tj <= ti;
    if (mask_lower == 2) return tj <= ((ti / bc.tpb) * bc.P + bc.c) * bc.tpb + ti % bc.tpb;
    return true;
  }
}

// ---- C op= A * B^T --------------------------------------------------------------
template <int BT, bool TRI, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, int tiles_m, int tiles_n, int sh, int mask_lower,
    BcMask bc, int K) {
  __shared__ __attribute__((aligned(16))) double smem[TileShapeG<BT, BT>::SMEM_DOUBLES];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  int ti, tj;
  if (!tile_coords<TRI>(lin, tiles_m, tiles_n, sh, mask_lower, bc, ti, tj)) return;
  v4d acc[BT / 32][BT / 32];
  zero_acc(acc);
  gemm_tile_g<BT, BT>(A + (int64_t)ti * BT * lda, lda, B + (int64_t)tj * BT * ldb, ldb, K,
                           acc, smem);
  store_tile<BT, BT, MODE>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
}

// ---- C -= A * B, B stored [k][n]; 64x64 tiles --------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(double* __restrict__ C, int64_t ldc,
                                                         const double* __restrict__ A, int64_t lda,
                                                         const double* __restrict__ B, int64_t ldb,
                                                         int tiles_m, int tiles_n, int sh, int K) {
  __shared__ __attribute__((aligned(16))) double smem[TileShape<64, 64, true>::SMEM_DOUBLES];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  int ti, tj;
  if (!tile_coords<false>(lin, tiles_m, tiles_n, sh, 0, BcMask{0, 1, 0}, ti, tj)) return;
  v4d acc[2][2];
  zero_acc(acc);
  gemm_tile<64, 64, true>(A + (int64_t)ti * 64 * lda, lda, B + (int64_t)tj * 64, ldb, K, acc,
                          smem);
  store_tile<64, 64, 0>(C + (int64_t)ti * 64 * ldc + (int64_t)tj * 64, ldc, acc);
}

// ---- POTF2 of one 64x64 block + explicit inverse -------------------------------------
// One workgroup, everything in LDS.  Factorisation in 16 steps of 4 columns:
//   phase A  every thread factors the 4x4 diagonal block redundantly in registers
//            (rsqrt-based, no divisions); thread i < 64 solves its row of the 4-column
//            panel and drops it in PB[64][4];
//   phase B  the rank-4 trailing update C -= PB PB^T is ONE v_mfma_f64_16x16x4_f64 per
//            16x16 tile (<= 10 lower tiles over 4 waves); columns already final are
//            masked through a zero B operand and predicated stores.
// Inverse W = L^-1 by recursive blocking: four 16x16 diagonal inverses (one column per
// lane, reciprocal pivots reused from the factorisation), then W21 = -W22 (L21 W11) at
// 32 and at 64 with MFMA products through a small LDS scratch tile.
// LDS: 2 x 64x66 + scratch = 77 KB — fits the slot of one retiring SYRK workgroup (73.7
// KB + 16 KB spare per CU), which lets the look-ahead stream run beside the trailing
// update; s_setprio(3) keeps its waves ahead of the co-resident SYRK waves.
// A non-positive / NaN pivot records (global index + 1) in *info by atomicMin and lets
// NaN propagate (LAPACK potrf info convention).
constexpr int PLD = 66;
constexpr int TLD = 34;

__device__ __forceinline__ v4d mfma0(double a, double b) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, (v4d){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
}

__global__ __launch_bounds__(256) void potf2_64_kernel(double* __restrict__ A, int64_t lda,
                                                       double* __restrict__ Winv, int64_t gidx0,
                                                       int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double Wk[64 * PLD];  // working matrix -> L (lower)
  __shared__ __attribute__((aligned(16))) double Wi[64 * PLD];  // inverse
  __shared__ __attribute__((aligned(16))) double PB[64 * 4];    // current 4-column panel
  __shared__ __attribute__((aligned(16))) double Tm[32 * TLD];  // product scratch
  __shared__ double Rinv[64];                                    // 1 / L[i][i]
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int e = tid; e < 4096; e += 256) {
    const int i = e >> 6, k = e & 63;
    Wk[i * PLD + k] = (k <= i) ? A[(int64_t)i * lda + k] : 0.0;
    Wi[i * PLD + k] = 0.0;
  }
  __syncthreads();

  for (int j = 0; j < 64; j += 4) {
    // ---- phase A: 4x4 diagonal factor (redundant per thread) + this thread's panel row
    const double* D = Wk + j * PLD + j;
    const double a00 = D[0];
    const double a10 = D[PLD], a11 = D[PLD + 1];
    const double a20 = D[2 * PLD], a21 = D[2 * PLD + 1], a22 = D[2 * PLD + 2];
    const double a30 = D[3 * PLD], a31 = D[3 * PLD + 1], a32 = D[3 * PLD + 2], a33 = D[3 * PLD + 3];
    const double rs0 = rsqrt(a00);
    const double l10 = a10 * rs0, l20 = a20 * rs0, l30 = a30 * rs0;
    const double b11 = a11 - l10 * l10;
    const double rs1 = rsqrt(b11);
    const double l21 = (a21 - l20 * l10) * rs1, l31 = (a31 - l30 * l10) * rs1;
    const double b22 = a22 - l20 * l20 - l21 * l21;
    const double rs2 = rsqrt(b22);
    const double l32 = (a32 - l30 * l20 - l31 * l21) * rs2;
    const double b33 = a33 - l30 * l30 - l31 * l31 - l32 * l32;
    const double rs3 = rsqrt(b33);
    if (tid == 0) {
      const int bad = !(a00 > 0.0) ? 1 : !(b11 > 0.0) ? 2 : !(b22 > 0.0) ? 3 : !(b33 > 0.0) ? 4 : 0;
      if (bad) atomicMin(info, (int)(gidx0 + j + bad));
      Rinv[j] = rs0;
      Rinv[j + 1] = rs1;
      Rinv[j + 2] = rs2;
      Rinv[j + 3] = rs3;
    }
    if (tid < 64) {
      const int i = tid;
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
      if (i >= j) {
        const double* row = Wk + i * PLD + j;
        x0 = row[0] * rs0;
        x1 = (row[1] - x0 * l10) * rs1;
        x2 = (row[2] - x0 * l20 - x1 * l21) * rs2;
        x3 = (row[3] - x0 * l30 - x1 * l31 - x2 * l32) * rs3;
        const int c = i - j;  // rows of the diagonal block: strictly-upper part is zero
        if (c < 1) x1 = 0.0;
        if (c < 2) x2 = 0.0;
        if (c < 3) x3 = 0.0;
      }
      PB[i * 4 + 0] = x0;
      PB[i * 4 + 1] = x1;
      PB[i * 4 + 2] = x2;
      PB[i * 4 + 3] = x3;
    }
    __syncthreads();
    // ---- phase B: commit the panel, rank-4 update of the trailing lower tiles
    if (tid < 64 && tid >= j) {
      double* row = Wk + tid * PLD + j;
      row[0] = PB[tid * 4 + 0];
      row[1] = PB[tid * 4 + 1];
      row[2] = PB[tid * 4 + 2];
      row[3] = PB[tid * 4 + 3];
    }
    const int jn = j + 4;
    const int t0 = jn >> 4;
    int idx = 0;
    for (int tr = t0; tr < 4; ++tr)
      for (int tc = t0; tc <= tr; ++tc, ++idx) {
        if ((idx & 3) != wave) continue;
        const int colg = tc * 16 + l15;
        const double a = PB[(tr * 16 + l15) * 4 + l4];
        const double b = (colg >= jn) ? PB[colg * 4 + l4] : 0.0;
        const v4d acc = mfma0(a, b);
        if (colg >= jn) {
          double* Cp = Wk + (tr * 16 + l4) * PLD + colg;
#pragma unroll
          for (int r = 0; r < 4; ++r) Cp[4 * r * PLD] -= acc[r];
        }
      }
    __syncthreads();
  }

  // ---- inverse, level 0: wave w inverts diagonal block w; lane c < 16 owns column c
  {
    const int b0 = wave * 16;
    if (lane < 16) {
      double w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double sacc = (i == lane) ? 1.0 : 0.0;
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
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wk[(r1 + l15) * PLD + c0 + ks * 4 + l4],
                                                 Wi[(c0 + ks * 4 + l4) * PLD + c0 + l15], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Tm[(wave * 16 + l4 + 4 * r) * TLD + l15] = acc[r];
  }
  __syncthreads();
  if (wave < 2) {
    const int c0 = wave * 32, r1 = c0 + 16;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wi[(r1 + l15) * PLD + r1 + ks * 4 + l4],
                                                 Tm[(wave * 16 + ks * 4 + l4) * TLD + l15], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wi[(r1 + l4 + 4 * r) * PLD + c0 + l15] = -acc[r];
  }
  __syncthreads();
  // ---- level 2: W_BA = -W_BB (L_BA W_AA), 32x32 blocks, one 16x16 tile per wave
  {
    const int tr = wave >> 1, tc = wave & 1;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wk[(32 + tr * 16 + l15) * PLD + ks * 4 + l4],
                                                 Wi[(ks * 4 + l4) * PLD + tc * 16 + l15], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) Tm[(tr * 16 + l4 + 4 * r) * TLD + tc * 16 + l15] = acc[r];
    __syncthreads();
    v4d acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Wi[(32 + tr * 16 + l15) * PLD + 32 + ks * 4 + l4],
                                                  Tm[(ks * 4 + l4) * TLD + tc * 16 + l15], acc2, 0, 0, 0);
    __syncthreads();  // every read of W_BB / Tm done before W_BA lands next to them
#pragma unroll
    for (int r = 0; r < 4; ++r) Wi[(32 + tr * 16 + l4 + 4 * r) * PLD + tc * 16 + l15] = -acc2[r];
  }
  __syncthreads();
  for (int e = tid; e < 4096; e += 256) {
    const int ii = e >> 6, k = e & 63;
    if (k <= ii) A[(int64_t)ii * lda + k] = Wk[ii * PLD + k];
    Winv[e] = Wi[ii * PLD + k];
  }
}

// ---- X <- X * L^-T (right, lower, transposed): ascending 64-column blocks --------------
// One workgroup owns a 64-row slab of X and walks the nb/64 column blocks in order:
//   T    = X[:, jb] - sum_{kb<jb} X[:, kb] L[jb, kb]^T      (MFMA, K = 64*jb)
//   X_jb = T * Winv_jb^T                                       (MFMA, K = 64)
// T makes a round trip through the slab's own global tile (L2-resident) so both
// products run on the same tile engine; a slab is private to its workgroup, so the
// only ordering needed is the workgroup barrier.
__global__ __launch_bounds__(256, 2) void trsm_rlt_kernel(double* X, int64_t ldx, const double* L,
                                                          int64_t ldl, const double* Winv, int nbq,
                                                          double* P, int64_t ldp) {
  __shared__ __attribute__((aligned(16))) double smem[TileShapeG<64, 64>::SMEM_DOUBLES];
  __builtin_amdgcn_s_setprio(2);  // panel solve is on the critical path of the look-ahead
  double* Xs = X + (int64_t)blockIdx.x * 64 * ldx;
  double* Ps = P ? P + (int64_t)blockIdx.x * 64 * ldp : nullptr;
  v4d acc[2][2];
  for (int jb = 0; jb < nbq; ++jb) {
    double* Xj = Xs + jb * 64;
    if (jb > 0) {
      zero_acc(acc);
      gemm_tile_g<64, 64>(Xs, ldx, L + (int64_t)jb * 64 * ldl, ldl, jb * 64, acc, smem);
      store_tile<64, 64, 0>(Xj, ldx, acc);
      __syncthreads();
    }
    zero_acc(acc);
    gemm_tile_g<64, 64>(Xj, ldx, Winv + (int64_t)jb * 4096, 64, 64, acc, smem);
    // gemm_tile ends with a barrier: every read of T is complete
    store_tile<64, 64, 1>(Xj, ldx, acc);
    if (Ps) store_tile<64, 64, 1>(Ps + jb * 64, ldp, acc);
    __syncthreads();
  }
}

// ---- X <- X * L^-1 (right, lower, no transpose): descending blocks ----------------------
//   T   = X[:, q] - sum_{k>q} X[:, k] L[k, q]        (B operand is L stored [k][n])
//   X_q = T * Winv_q
__global__ __launch_bounds__(256, 2) void trsm_rln_kernel(double* X, int64_t ldx, const double* L,
                                                          int64_t ldl, const double* Winv, int nbq) {
  __shared__ __attribute__((aligned(16))) double smem[TileShape<64, 64, true>::SMEM_DOUBLES];
  double* Xs = X + (int64_t)blockIdx.x * 64 * ldx;
  v4d acc[2][2];
  for (int q = nbq - 1; q >= 0; --q) {
    double* Xq = Xs + q * 64;
    if (q < nbq - 1) {
      zero_acc(acc);
      gemm_tile<64, 64, true>(Xs + (q + 1) * 64, ldx, L + (int64_t)(q + 1) * 64 * ldl + q * 64, ldl,
                              (nbq - 1 - q) * 64, acc, smem);
      store_tile<64, 64, 0>(Xq, ldx, acc);
      __syncthreads();
    }
    zero_acc(acc);
    gemm_tile<64, 64, true>(Xq, ldx, Winv + (int64_t)q * 4096, 64, 64, acc, smem);
    store_tile<64, 64, 1>(Xq, ldx, acc);
    __syncthreads();
  }
}

}  // namespace

void launch_potf2_64(double* A, int64_t lda, double* Winv, int64_t gidx0, int* info,
                     hipStream_t st) {
  hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(256), 0, st, A, lda, Winv, gidx0, info);
}

void launch_trsm_rlt(double* X, int64_t ldx, int64_t rows, const double* L, int64_t ldl,
                     const double* Winv, int nb, double* P, int64_t ldp, hipStream_t st) {
  hipLaunchKernelGGL(trsm_rlt_kernel, dim3((unsigned)(rows / 64)), dim3(256), 0, st, X, ldx, L, ldl,
                     Winv, nb / 64, P, ldp);
}

void launch_trsm_rln(double* X, int64_t ldx, int64_t rows, const double* L, int64_t ldl,
                     const double* Winv, int nb, hipStream_t st) {
  hipLaunchKernelGGL(trsm_rln_kernel, dim3((unsigned)(rows / 64)), dim3(256), 0, st, X, ldx, L, ldl,
                     Winv, nb / 64);
}

static int64_t rect_grid(int64_t tm, int64_t tn, int& sh) {
  sh = tm >= 8 ? 8 : 1;
  const int sw = 64 / sh;
  return ((tm + sh - 1) / sh) * ((tn + sw - 1) / sw) * 64;
}

template <int BT>
static void launch_gemm_nt_t(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                             int64_t ldb, int64_t m, int64_t n, int64_t k, int lower, int mode,
                             BcMask bc, hipStream_t st) {
  const int64_t tm = m / BT, tn = n / BT;
  dim3 block(256);
  if (lower == 1) {  // full lower triangle, triangular super-tile enumeration
    const int64_t ts = (tm + 7) / 8;
    dim3 grid((unsigned)(ts * (ts + 1) / 2 * 64));
    if (mode == 0)
      hipLaunchKernelGGL((gemm_nt_kernel<BT, true, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, 8, 0, bc, (int)k);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<BT, true, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, 8, 0, bc, (int)k);
  } else {           // rectangle; lower == 2: masked to tj <= ti
    int sh;
    dim3 grid((unsigned)rect_grid(tm, tn, sh));
    const int mask = lower == 2 ? 1 : lower == 3 ? 2 : 0;
    if (mode == 0)
      hipLaunchKernelGGL((gemm_nt_kernel<BT, false, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, mask, bc, (int)k);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<BT, false, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)tn, sh, mask, bc, (int)k);
  }
}

void launch_gemm_nt(int tile, double* C, int64_t ldc, const double* A, int64_t lda,
                    const double* B, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower,
                    int mode, hipStream_t st) {
  if (m <= 0 || n <= 0) return;
  const BcMask bc{0, 1, 0};
  if (tile == 128)
    launch_gemm_nt_t<128>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, bc, st);
  else
    launch_gemm_nt_t<64>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, bc, st);
}

void launch_gemm_nt_bc(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                       int64_t ldb, int64_t m, int64_t n, int64_t k, int bc_P, int bc_tpb, int bc_c,
                       hipStream_t st) {
  if (m <= 0 || n <= 0) return;
  launch_gemm_nt_t<128>(C, ldc, A, lda, B, ldb, m, n, k, 3, 0, BcMask{bc_P, bc_tpb, bc_c}, st);
}

void launch_gemm_nn(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                    int64_t ldb, int64_t m, int64_t n, int64_t k, hipStream_t st) {
  if (m <= 0 || n <= 0) return;
  const int64_t tm = m / 64, tn = n / 64;
  int sh;
  const int64_t nblk = rect_grid(tm, tn, sh);
  hipLaunchKernelGGL(gemm_nn_kernel, dim3((unsigned)nblk), dim3(256), 0, st, C, ldc, A, lda, B,
                     ldb, (int)tm, (int)tn, sh, (int)k);
}

}  // namespace gpx
