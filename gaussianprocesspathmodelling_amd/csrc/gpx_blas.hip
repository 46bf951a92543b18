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

// ---- C op= A * B^T --------------------------------------------------------------
template <int BT, bool LOWER, int MODE>
__global__ __launch_bounds__(256, (BT == 128 ? 2 : 2)) void gemm_nt_kernel(
    double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, int tiles_m, int K) {
  __shared__ __attribute__((aligned(16))) double smem[TileShape<BT, BT, false>::SMEM_DOUBLES];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  int ti, tj;
  if (LOWER) {
    tri_coords(lin, ti, tj);
  } else {
    // consecutive ids walk down a column of tiles: they share the B rows
    tj = (int)(lin / tiles_m);
    ti = (int)(lin - (int64_t)tj * tiles_m);
  }
  v4d acc[BT / 32][BT / 32];
  zero_acc(acc);
  gemm_tile<BT, BT, false>(A + (int64_t)ti * BT * lda, lda, B + (int64_t)tj * BT * ldb, ldb, K,
                           acc, smem);
  store_tile<BT, BT, MODE>(C + (int64_t)ti * BT * ldc + (int64_t)tj * BT, ldc, acc);
}

// ---- C -= A * B, B stored [k][n]; 64x64 tiles --------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(double* __restrict__ C, int64_t ldc,
                                                         const double* __restrict__ A, int64_t lda,
                                                         const double* __restrict__ B, int64_t ldb,
                                                         int tiles_m, int K) {
  __shared__ __attribute__((aligned(16))) double smem[TileShape<64, 64, true>::SMEM_DOUBLES];
  const int64_t lin = xcd_chunk_id(blockIdx.x, gridDim.x);
  const int tj = (int)(lin / tiles_m);
  const int ti = (int)(lin - (int64_t)tj * tiles_m);
  v4d acc[2][2];
  zero_acc(acc);
  gemm_tile<64, 64, true>(A + (int64_t)ti * 64 * lda, lda, B + (int64_t)tj * 64, ldb, K, acc,
                          smem);
  store_tile<64, 64, 0>(C + (int64_t)ti * 64 * ldc + (int64_t)tj * 64, ldc, acc);
}

// ---- POTF2 of one 64x64 block + explicit inverse -------------------------------------
// Right-looking column Cholesky in LDS (one barrier per column), then the inverse of
// the factor row by row (W L = I), both fp64.  A non-positive / NaN pivot records
// (global index + 1) in *info by atomicMin and lets NaN propagate (LAPACK potrf info).
constexpr int PLD = 65;
__global__ __launch_bounds__(256) void potf2_64_kernel(double* __restrict__ A, int64_t lda,
                                                       double* __restrict__ Winv, int64_t gidx0,
                                                       int* __restrict__ info) {
  __shared__ double Wk[64 * PLD];
  __shared__ double Lo[64 * PLD];
  __shared__ double Wi[64 * PLD];
  const int tid = threadIdx.x;
  for (int e = tid; e < 4096; e += 256) {
    const int i = e >> 6, k = e & 63;
    Wk[i * PLD + k] = (k <= i) ? A[(int64_t)i * lda + k] : 0.0;
    Lo[i * PLD + k] = 0.0;
    Wi[i * PLD + k] = 0.0;
  }
  __syncthreads();
  const int i = tid >> 2, part = tid & 3;
  for (int j = 0; j < 64; ++j) {
    const double ajj = Wk[j * PLD + j];
    if (!(ajj > 0.0) && tid == 0) atomicMin(info, (int)(gidx0 + j + 1));
    const double dj = sqrt(ajj);
    const double inv = 1.0 / dj;
    if (i > j) {
      const double lij = Wk[i * PLD + j] * inv;
      if (part == 0) Lo[i * PLD + j] = lij;
      for (int k = j + 1 + part; k <= i; k += 4) {
        const double lkj = Wk[k * PLD + j] * inv;
        Wk[i * PLD + k] -= lij * lkj;
      }
    } else if (i == j && part == 0) {
      Lo[j * PLD + j] = dj;
    }
    __syncthreads();
  }
  // inverse: row r of W from rows < r
  const int c = i;
  for (int r = 0; r < 64; ++r) {
    double s = 0.0;
    if (c <= r)
      for (int k = c + part; k < r; k += 4) s += Lo[r * PLD + k] * Wi[k * PLD + c];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (c <= r && part == 0) Wi[r * PLD + c] = ((c == r ? 1.0 : 0.0) - s) / Lo[r * PLD + r];
    __syncthreads();
  }
  for (int e = tid; e < 4096; e += 256) {
    const int ii = e >> 6, k = e & 63;
    if (k <= ii) A[(int64_t)ii * lda + k] = Lo[ii * PLD + k];
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
  __shared__ __attribute__((aligned(16))) double smem[TileShape<64, 64, false>::SMEM_DOUBLES];
  double* Xs = X + (int64_t)blockIdx.x * 64 * ldx;
  double* Ps = P ? P + (int64_t)blockIdx.x * 64 * ldp : nullptr;
  v4d acc[2][2];
  for (int jb = 0; jb < nbq; ++jb) {
    double* Xj = Xs + jb * 64;
    if (jb > 0) {
      zero_acc(acc);
      gemm_tile<64, 64, false>(Xs, ldx, L + (int64_t)jb * 64 * ldl, ldl, jb * 64, acc, smem);
      store_tile<64, 64, 0>(Xj, ldx, acc);
      __syncthreads();
    }
    zero_acc(acc);
    gemm_tile<64, 64, false>(Xj, ldx, Winv + (int64_t)jb * 4096, 64, 64, acc, smem);
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

template <int BT>
static void launch_gemm_nt_t(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                             int64_t ldb, int64_t m, int64_t n, int64_t k, int lower, int mode,
                             hipStream_t st) {
  const int64_t tm = m / BT, tn = n / BT;
  const int64_t nblk = lower ? tm * (tm + 1) / 2 : tm * tn;
  dim3 grid((unsigned)nblk), block(256);
  if (lower) {
    if (mode == 0)
      hipLaunchKernelGGL((gemm_nt_kernel<BT, true, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)k);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<BT, true, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)k);
  } else {
    if (mode == 0)
      hipLaunchKernelGGL((gemm_nt_kernel<BT, false, 0>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)k);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<BT, false, 1>), grid, block, 0, st, C, ldc, A, lda, B, ldb, (int)tm, (int)k);
  }
}

void launch_gemm_nt(int tile, double* C, int64_t ldc, const double* A, int64_t lda,
                    const double* B, int64_t ldb, int64_t m, int64_t n, int64_t k, int lower,
                    int mode, hipStream_t st) {
  if (m <= 0 || n <= 0) return;
  if (tile == 128)
    launch_gemm_nt_t<128>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, st);
  else
    launch_gemm_nt_t<64>(C, ldc, A, lda, B, ldb, m, n, k, lower, mode, st);
}

void launch_gemm_nn(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                    int64_t ldb, int64_t m, int64_t n, int64_t k, hipStream_t st) {
  if (m <= 0 || n <= 0) return;
  const int64_t tm = m / 64, tn = n / 64;
  hipLaunchKernelGGL(gemm_nn_kernel, dim3((unsigned)(tm * tn)), dim3(256), 0, st, C, ldc, A, lda, B,
                     ldb, (int)tm, (int)k);
}

}  // namespace gpx
