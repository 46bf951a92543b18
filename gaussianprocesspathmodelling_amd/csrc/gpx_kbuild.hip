// gpx_kbuild.hip — kernel-matrix builder (SURVEY.md §8 rows a1 "K1" and a2 "K1'").
//
// The reference has no kernel-matrix code; its nearest relative is the pairwise
// distance loop trajectories.calc_distance (GPmap.py:114-121).  Formula and
// operation order follow oracle/gp_oracle.py:kernel_matrix.
//
// HBM-write-bound: one 64x64 tile of K per 256-thread workgroup, input points staged
// once in LDS, each lane writes 16-byte (2 x f64) pieces so a half-wave covers 512
// contiguous bytes of one row.  Algorithmic traffic: 8 B per entry written, the
// points (N*d*8 B) read once per tile row/column from L2.
#include "gpx_internal.h"

namespace gpx {
namespace {

constexpr int KT = 64;      // tile edge
constexpr int MAXD = 32;    // max input dimension staged in LDS
constexpr double SQRT5 = 2.23606797749978969640917366873128;

template <int KERNEL, typename T>
__device__ __forceinline__ T kfun(T r2, T sf2) {
  if (KERNEL == 0) {
    return sf2 * exp((T)-0.5 * r2);
  } else {
    const T s = (T)SQRT5 * sqrt(r2);
    return sf2 * (((T)1 + s + s * s / (T)3) * exp(-s));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_points_kernel(const T* __restrict__ X, int64_t n,
                                                          int64_t npad, int d,
                                                          const double* __restrict__ ls, int n_ls,
                                                          T* __restrict__ Xs) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npad * d) return;
  const int64_t i = idx / d;
  const int c = (int)(idx - i * d);
  Xs[idx] = (i < n) ? X[idx] / (T)ls[n_ls == 1 ? 0 : c] : (T)0;
}

// linear index over the lower triangle (row-major) -> (ti, tj), tj <= ti
__device__ __forceinline__ void tri_coords(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > t) --i;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

// SYM: lower tiles of a square matrix built from one point set (As == Bs), diagonal
// gets + diag_add, padding becomes identity.  !SYM: all tiles, padding is zero.
template <typename T, int KERNEL, bool SYM, int D>
__global__ __launch_bounds__(256) void kbuild_kernel(const T* __restrict__ As, int64_t m,
                                                    const T* __restrict__ Bs, int64_t n, int d_rt,
                                                    int tiles_n, T sf2, T diag_add, T* __restrict__ K,
                                                    int64_t ld) {
  const int d = (D > 0) ? D : d_rt;
  // LDS sized by the instantiation (d = 3: 2 x 1.5 KB, not 2 x 16 KB): the occupancy is then the waves', not the LDS's
  __shared__ T xa[KT * (D > 0 ? D : MAXD)];
  __shared__ T xb[KT * (D > 0 ? D : MAXD)];
  int ti, tj;
  if (SYM) {
    tri_coords((int64_t)blockIdx.x, ti, tj);
  } else {
    ti = (int)(blockIdx.x / tiles_n);
    tj = (int)(blockIdx.x - (int64_t)ti * tiles_n);
  }
  const int64_t i0 = (int64_t)ti * KT, j0 = (int64_t)tj * KT;
  const int tid = threadIdx.x;
  for (int e = tid; e < KT * d; e += 256) {
    xa[e] = As[i0 * d + e];  // rows of the padded point array are always readable
    xb[e] = Bs[j0 * d + e];
  }
  __syncthreads();
  const int c2 = (tid & 31) * 2;
  const int rg = tid >> 5;
  T bj0[D > 0 ? D : MAXD], bj1[D > 0 ? D : MAXD];
  if (D > 0) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      bj0[c] = xb[c2 * D + c];
      bj1[c] = xb[(c2 + 1) * D + c];
    }
  }
  const int64_t col0 = j0 + c2, col1 = col0 + 1;
#pragma unroll 2
  for (int r = 0; r < 8; ++r) {
    const int il = rg + 8 * r;
    const int64_t row = i0 + il;
    T s0 = (T)0, s1 = (T)0;
    if (D > 0) {
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const T a = xa[il * D + c];
        const T e0 = a - bj0[c], e1 = a - bj1[c];
        s0 += e0 * e0;
        s1 += e1 * e1;
      }
    } else {
      for (int c = 0; c < d; ++c) {
        const T a = xa[il * d + c];
        const T e0 = a - xb[c2 * d + c], e1 = a - xb[(c2 + 1) * d + c];
        s0 += e0 * e0;
        s1 += e1 * e1;
      }
    }
    T v0 = kfun<KERNEL, T>(s0, sf2), v1 = kfun<KERNEL, T>(s1, sf2);
    if (SYM) {
      if (row == col0) v0 += diag_add;
      if (row == col1) v1 += diag_add;
      if (row >= m || col0 >= n) v0 = (row == col0) ? (T)1 : (T)0;
      if (row >= m || col1 >= n) v1 = (row == col1) ? (T)1 : (T)0;
    } else {
      if (row >= m || col0 >= n) v0 = (T)0;
      if (row >= m || col1 >= n) v1 = (T)0;
    }
    typedef T pair_t __attribute__((ext_vector_type(2)));
    pair_t out = {v0, v1};
    // Non-temporal 16-byte stores: nothing re-reads a tile before the whole matrix (17 GB) has gone by.  Round 4,
    // tools/kbuild_variants.hip, twelve variants side by side in one process at N = 65536 (profiles/r04_kbuild_variants.txt):
    // this kernel 3.36 ms = 5.1 TB/s against 3.52 with plain stores and 3.58 with the 2 x 16 KB static LDS of rounds 1-3;
    // unrolling (1 / 2 / 4 / 8 rows), persistent tile walks, row strips and an occupancy hint all land within 3.4-3.9 ms.
    __builtin_nontemporal_store(out, reinterpret_cast<pair_t*>(K + row * ld + col0));
  }
}

template <typename T, int KERNEL, bool SYM>
void dispatch_d(const T* As, int64_t m, const T* Bs, int64_t n, int d, int64_t nblocks, int tiles_n,
                double sf2, double diag_add, T* K, int64_t ld, hipStream_t st) {
  dim3 grid((unsigned)nblocks), block(256);
  const T s = (T)sf2, da = (T)diag_add;
  switch (d) {
    case 1: hipLaunchKernelGGL((kbuild_kernel<T, KERNEL, SYM, 1>), grid, block, 0, st, As, m, Bs, n, d, tiles_n, s, da, K, ld); break;
    case 2: hipLaunchKernelGGL((kbuild_kernel<T, KERNEL, SYM, 2>), grid, block, 0, st, As, m, Bs, n, d, tiles_n, s, da, K, ld); break;
    case 3: hipLaunchKernelGGL((kbuild_kernel<T, KERNEL, SYM, 3>), grid, block, 0, st, As, m, Bs, n, d, tiles_n, s, da, K, ld); break;
    default: hipLaunchKernelGGL((kbuild_kernel<T, KERNEL, SYM, 0>), grid, block, 0, st, As, m, Bs, n, d, tiles_n, s, da, K, ld); break;
  }
}

}  // namespace

template <typename T>
void launch_scale_points(const T* X, int64_t n, int64_t npad, int d, const double* ls, int n_ls, T* Xs,
                         hipStream_t st) {
  const int64_t total = npad * d;
  hipLaunchKernelGGL(scale_points_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X,
                     n, npad, d, ls, n_ls, Xs);
}

template <typename T>
void launch_kbuild_sym(int kernel, const T* Xs, int64_t n, int64_t npad, int d, double sf2,
                       double diag_add, T* K, int64_t ld, hipStream_t st) {
  const int64_t TT = npad / KT;
  const int64_t nblocks = TT * (TT + 1) / 2;
  if (kernel == 0)
    dispatch_d<T, 0, true>(Xs, n, Xs, n, d, nblocks, (int)TT, sf2, diag_add, K, ld, st);
  else
    dispatch_d<T, 1, true>(Xs, n, Xs, n, d, nblocks, (int)TT, sf2, diag_add, K, ld, st);
}

template <typename T>
void launch_kbuild_cross(int kernel, const T* As, int64_t m, int64_t mpad, const T* Bs, int64_t n,
                         int64_t npad, int d, double sf2, T* K, int64_t ld, hipStream_t st) {
  const int64_t tm = mpad / KT, tn = npad / KT;
  if (kernel == 0)
    dispatch_d<T, 0, false>(As, m, Bs, n, d, tm * tn, (int)tn, sf2, 0.0, K, ld, st);
  else
    dispatch_d<T, 1, false>(As, m, Bs, n, d, tm * tn, (int)tn, sf2, 0.0, K, ld, st);
}

#define GPX_INSTANTIATE_KBUILD(T)                                                                     \
  template void launch_scale_points<T>(const T*, int64_t, int64_t, int, const double*, int, T*,       \
                                       hipStream_t);                                                  \
  template void launch_kbuild_sym<T>(int, const T*, int64_t, int64_t, int, double, double, T*,        \
                                     int64_t, hipStream_t);                                           \
  template void launch_kbuild_cross<T>(int, const T*, int64_t, int64_t, const T*, int64_t, int64_t,   \
                                       int, double, T*, int64_t, hipStream_t);
GPX_INSTANTIATE_KBUILD(double)
GPX_INSTANTIATE_KBUILD(float)

}  // namespace gpx
