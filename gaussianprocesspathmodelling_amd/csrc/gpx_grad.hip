// gpx_grad.hip — analytic gradient of the log marginal likelihood w.r.t. the log
// hyper-parameters ("hyper-parameter gradient hooks"; the reference has
// no counterpart — GPmap.py contains no GP code).  Restated on the CPU by
// oracle/gp_oracle.py:OracleGP.lml_gradient (R&W eq. 5.9):
//
//     dLML/dlog(theta) = 1/2 sum_ij (sum_c alpha_ic alpha_jc - k K^-1_ij) (dK/dlog theta)_ij
//
//   dK/dlog l_c  = kd_ij * ((x_ic - x_jc) / l_c)^2   RBF: kd = Kf;  Matern-5/2: kd = sf2 (5/3)(1+s) e^-s
//   dK/dlog sf2  = Kf_ij = sf2 k(r_ij)
//   dK/dlog sn2  = sn2 delta_ij
//
// K^-1 is never stored.  With ZT = L^-T (upper triangular, built from the factor already held
// by a forward substitution on the identity that skips the structural zeros: N^3/3 flops),
// K^-1 = ZT ZT^T is an NT product on the MFMA tile engine whose k-range starts at the tile's
// first row (another N^3/3), and each 128x128 tile of it is consumed in the epilogue of the
// very workgroup that produced it: the dK/dtheta tile is regenerated from the scaled points
// (as the kernel build does), multiplied in, reduced over the workgroup and written as one
// partial sum per (tile, theta).  The alpha alpha^T term is a separate O(N^2 k) pass of the
// same epilogue over 64x64 tiles.  Partials are summed in a fixed order: deterministic.
#include "gpx_internal.h"
#include "gpx_tile.h"

namespace gpx {
namespace {

constexpr double SQRT5 = 2.23606797749978969640917366873128;
constexpr int GMAXD = 32;

// kf = sf2 k(r), kd = the factor of d_c^2 in dK/dlog l_c
template <int KERNEL>
__device__ __forceinline__ void kvals(double r2, double sf2, double& kf, double& kd) {
  if (KERNEL == 0) {
    kf = sf2 * exp(-0.5 * r2);
    kd = kf;
  } else {
    const double s = SQRT5 * sqrt(r2);
    const double e = sf2 * exp(-s);
    kf = (1.0 + s + s * s / 3.0) * e;
    kd = (5.0 / 3.0) * (1.0 + s) * e;
  }
}

// sum over the 256 threads of a workgroup, every thread gets the result (fixed order)
__device__ __forceinline__ double wg_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // red[] may still be read from the previous call
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- fused K^-1 tile + trace epilogue -------------------------------------------------------
// grid: the hole-free triangular super-tile map of the SYRK (tile_coords<true>), 128x128 tiles.
// part[lin * ntheta + t], t = (lengthscales..., sf2, sn2): sum over the tile of
// K^-1_ij (dK/dlog theta_t)_ij, off-diagonal tiles counted twice (symmetry).
template <int KERNEL, int D>
__global__ __launch_bounds__(256, 2) void kinv_trace_kernel(const double* __restrict__ ZT, int64_t ld,
                                                            int tiles, int64_t npad, int64_t n,
                                                            const double* __restrict__ Xs, int d_rt, int ard,
                                                            double sf2, double sn2, double* __restrict__ part,
                                                            int ntheta, int64_t nslots, int P, int rank,
                                                            int dc_nb, int dc_P, int dc_r, int64_t dc_cols, int dc_snake) {
  constexpr int BT = 128;
  __shared__ __attribute__((aligned(16))) double smem[TileShapeG<double, BT, BT>::SMEM_ELEMS];
  __shared__ double red[4];
  const int d = (D > 0) ? D : d_rt;
  // Tile (ti, tj) costs (tiles - ti) k-ranges, so handing each XCD one contiguous eighth of the
  // tile order (xcd_chunk_id, right for the uniform SYRK) gives the first XCD 29 % of the work
  // and the kernel ran at 31.6 TF.  Deal whole 64-slot groups (one super-tile: what an XCD runs
  // concurrently, so the L2 sharing inside a group is kept) round-robin over the XCDs instead:
  // neighbouring groups cost about the same, and the heavy ones still come first.
  // Sharded call (P ranks, each holding the whole L^-T): whole octets of groups — one group per
  // XCD — are dealt round-robin over the ranks, so every rank keeps all eight XCDs busy and its
  // share of heavy and light tiles; rank r launches only its own octets.
  const int64_t b = blockIdx.x, xl = b >> 3;
  const int64_t lin = ((((xl >> 6) * P + rank) << 3) + (b & 7)) * 64 + (xl & 63);
  int ti, tj;
  if (lin >= nslots || !tile_coords<true>(lin, tiles, tiles, 8, 0, BcMask{0, 1, 0}, ti, tj)) return;
  Num<double>::v4 acc[4][4];
  zero_acc(acc);
  // rows ti, tj of ZT are zero left of column ti*BT (tj <= ti): start the contraction there.
  // Distributed-column mode (dc_P > 0; the factor is only held distributed): this rank holds the
  // columns of L^-T that belong to ITS row blocks of L (height dc_nb, its block number lb — Deal.global(dc_r, lb) —
  // at local column lb dc_nb) for ALL rows, and contributes the part of the contraction over those
  // columns; the first own block that is not left of the tile's rows starts it.  The partial sums of
  // the ranks add up to the trace (one all-reduce of ntheta numbers).
  int64_t koff = (int64_t)ti * BT, klen = npad - koff;
  if (dc_P > 0) {
    const int64_t gi = koff / dc_nb;                                        // block of the tile's first row
    const int64_t before = Deal{dc_P, dc_snake}.upto(gi - 1, dc_r);         // own blocks with index < gi
    koff = before * dc_nb;
    klen = dc_cols - koff;
    if (klen <= 0) return;  // nothing of this tile lives here (its slot of `part` was zeroed)
  }
  gemm_tile_g<double, BT, BT>(ZT + (int64_t)ti * BT * ld + koff, ld, ZT + (int64_t)tj * BT * ld + koff, ld,
                              (int)klen, acc, smem);
  // gemm_tile_g ends with a barrier: the staging buffers are free -> scaled points of the
  // tile's rows and columns
  double* xa = smem;
  double* xb = smem + BT * GMAXD;
  const int tid = threadIdx.x;
  for (int e = tid; e < BT * d; e += 256) {
    xa[e] = Xs[(int64_t)ti * BT * d + e];
    xb[e] = Xs[(int64_t)tj * BT * d + e];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  double Sf = 0.0, Sn = 0.0, Sl = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = wr * 64 + m * 16 + l4 + 4 * r;
      const int64_t gi = (int64_t)ti * BT + il;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        const int jl = wc * 64 + nn * 16 + l15;
        const int64_t gj = (int64_t)tj * BT + jl;
        double r2 = 0.0;
        if (D > 0) {
#pragma unroll
          for (int c = 0; c < D; ++c) {
            const double e = xa[il * D + c] - xb[jl * D + c];
            r2 += e * e;
          }
        } else {
          for (int c = 0; c < d; ++c) {
            const double e = xa[il * d + c] - xb[jl * d + c];
            r2 += e * e;
          }
        }
        double v = acc[m][nn][r];
        if (gi >= n || gj >= n) v = 0.0;  // padded rows / columns are not part of K
        double kf, kd;
        kvals<KERNEL>(r2, sf2, kf, kd);
        Sf += v * kf;
        if (gi == gj) Sn += v;
        const double t = v * kd;
        Sl += t * r2;
        acc[m][nn][r] = t;  // kept for the per-dimension pass (ARD)
        // one element at a time: left alone, the scheduler interleaves all 64 evaluations of
        // this fully unrolled nest (the accumulators must stay in registers) and spills hundreds
        // of VGPRs; the epilogue is < 1 % of the tile's time either way
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  const double w = (ti == tj) ? 1.0 : 2.0;
  double* out = part + lin * ntheta;
  const int nls = ntheta - 2;
  if (!ard) {
    const double s = wg_sum(Sl, red);
    if (tid == 0) out[0] = w * s;
  } else {
    for (int c = 0; c < d; ++c) {
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = wr * 64 + m * 16 + l4 + 4 * r;
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) {
            const int jl = wc * 64 + nn * 16 + l15;
            const double e = xa[il * d + c] - xb[jl * d + c];
            s += acc[m][nn][r] * e * e;
          }
        }
      s = wg_sum(s, red);
      if (tid == 0) out[c] = w * s;
    }
  }
  const double sf = wg_sum(Sf, red), sn = wg_sum(Sn, red);
  if (tid == 0) {
    out[nls] = w * sf;
    out[nls + 1] = sn * sn2;  // only diagonal tiles hold diagonal elements (w = 1 there)
  }
}

// ---- the alpha alpha^T term: sum_ij (sum_c alpha_ic alpha_jc) (dK/dlog theta)_ij -------------
// 64x64 tiles over the lower triangle (row-major triangular enumeration as the kernel build);
// alphaT (64 x ld): row c = target c.  Same partial layout as above.
__device__ __forceinline__ void tri_coords64(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > t) --i;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

template <int KERNEL, int D>
__global__ __launch_bounds__(256) void alpha_quad_kernel(const double* __restrict__ alphaT, int64_t ld, int k,
                                                        int64_t n, const double* __restrict__ Xs, int d_rt,
                                                        int ard, double sf2, double sn2,
                                                        double* __restrict__ part, int ntheta) {
  constexpr int KT = 64;
  __shared__ double xa[KT * GMAXD];
  __shared__ double xb[KT * GMAXD];
  __shared__ double red[4];
  const int d = (D > 0) ? D : d_rt;
  int ti, tj;
  tri_coords64((int64_t)blockIdx.x, ti, tj);
  const int64_t i0 = (int64_t)ti * KT, j0 = (int64_t)tj * KT;
  const int tid = threadIdx.x;
  for (int e = tid; e < KT * d; e += 256) {
    xa[e] = Xs[i0 * d + e];
    xb[e] = Xs[j0 * d + e];
  }
  __syncthreads();
  const int jl0 = (tid & 31) * 2, rg = tid >> 5;
  // weights of this thread's 8 rows x 2 columns
  double wt[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r) wt[r][0] = wt[r][1] = 0.0;
  for (int c = 0; c < k; ++c) {
    const double* a = alphaT + (int64_t)c * ld;
    const double b0 = a[j0 + jl0], b1 = a[j0 + jl0 + 1];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const double ai = a[i0 + rg + 8 * r];
      wt[r][0] += ai * b0;
      wt[r][1] += ai * b1;
    }
  }
  double Sf = 0.0, Sn = 0.0, Sl = 0.0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int il = rg + 8 * r;
    const int64_t gi = i0 + il;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int jl = jl0 + q;
      const int64_t gj = j0 + jl;
      double r2 = 0.0;
      for (int c = 0; c < d; ++c) {
        const double e = xa[il * d + c] - xb[jl * d + c];
        r2 += e * e;
      }
      double v = wt[r][q];
      if (gi >= n || gj >= n) v = 0.0;
      double kf, kd;
      kvals<KERNEL>(r2, sf2, kf, kd);
      Sf += v * kf;
      if (gi == gj) Sn += v;
      const double t = v * kd;
      Sl += t * r2;
      wt[r][q] = t;
    }
  }
  const double w = (ti == tj) ? 1.0 : 2.0;
  double* out = part + (int64_t)blockIdx.x * ntheta;
  const int nls = ntheta - 2;
  if (!ard) {
    const double s = wg_sum(Sl, red);
    if (tid == 0) out[0] = w * s;
  } else {
    for (int c = 0; c < d; ++c) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double e = xa[(rg + 8 * r) * d + c] - xb[(jl0 + q) * d + c];
          s += wt[r][q] * e * e;
        }
      s = wg_sum(s, red);
      if (tid == 0) out[c] = w * s;
    }
  }
  const double sf = wg_sum(Sf, red), sn = wg_sum(Sn, red);
  if (tid == 0) {
    out[nls] = w * sf;
    out[nls + 1] = sn * sn2;
  }
}

// out[t] = scale * sum_tile part[tile * ntheta + t]; one workgroup per theta, fixed order
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ part, int64_t ntile,
                                                             int ntheta, double scale, double* __restrict__ out) {
  __shared__ double red[4];
  const int t = blockIdx.x;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < ntile; i += 256) s += part[i * ntheta + t];
  s = wg_sum(s, red);
  if (threadIdx.x == 0) out[t] = scale * s;
}

// out[0] = sum_i sum_c y[i*k + c] * alphaT[c*ld + i]
__global__ __launch_bounds__(256) void dot_rhs_kernel(const double* __restrict__ y, const double* __restrict__ alphaT,
                                                     int64_t ld, int64_t n, int k, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t e = threadIdx.x; e < n * k; e += 256) {
    const int64_t i = e / k;
    const int c = (int)(e - i * k);
    s += y[e] * alphaT[(int64_t)c * ld + i];
  }
  s = wg_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void set_diag_one_kernel(double* A, int64_t lda, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) A[i * lda + i] = 1.0;
}

template <int KERNEL>
void launch_kinv_trace_k(const double* ZT, int64_t ld, int64_t npad, int64_t n, const double* Xs, int d, int ard,
                         double sf2, double sn2, double* part, int ntheta, int P, int rank, int dc_nb, int dc_P,
                         int dc_r, int64_t dc_cols, int dc_snake, hipStream_t st) {
  const int tiles = (int)(npad / 128);
  const int64_t ts = (tiles + 7) / 8;
  const int64_t nslots = ts * (ts - 1) / 2 * 64 + ts * 36;
  const int64_t octets = (nslots + 511) / 512, mine = octets > rank ? (octets - rank + P - 1) / P : 0;
  if (mine == 0) return;
  dim3 grid((unsigned)(mine * 512)), block(256);  // whole octets of 8 x 64 slots
  switch (d) {
    case 1: hipLaunchKernelGGL((kinv_trace_kernel<KERNEL, 1>), grid, block, 0, st, ZT, ld, tiles, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, nslots, P, rank, dc_nb, dc_P, dc_r, dc_cols, dc_snake); break;
    case 2: hipLaunchKernelGGL((kinv_trace_kernel<KERNEL, 2>), grid, block, 0, st, ZT, ld, tiles, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, nslots, P, rank, dc_nb, dc_P, dc_r, dc_cols, dc_snake); break;
    case 3: hipLaunchKernelGGL((kinv_trace_kernel<KERNEL, 3>), grid, block, 0, st, ZT, ld, tiles, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, nslots, P, rank, dc_nb, dc_P, dc_r, dc_cols, dc_snake); break;
    default: hipLaunchKernelGGL((kinv_trace_kernel<KERNEL, 0>), grid, block, 0, st, ZT, ld, tiles, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, nslots, P, rank, dc_nb, dc_P, dc_r, dc_cols, dc_snake); break;
  }
}

template <int KERNEL>
void launch_alpha_quad_k(const double* alphaT, int64_t ld, int k, int64_t npad, int64_t n, const double* Xs, int d,
                         int ard, double sf2, double sn2, double* part, int ntheta, hipStream_t st) {
  const int64_t T = npad / 64;
  dim3 grid((unsigned)(T * (T + 1) / 2)), block(256);
  if (d == 3)
    hipLaunchKernelGGL((alpha_quad_kernel<KERNEL, 3>), grid, block, 0, st, alphaT, ld, k, n, Xs, d, ard, sf2, sn2, part, ntheta);
  else
    hipLaunchKernelGGL((alpha_quad_kernel<KERNEL, 0>), grid, block, 0, st, alphaT, ld, k, n, Xs, d, ard, sf2, sn2, part, ntheta);
}

}  // namespace

int64_t kinv_trace_slots(int64_t npad) {
  const int64_t ts = (npad / 128 + 7) / 8;
  return ts * (ts - 1) / 2 * 64 + ts * 36;
}

int64_t alpha_quad_slots(int64_t npad) {
  const int64_t T = npad / 64;
  return T * (T + 1) / 2;
}

void launch_set_diag_one(double* A, int64_t lda, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(set_diag_one_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, lda, n);
}

void launch_kinv_trace(int kernel, const double* ZT, int64_t ld, int64_t npad, int64_t n, const double* Xs, int d,
                       int ard, double sf2, double sn2, double* part, int ntheta, int P, int rank, hipStream_t st) {
  if (kernel == 0)
    launch_kinv_trace_k<0>(ZT, ld, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, P, rank, 0, 0, 0, 0, 0, st);
  else
    launch_kinv_trace_k<1>(ZT, ld, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, P, rank, 0, 0, 0, 0, 0, st);
}

void launch_kinv_trace_cols(int kernel, const double* ZTc, int64_t ldc, int64_t npad, int64_t n, const double* Xs,
                            int d, int ard, double sf2, double sn2, double* part, int ntheta, int nb, int P, int rank,
                            int64_t ncols, int snake, hipStream_t st) {
  if (kernel == 0)
    launch_kinv_trace_k<0>(ZTc, ldc, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, 1, 0, nb, P, rank, ncols, snake, st);
  else
    launch_kinv_trace_k<1>(ZTc, ldc, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, 1, 0, nb, P, rank, ncols, snake, st);
}

void launch_alpha_quad(int kernel, const double* alphaT, int64_t ld, int k, int64_t npad, int64_t n,
                       const double* Xs, int d, int ard, double sf2, double sn2, double* part, int ntheta,
                       hipStream_t st) {
  if (kernel == 0)
    launch_alpha_quad_k<0>(alphaT, ld, k, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, st);
  else
    launch_alpha_quad_k<1>(alphaT, ld, k, npad, n, Xs, d, ard, sf2, sn2, part, ntheta, st);
}

void launch_reduce_partials(const double* part, int64_t ntile, int ntheta, double scale, double* out,
                            hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ntheta), dim3(256), 0, st, part, ntile, ntheta, scale, out);
}

void launch_dot_rhs(const double* y, const double* alphaT, int64_t ld, int64_t n, int k, double* out,
                    hipStream_t st) {
  hipLaunchKernelGGL(dot_rhs_kernel, dim3(1), dim3(256), 0, st, y, alphaT, ld, n, k, out);
}

}  // namespace gpx
