// gpx_mixed.hip — kernels of the mixed-precision mode (BASELINE.json configs[4]: "fp32 + ARD
// lengthscales — mixed-precision tolerance study").  The factorisation
// runs in fp32 on the fp32 MFMA engine (2x the fp64 rate); alpha is then refined in fp64:
//     r = y - K alpha        fp64, K regenerated from the scaled points on the fly (never stored)
//     L L^T delta = r        fp32 solves with the fp32 factor
//     alpha += delta         fp64
// and the posterior mean is K* alpha in fp64 through the same matrix-free product.  The
// reference has no counterpart (GPmap.py has no GP code); restated by tests against the fp64
// oracle (oracle/gp_oracle.py).
#include "gpx_internal.h"

namespace gpx {
namespace {

constexpr double SQRT5 = 2.23606797749978969640917366873128;
constexpr int XMAXD = 32;
constexpr int KMAX = 8;  // target columns of the mixed mode

// exp(x) for x <= 0, branch-free: Cody-Waite reduction x = n ln2 + r, |r| <= ln2 / 2, degree-13 Taylor
// polynomial of exp(r) (truncation 0.347^14 / 14! = 4e-18: below an ulp), v_ldexp for 2^n (underflows to 0
// by itself).  ~20 fp64 instructions against the library exp with its special cases; the matrix-free
// products of the mixed mode regenerate N^2 kernel values per pass.  Measured (round 3, N = 65536, six
// refinement iterations): 193 -> 178 ms — the pass is a smaller part of an iteration than the two 64-row
// triangular solves through the fp32 factor.  Relative error 2.2e-16 (checked against numpy over [-630, 0]).
__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);  // ln2 hi
  r = fma(n, -1.90821492927058770002e-10, r);         // ln2 lo
  double p = 1.0 / 6227020800.0;                      // 1 / 13!
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

template <int KERNEL>
__device__ __forceinline__ double kval(double r2, double sf2) {
  if (KERNEL == 0) return sf2 * exp_nonpos(-0.5 * r2);
  const double s = SQRT5 * sqrt(r2);
  return sf2 * ((1.0 + s + s * s / 3.0) * exp_nonpos(-s));
}

// outT[c][i] = (y ? y[i*k + c] : 0) + sign * (sum_j sf2 k(a_i, b_j) alphaT[c][j] + diag * alphaT[c][i])
// for i < m (0 beyond), c < k <= KMAX.  As (mpad x d), Bs (npad x d) scaled points, zero rows beyond
// m / n; alphaT (k x lda) must be zero beyond n.  One workgroup per 64 rows; wave g takes 16 of
// every 64 columns (lane = row: the column operands are LDS broadcasts); fixed reduction order.
// KC = compile-time bound on the number of target columns (1, 2 or KMAX): no predicated accumulators
template <int KERNEL, int D, int KC>
__global__ __launch_bounds__(256) void kmatvec_kernel(const double* __restrict__ As, int64_t m,
                                                     const double* __restrict__ Bs, int64_t npad, int d_rt,
                                                     double sf2, double diag, const double* __restrict__ y,
                                                     const double* __restrict__ alphaT, int64_t lda, int k,
                                                     double sign, double* __restrict__ outT, int64_t ldo) {
  // sized by the instantiation (round 3): with KMAX-sized buffers (146 KB) ONE workgroup fitted a CU — one wave per
  // SIMD walking a latency chain of global load -> barrier -> 16 kernel values -> barrier; 3.5 KB at k = 1, d = 3
  __shared__ double xb[64 * (D > 0 ? D : XMAXD)];
  __shared__ double ab[KC * 64];
  __shared__ double red[3 * 64 * KC];
  const int d = (D > 0) ? D : d_rt;
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  double xa[D > 0 ? D : XMAXD];
  for (int c = 0; c < d; ++c) xa[c] = As[i * d + c];  // padded rows are readable
  double acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = 0.0;
  for (int64_t j0 = 0; j0 < npad; j0 += 64) {
    for (int e = tid; e < 64 * d; e += 256) xb[e] = Bs[j0 * d + e];
    for (int e = tid; e < k * 64; e += 256) ab[e] = alphaT[(int64_t)(e >> 6) * lda + j0 + (e & 63)];
    __syncthreads();
#pragma unroll 8
    for (int jj = g * 16; jj < g * 16 + 16; ++jj) {
      double r2 = 0.0;
      if (D > 0) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
          const double e = xa[c] - xb[jj * D + c];
          r2 += e * e;
        }
      } else {
        for (int c = 0; c < d; ++c) {
          const double e = xa[c] - xb[jj * d + c];
          r2 += e * e;
        }
      }
      const double kf = kval<KERNEL>(r2, sf2);
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if (KC <= 2 || c < k) acc[c] += kf * ab[c * 64 + jj];
    }
    __syncthreads();
  }
  if (g > 0) {
#pragma unroll
    for (int c = 0; c < KC; ++c) red[((g - 1) * 64 + lane) * KC + c] = acc[c];
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c < k) {
        double v = acc[c];
        for (int q = 0; q < 3; ++q) v += red[(q * 64 + lane) * KC + c];
        double out = 0.0;
        if (i < m) out = (y ? y[i * k + c] : 0.0) + sign * (v + diag * alphaT[(int64_t)c * lda + i]);
        outT[(int64_t)c * ldo + i] = out;
      }
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out,
                                                     int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    out[i] = (TO)in[i];
}

// dst (R x ldd, TO) [r][i] = r < rows && i < n ? scale * src[r][i] + (acc ? dst : 0) : (acc ? dst : 0)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rows_axpy_kernel(const TI* __restrict__ src, int64_t lds, TO* __restrict__ dst,
                                                       int64_t ldd, int rows, int64_t n, int64_t npad, int acc) {
  const int r = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256) {
    const TO v = (r < rows && i < n) ? (TO)src[(int64_t)r * lds + i] : (TO)0;
    TO* p = dst + (int64_t)r * ldd + i;
    *p = acc ? *p + v : v;
  }
}

// out[0] = sum over rows r < rows, i < n of src[r][i]^2 (one workgroup, fixed order)
__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double* __restrict__ src, int64_t lds, int rows,
                                                        int64_t n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int r = 0; r < rows; ++r)
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double v = src[(int64_t)r * lds + i];
      s += v * v;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- triangular solves with FEW right-hand sides (the refinement's delta = (L L^T)^-1 r) -----------------------
// With k <= 8 right-hand sides a block substitution is a STREAM over the factor: block solve = a product with the
// explicit block inverse W_p, update = a matrix-vector product with the panel (forward: rows below, one wave per
// row) or with the row block (backward: columns to the left, one lane per column pair, sixteen waves over the
// block's rows) — bandwidth-bound launches of the whole GPU instead of slab solves and 64-row MFMA updates that are
// 20-30 us of launch and latency each.  Fixed summation order everywhere (no atomics).
constexpr int FEW_NJ = 1024;  // widest block (the panel width's maximum is 2048: wider falls back to the slab path)

// 16-byte row loads (4 floats / 2 doubles per lane), 8-byte column loads (2 floats / 1 double per lane)
template <typename T>
struct FewVec;
template <>
struct FewVec<float> {
  static constexpr int NR = 4, NC = 2;
  static __device__ __forceinline__ float dot(const float* a, const float* b) {
    const float4 m = *reinterpret_cast<const float4*>(a), x = *reinterpret_cast<const float4*>(b);
    return (m.x * x.x + m.y * x.y) + (m.z * x.z + m.w * x.w);
  }
  static __device__ __forceinline__ void ldc(const float* p, float (&v)[2]) {
    const float2 m = *reinterpret_cast<const float2*>(p);
    v[0] = m.x;
    v[1] = m.y;
  }
};
template <>
struct FewVec<double> {
  static constexpr int NR = 2, NC = 1;
  static __device__ __forceinline__ double dot(const double* a, const double* b) {
    const double2 m = *reinterpret_cast<const double2*>(a), x = *reinterpret_cast<const double2*>(b);
    return m.x * x.x + m.y * x.y;
  }
  static __device__ __forceinline__ void ldc(const double* p, double (&v)[1]) { v[0] = *p; }
};

// out[c * ldo + i] (ASSIGN: = ; else -=) sum_{j < nj} M[i * ldm + j] * v[c * ldv + j],  i < ni, c < k <= KC
template <typename T, int KC, bool ASSIGN>
__global__ __launch_bounds__(256) void rowdot_kernel(T* __restrict__ out, int64_t ldo, const T* __restrict__ M,
                                                     int64_t ldm, int64_t ni, int nj, const T* __restrict__ v,
                                                     int64_t ldv, int k) {
  constexpr int NR = FewVec<T>::NR;
  __shared__ __attribute__((aligned(16))) T vs[KC * FEW_NJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < KC * nj; e += 256) {
    const int c = e / nj, j = e - c * nj;
    vs[c * FEW_NJ + j] = c < k ? v[(int64_t)c * ldv + j] : (T)0;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= ni) return;
  T acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = (T)0;
  const T* row = M + i * ldm;
  for (int j = lane * NR; j < nj; j += 64 * NR) {
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] += FewVec<T>::dot(row + j, &vs[c * FEW_NJ + j]);
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c < k) {
        T* p = out + (int64_t)c * ldo + i;
        *p = ASSIGN ? acc[c] : *p - acc[c];
      }
  }
}

// out[c * ldo + i] (ASSIGN: = ; else -=) sum_{j < nj} M[j * ldm + i] * v[c * ldv + j],  i < ni (a multiple of 64 NC);
// workgroup = 64 NC columns (lane = NC of them) x 16 waves, wave w takes the rows j = w, w + 16, ...
template <typename T, int KC, bool ASSIGN>
__global__ __launch_bounds__(1024) void coldot_kernel(T* __restrict__ out, int64_t ldo, const T* __restrict__ M,
                                                      int64_t ldm, int64_t ni, int nj, const T* __restrict__ v,
                                                      int64_t ldv, int k) {
  constexpr int NC = FewVec<T>::NC, WC = 64 * NC;
  __shared__ __attribute__((aligned(16))) T vs[KC * FEW_NJ];
  __shared__ T red[16 * KC * WC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < KC * nj; e += 1024) {
    const int c = e / nj, j = e - c * nj;
    vs[c * FEW_NJ + j] = c < k ? v[(int64_t)c * ldv + j] : (T)0;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * WC + lane * NC;
  T a[KC][NC];
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int q = 0; q < NC; ++q) a[c][q] = (T)0;
  const T* col = M + i;
#pragma unroll 8
  for (int j = wave; j < nj; j += 16) {
    T m[NC];
    FewVec<T>::ldc(col + (int64_t)j * ldm, m);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const T x = vs[c * FEW_NJ + j];
#pragma unroll
      for (int q = 0; q < NC; ++q) a[c][q] += m[q] * x;
    }
  }
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int q = 0; q < NC; ++q) red[(wave * KC + c) * WC + lane * NC + q] = a[c][q];
  __syncthreads();
  for (int e = tid; e < KC * WC; e += 1024) {
    const int c = e / WC, cc = e - c * WC;
    if (c < k) {
      T sacc = (T)0;
#pragma unroll
      for (int w = 0; w < 16; ++w) sacc += red[(w * KC + c) * WC + cc];
      T* p = out + (int64_t)c * ldo + (int64_t)blockIdx.x * WC + cc;
      *p = ASSIGN ? sacc : *p - sacc;
    }
  }
}

template <typename T, int KC>
void launch_few_kc(bool cols, bool assign, T* out, int64_t ldo, const T* M, int64_t ldm, int64_t ni, int nj,
                   const T* v, int64_t ldv, int k, hipStream_t st) {
  if (ni <= 0) return;
  if (!cols) {
    dim3 grid((unsigned)((ni + 3) / 4)), block(256);
    if (assign)
      hipLaunchKernelGGL((rowdot_kernel<T, KC, true>), grid, block, 0, st, out, ldo, M, ldm, ni, nj, v, ldv, k);
    else
      hipLaunchKernelGGL((rowdot_kernel<T, KC, false>), grid, block, 0, st, out, ldo, M, ldm, ni, nj, v, ldv, k);
  } else {
    dim3 grid((unsigned)(ni / (64 * FewVec<T>::NC))), block(1024);
    if (assign)
      hipLaunchKernelGGL((coldot_kernel<T, KC, true>), grid, block, 0, st, out, ldo, M, ldm, ni, nj, v, ldv, k);
    else
      hipLaunchKernelGGL((coldot_kernel<T, KC, false>), grid, block, 0, st, out, ldo, M, ldm, ni, nj, v, ldv, k);
  }
}

template <int KERNEL, int KC>
void launch_kmatvec_kc(const double* As, int64_t m, int64_t mpad, const double* Bs, int64_t npad, int d, double sf2,
                       double diag, const double* y, const double* alphaT, int64_t lda, int k, double sign,
                       double* outT, int64_t ldo, hipStream_t st) {
  dim3 grid((unsigned)(mpad / 64)), block(256);
  if (d == 3)
    hipLaunchKernelGGL((kmatvec_kernel<KERNEL, 3, KC>), grid, block, 0, st, As, m, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo);
  else
    hipLaunchKernelGGL((kmatvec_kernel<KERNEL, 0, KC>), grid, block, 0, st, As, m, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo);
}

template <int KERNEL>
void launch_kmatvec_k(const double* As, int64_t m, int64_t mpad, const double* Bs, int64_t npad, int d, double sf2,
                      double diag, const double* y, const double* alphaT, int64_t lda, int k, double sign,
                      double* outT, int64_t ldo, hipStream_t st) {
  if (k == 1)
    launch_kmatvec_kc<KERNEL, 1>(As, m, mpad, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo, st);
  else if (k == 2)
    launch_kmatvec_kc<KERNEL, 2>(As, m, mpad, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo, st);
  else
    launch_kmatvec_kc<KERNEL, KMAX>(As, m, mpad, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo, st);
}

}  // namespace

void launch_kmatvec(int kernel, const double* As, int64_t m, int64_t mpad, const double* Bs, int64_t npad, int d,
                    double sf2, double diag, const double* y, const double* alphaT, int64_t lda, int k,
                    double sign, double* outT, int64_t ldo, hipStream_t st) {
  if (kernel == 0)
    launch_kmatvec_k<0>(As, m, mpad, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo, st);
  else
    launch_kmatvec_k<1>(As, m, mpad, Bs, npad, d, sf2, diag, y, alphaT, lda, k, sign, outT, ldo, st);
}

// rows (cols = false): out[c][i] (=|-=) sum_j M[i][j] v[c][j];  columns (cols = true): ... sum_j M[j][i] v[c][j]
template <typename T>
void launch_few_product(bool cols, bool assign, T* out, int64_t ldo, const T* M, int64_t ldm, int64_t ni, int nj,
                        const T* v, int64_t ldv, int k, hipStream_t st) {
  if (k == 1)
    launch_few_kc<T, 1>(cols, assign, out, ldo, M, ldm, ni, nj, v, ldv, k, st);
  else if (k == 2)
    launch_few_kc<T, 2>(cols, assign, out, ldo, M, ldm, ni, nj, v, ldv, k, st);
  else
    launch_few_kc<T, KMAX>(cols, assign, out, ldo, M, ldm, ni, nj, v, ldv, k, st);
}
template void launch_few_product<float>(bool, bool, float*, int64_t, const float*, int64_t, int64_t, int, const float*,
                                        int64_t, int, hipStream_t);
template void launch_few_product<double>(bool, bool, double*, int64_t, const double*, int64_t, int64_t, int,
                                         const double*, int64_t, int, hipStream_t);

void launch_f64_to_f32(const double* in, float* out, int64_t count, hipStream_t st) {
  if (count <= 0) return;
  const int64_t bx = (count + 255) / 256;
  hipLaunchKernelGGL((convert_kernel<double, float>), dim3((unsigned)(bx > 4096 ? 4096 : bx)), dim3(256), 0, st, in, out, count);
}

void launch_f32_to_f64(const float* in, double* out, int64_t count, hipStream_t st) {
  if (count <= 0) return;
  const int64_t bx = (count + 255) / 256;
  hipLaunchKernelGGL((convert_kernel<float, double>), dim3((unsigned)(bx > 4096 ? 4096 : bx)), dim3(256), 0, st, in, out, count);
}

void launch_rows_f64_to_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int rows, int R, int64_t n,
                            int64_t npad, hipStream_t st) {
  const int64_t bx = (npad + 255) / 256;
  hipLaunchKernelGGL((rows_axpy_kernel<double, float>), dim3((unsigned)(bx > 1024 ? 1024 : bx), (unsigned)R), dim3(256), 0, st, src, lds, dst, ldd, rows, n, npad, 0);
}

void launch_rows_add_f32_to_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int rows, int64_t n,
                                int64_t npad, int acc, hipStream_t st) {
  const int64_t bx = (npad + 255) / 256;
  hipLaunchKernelGGL((rows_axpy_kernel<float, double>), dim3((unsigned)(bx > 1024 ? 1024 : bx), (unsigned)rows), dim3(256), 0, st, src, lds, dst, ldd, rows, n, npad, acc);
}

void launch_rows_sumsq(const double* src, int64_t lds, int rows, int64_t n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3(1), dim3(256), 0, st, src, lds, rows, n, out);
}

}  // namespace gpx
