// gpx_misc.hip — small HBM-bound helpers around the dense path: right-hand-side
// packing, posterior variance row reduction (SURVEY.md §8 row a6 "K4"), log-det,
// plus the fp64 MFMA layout probe and the two microbenchmarks quoted beside the
// rooflines.
#include "gpx_internal.h"

#include <atomic>

namespace gpx {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// YT[r][i] = (r < k && i < n) ? y[i*k + r] : 0
template <typename T>
__global__ __launch_bounds__(256) void pack_rhs_kernel(const T* __restrict__ y, int64_t n, int k,
                                                      T* __restrict__ YT, int64_t ld, int64_t npad) {
  const int r = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256)
    YT[(int64_t)r * ld + i] = (r < k && i < n) ? y[i * k + r] : (T)0;
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_rhs_kernel(const T* __restrict__ YT, int64_t ld, int64_t n,
                                                        int k, double scale, T* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * k) return;
  const int64_t i = idx / k;
  const int r = (int)(idx - i * k);
  out[idx] = (T)(scale * (double)YT[(int64_t)r * ld + i]);
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const double s = ((red[0] + red[1]) + (red[2] + red[3]));
  __syncthreads();
  return s;
}

// one workgroup per row: var[i] = sf2 - sum_j VT[i][j]^2  (fp64 accumulation, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void var_rows_kernel(const T* __restrict__ VT, int64_t ld,
                                                      int64_t ncols, double sf2, T* __restrict__ var) {
  typedef T pair_t __attribute__((ext_vector_type(2)));
  __shared__ double red[4];
  const T* row = VT + (int64_t)blockIdx.x * ld;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t j = (int64_t)threadIdx.x * 2; j < ncols; j += 512) {
    const pair_t v = *reinterpret_cast<const pair_t*>(row + j);
    s0 += (double)v.x * (double)v.x;
    s1 += (double)v.y * (double)v.y;
  }
  const double s = block_sum(s0 + s1, red);
  if (threadIdx.x == 0) var[blockIdx.x] = (T)(sf2 - s);
}

// dst (rows x cols, ldd) = src (rows x cols, lds); cols * sizeof(T) a multiple of 16, rows 16-byte
// aligned.  Streaming (non-temporal) 16-byte pieces, one row segment of 4 KiB per workgroup pass:
// the copy of a solved panel back into the matrix runs beside the trailing update, and the
// runtime's own 2-D blit moved it at 1.9 TB/s, holding CU slots 2.7x longer than necessary.
template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src,
                                                    int64_t lds, int64_t rows, int64_t cols) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int64_t pieces = cols * (int64_t)sizeof(T) / 16;   // 16-byte pieces per row
  const int64_t total = rows * pieces;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / pieces, c = e - r * pieces;
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(src + r * lds) + c);
    __builtin_nontemporal_store(v, reinterpret_cast<v4*>(dst + r * ldd) + c);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void set_diag_one_kernel_t(T* A, int64_t lda, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) A[i * lda + i] = (T)1;
}

template <typename T>
__global__ __launch_bounds__(256) void logdet_kernel(const T* __restrict__ A, int64_t lda, int64_t n,
                                                    double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += log((double)A[i * lda + i]);
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = 2.0 * t;
}

// D(16x16) = A(16x4) * B(4x16) through one v_mfma_f64_16x16x4_f64
__global__ __launch_bounds__(64) void mfma_probe_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  v4d c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

// same probe for v_mfma_f32_16x16x4_f32: D reg r of lane l = D[4(l>>4) + r][l&15]
__global__ __launch_bounds__(64) void mfma_probe_f32_kernel(const float* A, const float* B, float* D) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int l = threadIdx.x;
  const float a = A[(l & 15) * 4 + (l >> 4)];
  const float b = B[(l >> 4) * 16 + (l & 15)];
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// register-resident fp64 MFMA loop: 16 independent accumulators per wave
__global__ __launch_bounds__(256, 2) void mfma_loop_kernel(double* sink, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;  // keep the loop alive
}

// streaming copy: ONE 16-byte element per lane, one pass, as many workgroups as it takes, non-temporal
// loads and stores — the shape that reaches the ~6.3 TB/s class of MI355X_MICROARCH.md's float4 copy
// (tools/copy_bw.hip, round 3: 6.3 TB/s plain / 6.6 nt; grid-stride loops with 16 k workgroups stay at
// 4.6-5.0, four loads in flight per lane at 5.3-5.4).  count2 must be a multiple of 256.
__global__ __launch_bounds__(256) void copy_kernel(const double2* __restrict__ src,
                                                  double2* __restrict__ dst, int64_t count2) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count2)
    __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v2*>(src) + i),
                                reinterpret_cast<v2*>(dst) + i);
}

template <typename T>
__global__ __launch_bounds__(256) void fix_diag_kernel(T* A, int64_t lda, int n, int nvalid, double add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T* p = A + (int64_t)i * lda + i;
  *p = (i < nvalid) ? *p + (T)add : (T)1;
}

template <typename T>
__global__ __launch_bounds__(256) void unpermute_panel_kernel(const T* __restrict__ G, int64_t ldp, T* __restrict__ Pglob,
                                                             int64_t ldd, int nb, Deal dl, int p, int64_t maxcnt) {
  // blockIdx.y = trailing block b (global block g = p+1+b), blockIdx.x strides rows of the block
  typedef float v4 __attribute__((ext_vector_type(4)));  // 16-byte pieces whatever the element type
  constexpr int E = 16 / (int)sizeof(T);
  const int g = p + 1 + blockIdx.y;
  const int rr = dl.owner(g);
  const int64_t src_row0 = (int64_t)rr * maxcnt + (dl.local(g) - dl.upto(p, rr)) * nb;
  const int64_t dst_row0 = (int64_t)blockIdx.y * nb;
  const int c2 = nb / E;  // pieces per row
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)nb * c2;
       e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / c2;
    const int col = (int)(e - row * c2) * E;
    *reinterpret_cast<v4*>(Pglob + (dst_row0 + row) * ldd + col) =
        *reinterpret_cast<const v4*>(G + (src_row0 + row) * ldp + col);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_rhs_local_kernel(const T* __restrict__ y, int64_t n, int k, T* __restrict__ YTloc,
                                                            int64_t ldy, int nb, int nlb, Deal dl, int rank) {
  const int r = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < (int64_t)nlb * nb;
       j += (int64_t)gridDim.x * 256) {
    const int64_t lb = j / nb;
    const int64_t gi = dl.global(rank, lb) * nb + (j - lb * nb);
    YTloc[(int64_t)r * ldy + j] = (r < k && gi < n) ? y[gi * k + r] : (T)0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_local_kernel(const T* __restrict__ Loc, int64_t ldl, T* __restrict__ Full,
                                                           int64_t ldf, int nb, int nlb, Deal dl, int rank) {
  const int r = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < (int64_t)nlb * nb;
       j += (int64_t)gridDim.x * 256) {
    const int64_t lb = j / nb;
    const int64_t gi = dl.global(rank, lb) * nb + (j - lb * nb);
    Full[(int64_t)r * ldf + gi] = Loc[(int64_t)r * ldl + j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_block_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src,
                                                       int64_t lds, int rows, int cols, double sign) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)rows * cols;
       e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols;
    const int64_t c = e - r * cols;
    dst[r * ldd + c] += (T)sign * src[r * lds + c];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_scalar_kernel(T* p, int64_t count, double v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    p[i] += (T)v;
}

// dst[i] = op over q < P of src[q * count + i], in rank order (in-process transport: every rank
// that reduces gets bit-identical results); op 0 = sum, 1 = min
template <typename T>
__global__ __launch_bounds__(256) void reduce_ranks_kernel(const T* __restrict__ src, T* __restrict__ dst, int P,
                                                          int64_t count, int op) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    T v = src[i];
    for (int q = 1; q < P; ++q) {
      const T w = src[(int64_t)q * count + i];
      v = op == 1 ? (w < v ? w : v) : v + w;
    }
    dst[i] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void logdet_acc_kernel(const T* __restrict__ A, int64_t lda, int n,
                                                        double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += log((double)A[i * lda + i]);
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) out[0] += 2.0 * t;
}

}  // namespace

template <typename T>
void launch_fix_diag(T* A, int64_t lda, int n, int nvalid, double add, hipStream_t st) {
  hipLaunchKernelGGL(fix_diag_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, lda, n, nvalid, add);
}

template <typename T>
void launch_unpermute_panel(const T* G, int64_t ldp, T* dst, int64_t ldd, int nb, Deal dl, int p, int nblk, int64_t maxcnt,
                            hipStream_t st) {
  const int ntb = nblk - p - 1;
  if (ntb <= 0) return;
  hipLaunchKernelGGL(unpermute_panel_kernel<T>, dim3(64, (unsigned)ntb), dim3(256), 0, st, G, ldp, dst, ldd, nb, dl, p, maxcnt);
}

template <typename T>
void launch_pack_rhs_local(const T* y, int64_t n, int k, T* YTloc, int64_t ldy, int nb, int nlb, Deal dl, int rank, int R,
                           hipStream_t st) {
  if (nlb <= 0) return;
  const int64_t bx = ((int64_t)nlb * nb + 255) / 256;
  hipLaunchKernelGGL(pack_rhs_local_kernel<T>, dim3((unsigned)(bx > 1024 ? 1024 : bx), (unsigned)R), dim3(256), 0, st, y, n, k, YTloc, ldy, nb, nlb, dl, rank);
}

template <typename T>
void launch_scatter_local(const T* Loc, int64_t ldl, T* Full, int64_t ldf, int nb, int nlb, Deal dl, int rank, int R,
                          hipStream_t st) {
  if (nlb <= 0) return;
  const int64_t bx = ((int64_t)nlb * nb + 255) / 256;
  hipLaunchKernelGGL(scatter_local_kernel<T>, dim3((unsigned)(bx > 1024 ? 1024 : bx), (unsigned)R), dim3(256), 0, st, Loc, ldl, Full, ldf, nb, nlb, dl, rank);
}

template <typename T>
void launch_add_block(T* dst, int64_t ldd, const T* src, int64_t lds, int rows, int cols, double sign, hipStream_t st) {
  const int64_t total = (int64_t)rows * cols;
  if (total <= 0) return;
  const int64_t bx = (total + 255) / 256;
  hipLaunchKernelGGL(add_block_kernel<T>, dim3((unsigned)(bx > 2048 ? 2048 : bx)), dim3(256), 0, st, dst, ldd, src, lds, rows, cols, sign);
}

template <typename T>
void launch_add_scalar(T* p, int64_t count, double v, hipStream_t st) {
  if (count <= 0) return;
  const int64_t bx = (count + 255) / 256;
  hipLaunchKernelGGL(add_scalar_kernel<T>, dim3((unsigned)(bx > 1024 ? 1024 : bx)), dim3(256), 0, st, p, count, v);
}

template <typename T>
void launch_reduce_ranks(const T* src, T* dst, int P, int64_t count, int op, hipStream_t st) {
  if (count <= 0) return;
  const int64_t bx = (count + 255) / 256;
  hipLaunchKernelGGL(reduce_ranks_kernel<T>, dim3((unsigned)(bx > 2048 ? 2048 : bx)), dim3(256), 0, st, src, dst, P, count, op);
}

template <typename T>
void launch_logdet_acc(const T* A, int64_t lda, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(logdet_acc_kernel<T>, dim3(1), dim3(256), 0, st, A, lda, n, out);
}

template <typename T>
void launch_pack_rhs(const T* y, int64_t n, int k, T* YT, int64_t ld, int64_t npad, int R, hipStream_t st) {
  const int64_t bx = (npad + 255) / 256;
  hipLaunchKernelGGL(pack_rhs_kernel<T>, dim3((unsigned)(bx > 1024 ? 1024 : bx), (unsigned)R), dim3(256), 0, st, y, n, k, YT, ld, npad);
}

template <typename T>
void launch_unpack_rhs(const T* YT, int64_t ld, int64_t n, int k, double scale, T* out, hipStream_t st) {
  const int64_t total = n * k;
  if (total <= 0) return;
  hipLaunchKernelGGL(unpack_rhs_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, YT, ld, n, k, scale, out);
}

template <typename T>
void launch_var_rows(const T* VT, int64_t ld, int64_t m, int64_t ncols, double sf2, T* var, hipStream_t st) {
  if (m <= 0) return;
  hipLaunchKernelGGL(var_rows_kernel<T>, dim3((unsigned)m), dim3(256), 0, st, VT, ld, ncols, sf2, var);
}

// ---- timing perturbation (test hook) -------------------------------------------------------------
// Every result of this library is deterministic whatever the relative timing of its streams — IF
// every cross-stream dependency is expressed.  gpx_debug_set_delay(seed != 0) makes the launchers
// put a short spin kernel (0..~170 us, one wave, bounded by the shader clock) in front of a random
// third of their launches, on the stream of that launch: streams then race each other differently
// on every seed, and a missing event shows up as a result that changes (tests/test_delay_gpu.py
// demands bit-identical outputs).  Off (seed 0) it is one relaxed atomic load per launch.
namespace {
std::atomic<uint64_t> g_delay_state{0};
// Parks a stream until a device-side counter has reached `target` (gemm_nt_fused_kernel's strip
// slots).  One wave, lane 0 polls with a device-scope load every ~2 us; every wave reaches the
// exit: after 2^23 polls (~15 s) it records the failure in *info (INT_MIN: reported by the API call)
// and returns, and every later wait of the same call returns at once.  The kernel boundary behind it
// is the acquire for whatever the stream runs next.
__global__ void wait_counter_kernel(const unsigned* ctr, unsigned target, int* info) {
  if (threadIdx.x == 0) {
    unsigned polls = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) break;  // an earlier wait gave up
      __builtin_amdgcn_s_sleep(64);
      if (++polls > (1u << 23)) {
        atomicMin(info, (int)0x80000000);
        break;
      }
    }
  }
  __threadfence();
}

// Self-test of the device-flag hand-over (flag_handover_probe, gpx_api.hip): the wait side polls for at most
// max_polls x ~2 us and reports whether it SAW the flag; the set side publishes it.  Where kernels of different
// streams do not run concurrently (AMD_SERIALIZE_KERNEL, counter collection, a tool that funnels the streams into
// one queue) the wait side runs out of polls before the set side starts: seen stays 0.
__global__ void flag_probe_wait_kernel(const unsigned* flag, unsigned* seen, unsigned max_polls) {
  if (threadIdx.x == 0) {
    unsigned polls = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && polls < max_polls) {
      __builtin_amdgcn_s_sleep(64);
      ++polls;
    }
    *seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0 && polls < max_polls ? 1u : 0u;
  }
}
__global__ void flag_probe_set_kernel(unsigned* flag) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void spin_kernel(long long cycles) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) {
  }
}
}  // namespace

void debug_set_delay(uint64_t seed) { g_delay_state.store(seed); }

void debug_delay(hipStream_t st) {
  uint64_t x = g_delay_state.load(std::memory_order_relaxed);
  if (x == 0) return;
  uint64_t nx;
  do {  // xorshift64*, advanced atomically (rank threads of a group launch concurrently)
    nx = x;
    nx ^= nx >> 12;
    nx ^= nx << 25;
    nx ^= nx >> 27;
    if (nx == 0) nx = 0x9E3779B97F4A7C15ull;
  } while (!g_delay_state.compare_exchange_weak(x, nx, std::memory_order_relaxed));
  const uint64_t r = nx * 0x2545F4914F6CDD1Dull;
  if ((r >> 60) % 3 != 0) return;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (long long)((r >> 20) % 400000));
}

template <typename T>
void launch_copy2d(T* dst, int64_t ldd, const T* src, int64_t lds, int64_t rows, int64_t cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return;
  debug_delay(st);
  const int64_t total = rows * (cols * (int64_t)sizeof(T) / 16);
  const int64_t bx = (total + 255) / 256;
  hipLaunchKernelGGL(copy2d_kernel<T>, dim3((unsigned)(bx > 8192 ? 8192 : bx)), dim3(256), 0, st, dst, ldd, src, lds, rows, cols);
}

void launch_flag_probe_wait(const unsigned* flag, unsigned* seen, unsigned max_polls, hipStream_t st) {
  hipLaunchKernelGGL(flag_probe_wait_kernel, dim3(1), dim3(64), 0, st, flag, seen, max_polls);
}
void launch_flag_probe_set(unsigned* flag, hipStream_t st) {
  hipLaunchKernelGGL(flag_probe_set_kernel, dim3(1), dim3(64), 0, st, flag);
}

void launch_wait_counter(const unsigned* ctr, unsigned target, int* info, hipStream_t st) {
  hipLaunchKernelGGL(wait_counter_kernel, dim3(1), dim3(64), 0, st, ctr, target, info);
}

template <typename T>
void launch_set_diag_one_t(T* A, int64_t lda, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(set_diag_one_kernel_t<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, lda, n);
}

template <typename T>
void launch_logdet(const T* A, int64_t lda, int64_t n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(logdet_kernel<T>, dim3(1), dim3(256), 0, st, A, lda, n, out);
}

#define GPX_INSTANTIATE_MISC(T)                                                                    \
  template void launch_pack_rhs<T>(const T*, int64_t, int, T*, int64_t, int64_t, int, hipStream_t); \
  template void launch_unpack_rhs<T>(const T*, int64_t, int64_t, int, double, T*, hipStream_t);     \
  template void launch_var_rows<T>(const T*, int64_t, int64_t, int64_t, double, T*, hipStream_t);   \
  template void launch_logdet<T>(const T*, int64_t, int64_t, double*, hipStream_t);                 \
  template void launch_set_diag_one_t<T>(T*, int64_t, int64_t, hipStream_t);                        \
  template void launch_copy2d<T>(T*, int64_t, const T*, int64_t, int64_t, int64_t, hipStream_t);           \
  template void launch_fix_diag<T>(T*, int64_t, int, int, double, hipStream_t);                           \
  template void launch_unpermute_panel<T>(const T*, int64_t, T*, int64_t, int, Deal, int, int, int64_t, hipStream_t); \
  template void launch_pack_rhs_local<T>(const T*, int64_t, int, T*, int64_t, int, int, Deal, int, int, hipStream_t); \
  template void launch_scatter_local<T>(const T*, int64_t, T*, int64_t, int, int, Deal, int, int, hipStream_t); \
  template void launch_add_block<T>(T*, int64_t, const T*, int64_t, int, int, double, hipStream_t);        \
  template void launch_add_scalar<T>(T*, int64_t, double, hipStream_t);                                    \
  template void launch_reduce_ranks<T>(const T*, T*, int, int64_t, int, hipStream_t);                      \
  template void launch_logdet_acc<T>(const T*, int64_t, int, double*, hipStream_t);
GPX_INSTANTIATE_MISC(double)
GPX_INSTANTIATE_MISC(float)

void launch_mfma_probe_f32(const float* A, const float* B, float* D, hipStream_t st) {
  hipLaunchKernelGGL(mfma_probe_f32_kernel, dim3(1), dim3(64), 0, st, A, B, D);
}

void launch_mfma_probe(const double* A, const double* B, double* D, hipStream_t st) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, st, A, B, D);
}

void launch_mfma_loop(double* sink, int iters, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(mfma_loop_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sink, iters);
}

void launch_copy(const double* src, double* dst, int64_t count, hipStream_t st) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((count / 2 + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst), count / 2);
}

}  // namespace gpx
