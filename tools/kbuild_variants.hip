// Kernel-build variants side by side (round 4): the symmetric RBF build at N = 65536, d = 3, fp64 —
// the library's kernel (V0) against variants of its staging / unrolling / tile walk, each timed with hipEvents and
// compared bit for bit with V0.  Diagnostic only; what wins goes into csrc/gpx_kbuild.hip.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/kbv tools/kbuild_variants.hip && /tmp/kbv [N]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int KT = 64, D = 3;
typedef double pair_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tri_coords(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > t) --i;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

__device__ __forceinline__ double kfun(double r2, double sf2) { return sf2 * exp(-0.5 * r2); }

// one 64x64 tile: LDSW = doubles of LDS per point array, UNR = unroll of the row loop, NT = non-temporal stores
template <int LDSW, int UNR, bool NT>
__device__ __forceinline__ void tile(const double* __restrict__ Xs, int64_t m, int ti, int tj, double sf2, double diag_add,
                                     double* __restrict__ K, int64_t ld, double* xa, double* xb) {
  const int64_t i0 = (int64_t)ti * KT, j0 = (int64_t)tj * KT;
  const int tid = threadIdx.x;
  for (int e = tid; e < KT * D; e += 256) {
    xa[e] = Xs[i0 * D + e];
    xb[e] = Xs[j0 * D + e];
  }
  __syncthreads();
  const int c2 = (tid & 31) * 2, rg = tid >> 5;
  double bj0[D], bj1[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    bj0[c] = xb[c2 * D + c];
    bj1[c] = xb[(c2 + 1) * D + c];
  }
  const int64_t col0 = j0 + c2, col1 = col0 + 1;
#pragma unroll UNR
  for (int r = 0; r < 8; ++r) {
    const int il = rg + 8 * r;
    const int64_t row = i0 + il;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const double a = xa[il * D + c];
      const double e0 = a - bj0[c], e1 = a - bj1[c];
      s0 += e0 * e0;
      s1 += e1 * e1;
    }
    double v0 = kfun(s0, sf2), v1 = kfun(s1, sf2);
    if (row == col0) v0 += diag_add;
    if (row == col1) v1 += diag_add;
    if (row >= m || col0 >= m) v0 = (row == col0) ? 1.0 : 0.0;
    if (row >= m || col1 >= m) v1 = (row == col1) ? 1.0 : 0.0;
    pair_t out = {v0, v1};
    pair_t* dst = reinterpret_cast<pair_t*>(K + row * ld + col0);
    if (NT)
      __builtin_nontemporal_store(out, dst);
    else
      *dst = out;
  }
}

template <int LDSW, int UNR, bool NT, int MINB>
__global__ __launch_bounds__(256, MINB) void k_tile(const double* __restrict__ Xs, int64_t m, double sf2, double diag_add,
                                                    double* __restrict__ K, int64_t ld) {
  __shared__ double xa[KT * LDSW], xb[KT * LDSW];
  int ti, tj;
  tri_coords((int64_t)blockIdx.x, ti, tj);
  tile<LDSW, UNR, NT>(Xs, m, ti, tj, sf2, diag_add, K, ld, xa, xb);
}

// persistent walk: each workgroup takes tiles b, b + gridDim, ...
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_persist(const double* __restrict__ Xs, int64_t m, double sf2, double diag_add,
                                                 double* __restrict__ K, int64_t ld, int64_t ntiles) {
  __shared__ double xa[2][KT * D], xb[2][KT * D];
  int buf = 0;
  for (int64_t b = blockIdx.x; b < ntiles; b += gridDim.x, buf ^= 1) {
    int ti, tj;
    tri_coords(b, ti, tj);
    tile<D, UNR, NT>(Xs, m, ti, tj, sf2, diag_add, K, ld, xa[buf], xb[buf]);  // alternating buffers: one barrier per tile
  }
}

// a tile ROW strip per workgroup: 64 rows x (up to) `W` tile columns, the row points loaded once
template <int UNR, int W>
__global__ __launch_bounds__(256) void k_strip(const double* __restrict__ Xs, int64_t m, double sf2, double diag_add,
                                               double* __restrict__ K, int64_t ld, int TT) {
  // strips: for tile row ti the columns [0, ti] in chunks of W; enumerate (ti, chunk) by a simple 2-D grid and mask
  const int ti = blockIdx.y, ch = blockIdx.x;
  if (ch * W > ti) return;
  __shared__ double xa[KT * D], xb[2][KT * D];
  const int64_t i0 = (int64_t)ti * KT;
  const int tid = threadIdx.x;
  for (int e = tid; e < KT * D; e += 256) xa[e] = Xs[i0 * D + e];
  const int c2 = (tid & 31) * 2, rg = tid >> 5;
  int buf = 0;
  for (int tj = ch * W; tj < (ch + 1) * W && tj <= ti; ++tj, buf ^= 1) {
    const int64_t j0 = (int64_t)tj * KT;
    for (int e = tid; e < KT * D; e += 256) xb[buf][e] = Xs[j0 * D + e];
    __syncthreads();
    double bj0[D], bj1[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      bj0[c] = xb[buf][c2 * D + c];
      bj1[c] = xb[buf][(c2 + 1) * D + c];
    }
    const int64_t col0 = j0 + c2, col1 = col0 + 1;
#pragma unroll UNR
    for (int r = 0; r < 8; ++r) {
      const int il = rg + 8 * r;
      const int64_t row = i0 + il;
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double a = xa[il * D + c];
        const double e0 = a - bj0[c], e1 = a - bj1[c];
        s0 += e0 * e0;
        s1 += e1 * e1;
      }
      double v0 = kfun(s0, sf2), v1 = kfun(s1, sf2);
      if (row == col0) v0 += diag_add;
      if (row == col1) v1 += diag_add;
      if (row >= m || col0 >= m) v0 = (row == col0) ? 1.0 : 0.0;
      if (row >= m || col1 >= m) v1 = (row == col1) ? 1.0 : 0.0;
      pair_t out = {v0, v1};
      *reinterpret_cast<pair_t*>(K + row * ld + col0) = out;
    }
  }
}

#define CHK(x)                                                              \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
  const int64_t ld = N + 16, TT = N / KT, ntiles = TT * (TT + 1) / 2;
  std::vector<double> X((size_t)N * D);
  uint64_t s = 12345;
  for (auto& v : X) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v = (double)(s >> 11) / 9007199254740992.0 / 0.25;
  }
  double *dX, *K0, *K1;
  CHK(hipMalloc(&dX, X.size() * 8));
  CHK(hipMalloc(&K0, (size_t)N * ld * 8));
  CHK(hipMalloc(&K1, (size_t)N * ld * 8));
  CHK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemset(K0, 0, (size_t)N * ld * 8));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  const double bytes = 8.0 * ((double)N * (N + 1) / 2 + (double)N * D);
  int* dDiff;
  CHK(hipMalloc(&dDiff, 4));
  auto run = [&](const char* name, auto launch, double* out) -> int {
    CHK(hipMemset(out, 0, (size_t)N * ld * 8));
    launch(out);
    CHK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 6; ++it) {
      CHK(hipEventRecord(a));
      launch(out);
      CHK(hipEventRecord(b));
      CHK(hipEventSynchronize(b));
      float ms;
      CHK(hipEventElapsedTime(&ms, a, b));
      best = ms < best ? ms : best;
      sum += ms;
    }
    // bitwise comparison of sampled rows with V0's output (first call: out == K0)
    int bad = 0;
    if (out != K0) {
      std::vector<double> r0((size_t)ld), r1((size_t)ld);
      for (int64_t row : {(int64_t)0, (int64_t)63, (int64_t)64, N / 3, N / 2 + 17, N - 65, N - 1}) {
        CHK(hipMemcpy(r0.data(), K0 + row * ld, (size_t)ld * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r1.data(), out + row * ld, (size_t)ld * 8, hipMemcpyDeviceToHost));
        for (int64_t c = 0; c <= row; ++c) bad += r0[(size_t)c] != r1[(size_t)c];
      }
    }
    printf("%-44s best %.3f ms  mean %.3f ms  %.0f GB/s best  %s\n", name, best, sum / 6, bytes / (best * 1e-3) / 1e9,
           out == K0 ? "(reference)" : bad ? "DIFFERS" : "bit-identical on sampled rows");
    return 0;
  };
  const dim3 g((unsigned)ntiles), blk(256);
#define V(name, ...) \
  if (run(name, [&](double* o) { __VA_ARGS__; }, strcmp(name, "V0") ? K1 : K0)) return 1;
  auto L = [&](auto kern, double* o) { hipLaunchKernelGGL(kern, g, blk, 0, 0, dX, N, 1.5, 0.01, o, ld); };
  if (run("V0 library: LDS 2x16 KB static, unroll 2", [&](double* o) { L(k_tile<32, 2, false, 1>, o); }, K0)) return 1;
  if (run("V1 LDS sized by d (3 KB), unroll 2", [&](double* o) { L(k_tile<D, 2, false, 1>, o); }, K1)) return 1;
  if (run("V2 LDS by d, unroll 4", [&](double* o) { L(k_tile<D, 4, false, 1>, o); }, K1)) return 1;
  if (run("V3 LDS by d, unroll 8", [&](double* o) { L(k_tile<D, 8, false, 1>, o); }, K1)) return 1;
  if (run("V4 LDS by d, unroll 1", [&](double* o) { L(k_tile<D, 1, false, 1>, o); }, K1)) return 1;
  if (run("V5 LDS by d, unroll 2, non-temporal", [&](double* o) { L(k_tile<D, 2, true, 1>, o); }, K1)) return 1;
  if (run("V6 LDS by d, unroll 8, non-temporal", [&](double* o) { L(k_tile<D, 8, true, 1>, o); }, K1)) return 1;
  if (run("V7 LDS by d, unroll 2, min 8 blocks/CU", [&](double* o) { L(k_tile<D, 2, false, 8>, o); }, K1)) return 1;
  for (int wg : {256 * 4, 256 * 8, 256 * 16, 256 * 32})
    for (int nt = 0; nt < 2; ++nt) {
      char nm[96];
      snprintf(nm, sizeof nm, "V8 persistent %d workgroups, unroll 2%s", wg, nt ? ", non-temporal" : "");
      if (run(nm, [&](double* o) {
            if (nt)
              hipLaunchKernelGGL((k_persist<2, true>), dim3(wg), blk, 0, 0, dX, N, 1.5, 0.01, o, ld, ntiles);
            else
              hipLaunchKernelGGL((k_persist<2, false>), dim3(wg), blk, 0, 0, dX, N, 1.5, 0.01, o, ld, ntiles);
          }, K1))
        return 1;
    }
  if (run("V9 row strips of 8 tiles, unroll 2", [&](double* o) {
        hipLaunchKernelGGL((k_strip<2, 8>), dim3((unsigned)((TT + 7) / 8), (unsigned)TT), blk, 0, 0, dX, N, 1.5, 0.01, o, ld, (int)TT);
      }, K1))
    return 1;
  if (run("V10 row strips of 32 tiles, unroll 2", [&](double* o) {
        hipLaunchKernelGGL((k_strip<2, 32>), dim3((unsigned)((TT + 31) / 32), (unsigned)TT), blk, 0, 0, dX, N, 1.5, 0.01, o, ld, (int)TT);
      }, K1))
    return 1;
  if (run("V11 row strips of 8 tiles, unroll 4", [&](double* o) {
        hipLaunchKernelGGL((k_strip<4, 8>), dim3((unsigned)((TT + 7) / 8), (unsigned)TT), blk, 0, 0, dX, N, 1.5, 0.01, o, ld, (int)TT);
      }, K1))
    return 1;
  return 0;
}
