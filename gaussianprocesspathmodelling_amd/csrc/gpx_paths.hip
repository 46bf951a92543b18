// gpx_paths.hip — batched path-to-centroid distance, the
// reference's actual hot loop: trajectories.calc_distance (GPmap.py:114-121),
//     D[p][c] = sum_{i<L} || (x_p[i], y_p[i]) - (x_c[i], y_c[i]) ||_2 ,
// called (k+1)*P times per Lloyd iteration by kmeansclustering (GPmap.py:72-80) at 10.6 us
// per 2-D point of interpreter overhead.  Here: one wave per path, lane i owns point i
// (L <= 64; the reference keeps only paths of exactly 33 points, GPmap.py:189), the path
// is read once (coalesced 16 B per lane), centroids sit in LDS, and each D[p][c] is a
// wave reduction.  HBM-bound: 16*L bytes read per path, 8*C written.
#include "gpx_internal.h"

namespace gpx {
namespace {

constexpr int MAXC = 64;   // centroids per launch staged in LDS
constexpr int MAXL = 64;

__global__ __launch_bounds__(256) void path_distance_kernel(const double2* __restrict__ paths,
                                                           int64_t P, const double2* __restrict__ cents,
                                                           int C, int L, double* __restrict__ D,
                                                           int64_t ldd) {
  __shared__ double2 cs[MAXC * MAXL];
  for (int e = threadIdx.x; e < C * L; e += 256) cs[e] = cents[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < P; p += (int64_t)gridDim.x * 4) {
    double2 q = {0.0, 0.0};
    if (lane < L) q = paths[p * L + lane];
    for (int c = 0; c < C; ++c) {
      double v = 0.0;
      if (lane < L) {
        const double dx = q.x - cs[c * L + lane].x, dy = q.y - cs[c * L + lane].y;
        v = sqrt(dx * dx + dy * dy);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) D[p * ldd + c] = v;
    }
  }
}

}  // namespace

// paths (P, L, 2) and cents (C, L, 2) device arrays of (x, y); D (P, ldd) <- distances of
// centroids [c0, c0 + C) into columns [c0, c0 + C).
void launch_path_distance(const double* paths, int64_t P, const double* cents, int C, int L, double* D,
                          int64_t ldd, hipStream_t st) {
  if (P <= 0 || C <= 0) return;
  const int64_t blocks = (P + 3) / 4;
  hipLaunchKernelGGL(path_distance_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0,
                     st, reinterpret_cast<const double2*>(paths), P,
                     reinterpret_cast<const double2*>(cents), C, L, D, ldd);
}

}  // namespace gpx
