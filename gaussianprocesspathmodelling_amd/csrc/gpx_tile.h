// gpx_tile.h — the MFMA tile engine shared by the dense kernels (gpx_blas.hip) and the fused
// K^-1 trace kernel of the marginal-likelihood gradient (gpx_grad.hip): element traits, the NT
// engine with LDS-DMA staging, the accumulator store and the XCD-aware, hole-free tile maps.
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
//
// MFMA layouts (cdna_hip_programming.md §3), verified on hardware by gpx_mfma_probe:
//   A lane l = A[l&15][l>>4],  B lane l = B[l>>4][l&15]   (both types)
//   D reg r of lane l = D[(l>>4) + 4r][l&15]  for f64,   D[4(l>>4) + r][l&15]  for f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpx_internal.h"

namespace gpx {
namespace {

template <typename T>
struct Num;
template <>
struct Num<double> {
  typedef double v4 __attribute__((ext_vector_type(4)));    // accumulator of one 16x16 tile
  typedef double slot __attribute__((ext_vector_type(2)));  // one 16-byte LDS slot
  static constexpr int SLOT = 2;                            // elements per slot
  static constexpr int BK = 16;  // k-step: 8 slots = one 128-B line per row
  static constexpr int KS64 = 4; // k-steps per barrier of the 64-tile engine: 64 elements of K per barrier
  static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int l4, int r) { return l4 + 4 * r; }
  static __device__ __forceinline__ double rsq(double x) { return rsqrt(x); }
};
template <>
struct Num<float> {
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float slot __attribute__((ext_vector_type(4)));
  static constexpr int SLOT = 4;
  static constexpr int BK = 32;
  static constexpr int KS64 = 2;
  static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int l4, int r) { return 4 * l4 + r; }
  static __device__ __forceinline__ float rsq(float x) { return rsqrtf(x); }
};

// ---- NT tile engine with LDS-DMA staging ------------------------------------------------
// acc += A(BM x K) * B(BN x K)^T, both row-major with k contiguous.  Per k-step (one
// 128-byte line per row: 16 doubles / 32 floats) every wave issues BM/32 + BN/32
// `global_load_lds_dwordx4` (1 KiB = 8 rows x 128 B each, straight into LDS: no staging
// VGPRs, no ds_write) for step t+1 before the MFMAs of step t; `__syncthreads()` drains
// them (vmcnt(0)) once per step.
// LDS image: [row][8 slots of 16 B], physical slot = logical slot ^ swz(row).  LDS-DMA
// writes lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
// fragment reads (cdna_hip_programming.md rule 21).  swz() is chosen so that a
// ds_read_b128 lane group (rows {0-3,12-15} at k-group g with rows {4-11} at g+1, and the
// three analogous groups) hits 16 distinct 16-byte bank slots: conflict-free (measured:
// SQ_LDS_BANK_CONFLICT = 0).
// k permutation: lane group g = l>>4 consumes k = g*2*SLOT + s at MFMA step s (instead of
// 4s+g) — the same for A and B, so one ds_read_b128 pair per fragment feeds every step.
__device__ __forceinline__ int swz(int row) {
  const int t = ((row >> 1) + 2) & 7;
  return ((t & 3) << 1) | (t >> 2);
}

// KSUB k-steps share ONE barrier (KSUB = 2: the "BK = 64" fp32 variant of round 3): a stage then holds
// KSUB consecutive 128-byte lines per row as KSUB images of the layout below, back to back.
template <typename T, int BM, int BN, int KSUB = 1>
struct TileShapeG {
  static constexpr int BK = Num<T>::BK;
  static constexpr int A_STAGE = BM * BK * KSUB, B_STAGE = BN * BK * KSUB;
  static constexpr int SMEM_ELEMS = 2 * (A_STAGE + B_STAGE);
};

#define GPX_GLDS16(gptr, lptr)                                                             \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),  \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// WVM = waves along M (2: 256-thread workgroup, 2 x 2 waves; 4: 512 threads, 4 x 2 waves).
template <typename T, int BM, int BN, int WVM = 2, int KSUB = 1>
__device__ __forceinline__ void gemm_tile_g(const T* A, int64_t lda, const T* B, int64_t ldb, int K,
                                            typename Num<T>::v4 (&acc)[BM / (16 * WVM)][BN / 32], T* smem) {
  using S = TileShapeG<T, BM, BN, KSUB>;
  using slot_t = typename Num<T>::slot;
  constexpr int BK = S::BK, SL = Num<T>::SLOT;
  constexpr int MT = BM / (16 * WVM), NT = BN / 32, WM = BM / WVM, WN = BN / 2;
  constexpr int IA = BM / (16 * WVM), IB = BN / (16 * WVM);  // DMA instructions per wave per k-step
  constexpr int RQ = 8 * BK;                 // elements per DMA instruction (8 rows)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  T* As = smem;
  T* Bs = smem + 2 * S::A_STAGE;

  // DMA: instruction q of this wave covers rows 8*(wave*I + q) .. +7; lane -> (row, slot)
  const int drow = lane >> 3, dslot = lane & 7;
  const T* ga[IA];
  const T* gb[IB];
#pragma unroll
  for (int q = 0; q < IA; ++q) {
    const int row = (wave * IA + q) * 8 + drow;
    ga[q] = A + (int64_t)row * lda + (dslot ^ swz(row)) * SL;
  }
#pragma unroll
  for (int q = 0; q < IB; ++q) {
    const int row = (wave * IB + q) * 8 + drow;
    gb[q] = B + (int64_t)row * ldb + (dslot ^ swz(row)) * SL;
  }
  T* const la = As + wave * IA * RQ;  // wave-uniform LDS destinations
  T* const lb = Bs + wave * IB * RQ;

  // fragment reads: row = w*W + t*16 + l15, logical slots 2*l4 and 2*l4+1
  const int sw = swz(l15);
  const int a_off0 = (wr * WM + l15) * BK + ((2 * l4) ^ sw) * SL;
  const int a_off1 = (wr * WM + l15) * BK + ((2 * l4 + 1) ^ sw) * SL;
  const int b_off0 = (wc * WN + l15) * BK + ((2 * l4) ^ sw) * SL;
  const int b_off1 = (wc * WN + l15) * BK + ((2 * l4 + 1) ^ sw) * SL;

#pragma unroll
  for (int u = 0; u < KSUB; ++u) {
#pragma unroll
    for (int q = 0; q < IA; ++q) GPX_GLDS16(ga[q] + u * BK, la + u * BM * BK + q * RQ);
#pragma unroll
    for (int q = 0; q < IB; ++q) GPX_GLDS16(gb[q] + u * BK, lb + u * BN * BK + q * RQ);
  }
  __syncthreads();

  // Per k-step: 8 first-half fragment reads, then the MFMAs with everything else issued in
  // their shadow: the second-half fragment reads inside the first quarter of the burst, the
  // DMA of step t+1 inside the second (sched_group_barrier: 0x8 MFMA, 0x100 DS read, 0x20
  // VMEM read).  Measured on the SYRK: 65.1 TF with DMA + all 16 reads clumped before the
  // burst -> 67.9 TF interleaved.  (A 4-stage, one-workgroup-per-CU variant with counted
  // vmcnt and a second fragment set reached only 59.6 TF: per-wave wait time fell from 8.7 %
  // to 3.7 %, but nothing covers the tile epilogue and the launch tail any more.)  The DMA is unconditional (clamped to the last step, landing in
  // the buffer nobody reads again) so that the loop body stays one basic block.
  constexpr int HALF = SL * MT * NT;  // MFMAs per half step
  const int KT = K / (BK * KSUB);
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    const int64_t ko = (int64_t)(kt + 1 < KT ? kt + 1 : kt) * (BK * KSUB);
#pragma unroll
    for (int u = 0; u < KSUB; ++u) {
      const T* Ab = As + buf * S::A_STAGE + u * BM * BK;
      const T* Bb = Bs + buf * S::B_STAGE + u * BN * BK;
      slot_t a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a0[m] = *reinterpret_cast<const slot_t*>(Ab + a_off0 + m * 16 * BK);
#pragma unroll
      for (int n = 0; n < NT; ++n) b0[n] = *reinterpret_cast<const slot_t*>(Bb + b_off0 + n * 16 * BK);
#pragma unroll
      for (int m = 0; m < MT; ++m) a1[m] = *reinterpret_cast<const slot_t*>(Ab + a_off1 + m * 16 * BK);
#pragma unroll
      for (int n = 0; n < NT; ++n) b1[n] = *reinterpret_cast<const slot_t*>(Bb + b_off1 + n * 16 * BK);
      {
#pragma unroll
        for (int q = 0; q < IA; ++q)
          GPX_GLDS16(ga[q] + ko + u * BK, la + (buf ^ 1) * S::A_STAGE + u * BM * BK + q * RQ);
#pragma unroll
        for (int q = 0; q < IB; ++q)
          GPX_GLDS16(gb[q] + ko + u * BK, lb + (buf ^ 1) * S::B_STAGE + u * BN * BK + q * RQ);
      }
#pragma unroll
      for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = Num<T>::mfma(a0[m][s], b0[n][s], acc[m][n]);
#pragma unroll
      for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = Num<T>::mfma(a1[m][s], b1[n][s], acc[m][n]);
      __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);  // first-half fragments first
      // first quarter of the burst: one second-half fragment read per 2 MFMAs; second
      // quarter: the DMA of step t+1, one per 2 MFMAs; the second half is pure MFMA (covers
      // the DMA latency together with the co-resident workgroup's burst)
#pragma unroll
      for (int i = 0; i < MT + NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, HALF / (2 * (MT + NT)), 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < IA + IB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, HALF / (2 * (IA + IB)), 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
      }
      if (KSUB > 1) __builtin_amdgcn_sched_group_barrier(0x8, HALF, 0);  // the rest of this sub-step's MFMAs stay here
    }
    // keep the MFMAs ABOVE the barrier: hipcc otherwise sinks them below the vmcnt(0)
    // drain of __syncthreads() and the DMA latency is exposed on every k-step
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
}

template <typename V4, int MT, int NT>
__device__ __forceinline__ void zero_acc(V4 (&acc)[MT][NT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (V4){0, 0, 0, 0};
}

// MODE 0: C -= acc;  MODE 1: C = acc;  C -> tile origin.
// MODE 0 issues one no-return `global_atomic_add_f64/f32` of -acc per element instead of
// load -> wait -> subtract -> store: every element of C receives exactly ONE addend per
// launch (tiles are disjoint, launches are stream-ordered), so the result is bit-identical
// to the subtraction and deterministic, but the wave never waits for C to arrive
// (SYRK +0.7 %; the load/store epilogue it replaces cost 2.9 % in the ablation).  C is
// always the library's own hipMalloc'ed (coarse-grained) memory, where the hardware
// floating-point atomics are valid.
template <typename T, int BM, int BN, int MODE, int WVM = 2>
__device__ __forceinline__ void store_tile(T* C, int64_t ldc,
                                           const typename Num<T>::v4 (&acc)[BM / (16 * WVM)][BN / 32]) {
  constexpr int NT = BN / 32;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l4 = lane >> 4;
  T* Cw = C + (int64_t)(wr * (BM / WVM)) * ldc + wc * (BN / 2) + (lane & 15);
#pragma unroll
  for (int m = 0; m < BM / (16 * WVM); ++m) {
    T* Cm = Cw + (int64_t)(m * 16) * ldc;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T* dst = &Cm[(int64_t)Num<T>::drow(l4, r) * ldc + n * 16];
        if (MODE == 0)
          unsafeAtomicAdd(dst, -acc[m][n][r]);
        else
          *dst = acc[m][n][r];
      }
  }
}

// blocks are dealt round-robin over the 8 XCDs; give each XCD one contiguous chunk of
// the logical tile order so that neighbouring tiles share an L2 (speed only).
__host__ __device__ __forceinline__ int64_t xcd_chunk_id(int64_t bid, int64_t nblk) {
  const int64_t q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Single-precision estimate + exact integer correction (no fp64 VALU in a prologue that runs beside
// a partner saturating the fp64 MFMA pipe).
__host__ __device__ __forceinline__ void tri_coords(int64_t t, int& ti, int& tj) {
  int64_t i = (int64_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  if (i < 0) i = 0;
  while (i * (i + 1) / 2 > t) --i;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  ti = (int)i;
  tj = (int)(t - i * (i + 1) / 2);
}

// Logical tile order: 8x8 super-tiles (64 tiles = what one XCD runs concurrently at 2
// workgroups per CU), so the tiles in flight on an XCD share 8 A row-slices and 8 B
// row-slices through its L2 instead of re-fetching the panel per tile.
//   TRI : the lower triangle (m == n) without holes: the full super-tiles below the
//         diagonal, then the valid tiles of the diagonal super-tiles (see tile_coords).
//   !TRI: rectangular super-tile grid (sh x 64/sh tiles each); mask_lower 1 also drops
//         tiles with tj > ti (look-ahead strip of the SYRK), 2 applies the block-cyclic
//         row map of the sharded trailing update.
// (Deal / BcMask / bc_brow — the dealing of row blocks over the ranks of a shard — live in gpx_internal.h: the host side
//  of the shard uses them too)
template <bool TRI>
__host__ __device__ __forceinline__ bool tile_coords(int64_t lin, int tiles_m, int tiles_n, int sh,
                                            int mask_lower, const BcMask& bc, int& ti, int& tj) {
  const int64_t st = lin >> 6;
  const int inner = (int)(lin & 63);
  if (TRI) {
    // No holes: first every FULL super-tile strictly below the diagonal of the super-tile
    // grid, then the 36 valid tiles of each diagonal super-tile packed back to back.  With
    // the diagonal super-tiles enumerated in place, their 28 masked slots exited at once and
    // ran one whole tile ahead of their neighbours, which scrambles the k-phase the tiles
    // of a super-tile need to share their panel rows through L2.
    const int S = (tiles_m + 7) >> 3;
    const int64_t full = (int64_t)S * (S - 1) / 2 * 64;
    if (lin < full) {
      int i, j;
      tri_coords(st, i, j);  // strictly lower: super-row i + 1, super-column j
      ti = (i + 1) * 8 + (inner >> 3);
      tj = j * 8 + (inner & 7);
    } else {
      const int64_t id2 = lin - full;
      const int k = (int)(id2 / 36), e = (int)(id2 - (int64_t)k * 36);
      int r = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      if (r * (r + 1) / 2 > e) --r;
      if ((r + 1) * (r + 2) / 2 <= e) ++r;
      ti = k * 8 + r;
      tj = k * 8 + (e - r * (r + 1) / 2);
    }
    return ti < tiles_m && tj < tiles_n;  // ragged edge only (tiles_m not a multiple of 8)
  } else {
    const int sw = 64 / sh;  // sh in {1, 8}
    const int sn = (tiles_n + sw - 1) / sw;
    const int sr = (int)(st / sn), sc = (int)(st - (int64_t)sr * sn);
    ti = sr * sh + inner / sw;
    tj = sc * sw + inner % sw;
    if (ti >= tiles_m || tj >= tiles_n) return false;
    if (mask_lower == 1) return tj <= ti;
    if (mask_lower == 2) return tj <= bc.row_tile(ti);
    return true;
  }
}
}  // namespace
}  // namespace gpx
