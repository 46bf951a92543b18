// Cost of fine-grained two-stream pipelining on this platform: a chain of short kernels on stream A
// (a), alone; (b) with stream B running a dependent short kernel per step (B waits for A's event of
// the same step) and A waiting for B's event of TWO steps earlier (long signalled when reached);
// (c) A waiting for B's event of the SAME previous step (fresh).   hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int steps = 256;
  std::vector<hipEvent_t> ea(steps), eb(steps);
  for (int i = 0; i < steps; ++i) {
    CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming));
  }
  const long long ka = 2300 * 22, kb = 2300 * 10;  // ~22 us on A, ~10 us on B (cycles at ~2.3 GHz... s_memtime is 100 MHz-based? measured below)
  std::vector<hipEvent_t> et(steps);
  for (int i = 0; i < steps; ++i) CK(hipEventCreate(&et[i]));
  for (int mode = 0; mode < 8; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < steps; ++i) {
        if (mode == 1 && i >= 2) CK(hipStreamWaitEvent(A, eb[i - 2], 0));
        if (mode == 2 && i >= 1) CK(hipStreamWaitEvent(A, eb[i - 1], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, A, ka, nullptr);
        if (mode == 1 || mode == 2) {
          CK(hipEventRecord(ea[i], A));
          CK(hipStreamWaitEvent(B, ea[i], 0));
          hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, B, kb, nullptr);
          CK(hipEventRecord(eb[i], B));
        }
        if (mode == 3) hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, A, kb, nullptr);  // same work, one stream
        if (mode == 4) CK(hipEventRecord(et[i], A));                                     // record only, timing event
        if (mode == 5) CK(hipEventRecord(ea[i], A));                                     // record only, no-timing event
        if (mode == 6) { CK(hipEventRecord(ea[i], A)); CK(hipStreamWaitEvent(B, ea[i], 0)); hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, B, kb, nullptr); }  // B follows A, A never waits
        if (mode == 7) { CK(hipEventRecord(et[i], A)); CK(hipStreamWaitEvent(B, et[i], 0)); hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, B, kb, nullptr); }
      }
      CK(hipStreamSynchronize(A));
      CK(hipStreamSynchronize(B));
      auto t1 = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / steps;
      if (rep == 2)
        printf("mode %d (%s): %.2f us per step\n", mode,
               mode == 0 ? "A alone" : mode == 1 ? "A || B, A waits B[i-2]" : mode == 2 ? "A || B, A waits B[i-1]" : mode == 3 ? "A then B on one stream"
               : mode == 4 ? "A + record (timing event)" : mode == 5 ? "A + record (no-timing event)" : mode == 6 ? "A + record(no-timing), B waits and runs" : "A + record(timing), B waits and runs", us);
    }
  }
  return 0;
}
