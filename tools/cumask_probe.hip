// Which physical CUs does a stream CU mask (hipExtStreamCreateWithCUMask) remove on an MI355X in SPX mode?
// Every workgroup records (XCC_ID, HW_ID); the host prints the CUs used per XCC with and without the mask.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/cumask_probe.hip && /tmp/cumask_probe [bits_cleared]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ void where(unsigned* out, long long spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}
static void run(hipStream_t st, const char* tag) {
  const int G = 4096;
  unsigned* d;
  hipMalloc(&d, G * 8);
  hipLaunchKernelGGL(where, dim3(G), dim3(256), 65536, st, d, 20000LL);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2 * G);
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> cus;
  for (int i = 0; i < G; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus[xcc].insert(se * 32 + sh * 16 + cu);
  }
  size_t tot = 0;
  printf("%s\n", tag);
  {  // is workgroup i on XCC i % 8 (what the library's XCD-aware tile maps assume)?
    int same = 0;
    for (int i = 0; i < G; ++i) same += (h[2 * i] & 0xf) == (unsigned)(i % 8);
    printf("  workgroups with xcc == blockIdx %% 8: %d of %d; first 32 blocks' xcc:", same, G);
    for (int i = 0; i < 32; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
  }
  for (auto& kv : cus) {
    tot += kv.second.size();
    printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
    for (unsigned c : kv.second) printf(" %u.%u.%u", c / 32, (c / 16) & 1, c & 15);
    printf("\n");
  }
  printf("  total distinct CUs seen: %zu\n", tot);
  hipFree(d);
}
int main(int argc, char** argv) {
  const int cleared = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  hipStream_t s0, sm;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  run(s0, "unmasked stream");
  std::vector<uint32_t> mask((p.multiProcessorCount + 31) / 32, 0xffffffffu);
  for (int b = 0; b < cleared; ++b) mask[b / 32] &= ~(1u << (b % 32));
  hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data());
  printf("hipExtStreamCreateWithCUMask (bits 0..%d cleared): %s\n", cleared - 1, hipGetErrorString(e));
  if (e == hipSuccess) run(sm, "masked stream");
  return 0;
}
