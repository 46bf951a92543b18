// Host-side sanitizer driver (SURVEY.md §5 "sanitizers").  TEST INFRASTRUCTURE.
//
// Links the sanitizer-instrumented HOST build of libgpx (tests/test_host_sanitizers.py compiles it with
// `hipcc -Xarch_host -fsanitize=...`: device code untouched — GPU AddressSanitizer is not available on this pool) and
// drives everything of the library's host side that runs WITHOUT a GPU: the rank threads' rendezvous (LocalHub: barrier
// rounds, a rank that aborts), the host replay of the tile maps, argument validation of every entry point, gpx_create's
// failure paths.  Exit code 0 = every call returned what it should; the sanitizers abort the process on a finding.
//
//   driver asan   — the full list (address + undefined-behaviour build)
//   driver tsan   — the threaded parts only (thread-sanitizer build)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/gpx.h"

static int fails = 0;
#define EXPECT(cond)                                                \
  do {                                                              \
    if (!(cond)) {                                                  \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++fails;                                                      \
    }                                                               \
  } while (0)

static void hub_rounds() {
  int done = -1;
  EXPECT(gpx_debug_local_hub(4, 300, -1, 0, &done) == GPX_OK && done == 300);
  EXPECT(gpx_debug_local_hub(8, 100, -1, 0, &done) == GPX_OK && done == 100);
  // rank 2 leaves at round 17 and aborts the hub: everybody comes back, nobody deadlocks
  EXPECT(gpx_debug_local_hub(4, 100, 2, 17, &done) == GPX_OK && done >= 16 && done <= 17);
  EXPECT(gpx_debug_local_hub(2, 50, 0, 0, &done) == GPX_OK && done == 0);
  EXPECT(gpx_debug_local_hub(0, 5, -1, 0, &done) != GPX_OK);
  EXPECT(gpx_debug_local_hub(4, 5, -1, 0, nullptr) != GPX_OK);
}

static void tile_maps() {
  std::vector<int32_t> out(2 * 70000);
  int64_t n = 0;
  // kind 0: lower triangle of a tm x tm grid, every tile once
  for (int64_t tm : {1, 7, 8, 9, 37, 64, 130}) {
    EXPECT(gpx_debug_tile_map(0, tm, 0, 1, 1, 0, out.data(), 70000, &n) == GPX_OK);
    EXPECT(n == tm * (tm + 1) / 2);
    std::vector<char> seen((size_t)(tm * tm), 0);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t ti = out[2 * i], tj = out[2 * i + 1];
      EXPECT(ti >= 0 && ti < tm && tj >= 0 && tj <= ti);
      if (ti >= 0 && ti < tm && tj >= 0 && tj <= ti) {
        EXPECT(!seen[(size_t)(ti * tm + tj)]);
        seen[(size_t)(ti * tm + tj)] = 1;
      }
    }
  }
  // kind 1: block-cyclic staircase (P ranks, tpb tiles per block, offset c)
  for (int P : {2, 3, 8})
    for (int tpb : {2, 4, 8}) {
      const int64_t tm = 24, tn = (24 / tpb) * P * tpb + 16;
      EXPECT(gpx_debug_tile_map(1, tm, tn, P, tpb, 1, out.data(), 70000, &n) == GPX_OK);
      int64_t want = 0;
      for (int64_t ti = 0; ti < tm; ++ti) {
        const int64_t lim = ((ti / tpb) * P + 1) * tpb + ti % tpb;
        want += (lim < tn ? lim : tn - 1) + 1;
      }
      EXPECT(n == want);
    }
  // kind 2: fused strip + rest
  EXPECT(gpx_debug_tile_map(2, 40, 8, 1, 1, 0, out.data(), 70000, &n) == GPX_OK && n == 40 * 41 / 2);
  // capacity too small / bad arguments
  EXPECT(gpx_debug_tile_map(0, 64, 0, 1, 1, 0, out.data(), 10, &n) != GPX_OK);
  EXPECT(gpx_debug_tile_map(0, 0, 0, 1, 1, 0, out.data(), 10, &n) != GPX_OK);
  EXPECT(gpx_debug_tile_map(7, 8, 0, 1, 1, 0, out.data(), 100, &n) != GPX_OK);
  EXPECT(gpx_debug_tile_map(1, 8, 8, 0, 1, 0, out.data(), 100, &n) != GPX_OK);
  EXPECT(gpx_debug_tile_map(0, 8, 0, 1, 1, 0, nullptr, 100, &n) != GPX_OK);
}

static void argument_validation() {
  EXPECT(gpx_abi_version() == GPX_ABI_VERSION);
  gpx_handle* h = nullptr;
  gpx_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.world = 1;
  EXPECT(gpx_create(nullptr, &cfg) == GPX_E_ARG);
  EXPECT(gpx_create(&h, nullptr) == GPX_E_ARG && h == nullptr);
  EXPECT(std::strlen(gpx_last_error(nullptr)) > 0);
  struct Bad {
    int field, value;
  };
  // one field out of range at a time: 0 kernel, 1 dtype, 2 world, 3 rank, 4 ndev, 5 block, 6 refine
  for (const Bad& b : {Bad{0, 9}, Bad{1, 7}, Bad{2, 0}, Bad{2, 65}, Bad{3, 1}, Bad{3, -1}, Bad{4, -1}, Bad{4, GPX_MAX_GROUP + 1},
                       Bad{5, 100}, Bad{5, 4224}, Bad{5, 1000}, Bad{6, -1}, Bad{6, 51}}) {
    gpx_config c = cfg;
    int32_t* f[] = {&c.kernel, &c.dtype, &c.world, &c.rank, &c.ndev, &c.block, &c.refine};
    *f[b.field] = b.value;
    h = reinterpret_cast<gpx_handle*>(0x1);
    EXPECT(gpx_create(&h, &c) == GPX_E_ARG);
    EXPECT(h == nullptr);
  }
  // a valid configuration on a box WITHOUT a GPU: the library has no CPU path and says so (with one: a handle)
  int ndev = -1;
  const int rc_count = gpx_device_count(&ndev);
  EXPECT(gpx_device_count(nullptr) == GPX_E_ARG);
  const int rc = gpx_create(&h, &cfg);
  if (rc_count != GPX_OK || ndev <= 0) {
    EXPECT(rc == GPX_E_HIP && h == nullptr);
    EXPECT(std::strstr(gpx_last_error(nullptr), "no CPU fallback") != nullptr);
    gpx_config g = cfg;  // a device group: the same failure before any member is created
    g.ndev = 2;
    g.devices[0] = 0;
    g.devices[1] = 1;
    EXPECT(gpx_create(&h, &g) == GPX_E_HIP && h == nullptr);
  } else {
    EXPECT(rc == GPX_OK && h != nullptr);
    gpx_destroy(h);
  }
  // entry points on a null handle / null outputs
  double x = 0;
  int64_t info = 0;
  gpx_timings tm;
  EXPECT(gpx_fit(nullptr, &x, &x, 1, 1, 1, &x, 1, 1.0, 0.0, 0.0, GPX_MEM_HOST, &info) == GPX_E_ARG);
  EXPECT(gpx_fit_predict(nullptr, &x, &x, 1, 1, 1, &x, 1, 1.0, 0.0, 0.0, &x, 1, &x, &x, GPX_MEM_HOST, &info) == GPX_E_ARG);
  EXPECT(gpx_predict(nullptr, &x, 1, &x, &x, GPX_MEM_HOST) == GPX_E_ARG);
  EXPECT(gpx_get_alpha(nullptr, &x) == GPX_E_ARG);
  EXPECT(gpx_lml_grad(nullptr, &x, &x) == GPX_E_ARG);
  EXPECT(gpx_logdet(nullptr, &x) == GPX_E_ARG);
  EXPECT(gpx_get_timings(nullptr, &tm) == GPX_E_ARG);
  EXPECT(gpx_set_flags(nullptr, 0) == GPX_E_ARG);
  EXPECT(gpx_release_scratch(nullptr) == GPX_E_ARG);
  EXPECT(gpx_comm_init(nullptr, &x) == GPX_E_ARG);
  EXPECT(gpx_comm_init_host(nullptr, nullptr) == GPX_E_ARG);
  gpx_destroy(nullptr);
  // unit-test entry points: argument checks come before any device work
  EXPECT(gpx_kernel_matrix(0, nullptr, 4, nullptr, 0, 2, &x, 1, 1.0, 0.0, &x) == GPX_E_ARG);
  EXPECT(gpx_kernel_matrix(5, &x, 1, nullptr, 0, 1, &x, 1, 1.0, 0.0, &x) == GPX_E_ARG);
  EXPECT(gpx_potrf(&x, 63, 0, &info) == GPX_E_ARG);
  EXPECT(gpx_potrf(&x, 64, 100, &info) == GPX_E_ARG);
  EXPECT(gpx_trsm(&x, 64, &x, 60) == GPX_E_ARG);
  EXPECT(gpx_gemm_nt(&x, 64, 128, &x, &x, 16, 1) == GPX_E_ARG);
  EXPECT(gpx_mfma_probe(nullptr, &x, &x) == GPX_E_ARG);
  EXPECT(gpx_microbench(nullptr, &x) == GPX_E_ARG);
  EXPECT(gpx_path_distance(&x, 0, &x, 1, 4, &x, GPX_MEM_HOST) == GPX_E_ARG);
  EXPECT(gpx_debug_gemm_bench(GPX_F64, 100, 64, 1, 0, 1, &x) == GPX_E_ARG);
}

int main(int argc, char** argv) {
  const bool tsan = argc > 1 && std::strcmp(argv[1], "tsan") == 0;
  hub_rounds();
  if (!tsan) {
    tile_maps();
    argument_validation();
  }
  std::printf("host sanitizer driver (%s): %d failed expectation(s)\n", tsan ? "tsan" : "asan+ubsan", fails);
  return fails ? 1 : 0;
}
