"""Timing-only harness for engine ablations / schedule variants: gpx_fit at N through a given
build of libgpx.so (host buffers, profile flag), *info ignored* so that builds with
deliberately wrong data movement can still be timed; prints the best trailing-update rate.
   python tools/engine_ablation_harness.py path/to/libgpx_variant.so [N] [reps]
Results of the round-1 ablations are in DESIGN.md section 3.2."""
import ctypes as C, os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocesspathmodelling_amd import _abi
_abi.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _abi.load()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(1)
X = rng.random((N, 3)); y = np.sin(6.28 * X[:, 0]) + 0.1 * rng.standard_normal(N)
h = C.c_void_p()
cfg = _abi.GpxConfig(kernel=0, dtype=0, device=0, block=0, rank=0, world=1, flags=_abi.FLAG_PROFILE, reserved=0)
assert lib.gpx_create(C.byref(h), C.byref(cfg)) == 0
ls = np.array([0.25]); info = C.c_int64(0)
best = None
for _ in range(reps):
    rc = lib.gpx_fit(h, C.c_void_p(X.ctypes.data), C.c_void_p(y.ctypes.data), N, 3, 1, _abi.dptr(ls), 1, 1.5, 1e-2, 0.0,
                     _abi.MEM_HOST, C.byref(info))
    assert rc == 0, (rc, lib.gpx_last_error(h))
    t = _abi.GpxTimings(); lib.gpx_get_timings(h, C.byref(t))
    tf = t.syrk_flops / (t.chol_syrk * 1e-3) / 1e12
    r = {"syrk_tf": round(tf, 2), "syrk_ms": round(t.chol_syrk, 2), "chol_ms": round(t.chol, 2), "strip_ms": round(t.chol_strip, 2),
         "info": info.value}
    if best is None or r["syrk_tf"] > best["syrk_tf"]:
        best = r
lib.gpx_destroy(h)
# end-to-end fit+predict without the profile flag (look-ahead streams free-running)
import time
h = C.c_void_p()
cfg = _abi.GpxConfig(kernel=0, dtype=0, device=0, block=0, rank=0, world=1, flags=0, reserved=0)
assert lib.gpx_create(C.byref(h), C.byref(cfg)) == 0
M = 4096
Xs = rng.random((M, 3)); mean = np.empty(M); var = np.empty(M)
ts = []
for _ in range(reps + 1):
    t0 = time.perf_counter()
    assert lib.gpx_fit(h, C.c_void_p(X.ctypes.data), C.c_void_p(y.ctypes.data), N, 3, 1, _abi.dptr(ls), 1, 1.5, 1e-2,
                       0.0, _abi.MEM_HOST, C.byref(info)) == 0
    if info.value == 0:
        assert lib.gpx_predict(h, C.c_void_p(Xs.ctypes.data), M, C.c_void_p(mean.ctypes.data),
                               C.c_void_p(var.ctypes.data), _abi.MEM_HOST) == 0
    ts.append((time.perf_counter() - t0) * 1e3)
best["step_ms_min"] = round(min(ts[1:]), 1)
print(os.path.basename(sys.argv[1]), json.dumps(best))
lib.gpx_destroy(h)
