#!/usr/bin/env python3
"""Where the one-workgroup POTF2 (potf2_64_kernel) spends its cycles: a DIAGNOSTIC build of the
library (-DGPX_STAMPS: thread 0 writes s_memtime stamps at the phase boundaries; never in the
shipped .so) factors one 64x64 block; prints cycles per phase.
    python tools/potf2_stamps.py --build      (here: hipcc cross-compiles tools/_stamps/libgpx_stamps.so)
    python tools/potf2_stamps.py              (GPU box)"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocesspathmodelling_amd import _abi, build
OUT = os.path.join(ROOT, "tools", "_stamps")
LIB = os.path.join(OUT, "libgpx_stamps.so")

if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DGPX_STAMPS",
                    "-o", LIB] + srcs, check=True)
    print("built", LIB)
    sys.exit(0)

_abi._preload_torch_hip_runtime()
lib = C.CDLL(LIB)
rng = np.random.default_rng(0)
B = rng.standard_normal((64, 64))
K = B @ B.T + 64 * np.eye(64)
info = C.c_int64(0)
stamps = (C.c_longlong * 64)()
rows = []
for it in range(6):
    A = K.copy()
    assert lib.gpx_potrf(A.ctypes.data_as(C.POINTER(C.c_double)), 64, 0, C.byref(info)) == 0 and info.value == 0
    assert lib.gpx_debug_read_stamps(stamps, 64) == 0
    rows.append(np.array(stamps[:22], dtype=np.int64))
assert np.allclose(np.tril(A), np.linalg.cholesky(K), rtol=1e-12, atol=1e-12)
s = rows[-1]
print("total cycles (s_memtime ticks = shader cycles):", s[21] - s[0])
print("load + zero            :", s[1] - s[0])
for j in range(8):
    a = s[2 + 2 * j] - (s[1] if j == 0 else s[1 + 2 * j])
    b = s[3 + 2 * j] - s[2 + 2 * j]
    print(f"step {j}: phase A (wave 0; beside it the rest of the previous step's phase B) {a:6d}   export part of phase B {b:6d}")
print("inverse (3 levels)     :", s[20] - s[17])
print("store                  :", s[21] - s[20])
print("all runs, total:", [int(r[21] - r[0]) for r in rows])
