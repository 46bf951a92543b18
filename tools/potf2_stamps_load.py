#!/usr/bin/env python3
"""Which phase of the one-workgroup POTF2 stretches when a trailing update runs beside it?  The diagnostic
build (tools/potf2_stamps.py --build) factors 64x64 blocks in a loop, alone and while another thread keeps
the GPU busy with N = 16384 fits of the shipped library; prints median / p90 cycles per phase for both.
    python tools/potf2_stamps.py --build && python tools/potf2_stamps_load.py      (GPU box)"""
import ctypes as C, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocesspathmodelling_amd import _abi, GP
from bench import synthetic
LIB = os.path.join(ROOT, "tools", "_stamps", "libgpx_stamps.so")
_abi._preload_torch_hip_runtime()
lib = C.CDLL(LIB)
rng = np.random.default_rng(0)
B = rng.standard_normal((64, 64))
K = B @ B.T + 64 * np.eye(64)
info = C.c_int64(0)
stamps = (C.c_longlong * 64)()


def sample(n):
    out = []
    for _ in range(n):
        A = K.copy()
        assert lib.gpx_potrf(A.ctypes.data_as(C.POINTER(C.c_double)), 64, 0, C.byref(info)) == 0
        assert lib.gpx_debug_read_stamps(stamps, 64) == 0
        s = np.array(stamps[:22], dtype=np.int64)
        steps_a = sum(s[2 + 2 * j] - s[1 + 2 * j] for j in range(8))
        steps_b = sum(s[3 + 2 * j] - s[2 + 2 * j] for j in range(8))
        out.append([s[21] - s[0], s[1] - s[0], steps_a, steps_b, s[20] - s[17], s[21] - s[20]])
    return np.array(out)


def show(tag, a):
    names = ["total", "load", "phase A x8", "phase B x8", "inverse", "store"]
    print(tag)
    for i, n in enumerate(names):
        print(f"   {n:11s} median {int(np.median(a[:, i])):7d}   p90 {int(np.percentile(a[:, i], 90)):7d}   max {int(a[:, i].max()):7d}")


sample(5)
show("POTF2 alone (cycles)", sample(200))
stop = False
X, y, Xs = synthetic(16384, 3, 64, 1)


def load():
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        while not stop:
            gp.fit(X, y)


t = threading.Thread(target=load)
t.start()
time.sleep(3.0)
show("POTF2 beside N = 16384 fits of the shipped library (cycles)", sample(400))
stop = True
t.join()
