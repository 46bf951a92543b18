#!/usr/bin/env python3
"""The clock the chip really holds inside the SYRK's MFMA loop, fp64 against fp32: a DIAGNOSTIC
build (-DGPX_STAMPS, built by `python tools/potf2_stamps.py --build`) stamps s_memtime (shader
cycles) and s_memrealtime (100 MHz) around one tile of every REST launch; clock = ratio x 100 MHz
(MI355X_MICROARCH.md, DVFS give-back item 6).  After >= 2 s of back-to-back launches.
    python tools/syrk_clock.py"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocesspathmodelling_amd import _abi
import gaussianprocesspathmodelling_amd._abi as abi
abi.LIB_PATH = os.path.join(ROOT, "tools", "_stamps", "libgpx_stamps.so")     # the diagnostic copy
from gaussianprocesspathmodelling_amd import GP
from bench import synthetic
lib = abi.load()
lib.gpx_debug_read_syrk_clock.argtypes = [C.POINTER(C.c_longlong)]
N = 49152
X, y, _ = synthetic(N, 3, 16, 12345)
out = {}
for dtype in ("float64", "float32"):
    with GP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, dtype=dtype, profile=True) as gp:
        zero = (C.c_longlong * 8)()
        hip = C.CDLL(None)
        for _ in range(3):
            gp.fit(X, y)
        # mean over ALL workgroups of the big launches of the next fit (buffer slots 5, 6)
        base = (C.c_longlong * 8)()
        assert lib.gpx_debug_read_syrk_clock(base) == 0
        gp.fit(X, y)
        tm = gp.timings_
        buf = (C.c_longlong * 8)()
        assert lib.gpx_debug_read_syrk_clock(buf) == 0
        cyc, real, K = buf[0], buf[1], buf[2]
        kt = K // (16 if dtype == "float64" else 32)
        out[dtype] = {"shader_cycles": cyc, "realtime_ticks_100MHz": real, "clock_mhz": cyc / real * 100.0,
                      "k_steps": kt, "cycles_per_k_step": cyc / kt, "mean_prologue_cycles": (buf[7] - base[7]) / max(1, buf[6] - base[6]),
                      "mean_epilogue_issue_cycles": (buf[4] - base[4]) / max(1, buf[6] - base[6]),
                      "mean_cycles_per_k_step_all_workgroups": (buf[5] - base[5]) / max(1, buf[6] - base[6]) / kt,
                      "workgroups_averaged": buf[6] - base[6], "mfma_cycles_per_k_step_per_wave": 4096,
                      "syrk_tflops": tm["syrk_flops"] / (tm["chol_syrk"] * 1e-3) / 1e12}
print(json.dumps(out))
