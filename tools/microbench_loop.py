#!/usr/bin/env python3
"""Run gpx_microbench repeatedly for a few seconds (for tools/clock_sampler.py)."""
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import _abi
lib = _abi.load()
a, b = C.c_double(0), C.c_double(0)
t0 = time.time(); vals = []
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 4.0:
    assert lib.gpx_microbench(C.byref(a), C.byref(b)) == 0
    vals.append((a.value, b.value))
print(json.dumps({"runs": len(vals), "mfma_tflops_max": max(v[0] for v in vals),
                  "mfma_tflops_median": sorted(v[0] for v in vals)[len(vals) // 2],
                  "copy_gbs_max": max(v[1] for v in vals)}))
