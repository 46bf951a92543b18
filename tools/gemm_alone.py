#!/usr/bin/env python3
"""The dense tile engine ALONE (no look-ahead stream beside it), operands resident: TFLOP/s of the
triangular trailing-update launch by dtype, panel width K and epilogue (gpx_debug_gemm_bench).
    python tools/gemm_alone.py [n]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import _abi
lib = _abi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36864
rows = []
for dtype, name, peak in ((_abi.DTYPE_IDS["float64"], "f64", 78.6), (_abi.DTYPE_IDS["float32"], "f32", 157.3)):
    for lower in (1, 0):
        for mode in (0, 1):
            for k in (256, 512, 1024, 2048, 4096):
                nn = n if lower else n // 2
                ms = C.c_double(0)
                rc = lib.gpx_debug_gemm_bench(dtype, nn, k, lower, mode, 6, C.byref(ms))
                assert rc == 0, rc
                flops = (nn * (nn + 128) if lower else 2 * nn * nn) * k
                tf = flops / (ms.value * 1e-3) / 1e12
                rows.append({"dtype": name, "shape": "triangle" if lower else "square", "epilogue": "atomic C-=" if mode == 0 else "store C=",
                             "n": nn, "k": k, "ms": round(ms.value, 4), "tflops": round(tf, 2), "frac": round(tf / peak, 4)})
                print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(rows))
