#!/usr/bin/env python3
"""A/B of the trailing-update engine variants at N = 65536 (one process, alternating): default 4-wave
128 x 128 tiles, two workgroups per CU, against GPX_SYRK_W8=1 (eight waves, two k-steps per barrier, one
workgroup per CU: gemm_nt_w8_kernel).  fp32 (the case it was built for) and fp64.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
X, y, Xs = synthetic(65536, 3, 4096, 12345)
out = {}
for dtype in ("float32", "float64"):
    tdt = torch.float32 if dtype == "float32" else torch.float64
    Xd, yd = torch.from_numpy(X).to("cuda", tdt), torch.from_numpy(y).to("cuda", tdt)
    with GP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, dtype=dtype, profile=True) as gp:
        gp.fit(Xd, yd)
        for rnd in range(3):
            for w8 in ("0", "1"):
                os.environ["GPX_SYRK_W8"] = w8
                gp.fit(Xd, yd)
                tm = gp.timings_
                rec = out.setdefault(f"{dtype}_w8={w8}", {"chol_ms": [], "syrk_tflops": []})
                rec["chol_ms"].append(round(tm["chol"], 2))
                rec["syrk_tflops"].append(round(tm["syrk_flops"] / (tm["chol_syrk"] * 1e-3) / 1e12, 2))
os.environ["GPX_SYRK_W8"] = "0"
print(json.dumps(out))
