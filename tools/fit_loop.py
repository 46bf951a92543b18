#!/usr/bin/env python3
"""A few back-to-back fits at the bench size in one dtype (for tools/clock_sampler.py: which shader
clock / board power does the fp32 engine hold against the fp64 one?).
    python tools/clock_sampler.py OUT.json -- python tools/fit_loop.py float32"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
dtype = sys.argv[1] if len(sys.argv) > 1 else "float64"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fits = int(sys.argv[4]) if len(sys.argv) > 4 else 8
X, y, _ = synthetic(N, 3, 16, 12345)
tdt = torch.float32 if dtype == "float32" else torch.float64
Xd, yd = torch.from_numpy(X).to("cuda:0", tdt), torch.from_numpy(y).to("cuda:0", tdt)
with GP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, dtype=dtype, profile=True, block=block) as gp:
    for _ in range(fits):
        gp.fit(Xd, yd)
    tm = gp.timings_
print(json.dumps({"dtype": dtype, "N": N, "block": block, "fit_ms": tm["fit_total"], "chol_ms": tm["chol"], "chol_tflops": N ** 3 / 3 / (tm["chol"] * 1e-3) / 1e12,
                  "syrk_tflops": tm["syrk_flops"] / (tm["chol_syrk"] * 1e-3) / 1e12}))
