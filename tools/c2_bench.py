#!/usr/bin/env python3
"""BASELINE.json configs[1] (C2: N = 8192, d = 3, RBF, fp64, M = 4096) as a timed loop, inputs
resident in HBM — the small-N latency case where the serial diagonal / panel chain, not the MFMA
rate, sets the time.  Prints one JSON line.
    python tools/c2_bench.py [--steps 20] [--ntrain 8192]
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2prof -- python3 tools/c2_bench.py"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--ntrain", type=int, default=8192)
ap.add_argument("--mtest", type=int, default=4096)
ap.add_argument("--block", type=int, default=0, help="Cholesky panel width (0 = library default)")
ap.add_argument("--no-profile", action="store_true", help="no per-phase events inside the Cholesky (sub-phase times read 0)")
ap.add_argument("--fused", action="store_true", help="gp.fit_predict(X, y, Xs) (one factorisation pass) instead of fit + predict")
a = ap.parse_args()
N, M = a.ntrain, a.mtest
dev = torch.device("cuda", 0)
X, y, Xs = (torch.from_numpy(v).to(dev) for v in synthetic(N, 3, M, 12345))
with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, profile=not a.no_profile, block=a.block) as gp:
    step = (lambda: gp.fit_predict(X, y, Xs)) if a.fused else (lambda: gp.fit(X, y).predict(Xs))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        for k, v in gp.timings_.items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
ms = el / a.steps * 1e3
ph = {k: round(acc[k] / a.steps, 3) for k in ("kbuild", "chol", "chol_diag", "chol_trsm", "chol_strip", "chol_syrk", "fit_total",
                                              "kstar", "trsm", "mean", "var", "predict_total")}
print(json.dumps({"config": f"C2: N={N} d=3 RBF fp64 M={M}, inputs resident in HBM", "block": a.block, "profile": not a.no_profile, "step": "fit_predict" if a.fused else "fit + predict", "ms_per_step": ms,
                  "points_per_s": (N + M) / (ms * 1e-3), "cholesky_tflops": N ** 3 / 3 / (ph["chol"] * 1e-3) / 1e12,
                  "cholesky_flops_at_peak_ms": N ** 3 / 3 / 78.6e12 * 1e3, "phases_ms": ph}))
