#!/usr/bin/env python3
"""BASELINE.json configs[4]: N=65536, d=3, RBF, fp32 + ARD lengthscales on one MI355X —
mixed-precision tolerance study ("report max/median rel error of fp32 vs
fp64, a study, not a 1e-6 gate").  The fp64 side is the HIP path itself (parity-tested
against the oracle at smaller N); a full CPU oracle at this size needs 34 GB and minutes.

    python tools/precision_study.py [--ntrain 65536] [--mtest 4096] [--noise 1e-2]
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntrain", type=int, default=65536)
    ap.add_argument("--mtest", type=int, default=4096)
    ap.add_argument("--noise", type=float, default=1e-2)
    args = ap.parse_args()
    from gaussianprocesspathmodelling_amd import GP
    from bench import synthetic
    N, M = args.ntrain, args.mtest
    X, y, Xs = synthetic(N, 3, M, 12345)
    ls, sf2 = (0.3, 0.2, 0.25), 1.5
    out = {"config": f"C5: N={N} d=3 RBF ARD ls={ls} sf2={sf2} sn2={args.noise} M={M}"}
    res = {}
    for dtype in ("float64", "float32"):
        with GP("rbf", ls, sf2, args.noise, jitter=0.0, dtype=dtype, profile=True) as gp:
            gp.fit(X, y); gp.predict(Xs)            # warm-up (allocations)
            t0 = time.perf_counter()
            gp.fit(X, y)
            mean, var = gp.predict(Xs)
            dt = time.perf_counter() - t0
            tm = gp.timings_
            res[dtype] = (mean.astype(np.float64), var.astype(np.float64), gp.alpha_.astype(np.float64),
                          gp.log_det_, gp.info_)
            out[dtype] = {"step_ms": dt * 1e3, "fit_ms": tm["fit_total"], "predict_ms": tm["predict_total"],
                          "chol_ms": tm["chol"], "chol_tflops": N ** 3 / 3 / (tm["chol"] * 1e-3) / 1e12,
                          "syrk_tflops": tm["syrk_flops"] / (tm["chol_syrk"] * 1e-3) / 1e12,
                          "kbuild_ms": tm["kbuild"], "info": gp.info_}
    m64, v64, a64, ld64, _ = res["float64"]
    m32, v32, a32, ld32, _ = res["float32"]
    rm = np.abs(m32 - m64) / np.maximum(np.abs(m64), 1e-6)
    rv = np.abs(v32 - v64) / np.maximum(v64, 1e-6 * sf2)
    ra = np.abs(a32 - a64) / np.max(np.abs(a64))
    out["fp32_vs_fp64"] = {
        "mean_rel_max": float(rm.max()), "mean_rel_median": float(np.median(rm)),
        "mean_abs_max": float(np.abs(m32 - m64).max()),
        "var_rel_max": float(rv.max()), "var_rel_median": float(np.median(rv)),
        "var_abs_max": float(np.abs(v32 - v64).max()),
        "alpha_err_over_max": float(ra.max()), "logdet_rel": abs(ld32 - ld64) / abs(ld64),
        "speedup_step": out["float64"]["step_ms"] / out["float32"]["step_ms"]}
    # mixed precision: fp32 factorisation + fp64 refinement of alpha, fp64 mean (variance stays fp32)
    out["mixed"] = []
    for it in (1, 2, 3, 5, 0):    # 0 = the default: adaptive
        with GP("rbf", ls, sf2, args.noise, jitter=0.0, dtype="mixed", refine=it, profile=True) as gp:
            gp.fit(X, y); gp.predict(Xs)
            t0 = time.perf_counter()
            gp.fit(X, y)
            mean, var = gp.predict(Xs)
            dt = time.perf_counter() - t0
            tm = gp.timings_
            rm = np.abs(mean - m64) / np.maximum(np.abs(m64), 1e-6)
            out["mixed"].append({
                "refine_iterations": ("adaptive" if it == 0 else it), "iterations_run": tm["refine_iters"], "step_ms": dt * 1e3, "fit_ms": tm["fit_total"], "refine_ms": tm["refine"],
                "predict_ms": tm["predict_total"], "speedup_step_vs_fp64": out["float64"]["step_ms"] / (dt * 1e3),
                "residual_before": tm["refine_resid0"], "residual_after": tm["refine_resid"],
                "mean_rel_max": float(rm.max()), "mean_rel_median": float(np.median(rm)),
                "mean_abs_max": float(np.abs(mean - m64).max()),
                "alpha_err_over_max": float(np.abs(gp.alpha_ - a64).max() / np.abs(a64).max()),
                "var_rel_median": float(np.median(np.abs(var - v64) / np.maximum(v64, 1e-6 * sf2))),
                "var_abs_max": float(np.abs(var - v64).max())})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
