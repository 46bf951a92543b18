#!/usr/bin/env python3
"""Cost of the analytic LML gradient (gpx_lml_grad) at the bench size: one fit, then the gradient
(L^-T by structured forward substitution + fused K^-1 trace pass), against central differences of
the GPU's own LML for two parameters.   python tools/grad_bench.py [--ntrain 65536]
With --devices n: the same gradient on a device group of n ranks (one GPU box: the ranks share the
card, so this checks the sharded gradient at scale against the single-GPU one, it is not a speed-up)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP

ap = argparse.ArgumentParser()
ap.add_argument("--ntrain", type=int, default=65536)
ap.add_argument("--devices", type=int, default=0)
a = ap.parse_args()
N = a.ntrain
X, y, _ = synthetic(N, 3, 16, 12345)
ls, sf2, sn2 = np.array([0.3, 0.2, 0.25]), 1.5, 1e-2
if a.devices > 1:
    with GP("rbf", ls, sf2, sn2, jitter=0.0) as g1:
        lml1, grad1 = g1.fit(X, y).lml_gradient()
        t1 = g1.timings_
    with GP("rbf", ls, sf2, sn2, jitter=0.0, devices=a.devices, oversubscribe=True) as gp:
        gp.fit(X, y); gp.lml_gradient()
        t0 = time.perf_counter(); lml, grad = gp.lml_gradient(); dt = time.perf_counter() - t0
        tm = gp.timings_
    print(json.dumps({"config": f"N={N} d=3 RBF ARD fp64, {a.devices} ranks sharing one GPU (in-process transport)",
                      "single_gpu": {"grad_ms": t1["grad_total"], "trtri_ms": t1["grad_trtri"], "trace_ms": t1["grad_trace"]},
                      "group_rank0": {"grad_ms": tm["grad_total"], "grad_wall_ms": dt * 1e3,
                                      "trtri_plus_allgather_ms": tm["grad_trtri"], "trace_ms": tm["grad_trace"]},
                      "lml_rel_diff": abs(lml - lml1) / abs(lml1),
                      "grad_rel_diff": float(np.max(np.abs(grad - grad1)) / np.max(np.abs(grad1))),
                      "grad": grad.tolist(), "grad_single": grad1.tolist()}))
    sys.exit(0)
with GP("rbf", ls, sf2, sn2, jitter=0.0) as gp:
    gp.fit(X, y); gp.lml_gradient()                       # warm-up (allocations)
    gp.fit(X, y)
    t0 = time.perf_counter(); lml, grad = gp.lml_gradient(); dt = time.perf_counter() - t0
    tm = gp.timings_
    fd = {}
    h = 1e-4
    v0 = np.log(np.concatenate([ls, [sf2, sn2]]))
    for i in (0, 4):
        f = []
        for sgn in (+1, -1):
            v = v0.copy(); v[i] += sgn * h
            gp.lengthscale, gp.variance, gp.noise = np.exp(v[:3]), float(np.exp(v[3])), float(np.exp(v[4]))
            f.append(gp.fit(X, y).log_marginal_likelihood(y))
        fd[i] = (f[0] - f[1]) / (2 * h)
print(json.dumps({"config": f"N={N} d=3 RBF ARD fp64", "fit_ms": tm["fit_total"], "grad_ms": tm["grad_total"],
                  "grad_wall_ms": dt * 1e3, "trtri_ms": tm["grad_trtri"], "trace_ms": tm["grad_trace"],
                  "trtri_tflops": N ** 3 / 3 / (tm["grad_trtri"] * 1e-3) / 1e12,
                  "trace_tflops": N ** 3 / 3 / (tm["grad_trace"] * 1e-3) / 1e12,
                  "grad_over_fit": tm["grad_total"] / tm["fit_total"], "lml": lml, "grad": grad.tolist(),
                  "central_difference": {str(k): v for k, v in fd.items()},
                  "fd_rel_err": {str(k): abs(v - grad[k]) / np.abs(grad).max() for k, v in fd.items()}}))
