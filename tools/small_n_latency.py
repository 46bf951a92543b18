#!/usr/bin/env python3
"""Steady-state fit / predict latency at small N (host NumPy in and out).  python tools/small_n_latency.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import synthetic_problem
for N, M in ((512, 256), (2048, 512), (8192, 4096)):
    X, y, Xs = synthetic_problem(N, 3, M, seed=1)
    with GP("rbf", 0.25, 1.5, 1e-2) as gp:
        gp.fit(X, y).predict(Xs)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); gp.fit(X, y); t1 = time.perf_counter(); gp.predict(Xs); t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        f = min(t[0] for t in ts) * 1e3; p = min(t[1] for t in ts) * 1e3
        tm = gp.timings_
        ts1 = []
        for _ in range(10):
            t0 = time.perf_counter(); gp.fit_predict(X, y, Xs); ts1.append(time.perf_counter() - t0)
        print(f"N={N} M={M}: fit {f:.3f} ms predict {p:.3f} ms (device: fit_total {tm['fit_total']:.3f} predict_total {tm['predict_total']:.3f}); "
              f"one pass (fit_predict) {min(ts1) * 1e3:.3f} ms against {f + p:.3f}")
