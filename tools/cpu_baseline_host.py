#!/usr/bin/env python3
"""CPU baseline details asked for by SURVEY.md §8(d): host CPU model, core count, BLAS vendor /
threads, the NumPy/SciPy oracle at N = 8192 (M = 4096) with all BLAS threads (median of 3) and
with ONE thread (per-core figure).  Runs on the GPU box's host; no GPU involved.
    python tools/cpu_baseline_host.py > profiles/r01_cpu_baseline_host.json"""
import json, os, platform, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from threadpoolctl import threadpool_info, threadpool_limits
from oracle.gp_oracle import OracleGP, synthetic_problem


def run(N, M):
    X, y, Xs = synthetic_problem(N, 3, M, seed=12345)
    t0 = time.perf_counter()
    gp = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    t1 = time.perf_counter()
    gp.predict(Xs)
    t2 = time.perf_counter()
    return {"fit_s": t1 - t0, "predict_s": t2 - t1, "points_per_s": (N + M) / (t2 - t0),
            "phases_ms": {k: round(v, 1) for k, v in gp.timings_.items()}}


def main():
    try:
        model = [l.split(":", 1)[1].strip() for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
                 if l.startswith("Model name")][0]
    except Exception:
        model = platform.processor()
    out = {"cpu_model": model, "os_cpu_count": os.cpu_count(),
           "blas": [{k: p.get(k) for k in ("internal_api", "num_threads", "version", "threading_layer")} for p in threadpool_info()]}
    N, M = 8192, 4096
    runs = [run(N, M) for _ in range(3)]
    runs.sort(key=lambda r: r["points_per_s"])
    out["all_threads_N8192"] = runs[1]
    with threadpool_limits(limits=1):
        out["one_thread_N8192"] = run(N, M)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
