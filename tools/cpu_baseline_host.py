#!/usr/bin/env python3
"""CPU baseline details asked for by SURVEY.md §8(d), reproducible from this one command (runs on
the GPU box's host; no GPU involved):

    python tools/cpu_baseline_host.py > profiles/r02_cpu_baseline_host.json

Records the host CPU model, os.cpu_count(), the scheduler affinity and the cgroup CPU quota of
THIS process (the box's share of the host is smaller than the host), every BLAS pool
(threadpoolctl), and then
  * a thread sweep of the O(N^3) building blocks at N = 8192 — SciPy `cholesky` (LAPACK potrf of
    SciPy's bundled OpenBLAS), NumPy `cholesky` (NumPy's OpenBLAS), a 4096^3 dgemm and the oracle's
    level-3 blocked Cholesky — which shows where the Cholesky stops scaling (round 1 measured
    potrf at 30 GF/s with 64 threads: it does not parallelise at this size at ANY thread count,
    while dgemm — and with it the blocked variant — does);
  * the NumPy/SciPy oracle (oracle/gp_oracle.py, blocked Cholesky) at N = 8192, M = 4096 with the
    thread count that sweep found best (median of 3) and with ONE thread (per-core figure).
"""
import json, os, platform, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from threadpoolctl import threadpool_info, threadpool_limits
from oracle.gp_oracle import OracleGP, chol_lower_blocked, synthetic_problem


def cpu_budget():
    """CPUs this process may actually use: min(affinity, cgroup quota)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:                                        # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    return {"affinity_cpus": aff, "cgroup_cpu_quota": quota,
            "usable_cpus": int(max(1, min(aff, quota if quota else aff)))}


CHOL = "blocked"      # the oracle's level-3 blocked Cholesky (LAPACK potrf does not scale here)


def run(N, M):
    X, y, Xs = synthetic_problem(N, 3, M, seed=12345)
    t0 = time.perf_counter()
    gp = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, chol=CHOL).fit(X, y)
    t1 = time.perf_counter()
    gp.predict(Xs)
    t2 = time.perf_counter()
    return {"fit_s": t1 - t0, "predict_s": t2 - t1, "points_per_s": (N + M) / (t2 - t0),
            "phases_ms": {k: round(v, 1) for k, v in gp.timings_.items()}}


def sweep(budget):
    import scipy.linalg as sl
    N = 8192
    rng = np.random.default_rng(0)
    A = rng.standard_normal((N, 256))
    K = A @ A.T + N * np.eye(N)
    B = rng.standard_normal((4096, 4096))
    rows = []
    cand = sorted({1, 4, 8, 16, 32, 64, budget["usable_cpus"]})
    for t in cand:
        with threadpool_limits(limits=t):
            t0 = time.perf_counter(); sl.cholesky(K, lower=True, check_finite=False); t1 = time.perf_counter()
            np.linalg.cholesky(K); t2 = time.perf_counter()
            B @ B; t3 = time.perf_counter()
            chol_lower_blocked(K.copy()); t4 = time.perf_counter()
        rows.append({"threads": t, "scipy_potrf_gflops": N ** 3 / 3 / (t1 - t0) / 1e9,
                     "numpy_potrf_gflops": N ** 3 / 3 / (t2 - t1) / 1e9,
                     "dgemm_4096_gflops": 2 * 4096 ** 3 / (t3 - t2) / 1e9,
                     "blocked_cholesky_gflops": N ** 3 / 3 / (t4 - t3) / 1e9})
        print(rows[-1], file=sys.stderr, flush=True)
    return rows


def main():
    try:
        model = [l.split(":", 1)[1].strip() for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
                 if l.startswith("Model name")][0]
    except Exception:
        model = platform.processor()
    budget = cpu_budget()
    out = {"command": "python tools/cpu_baseline_host.py", "cpu_model": model, "os_cpu_count": os.cpu_count(), **budget,
           "loadavg": open("/proc/loadavg").read().split()[:3],
           "env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")},
           "blas": [{k: p.get(k) for k in ("internal_api", "num_threads", "version", "threading_layer", "filepath")} for p in threadpool_info()]}
    out["thread_sweep_N8192"] = sw = sweep(budget)
    best = max(sw, key=lambda r: r["blocked_cholesky_gflops"])["threads"]
    out["best_potrf_threads"] = best
    N, M = 8192, 4096
    with threadpool_limits(limits=best):
        runs = [run(N, M) for _ in range(3)]
    runs.sort(key=lambda r: r["points_per_s"])
    out["best_threads_N8192"] = dict(runs[1], threads=best)
    with threadpool_limits(limits=1):
        out["one_thread_N8192"] = dict(run(N, M), threads=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
