#!/usr/bin/env python3
"""FULL-SIZE golden fixtures: the CPU oracle at BASELINE.json's own sizes.  TEST INFRASTRUCTURE.

    python oracle/make_golden_full.py --config C3 --out gpurun_out/G5.npz
    python oracle/make_golden_full.py --config C5 --out gpurun_out/G6.npz

Writes what the oracle (oracle/gp_oracle.py: R&W Alg. 2.1 with cdist -> exp -> blocked level-3
Cholesky -> solve_triangular; the reference itself has no GP code) produces for

    C3 = BASELINE.json configs[2]: N=65536, d=3, RBF  l=0.25,            sf2=1.5, sn2=1e-2, M=4096
    C5 = BASELINE.json configs[4]: N=65536, d=3, RBF  l=(0.3,0.2,0.25),  sf2=1.5, sn2=1e-2, M=4096

on the synthetic workload of SURVEY.md §8(d) (seed 12345; the inputs are regenerated from the seed
by the tests, only digests of them are stored): mean[M], var[M], logdet, alpha on 1024 fixed rows,
max|alpha|.  About 60 KB per config.  The matrix (34.4 GB) is built and factorised in place; the
BLAS pools are limited to the CPUs this process may use.  Also prints one JSON line with the
oracle's own wall times: the measured FULL-SIZE CPU baseline bench.py quotes (`cpu_baseline`).

The HIP path is not involved: `tests/test_full_size_gpu.py` compares it with these files.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {
    "C3": dict(kernel="rbf", lengthscale=(0.25,), golden="G5"),
    "C5": dict(kernel="rbf", lengthscale=(0.3, 0.2, 0.25), golden="G6"),
}
SF2, SN2, SEED, DIM = 1.5, 1e-2, 12345, 3
ALPHA_ROWS_SEED, ALPHA_ROWS = 777, 1024


def log(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_budget():
    """CPUs this process may use: min(scheduler affinity, cgroup quota)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        pass
    return aff, quota, int(max(1, min(aff, quota if quota else aff)))


def alpha_rows(N):
    return np.sort(np.random.default_rng(ALPHA_ROWS_SEED).choice(N, size=min(ALPHA_ROWS, N), replace=False))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), required=True)
    ap.add_argument("--ntrain", type=int, default=65536)
    ap.add_argument("--mtest", type=int, default=4096)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from scipy.linalg import solve_triangular
    from scipy.spatial import distance as dst
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle.gp_oracle import chol_lower_blocked, synthetic_problem
    cfg = CONFIGS[a.config]
    aff, quota, usable = cpu_budget()
    blas_max = max((int(p.get("num_threads") or 1) for p in threadpool_info() if p.get("user_api") == "blas"), default=1)
    threads = max(1, min(usable, blas_max))
    threadpool_limits(limits=threads, user_api="blas")
    N, M = a.ntrain, a.mtest
    ls = np.asarray(cfg["lengthscale"], dtype=np.float64)
    X, y, Xs = synthetic_problem(N, DIM, M, SEED)
    t0 = time.time()
    Xl = X / ls
    K = np.empty((N, N))
    step = 4096
    for i in range(0, N, step):                       # row blocks: progress + bounded temporaries
        blk = dst.cdist(Xl[i:i + step], Xl, "sqeuclidean")
        blk *= -0.5
        np.exp(blk, out=blk)
        blk *= SF2
        K[i:i + step] = blk
        if (i // step) % 4 == 0:
            log(f"kernel rows {i}/{N}")
    K[np.diag_indices_from(K)] += SN2
    t1 = time.time()
    log("cholesky ...")
    L = chol_lower_blocked(K)
    t2 = time.time()
    log(f"cholesky done in {t2 - t1:.1f} s")
    z = solve_triangular(L, y, lower=True, check_finite=False)
    alpha = solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    t3 = time.time()
    log("predict ...")
    Ks = dst.cdist(Xs / ls, Xl, "sqeuclidean")
    Ks *= -0.5
    np.exp(Ks, out=Ks)
    Ks *= SF2
    mean = Ks @ alpha
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False, overwrite_b=True)
    var = SF2 - np.einsum("ij,ij->j", V, V)
    t4 = time.time()
    rows = alpha_rows(N)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(
        a.out, config=a.config, kernel=cfg["kernel"], lengthscale=ls, sf2=SF2, sn2=SN2, jitter=0.0,
        N=N, d=DIM, M=M, seed=SEED, mean=mean, var=var, logdet=logdet, alpha_rows=rows, alpha_sel=alpha[rows],
        alpha_absmax=float(np.abs(alpha).max()),
        digest=np.array([X.sum(), y.sum(), Xs.sum(), float(X[N // 2, 1]), float(y[N - 1])]))
    rec = {"config": f"{a.config} full-size oracle: N={N} d={DIM} {cfg['kernel']} ls={[float(v) for v in ls]} fp64 M={M}",
           "oracle_fit_predict_s": t4 - t0, "oracle_points_per_s": (N + M) / (t4 - t0),
           "oracle_kbuild_s": t1 - t0, "oracle_cholesky_s": t2 - t1, "oracle_solve_s": t3 - t2,
           "oracle_predict_s": t4 - t3, "cholesky_gflops": N ** 3 / 3.0 / (t2 - t1) / 1e9,
           "blas_threads": threads, "affinity_cpus": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
           "golden": os.path.basename(a.out), "command": "python " + " ".join(sys.argv)}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
