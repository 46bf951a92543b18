#!/usr/bin/env python3
"""One-off: FULL CPU oracle at the bench size (BASELINE.json configs[2]: N=65536, d=3, RBF,
fp64, M=4096) against the HIP path — SURVEY.md §4/§8d "65536 (full oracle once; 34 GB host
RAM)".  Same arithmetic as oracle/gp_oracle.py (cdist -> exp -> Cholesky -> solve_triangular),
done in place to fit one 34.4 GB matrix; the Cholesky is the oracle's level-3 blocked variant
(chol_lower_blocked: LAPACK potrf on 2048-blocks, BLAS-3 for the rest — the monolithic potrf of
the bundled OpenBLAS crashed on this matrix with a 16-thread pool and does not scale anyway).  The BLAS pools are limited to the
CPUs this process may use (bench.cpu_budget: affinity and cgroup quota).  Prints progress lines
and one JSON line: parity of the HIP path at the bench size AND the measured full-size CPU
baseline that bench.py quotes (oracle_points_per_s).
    python tools/full_oracle_c3.py > profiles/r02_c3_full_oracle_parity.json
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def log(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntrain", type=int, default=65536)
    ap.add_argument("--mtest", type=int, default=4096)
    a = ap.parse_args()
    from scipy.linalg import solve_triangular
    from oracle.gp_oracle import chol_lower_blocked
    from scipy.spatial import distance as dst
    from threadpoolctl import threadpool_info, threadpool_limits
    from bench import cpu_budget, synthetic
    from gaussianprocesspathmodelling_amd import GP
    aff, quota, usable = cpu_budget()
    blas_max = max((int(p.get("num_threads") or 1) for p in threadpool_info() if p.get("user_api") == "blas"), default=1)
    threads = max(1, min(usable, blas_max))
    threadpool_limits(limits=threads, user_api="blas")
    N, M, ls, sf2, sn2 = a.ntrain, a.mtest, 0.25, 1.5, 1e-2
    X, y, Xs = synthetic(N, 3, M, 12345)
    t0 = time.time()
    with GP("rbf", ls, sf2, sn2, jitter=0.0) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        alpha, logdet = gp.alpha_.copy(), gp.log_det_
    log(f"HIP path done in {time.time() - t0:.1f} s")
    t_cpu = time.time()
    Xl = X / ls
    K = np.empty((N, N))
    step = 4096
    for i in range(0, N, step):                       # row blocks: progress + bounded temporaries
        blk = dst.cdist(Xl[i:i + step], Xl, "sqeuclidean")
        blk *= -0.5
        np.exp(blk, out=blk)
        blk *= sf2
        K[i:i + step] = blk
        if (i // step) % 4 == 0:
            log(f"kernel rows {i}/{N}")
    K[np.diag_indices_from(K)] += sn2
    log("cholesky ...")
    t1 = time.time()
    L = chol_lower_blocked(K)
    t_chol = time.time() - t1
    log(f"cholesky done in {t_chol:.1f} s")
    z = solve_triangular(L, y, lower=True, check_finite=False)
    a_ref = solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    ld_ref = 2.0 * float(np.sum(np.log(np.diag(L))))
    log("predict ...")
    Ks = dst.cdist(Xs / ls, Xl, "sqeuclidean")
    Ks *= -0.5
    np.exp(Ks, out=Ks)
    Ks *= sf2
    m_ref = Ks @ a_ref
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False, overwrite_b=True)
    v_ref = sf2 - np.einsum("ij,ij->j", V, V)
    cpu_s = time.time() - t_cpu
    rm = np.abs(mean - m_ref) / np.maximum(np.abs(m_ref), 1e-6)
    rv = np.abs(var - v_ref) / np.maximum(v_ref, 1e-6 * sf2)
    out = {"config": f"C3 full oracle: N={N} d=3 RBF fp64 M={M}", "mean_rel_max": float(rm.max()),
           "var_rel_max": float(rv.max()), "alpha_err_over_max": float(np.abs(alpha - a_ref).max() / np.abs(a_ref).max()),
           "logdet_rel": abs(logdet - ld_ref) / abs(ld_ref), "tolerance": 1e-6,
           "pass": bool(rm.max() <= 1e-6 and rv.max() <= 1e-6),
           "oracle_fit_predict_s": cpu_s, "oracle_points_per_s": (N + M) / cpu_s, "oracle_cholesky_s": t_chol,
           "blas_threads": threads, "affinity_cpus": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
           "command": "python tools/full_oracle_c3.py"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
