#!/usr/bin/env python3
"""One-off: FULL CPU oracle at the bench size (BASELINE.json configs[2]: N=65536, d=3, RBF,
fp64, M=4096) against the HIP path — SURVEY.md §4/§8d "65536 (full oracle once; 34 GB host
RAM)".  Same arithmetic as oracle/gp_oracle.py (cdist -> exp -> scipy cholesky ->
solve_triangular), done in place to fit one 34.4 GB matrix.  Prints progress lines and one
JSON line.   python tools/full_oracle_c3.py [--ntrain 65536]
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
    from scipy.linalg import cholesky, solve_triangular
    from scipy.spatial import distance as dst
    from bench import synthetic
    from gaussianprocesspathmodelling_amd import GP
    N, M, ls, sf2, sn2 = a.ntrain, a.mtest, 0.25, 1.5, 1e-2
    X, y, Xs = synthetic(N, 3, M, 12345)
    t0 = time.time()
    with GP("rbf", ls, sf2, sn2, jitter=0.0) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        alpha, logdet = gp.alpha_.copy(), gp.log_det_
    log(f"HIP path done in {time.time() - t0:.1f} s")
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
    L = cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    log(f"cholesky done in {time.time() - t1:.1f} s")
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
    rm = np.abs(mean - m_ref) / np.maximum(np.abs(m_ref), 1e-6)
    rv = np.abs(var - v_ref) / np.maximum(v_ref, 1e-6 * sf2)
    out = {"config": f"C3 full oracle: N={N} d=3 RBF fp64 M={M}", "mean_rel_max": float(rm.max()),
           "var_rel_max": float(rv.max()), "alpha_err_over_max": float(np.abs(alpha - a_ref).max() / np.abs(a_ref).max()),
           "logdet_rel": abs(logdet - ld_ref) / abs(ld_ref), "tolerance": 1e-6,
           "pass": bool(rm.max() <= 1e-6 and rv.max() <= 1e-6), "oracle_seconds": time.time() - t0}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
