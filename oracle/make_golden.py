"""Generate tests/golden/G{1,2,3}.npz — run in the BUILD container only.

TEST INFRASTRUCTURE.  The reference (``/root/reference/GPmap.py``) has no GP path
and no fixtures, so these goldens pin the build-authored
oracle (``oracle/gp_oracle.py``) against scikit-learn 1.7.2's
``GaussianProcessRegressor`` (third-party; present in the build container, not
assumed on the GPU box).  The sklearn results are stored next to the oracle's so
the CPU test-suite can re-check the oracle without sklearn.

    python oracle/make_golden.py          # rewrites tests/golden/G*.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.gp_oracle import OracleGP, synthetic_problem  # noqa: E402

CASES = {
    # name: (N, d, M, kernel, lengthscale, sf2, sn2)         SURVEY.md §8(c)
    "G1": (512, 2, 128, "rbf", 0.25, 1.5, 1e-2),
    "G2": (512, 3, 128, "matern52", 0.25, 1.5, 1e-2),
    "G3": (2048, 3, 512, "rbf", (0.3, 0.2, 0.25), 1.5, 1e-2),
}


def sklearn_reference(X, y, Xs, kernel, ls, sf2, sn2):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern
    ls = np.atleast_1d(np.asarray(ls, float))
    ls = float(ls[0]) if ls.size == 1 else ls
    base = (RBF(ls, "fixed") if kernel == "rbf" else Matern(ls, "fixed", nu=2.5))
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(sf2, "fixed") * base,
                                   alpha=sn2, optimizer=None, normalize_y=False)
    gpr.fit(X, y)
    mean, std = gpr.predict(Xs, return_std=True)
    lml = gpr.log_marginal_likelihood_value_
    return mean, std ** 2, lml


def sklearn_lml_gradient(X, y, kernel, ls, sf2, sn2):
    """scikit-learn's log marginal likelihood and its gradient w.r.t. the log hyper-parameters,
    re-ordered to (lengthscale[0..n_ls), variance, noise).  sklearn's theta of
    Constant * RBF/Matern + White is (log constant, log length_scale..., log noise_level)."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel
    ls = np.atleast_1d(np.asarray(ls, float))
    lsk = float(ls[0]) if ls.size == 1 else ls
    base = RBF(lsk) if kernel == "rbf" else Matern(lsk, nu=2.5)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(sf2) * base + WhiteKernel(sn2), alpha=0.0,
                                   optimizer=None, normalize_y=False)
    gpr.fit(X, y)
    lml, g = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    return float(lml), np.concatenate([g[1:1 + ls.size], [g[0], g[-1]]])


def main():
    outdir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, (N, d, M, kernel, ls, sf2, sn2) in CASES.items():
        X, y, Xs = synthetic_problem(N, d, M)
        gp = OracleGP(kernel=kernel, lengthscale=ls, variance=sf2, noise=sn2, jitter=0.0)
        gp.fit(X, y, keep_K_corner=8)
        mean, var = gp.predict(Xs)
        sk_mean, sk_var, sk_lml = sklearn_reference(X, y, Xs, kernel, ls, sf2, sn2)
        rel_m = np.max(np.abs(mean - sk_mean) / np.maximum(np.abs(sk_mean), 1e-6))
        rel_v = np.max(np.abs(var - sk_var) / np.maximum(np.abs(sk_var), 1e-6 * sf2))
        lml = gp.log_marginal_likelihood()
        print(f"{name}: N={N} d={d} M={M} {kernel}: oracle vs sklearn  "
              f"mean rel {rel_m:.2e}  var rel {rel_v:.2e}  lml {lml:.9f} vs {sk_lml:.9f}")
        assert rel_m < 1e-8 and rel_v < 1e-8, "oracle disagrees with scikit-learn"
        assert abs(lml - sk_lml) < 1e-8 * abs(sk_lml)
        grad = gp.lml_gradient()
        sk_lml2, sk_grad = sklearn_lml_gradient(X, y, kernel, ls, sf2, sn2)
        gerr = np.max(np.abs(grad - sk_grad) / np.maximum(np.abs(sk_grad), 1e-6 * np.abs(sk_grad).max()))
        print(f"      lml gradient (d/dlog ls.., sf2, sn2): oracle {grad}  sklearn {sk_grad}  rel {gerr:.2e}")
        assert abs(sk_lml2 - lml) < 1e-8 * abs(lml) and gerr < 1e-6, "oracle gradient disagrees with scikit-learn"
        np.savez_compressed(
            os.path.join(outdir, f"{name}.npz"),
            X=X, y=y, Xs=Xs, kernel=np.array(kernel), lengthscale=np.atleast_1d(np.asarray(ls, float)),
            variance=np.float64(sf2), noise=np.float64(sn2), jitter=np.float64(0.0),
            K_corner=gp.K_corner_, diagL=np.diag(gp.L_)[:16].copy(), alpha=gp.alpha_,
            mean=mean, var=var, logdet=np.float64(gp.log_det_), lml=np.float64(lml),
            sk_mean=sk_mean, sk_var=sk_var, sk_lml=np.float64(sk_lml),
            lml_grad=grad, sk_lml_grad=sk_grad)
    print("wrote", outdir)


if __name__ == "__main__":
    main()
