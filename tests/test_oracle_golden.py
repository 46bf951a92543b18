"""CPU: the oracle (oracle/gp_oracle.py) against the committed golden fixtures.

The reference has no GP fixtures; tests/golden/G*.npz were written by
oracle/make_golden.py in the build container and carry both the oracle's outputs and
scikit-learn 1.7.2's for the same inputs, so this re-pins the oracle without sklearn.
"""
import os

import numpy as np
import pytest

from oracle.gp_oracle import (OracleGP, chol_lower, chol_lower_blocked, kernel_matrix, synthetic_problem,
                              trsm_right_lower_trans)

CASES = ["G1", "G2", "G3"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(golden_dir, name):
    g = load(golden_dir, name)
    gp = OracleGP(kernel=str(g["kernel"]), lengthscale=g["lengthscale"], variance=float(g["variance"]),
                  noise=float(g["noise"]), jitter=float(g["jitter"]))
    gp.fit(g["X"], g["y"], keep_K_corner=8)
    mean, var = gp.predict(g["Xs"])
    np.testing.assert_allclose(gp.K_corner_, g["K_corner"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(np.diag(gp.L_)[:16], g["diagL"], rtol=1e-12)
    np.testing.assert_allclose(gp.alpha_, g["alpha"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, g["var"], rtol=1e-8, atol=1e-12)
    assert abs(gp.log_det_ - float(g["logdet"])) <= 1e-10 * abs(float(g["logdet"]))
    assert abs(gp.log_marginal_likelihood() - float(g["lml"])) <= 1e-9 * abs(float(g["lml"]))


@pytest.mark.parametrize("name", CASES)
def test_golden_matches_sklearn_values(golden_dir, name):
    """1e-6 relative (the north_star tolerance) between oracle and sklearn outputs."""
    g = load(golden_dir, name)
    sf2 = float(g["variance"])
    assert np.all(np.abs(g["mean"] - g["sk_mean"]) <= 1e-6 * np.maximum(np.abs(g["sk_mean"]), 1e-6))
    assert np.all(np.abs(g["var"] - g["sk_var"]) <= 1e-6 * np.maximum(g["sk_var"], 1e-6 * sf2))
    assert abs(float(g["lml"]) - float(g["sk_lml"])) <= 1e-9 * abs(float(g["sk_lml"]))


def test_synthetic_problem_is_the_golden_input(golden_dir):
    g = load(golden_dir, "G1")
    X, y, Xs = synthetic_problem(512, 2, 128)
    assert np.array_equal(X, g["X"]) and np.array_equal(y, g["y"]) and np.array_equal(Xs, g["Xs"])


@pytest.mark.parametrize("kernel", ["rbf", "matern52"])
def test_kernel_properties(kernel):
    rng = np.random.default_rng(0)
    A = rng.uniform(size=(40, 3))
    K = kernel_matrix(A, A, kernel, (0.3, 0.2, 0.25), 1.5)
    assert np.allclose(K, K.T, rtol=0, atol=1e-15)
    assert np.allclose(np.diag(K), 1.5)
    assert np.linalg.eigvalsh(K + 1e-8 * np.eye(40)).min() > 0
    # permutation equivariance
    p = rng.permutation(40)
    assert np.allclose(kernel_matrix(A[p], A[p], kernel, (0.3, 0.2, 0.25), 1.5), K[np.ix_(p, p)])


def test_interpolation_at_small_noise():
    X, y, _ = synthetic_problem(64, 2, 4, seed=3)
    gp = OracleGP("matern52", 0.3, 1.0, noise=1e-9, jitter=0.0).fit(X, y)
    mean, var = gp.predict(X)
    assert np.max(np.abs(mean - y)) < 1e-3      # residual = noise * alpha
    assert np.max(np.abs(var)) < 1e-8           # latent variance at a training point ~ noise


def test_not_positive_definite_raises():
    X = np.zeros((8, 2))  # identical points, zero noise -> singular
    with pytest.raises(np.linalg.LinAlgError):
        OracleGP("rbf", 1.0, 1.0, noise=0.0, jitter=0.0, max_tries=1).fit(X, np.zeros(8))


def test_multi_output_shares_factor():
    X, y, Xs = synthetic_problem(96, 3, 16, seed=5)
    Y = np.stack([y, 2.0 * y - 1.0], axis=1)
    gp2 = OracleGP("matern52", 0.4, 1.2).fit(X, Y)
    m2, v2 = gp2.predict(Xs)
    gp1 = OracleGP("matern52", 0.4, 1.2).fit(X, Y[:, 1])
    m1, v1 = gp1.predict(Xs)
    assert np.allclose(m2[:, 1], m1, rtol=1e-12) and np.allclose(v2, v1, rtol=1e-12)


def test_linear_algebra_pieces():
    rng = np.random.default_rng(1)
    B = rng.standard_normal((64, 64))
    K = B @ B.T + 64 * np.eye(64)
    L = chol_lower(K)
    assert np.allclose(L @ L.T, K)
    A = rng.standard_normal((32, 64))
    X = trsm_right_lower_trans(A, L)
    assert np.allclose(X @ L.T, A)


@pytest.mark.parametrize("name", ["G1", "G2"])
def test_oracle_lml_gradient_matches_golden_and_sklearn(golden_dir, name):
    """d LML / d log(lengthscale.., variance, noise): oracle vs its stored value and vs
    scikit-learn's ``log_marginal_likelihood(theta, eval_gradient=True)`` (same fixture)."""
    g = load(golden_dir, name)
    gp = OracleGP(kernel=str(g["kernel"]), lengthscale=g["lengthscale"], variance=float(g["variance"]),
                  noise=float(g["noise"]), jitter=0.0).fit(g["X"], g["y"])
    grad = gp.lml_gradient()
    scale = np.abs(g["sk_lml_grad"]).max()
    assert np.max(np.abs(grad - g["lml_grad"])) <= 1e-9 * scale
    assert np.max(np.abs(grad - g["sk_lml_grad"])) <= 1e-8 * scale


@pytest.mark.parametrize("kernel,ls", [("rbf", 0.3), ("matern52", (0.3, 0.5)), ("rbf", (0.2, 0.4))])
def test_oracle_lml_gradient_matches_central_differences(kernel, ls):
    X, y, _ = synthetic_problem(150, 2, 1, seed=8)
    Y = np.stack([y, np.cos(2 * y)], axis=1)          # two targets share the factor
    sf2, sn2 = 1.3, 3e-2
    ls = np.atleast_1d(np.asarray(ls, float))

    def lml(v):
        return OracleGP(kernel, np.exp(v[:ls.size]), np.exp(v[-2]), np.exp(v[-1]), jitter=0.0).fit(X, Y).log_marginal_likelihood()

    v0 = np.log(np.concatenate([ls, [sf2, sn2]]))
    grad = OracleGP(kernel, ls, sf2, sn2, jitter=0.0).fit(X, Y).lml_gradient()
    h = 1e-5
    for i in range(v0.size):
        e = np.zeros_like(v0)
        e[i] = h
        fd = (lml(v0 + e) - lml(v0 - e)) / (2 * h)
        assert abs(fd - grad[i]) <= 1e-6 * max(1.0, np.abs(grad).max()), (i, fd, grad[i])


def test_blocked_cholesky_matches_lapack_and_golden(golden_dir):
    """The level-3 blocked CPU Cholesky (the full-size parity oracle and the stronger CPU baseline)
    against LAPACK potrf and, through OracleGP(chol="blocked"), against the G3 fixture."""
    rng = np.random.default_rng(4)
    for n, nb, rb in ((1, 8, 8), (130, 32, 48), (700, 128, 256), (1000, 2048, 4096)):
        A = rng.standard_normal((n, max(2, n // 3)))
        K = A @ A.T + n * np.eye(n)
        L = np.tril(chol_lower_blocked(K.copy(), nb, rb))
        assert np.max(np.abs(L - chol_lower(K))) <= 1e-13 * np.abs(L).max()
    g = load(golden_dir, "G3")
    gp = OracleGP(str(g["kernel"]), g["lengthscale"], float(g["variance"]), float(g["noise"]), jitter=0.0,
                  chol="blocked").fit(g["X"], g["y"])
    mean, var = gp.predict(g["Xs"])
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, g["var"], rtol=1e-8, atol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        chol_lower_blocked(np.zeros((8, 8)), 4, 4)


@pytest.mark.parametrize("name,ls", [("G5.npz", (0.25,)), ("G6.npz", (0.3, 0.2, 0.25))])
def test_full_size_fixtures_are_what_the_generator_describes(name, ls):
    """G5 / G6 (oracle/make_golden_full.py, N = 65536: 34 GB and minutes of CPU, so not recomputed here):
    the stored digests match the inputs the seed regenerates, the shapes and hyper-parameters are the
    configs', and the stored posterior is a posterior (0 < var < sf2, alpha rows sorted and unique).
    The full-size HIP parity against them is tests/test_full_size_gpu.py."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), allow_pickle=False)
    N, d, M = int(g["N"]), int(g["d"]), int(g["M"])
    assert (N, d, M) == (65536, 3, 4096) and str(g["kernel"]) == "rbf"
    assert tuple(float(v) for v in g["lengthscale"]) == ls and float(g["sf2"]) == 1.5 and float(g["sn2"]) == 1e-2
    X, y, Xs = synthetic_problem(N, d, M, seed=int(g["seed"]))
    digest = np.array([X.sum(), y.sum(), Xs.sum(), float(X[N // 2, 1]), float(y[N - 1])])
    assert np.allclose(digest, g["digest"], rtol=1e-13, atol=0)
    assert g["mean"].shape == (M,) and g["var"].shape == (M,) and g["alpha_sel"].shape == (1024,)
    assert np.all(g["var"] > 0) and np.all(g["var"] < 1.5) and np.all(np.isfinite(g["mean"]))
    rows = g["alpha_rows"]
    assert np.all(np.diff(rows) > 0) and rows[0] >= 0 and rows[-1] < N
    assert np.abs(g["alpha_sel"]).max() <= float(g["alpha_absmax"])
    # a posterior mean of this smooth target stays within the data's range
    assert np.abs(g["mean"]).max() < np.abs(y).max()
