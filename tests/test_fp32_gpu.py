"""GPU: the fp32 instantiation of the whole path (BASELINE.json configs[4], "fp32 + ARD
lengthscales — mixed-precision tolerance study").  Not a 1e-6 path: every kernel,
including the Cholesky, runs in fp32, so the error against the fp64 oracle is
~cond(K) * 6e-8.  The tests pin (i) that the fp32 kernels compute the right thing on a
well-conditioned problem and (ii) the error level the study reports."""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, synthetic_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,kernel,ls,noise", [
    (1000, 200, "rbf", (0.3, 0.2, 0.25), 1e-1),
    (2500, 130, "matern52", (0.3, 0.2, 0.25), 1e-2),
    (129, 64, "rbf", 0.5, 1e-1),
])
def test_fp32_path_tracks_fp64_oracle(N, M, kernel, ls, noise):
    X, y, Xs = synthetic_problem(N, 3, M, seed=N)
    ref = OracleGP(kernel, ls, 1.5, noise, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP(kernel, ls, 1.5, noise, jitter=0.0, dtype="float32") as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        assert gp.info_ == 0
        assert mean.dtype == np.float32 and var.dtype == np.float32 and gp.alpha_.dtype == np.float32
        em = np.max(np.abs(mean - mr)) / np.max(np.abs(mr))
        ev = np.max(np.abs(var - vr)) / 1.5
        el = abs(gp.log_det_ - ref.log_det_) / abs(ref.log_det_)
        print(f"fp32 N={N} {kernel}: mean err {em:.2e} (of max|mean|), var err {ev:.2e} (of sf2), logdet rel {el:.2e}")
        assert em <= 2e-3 and ev <= 2e-3 and el <= 1e-3


@pytest.mark.parametrize("ndev,N,M,kernel,nb,repl", [
    (2, 1000, 200, "rbf", 128, 1), (3, 2500, 130, "matern52", 256, 0), (4, 3300, 300, "rbf", 256, 1),
    (4, 3300, 300, "rbf", 256, 0), (8, 2000, 77, "matern52", 128, 0), (2, 9000, 500, "rbf", 0, -1),
])
def test_fp32_shard_tracks_fp64_oracle_and_the_unsharded_fp32_path(monkeypatch, ndev, N, M, kernel, nb, repl):
    """Round 4: the row-block shard in the handle's element type (it was fp64 only).  fp32 on 2-8 ranks sharing the
    card, both solve modes, against the fp64 oracle at the study's tolerance (test above) and against the unsharded fp32
    handle (same kernels, other blocking: agreement well inside the fp32 error level)."""
    if nb:
        monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    else:
        monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    if repl >= 0:
        monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    ls, noise = (0.3, 0.2, 0.25), 1e-1
    X, y, Xs = synthetic_problem(N, 3, M, seed=N + ndev)
    ref = OracleGP(kernel, ls, 1.5, noise, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP(kernel, ls, 1.5, noise, jitter=0.0, dtype="float32") as one:
        m1, v1 = one.fit(X, y).predict(Xs)
        a1, ld1 = one.alpha_.copy(), one.log_det_
    with GP(kernel, ls, 1.5, noise, jitter=0.0, dtype="float32", devices=ndev, oversubscribe=True) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        assert gp.info_ == 0 and mean.dtype == np.float32 and var.dtype == np.float32 and gp.alpha_.dtype == np.float32
        em = np.max(np.abs(mean - mr)) / np.max(np.abs(mr))
        ev = np.max(np.abs(var - vr)) / 1.5
        el = abs(gp.log_det_ - ref.log_det_) / abs(ref.log_det_)
        assert em <= 2e-3 and ev <= 2e-3 and el <= 1e-3, (em, ev, el)
        assert np.max(np.abs(mean - m1)) <= 1e-3 * np.max(np.abs(m1)) and np.max(np.abs(var - v1)) <= 1e-3 * 1.5
        assert np.max(np.abs(gp.alpha_ - a1)) <= 2e-3 * np.max(np.abs(a1)) and abs(gp.log_det_ - ld1) <= 1e-4 * abs(ld1)
        m2 = gp.predict(Xs, return_var=False)                  # mean-only predict: alpha on the shard
        assert np.max(np.abs(m2 - mean)) <= 1e-3 * np.max(np.abs(mean))
        m3, v3 = gp.fit(X, y).predict(Xs)                        # refit: bit-identical
        assert np.array_equal(m3, mean) and np.array_equal(v3, var)
