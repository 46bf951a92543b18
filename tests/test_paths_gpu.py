"""GPU: the batched path-distance kernel and the k-means loop against the reference's own
numbers (tests/golden/G4.json) and the oracle's restatement."""
import json
import os

import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import paths as gpaths
from oracle import gp_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g4(golden_dir):
    return json.load(open(os.path.join(golden_dir, "G4.json")))


def test_distance_kernel_matches_reference_calc_distance(g4):
    t = gpaths.read_csv(g4["csv"], 10)
    keys = t.keys()
    xy = np.ascontiguousarray(t.as_array()[:, :, 1:3])
    D = gpaths.path_distance_matrix(xy, xy)
    for a, b, ref in g4["distance_pairs"]:
        assert abs(D[keys.index(a), keys.index(b)] - ref) <= 1e-12 * ref
    assert np.all(np.diag(D) == 0.0) and np.allclose(D, D.T, rtol=1e-15)


@pytest.mark.parametrize("P,C,L", [(1, 1, 33), (1003, 3, 33), (257, 70, 33), (64, 5, 64), (10, 2, 1)])
def test_distance_kernel_vs_oracle(P, C, L):
    rng = np.random.default_rng(P + C + L)
    paths = rng.uniform(-5e4, 5e4, (P, L, 2))
    cents = rng.uniform(-5e4, 5e4, (C, L, 2))
    D = gpaths.path_distance_matrix(paths, cents)
    ref = gp_oracle.path_distance_matrix(paths, cents)
    assert np.max(np.abs(D - ref) / ref) <= 1e-13


def test_kmeans_reproduces_reference_partition(g4):
    t = gpaths.read_csv(g4["csv"], 10)
    clusters = gpaths.kmeans(t, 3, init_keys=g4["kmeans_init_keys"])
    assert sorted(sorted(v) for v in clusters.values()) == g4["kmeans_partition"]
    # same draw as the reference when Python's random is seeded the same way
    import random
    random.seed(123)
    clusters2 = t.kmeansclustering(3)
    assert sorted(sorted(v) for v in clusters2.values()) == g4["kmeans_partition"]


def test_kmeans_separates_obvious_groups_and_feeds_the_gp(g4):
    from gaussianprocesspathmodelling_amd import GP
    t = gpaths.read_csv(g4["csv"])                     # all 12 valid paths, 3 spatial groups
    groups = gpaths.kmeans(t, 3, init_keys=["P00", "P01", "P03"])
    assert sum(len(v) for v in groups.values()) == 12
    # one GP per cluster: x(t), y(t) as two targets sharing one factorisation
    ids = max(groups.values(), key=len)
    X, Y, _ = gpaths.to_gp_inputs(t, keys=ids)
    with GP("matern52", 0.3, variance=float(Y.var()), noise=1e-4 * float(Y.var())) as gp:
        mean, var = gp.fit(X, Y - Y.mean(0)).predict(X[:5])
        assert mean.shape == (5, 2) and np.all(np.isfinite(mean)) and np.all(var >= 0)


def _synthetic_groups(n_groups=3, per_group=7, seed=5):
    """Paths of 33 points in a few spatial groups: smooth curves + per-path offsets + noise."""
    rng = np.random.default_rng(seed)
    t = gpaths.Trajectories()
    truth = {}
    tt = np.arange(33, dtype=float) * 40.0
    for g in range(n_groups):
        cx, cy = 4000.0 * g, 2500.0 * (g % 2)
        for p in range(per_group):
            tr = gpaths.Trajectory()
            ox, oy = rng.normal(0, 60, 2)
            for i in range(33):
                s = i / 32.0
                tr.add_point(tt[i], cx + ox + 1500.0 * s + 200.0 * np.sin(3 * s + g) + rng.normal(0, 15),
                             cy + oy + 900.0 * s * s + rng.normal(0, 15))
            key = f"G{g}P{p}"
            t.add_trajectory(key, tr)
            truth.setdefault(g, []).append(key)
    return t, truth


def _oracle_model(t, keys, kernel, ls, var, noise, q):
    X, Y, (lo, span) = gpaths.to_gp_inputs(t, keys)
    mu, sd = Y.mean(0), Y.std(0)
    o = gp_oracle.OracleGP(kernel, ls, var, noise, jitter=1e-10 * var).fit(X, (Y - mu) / sd)
    m, v = o.predict((q.reshape(-1, 1) - lo) / span)
    return m * sd + mu, v[:, None] * sd ** 2


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_one_gp_per_cluster_matches_the_oracle(devices):
    """fit_path_models: k-means clusters -> one two-target GP each; replicas dealt over the
    device list (here two worker threads sharing GPU 0), each checked against the CPU oracle."""
    t, truth = _synthetic_groups()
    clusters = gpaths.kmeans(t, 3, init_keys=["G0P0", "G1P0", "G2P0"])
    assert sorted(sorted(v) for v in clusters.values()) == sorted(sorted(v) for v in truth.values())
    models = gpaths.fit_path_models(t, clusters, devices=devices, kernel="matern52", lengthscale=0.3,
                                    variance=1.0, noise=0.02)
    try:
        assert list(models) == list(clusters)
        q = np.linspace(-50.0, 1400.0, 57)           # raw time stamps, inside and outside the data
        for cid, m in models.items():
            assert m.keys == clusters[cid]
            mean, var = m.predict(q)
            om, ov = _oracle_model(t, clusters[cid], "matern52", 0.3, 1.0, 0.02, q)
            assert mean.shape == (57, 2) and var.shape == (57, 2)
            assert np.max(np.abs(mean - om)) <= 1e-6 * np.max(np.abs(om))
            assert np.max(np.abs(var - ov)) <= 1e-6 * np.max(np.abs(ov))
            # mean only = K* alpha, with the variance = V^T z: two summation orders of the same mean
            assert np.max(np.abs(m.predict(q, return_var=False) - mean)) <= 1e-8 * np.max(np.abs(mean))
            # the model follows its own cluster: the posterior mean stays within the spread of the paths
            arr = t.as_array(clusters[cid])
            mid, _ = m.predict(arr[0, :, 0])
            assert np.max(np.abs(mid - arr[:, :, 1:3].mean(0))) < 150.0
    finally:
        for m in models.values():
            m.close()


def test_path_models_with_fitted_hyperparameters():
    t, truth = _synthetic_groups(n_groups=2, per_group=6, seed=9)
    clusters = {g: keys for g, keys in truth.items()}
    clusters["empty"] = []                          # skipped
    base = gpaths.fit_path_models(t, clusters, lengthscale=0.05, noise=0.5)
    opt = gpaths.fit_path_models(t, clusters, lengthscale=0.05, noise=0.5, optimize=True)
    try:
        assert list(opt) == [0, 1]
        for cid in opt:
            X, Y, _ = gpaths.to_gp_inputs(t, clusters[cid])
            Yn = (Y - Y.mean(0)) / Y.std(0)
            assert opt[cid].gp.log_marginal_likelihood(Yn) > base[cid].gp.log_marginal_likelihood(Yn) + 1.0
            assert opt[cid].gp.noise < 0.5            # the data are far less noisy than the start says
    finally:
        for m in list(base.values()) + list(opt.values()):
            m.close()
