"""GPU: end-to-end fit()/predict() parity of the HIP path against the oracle and the
golden fixtures.  Bar: 1e-6 relative fp64 (north_star), elementwise
  |d mean| <= 1e-6 * max(|mean_ref|, 1e-6),  |d var| <= 1e-6 * max(var_ref, 1e-6 * sf2).
"""
import os

import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, kernel_matrix, synthetic_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def assert_parity(mean, var, mean_ref, var_ref, sf2):
    dm = np.abs(mean - mean_ref) / np.maximum(np.abs(mean_ref), 1e-6)
    dv = np.abs(var - var_ref) / np.maximum(var_ref, 1e-6 * sf2)
    assert dm.max() <= RTOL, f"mean rel err {dm.max():.3e}"
    assert dv.max() <= RTOL, f"var rel err {dv.max():.3e}"
    return dm.max(), dv.max()


@pytest.mark.parametrize("name", ["G1", "G2", "G3"])
def test_golden_fixtures(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    sf2 = float(z["variance"])
    gp = GP(kernel=str(z["kernel"]), lengthscale=z["lengthscale"], variance=sf2,
            noise=float(z["noise"]), jitter=float(z["jitter"]))
    gp.fit(z["X"], z["y"])
    assert gp.info_ == 0
    mean, var = gp.predict(z["Xs"])
    assert_parity(mean, var, z["mean"], z["var"], sf2)
    assert_parity(mean, var, z["sk_mean"], z["sk_var"], sf2)   # scikit-learn's numbers too
    assert np.max(np.abs(gp.alpha_ - z["alpha"])) <= 1e-7 * np.max(np.abs(z["alpha"]))
    assert abs(gp.log_det_ - float(z["logdet"])) <= 1e-10 * abs(float(z["logdet"]))
    assert abs(gp.log_marginal_likelihood(z["y"]) - float(z["lml"])) <= 1e-9 * abs(float(z["lml"]))
    m_only = gp.predict(z["Xs"], return_var=False)   # K* alpha; with variance the mean is V^T z
    assert np.max(np.abs(m_only - mean)) <= 1e-9 * max(1.0, np.abs(mean).max())
    gp.close()


@pytest.mark.parametrize("N,d,M,kernel,ls,block", [
    (1000, 3, 77, "rbf", 0.25, 0),                 # ragged: N, M not multiples of any tile
    (1, 2, 1, "rbf", 0.25, 0),                     # smallest problem
    (129, 1, 300, "matern52", 0.4, 128),
    (1500, 3, 130, "matern52", (0.3, 0.2, 0.25), 256),
    (2500, 2, 64, "rbf", (0.3, 0.2), 512),
    (300, 32, 50, "rbf", 2.0, 0),                  # largest supported input dimension (generic-d kernel)
    (700, 7, 33, "matern52", (1.0, 0.8, 1.2, 0.9, 1.1, 1.0, 0.7), 0),
    (64, 3, 4097, "rbf", 0.25, 0),                 # more query points than training points
])
def test_ragged_sizes_vs_oracle(N, d, M, kernel, ls, block):
    X, y, Xs = synthetic_problem(N, d, M, seed=N + M)
    ref = OracleGP(kernel, ls, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0, block=block) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        assert_parity(mean, var, mr, vr, 1.5)
        assert abs(gp.log_det_ - ref.log_det_) <= 1e-9 * max(1.0, abs(ref.log_det_))


def test_config_C2_n8192_full_oracle():
    """BASELINE.json configs[1]: N=8192, d=3, RBF, fp64, M=4096 — full CPU oracle."""
    X, y, Xs = synthetic_problem(8192, 3, 4096)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        em, ev = assert_parity(mean, var, mr, vr, 1.5)
        print(f"C2 parity: mean {em:.2e} var {ev:.2e}; timings {gp.timings_}")
        assert abs(gp.log_det_ - ref.log_det_) <= 1e-9 * abs(ref.log_det_)


@pytest.mark.parametrize("batch", [128, 256, 640])
def test_predict_in_batches_is_identical(monkeypatch, batch):
    """Query rows are independent: batches of GPX_PRED_BATCH rows through one V^T buffer give
    the very same numbers as one pass (ragged last batch, mean-only route included)."""
    X, y, Xs = synthetic_problem(1500, 3, 700, seed=21)
    Y = np.stack([y, np.sin(3 * y)], axis=1)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0) as gp:
        gp.fit(X, Y)
        m1, v1 = gp.predict(Xs)
        mo1 = gp.predict(Xs, return_var=False)
        monkeypatch.setenv("GPX_PRED_BATCH", str(batch))
        m2, v2 = gp.predict(Xs)
        mo2 = gp.predict(Xs, return_var=False)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2) and np.array_equal(mo1, mo2)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, Y)
    mr, vr = ref.predict(Xs)
    for c in range(2):
        assert_parity(m2[:, c], v2, mr[:, c], vr, 1.5)


def test_n32768_full_oracle():
    """Half the bench size against the FULL CPU oracle (level-3 blocked Cholesky, ~30 s on the box's
    16 CPUs, 8.6 GB): a driver-run parity point between C2 and the one-off full-size run of
    tools/full_oracle_c3.py (profiles/r02_c3_full_oracle_parity.json)."""
    N, M = 32768, 1024
    X, y, Xs = synthetic_problem(N, 3, M)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, chol="blocked").fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        em, ev = assert_parity(mean, var, mr, vr, 1.5)
        assert np.max(np.abs(gp.alpha_ - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
        assert abs(gp.log_det_ - ref.log_det_) <= 1e-9 * abs(ref.log_det_)
        print(f"N=32768 parity: mean {em:.2e} var {ev:.2e}")


def test_multi_output_and_noise_flag():
    X, y, Xs = synthetic_problem(700, 3, 50, seed=11)
    Y = np.stack([y, np.cos(y), 2 * y - 1], axis=1)
    ref = OracleGP("rbf", 0.3, 1.2, 2e-2, jitter=0.0).fit(X, Y)
    mr, vr = ref.predict(Xs, include_noise=True)
    with GP("rbf", 0.3, 1.2, 2e-2, jitter=0.0) as gp:
        mean, var = gp.fit(X, Y).predict(Xs, include_noise=True)
        assert mean.shape == (50, 3)
        for c in range(3):
            assert_parity(mean[:, c], var, mr[:, c], vr, 1.2)
        assert gp.alpha_.shape == (700, 3)


def test_max_targets_share_one_factorisation():
    """k = 64 right-hand sides (the ABI maximum) through one Cholesky factor."""
    X, y, Xs = synthetic_problem(900, 3, 40, seed=5)
    rng = np.random.default_rng(1)
    Y = y[:, None] * rng.uniform(0.5, 2.0, 64)[None, :] + 0.05 * rng.standard_normal((900, 64))
    ref = OracleGP("matern52", 0.3, 1.0, 1e-2, jitter=0.0).fit(X, Y)
    mr, vr = ref.predict(Xs)
    with GP("matern52", 0.3, 1.0, 1e-2, jitter=0.0) as gp:
        mean, var = gp.fit(X, Y).predict(Xs)
        assert mean.shape == (40, 64)
        for c in (0, 31, 63):
            assert_parity(mean[:, c], var, mr[:, c], vr, 1.0)
        assert np.max(np.abs(gp.alpha_ - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
    with pytest.raises(Exception):
        GP("rbf", 0.3).fit(X, np.zeros((900, 65)))


def test_release_scratch_keeps_the_fit():
    torch = pytest.importorskip("torch")
    X, y, Xs = synthetic_problem(3000, 3, 500, seed=6)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        m1, v1 = gp.fit(X, y).predict(Xs)
        l1, g1 = gp.lml_gradient()
        free_before = torch.cuda.mem_get_info(0)[0]
        gp.release_scratch()
        # L^-T, V^T, ... are gone.  (">=": after a test that held > 200 GB the runtime may serve and take back
        # these 72 MB allocations from memory it keeps mapped, and the driver's free count does not move)
        assert torch.cuda.mem_get_info(0)[0] >= free_before
        m2, v2 = gp.predict(Xs)
        l2, g2 = gp.lml_gradient()
        assert np.array_equal(m1, m2) and np.array_equal(v1, v2) and l1 == l2 and np.array_equal(g1, g2)


def test_state_round_trip_is_bit_identical():
    """get_state -> JSON -> from_state -> fit reproduces the model bit for bit (the path is
    deterministic), including ARD lengthscales and the escalated jitter bookkeeping."""
    import json
    X, y, Xs = synthetic_problem(1500, 3, 200, seed=8)
    with GP("matern52", (0.3, 0.2, 0.25), 1.3, 5e-3, block=256) as gp:
        assert "fitted" not in gp.get_state()
        m1, v1 = gp.fit(X, y).predict(Xs)
        a1 = gp.alpha_.copy()
        st = json.loads(json.dumps(gp.get_state()))
    assert st["fitted"] == {"N": 1500, "d": 3, "k": 1, "jitter_used": st["jitter"], "log_det": st["fitted"]["log_det"]}
    with GP.from_state(st) as gp2:
        assert gp2.kernel == "matern52" and gp2.block == 256 and list(gp2.lengthscale) == [0.3, 0.2, 0.25]
        m2, v2 = gp2.fit(X, y).predict(Xs)
        assert np.array_equal(m1, m2) and np.array_equal(v1, v2) and np.array_equal(a1, gp2.alpha_)
        assert gp2.log_det_ == st["fitted"]["log_det"]
    with pytest.raises(ValueError):
        GP.from_state({"format": 7})


def test_refit_reuses_handle_and_permutation_invariance():
    X, y, Xs = synthetic_problem(640, 2, 40, seed=2)
    with GP("matern52", 0.35, 1.0, 1e-2) as gp:
        m1, v1 = gp.fit(X, y).predict(Xs)
        p = np.random.default_rng(0).permutation(640)
        m2, v2 = gp.fit(X[p], y[p]).predict(Xs)
        assert np.max(np.abs(m1 - m2)) <= 1e-9 and np.max(np.abs(v1 - v2)) <= 1e-9
        assert np.all(v1 > -1e-9) and np.all(v1 <= 1.0 + 1e-12)


def test_not_positive_definite_escalates_then_raises():
    X = np.zeros((200, 2))
    X[:, 0] = np.repeat(np.linspace(0, 1, 100), 2)       # duplicated points, zero noise
    y = np.sin(X[:, 0])
    with GP("rbf", 0.5, 1.0, noise=0.0, jitter=0.0, max_tries=1) as gp:
        with pytest.raises(np.linalg.LinAlgError):
            gp.fit(X, y)
        assert gp.info_ > 0
        with pytest.raises(RuntimeError):
            gp.predict(X[:3])
    with GP("rbf", 0.5, 1.0, noise=0.0, jitter=0.0, max_tries=12) as gp:
        gp.fit(X, y)                                       # jitter x10 per try until PD
        assert gp.info_ == 0 and gp.jitter_used_ > 0.0


def test_device_tensor_inputs():
    torch = pytest.importorskip("torch")
    X, y, Xs = synthetic_problem(900, 3, 120, seed=4)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    dev = torch.device("cuda:0")
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0) as gp:
        gp.fit(torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev))
        mean, var = gp.predict(torch.from_numpy(Xs).to(dev))
        assert mean.is_cuda and var.is_cuda
        assert_parity(mean.cpu().numpy(), var.cpu().numpy(), mr, vr, 1.5)


def test_config_C3_n65536_properties():
    """BASELINE.json configs[2] (the bench workload): N=65536, d=3, RBF, M=4096.  No full CPU
    oracle at this size in the test-suite (34 GB, minutes); size-independent properties:
      (i)  residual  || K alpha - y || on 512 random rows, K rows regenerated by the oracle kernel;
      (ii) predicting AT training points: mean_i = y_i - (sn2 + jitter) alpha_i  (exact identity);
      (iii) 0 <= var <= sf2 and var at training points <= sn2-ish (posterior shrinkage).
    """
    N, d, M = 65536, 3, 4096
    X, y, Xs = synthetic_problem(N, d, M)
    sf2, sn2 = 1.5, 1e-2
    import torch
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]             # what earlier tests left allocated is not this handle's
    with GP("rbf", 0.25, sf2, sn2, jitter=0.0, profile=True) as gp:
        gp.fit(X, y)
        assert gp.info_ == 0
        alpha = gp.alpha_
        rows = np.random.default_rng(1).choice(N, 512, replace=False)
        Kr = kernel_matrix(X[rows], X, "rbf", 0.25, sf2)
        Kr[np.arange(512), rows] += sn2
        resid = np.abs(Kr @ alpha - y[rows])
        assert resid.max() <= 1e-8 * np.abs(y).max(), resid.max()
        mean_t, var_t = gp.predict(X[rows])
        assert np.max(np.abs(mean_t - (y[rows] - sn2 * alpha[rows]))) <= 1e-8
        assert np.all(var_t > 0) and np.all(var_t < sn2)
        mean, var = gp.predict(Xs)
        assert np.all(np.isfinite(mean)) and np.all(var > 0) and np.all(var < sf2)
        # mean through an independent route: K* alpha on the CPU for 256 test points
        Ks = kernel_matrix(Xs[:256], X, "rbf", 0.25, sf2)
        assert np.max(np.abs(Ks @ alpha - mean[:256])) <= 1e-9 * max(1.0, np.abs(mean).max())
        print("C3 timings:", gp.timings_)
        # M = 65536 query points at N = 65536: K* / V^T would be 34 GB in one piece; predict
        # streams them in batches of 8192 rows through one 4.3 GB buffer.  Stated budget for
        # the handle (round 4: 2048-wide panels at this size): factor 34.4 GB + panels 2.2 GB + V^T batch 4.3 GB + block
        # inverses 1.1 GB + compact blocks and split-K partial tiles of one batch 0.35 GB + alpha / right-hand-side
        # rows, points, outputs < 44 GB (with K* / V^T in one piece: 34 GB more).
        Xbig = np.random.default_rng(2).uniform(0, 1, (65536, d))
        Xbig[:M] = Xs
        mb, vb = gp.predict(Xbig)
        used = free0 - torch.cuda.mem_get_info(0)[0]
        assert used <= 44e9, f"the handle holds {used / 1e9:.1f} GB"
        assert np.array_equal(mb[:M], mean) and np.array_equal(vb[:M], var)   # same rows, batched or not
        assert np.all(np.isfinite(mb)) and np.all(vb > 0) and np.all(vb < sf2)
        print("M=65536 predict:", {k_: round(v_, 1) for k_, v_ in gp.timings_.items() if k_ in ("kstar", "trsm", "mean", "var", "predict_total")})


@pytest.mark.parametrize("name", ["G1", "G2", "G3"])
def test_lml_gradient_matches_golden_fixtures(golden_dir, name):
    """gpx_lml_grad against the gradient fixtures: the oracle's R&W eq. 5.9 value and
    scikit-learn's ``log_marginal_likelihood(theta, eval_gradient=True)`` for the same inputs
    (oracle/make_golden.py).  RBF scalar, Matern-5/2 scalar, RBF ARD."""
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    with GP(kernel=str(z["kernel"]), lengthscale=z["lengthscale"], variance=float(z["variance"]),
            noise=float(z["noise"]), jitter=0.0) as gp:
        gp.fit(z["X"], z["y"])
        lml, grad = gp.lml_gradient()
        assert abs(lml - float(z["lml"])) <= 1e-9 * abs(float(z["lml"]))
        scale = np.abs(z["sk_lml_grad"]).max()
        assert np.max(np.abs(grad - z["lml_grad"])) <= 1e-8 * scale, (grad, z["lml_grad"])
        assert np.max(np.abs(grad - z["sk_lml_grad"])) <= 1e-8 * scale
        lml2, grad2 = gp.lml_gradient()                     # deterministic: fixed-order reductions
        assert lml2 == lml and np.array_equal(grad, grad2)
        mean, var = gp.predict(z["Xs"])                      # the factor is untouched by the gradient
        assert_parity(mean, var, z["mean"], z["var"], float(z["variance"]))


@pytest.mark.parametrize("N,d,k,kernel,ls,block", [
    (1000, 3, 1, "matern52", (0.3, 0.2, 0.25), 0),     # ragged N: padded rows must not contribute
    (700, 2, 3, "rbf", 0.3, 0),                        # three targets share K^-1
    (2300, 3, 2, "rbf", (0.3, 0.2, 0.25), 256),        # several 1024-blocks of the L^-T look-ahead chain
    (300, 7, 1, "matern52", (1.0, 0.8, 1.2, 0.9, 1.1, 1.0, 0.7), 0),   # generic-d epilogue
    (129, 1, 1, "rbf", 0.4, 0),
])
def test_lml_gradient_vs_oracle(N, d, k, kernel, ls, block):
    X, y, _ = synthetic_problem(N, d, 1, seed=N)
    Y = y if k == 1 else np.stack([np.cos(c * y) + 0.3 * c for c in range(k)], axis=1)
    ref = OracleGP(kernel, ls, 1.3, 2e-2, jitter=0.0).fit(X, Y)
    want = ref.lml_gradient()
    with GP(kernel, ls, 1.3, 2e-2, jitter=0.0, block=block) as gp:
        lml, grad = gp.fit(X, Y).lml_gradient()
    assert abs(lml - ref.log_marginal_likelihood()) <= 1e-9 * abs(ref.log_marginal_likelihood())
    assert np.max(np.abs(grad - want)) <= 1e-8 * np.abs(want).max(), (grad, want)


def test_lml_gradient_at_scale_against_central_differences():
    """N = 16384: no CPU oracle for K^-1 at this size in the suite; the analytic gradient must
    match central differences of the GPU's own log marginal likelihood, and cost about two
    fits — not one fit per parameter."""
    N = 16384
    X, y, _ = synthetic_problem(N, 3, 1)
    ls, sf2, sn2 = np.array([0.3, 0.2, 0.25]), 1.5, 1e-2
    with GP("rbf", ls, sf2, sn2, jitter=0.0) as gp:
        gp.fit(X, y)
        lml, grad = gp.lml_gradient()
        tm = gp.timings_
        assert abs(lml - gp.log_marginal_likelihood(y)) <= 1e-10 * abs(lml)
    h = 1e-4
    v0 = np.log(np.concatenate([ls, [sf2, sn2]]))
    for i in range(5):
        f = []
        for sgn in (+1, -1):
            v = v0.copy()
            v[i] += sgn * h
            with GP("rbf", np.exp(v[:3]), float(np.exp(v[3])), float(np.exp(v[4])), jitter=0.0) as g2:
                f.append(g2.fit(X, y).log_marginal_likelihood(y))
        fd = (f[0] - f[1]) / (2 * h)
        assert abs(fd - grad[i]) <= 2e-5 * np.abs(grad).max(), (i, fd, grad[i])
    print(f"N={N}: fit {tm['fit_total']:.1f} ms, gradient {tm['grad_total']:.1f} ms "
          f"(L^-T {tm['grad_trtri']:.1f}, fused K^-1 trace {tm['grad_trace']:.1f})")


def test_hyperparameter_optimisation_climbs_the_marginal_likelihood():
    """§8(f) rank 1: L-BFGS-B over log-hyper-parameters with the analytic gradient, every
    evaluation one GPU fit + one gpx_lml_grad.  The likelihood it reports must be the oracle's at
    the same point, must not be below the start, and must reach the neighbourhood of the
    generating parameters' likelihood."""
    rng = np.random.default_rng(3)
    N = 500
    X = rng.uniform(0, 1, (N, 2))
    Ktrue = kernel_matrix(X, X, "rbf", (0.2, 0.4), 1.3) + 0.02 * np.eye(N)
    y = np.linalg.cholesky(Ktrue) @ rng.standard_normal(N)
    truth = OracleGP("rbf", (0.2, 0.4), 1.3, 0.02, jitter=0.0).fit(X, y).log_marginal_likelihood()
    with GP("rbf", (1.0, 1.0), 0.5, 0.2, jitter=0.0) as gp:
        start = gp.fit(X, y).log_marginal_likelihood(y)
        res = gp.optimize(X, y, maxiter=60)
        final = gp.log_marginal_likelihood(y)
        assert abs(final + res.fun) <= 1e-9 * abs(final)
        assert final >= start and final >= truth - 3.0          # within a few nats of the truth
        ref = OracleGP("rbf", gp.lengthscale, gp.variance, gp.noise, jitter=0.0).fit(X, y)
        assert abs(final - ref.log_marginal_likelihood()) <= 1e-8 * abs(final)
        assert 0.1 < gp.lengthscale[0] < 0.4 and 0.2 < gp.lengthscale[1] < 0.8


_SCHED_BASE = []


def schedule_variants_baseline():
    """(X, y, Xs, oracle mean / var, default-schedule mean / var / alpha / logdet) at N = 12288 — computed by the first
    variant that runs (no schedule switch is set at that point: monkeypatch sets them after this call), kept for the others."""
    if not _SCHED_BASE:
        X, y, Xs = synthetic_problem(12288, 3, 300, seed=31)
        ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
        mr, vr = ref.predict(Xs)
        with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
            m0, v0 = gp.fit(X, y).predict(Xs)
            _SCHED_BASE.append((X, y, Xs, mr, vr, m0, v0, gp.alpha_.copy(), gp.log_det_))
    return _SCHED_BASE[0]


@pytest.mark.parametrize("env", [{"GPX_FUSED_STRIP": "1"}, {"GPX_DIAG_STEP": "64"},
                                 {"GPX_FUSED_STRIP": "1", "GPX_DIAG_STEP": "64"}, {"GPX_CU_SELF_RESERVE": "1"}, {"GPX_CU_SELF_RESERVE": "4"},
                                 {"GPX_CHAIN_FLAG": "0"},
                                 {"GPX_SPLIT_STRIP": "0"}, {"GPX_SPLIT_STRIP": "0", "GPX_CHAIN_FLAG": "0"},
                                 {"GPX_REST_SPLIT": "0"}, {"GPX_REST_SPLIT": "4"}, {"GPX_SOLVE_TOP": "0"}])
def test_schedule_variants_give_the_same_factorisation(monkeypatch, env):
    """Round-3 schedule switches of the blocked Cholesky: the fused trailing update (strip + rest in ONE
    launch, device-counter hand-over to the look-ahead stream: gemm_nt_fused_kernel / wait_counter_kernel)
    the 64-wide diagonal stepping (the default steps 128 columns per launch: potf2_128_kernel), the self-reserving
    trailing update of the chain-bound panels (round 4: persistent workgroups that leave k CUs per XCD to the diagonal
    chain and take their tiles from device counters: gemm_nt_resv_kernel, GPX_CU_SELF_RESERVE=k), and the
    hipEvent hand-over of the diagonal chain (GPX_CHAIN_FLAG=0; the default is a device flag: what `rocprofv3 --pmc` needs).
    N = 12288 with 1024-panels: 11 trailing updates, the first 8 of them large enough to fuse.  Against
    the oracle at 1e-6 and against the default schedule: the fused launch does the same arithmetic per
    tile (bit-identical); the diagonal stepping changes the association inside a 128 x 128 tile only."""
    X, y, Xs, mr, vr, m0, v0, a0, ld0 = schedule_variants_baseline()   # the oracle and the default schedule: once for all variants
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        assert gp.info_ == 0
        assert_parity(mean, var, mr, vr, 1.5)
        if "GPX_DIAG_STEP" not in env:         # same arithmetic per element, other launch / wave geometry
            assert np.array_equal(mean, m0) and np.array_equal(var, v0) and np.array_equal(gp.alpha_, a0)
        else:
            assert np.max(np.abs(mean - m0)) <= 1e-9 * np.abs(m0).max() and abs(gp.log_det_ - ld0) <= 1e-12 * abs(ld0)
        m2, v2 = gp.fit(X, y).predict(Xs)                  # refit: the strip counter is reset per fit
        assert np.array_equal(m2, mean) and np.array_equal(v2, var)
