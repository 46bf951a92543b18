"""GPU: gpx_fit_predict — fit and predict (mean + variance) as ONE factorisation pass, the query points' cross-kernel
rows riding through the blocked Cholesky as bordered rows (include/gpx.h, ABI v4).

Bars: against the oracle north_star's 1e-6 (mean and variance, elementwise relative with the usual floors) and
against the two-call path of the same library 1e-9 (the same arithmetic in another summation order).  The reference
holds no GP code: parity unpinned by the reference, as everywhere.
"""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, synthetic_problem

pytestmark = pytest.mark.gpu


def rel(a, b, floor):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.mark.parametrize("N,M,d,k,kernel,ard,block", [
    (8192, 4096, 3, 1, "rbf", False, 0),        # BASELINE.json configs[1] (C2)
    (3000, 700, 3, 1, "rbf", False, 0),         # ragged N and M: padded rows and columns
    (2500, 130, 5, 3, "matern52", True, 512),   # several targets, ARD, 512-panels
    (1024, 64, 2, 1, "rbf", False, 0),          # a single panel: no trailing update at all
    (5000, 1, 3, 1, "rbf", False, 256),         # one query point
    (4224, 8192, 3, 2, "rbf", True, 0),         # as many query points as one predict batch holds
    (1000, 3000, 3, 1, "matern52", False, 256), # round 4: more query rows than matrix rows (the bordered part is the larger one)
    (700, 300, 2, 64, "rbf", False, 128),       # round 4: all 64 right-hand-side rows in use beside the query rows
])
def test_fit_predict_matches_the_oracle_and_the_two_calls(N, M, d, k, kernel, ard, block):
    X, y, Xs = synthetic_problem(N, d, M, seed=N + M)
    if k > 1:
        rng = np.random.default_rng(5)
        y = np.stack([y] + [np.sin((c + 2) * X[:, 0]) + 0.1 * rng.standard_normal(N) for c in range(k - 1)], axis=1)
    ls = tuple(0.2 + 0.05 * i for i in range(d)) if ard else 0.25
    sf2, sn2 = 1.5, 1e-2
    ref = OracleGP(kernel, ls, sf2, sn2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP(kernel, ls, sf2, sn2, jitter=0.0, block=block) as gp:
        m2, v2 = gp.fit(X, y).predict(Xs)
        a2, ld2 = gp.alpha_.copy(), gp.log_det_
        mean, var = gp.fit_predict(X, y, Xs)
        assert gp.info_ == 0
        assert rel(mean, mr, 1e-6) <= 1e-6 and rel(var, vr, 1e-6 * sf2) <= 1e-6          # north_star's criterion
        # the two forms of the same library: elementwise 1e-9 (means that cross zero among 64 targets: 1e-10 of the largest)
        assert rel(mean, m2, 1e-6) <= 1e-9 or np.max(np.abs(mean - m2)) <= 1e-10 * np.max(np.abs(m2))
        assert rel(var, v2, 1e-6 * sf2) <= 1e-9
        # the handle is fitted exactly as after fit(): factor, log-determinant, alpha, further predicts
        assert gp.log_det_ == ld2
        assert np.array_equal(gp.alpha_, a2)
        m3, v3 = gp.predict(Xs)
        assert np.array_equal(m3, m2) and np.array_equal(v3, v2)
        if N <= 3000:
            lml, grad = gp.lml_gradient()
            assert abs(lml - ref.log_marginal_likelihood()) <= 1e-9 * abs(lml)
            assert np.max(np.abs(grad - ref.lml_gradient())) <= 1e-6 * np.max(np.abs(grad))


def test_fit_predict_fp32_and_device_tensors():
    torch = pytest.importorskip("torch")
    X, y, Xs = synthetic_problem(4096, 3, 512, seed=3)
    dev = torch.device("cuda", 0)
    Xd, yd, Xsd = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        m64, v64 = gp.fit(X, y).predict(Xs)
        md, vd = gp.fit_predict(Xd, yd, Xsd)
        assert md.is_cuda and rel(md.cpu().numpy(), m64, 1e-6) <= 1e-9 and rel(vd.cpu().numpy(), v64, 1.5e-6) <= 1e-9
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, dtype="float32") as gp:
        m2, v2 = gp.fit(X, y).predict(Xs)
        mean, var = gp.fit_predict(X, y, Xs)
        assert mean.dtype == np.float32
        # fp32: the two summation orders differ at fp32 rounding of an ill-conditioned solve (cond ~ 1e5)
        assert np.max(np.abs(mean - m2)) <= 2e-2 * np.max(np.abs(m2)) and np.max(np.abs(var - v2)) <= 2e-3 * 1.5


def test_fit_predict_reports_a_bad_pivot_and_falls_back_where_unsupported():
    X, y, Xs = synthetic_problem(2048, 3, 100, seed=9)
    X[1500] = X[200]                                       # duplicate point, no noise, no jitter: singular
    with GP("rbf", 0.25, 1.0, 0.0, jitter=0.0, max_tries=1) as gp:
        with pytest.raises(np.linalg.LinAlgError):
            gp.fit_predict(X, y, Xs)
        assert gp.info_ > 0
    X, y, Xs = synthetic_problem(3000, 3, 200, seed=10)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for kw in ({"dtype": "mixed"}, {"devices": 2, "oversubscribe": True}):     # two calls under the hood
        with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, **kw) as gp:
            mean, var = gp.fit_predict(X, y, Xs)
            assert rel(np.asarray(mean, np.float64), mr, 1e-6) <= 1e-6


@pytest.mark.parametrize("env", [{"GPX_SPLIT_STRIP": "0"}, {"GPX_FUSED_STRIP": "1"}, {"GPX_SOLVE_TOP": "0"},
                                 {"GPX_REST_SPLIT": "0"}, {"GPX_CHAIN_FLAG": "0"}, {"GPX_DIAG_STEP": "64"}])
def test_fit_predict_under_every_schedule_switch(monkeypatch, env):
    """The query rows are bordered rows in EVERY schedule of the factorisation (in the round-2 and the fused one they
    ride in the chain's panel solves and the generic bordered update): same answer, 12 panels of 1024."""
    X, y, Xs = synthetic_problem(12288, 3, 300, seed=41)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        m0, v0 = gp.fit_predict(X, y, Xs)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        mean, var = gp.fit_predict(X, y, Xs)
        assert gp.info_ == 0
        if "GPX_DIAG_STEP" in env:      # another association inside the 128 x 128 diagonal tiles
            assert rel(mean, m0, 1e-6) <= 1e-9 and rel(var, v0, 1.5e-6) <= 1e-9
        else:
            assert np.array_equal(mean, m0) and np.array_equal(var, v0)


def test_fit_predict_jitter_escalation_many_query_points_and_buffer_reuse():
    X, y, Xs = synthetic_problem(2048, 3, 100, seed=9)
    X[1500] = X[200]                                       # duplicate point, no noise: needs jitter
    y[1500] = y[200]
    with GP("rbf", 0.25, 1.0, 0.0, jitter=0.0, max_tries=12) as gp:
        mean, var = gp.fit_predict(X, y, Xs)
        assert gp.info_ == 0 and gp.jitter_used_ > 0.0
        jit = gp.jitter_used_
        m2, v2 = gp.fit(X, y).predict(Xs)                  # the two calls escalate the same way
        assert gp.jitter_used_ == jit
        # cond(K) ~ 1 / jitter ~ 1e11: the oracle is no yardstick here; the two forms of the same library are
        assert np.max(np.abs(mean - m2)) <= 1e-6 * np.max(np.abs(m2)) and np.max(np.abs(var - v2)) <= 1e-6
    Xb, yb, Xsb = synthetic_problem(5000, 3, 8200, seed=12)   # more than one predict batch: the first rides, the rest is predicted (round 4)
    Xc, yc, Xsc = synthetic_problem(1500, 3, 90, seed=13)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
        mb, vb = gp.fit_predict(Xb, yb, Xsb)
        m2, v2 = gp.fit(Xb, yb).predict(Xsb)
        assert rel(mb, m2, 1e-6) <= 1e-9 and rel(vb, v2, 1.5e-6) <= 1e-9         # the riding batch: another summation order
        assert np.array_equal(mb[8192:], m2[8192:]) and np.array_equal(vb[8192:], v2[8192:])   # the rest: the ordinary predict
        big = gp.fit_predict(Xb, yb, Xsb[:4096])                                # grows the buffers ...
        small = gp.fit_predict(Xc, yc, Xsc)                                     # ... a smaller problem in them
        refc = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(Xc, yc)
        mr, vr = refc.predict(Xsc)
        assert rel(small[0], mr, 1e-6) <= 1e-6 and rel(small[1], vr, 1.5e-6) <= 1e-6
        m3, v3 = gp.fit(Xc, yc).predict(Xsc)                                    # and a plain fit after it
        assert rel(m3, mr, 1e-6) <= 1e-6 and rel(v3, vr, 1.5e-6) <= 1e-6
        assert rel(big[0], m2[:4096], 1e-6) <= 1e-9


@pytest.mark.parametrize("kw", [{"devices": 3, "oversubscribe": True}, {"dtype": "mixed"},
                                {"device": 0, "world": 1, "rank": 0, "comm": "rccl"}])
def test_c_abi_fit_predict_on_groups_shards_and_mixed_handles(kw):
    """ABI v5: gpx_fit_predict is accepted by EVERY handle (v4: GPX_E_UNSUPPORTED on groups, shards, mixed).  Mixed: it
    runs as gpx_fit + gpx_predict, same bits.  Groups and shards: the query rows ride through the sharded factorisation
    (tests/test_group_gpu.py), equal to the two calls at 1e-9.  Called through the C ABI directly."""
    import ctypes as C
    from gaussianprocesspathmodelling_amd import _abi
    X, y, Xs = synthetic_problem(2000, 3, 150, seed=17)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, **kw) as gp:
        m2, v2 = gp.fit(X, y).predict(Xs)
        mean, var = np.empty(150), np.empty(150)
        ls, info = (C.c_double * 1)(0.25), C.c_int64(-1)
        rc = gp._lib.gpx_fit_predict(gp._h, X.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 2000, 3, 1, ls, 1,
                                     1.5, 1e-2, 0.0, Xs.ctypes.data_as(C.c_void_p), 150, mean.ctypes.data_as(C.c_void_p),
                                     var.ctypes.data_as(C.c_void_p), _abi.MEM_HOST, C.byref(info))
        assert rc == 0 and info.value == 0
        if kw.get("dtype") == "mixed":
            assert np.array_equal(mean, m2) and np.array_equal(var, v2)
        else:
            assert np.max(np.abs(mean - m2)) <= 1e-9 * np.abs(m2).max() and np.max(np.abs(var - v2)) <= 1e-9 * 1.5
