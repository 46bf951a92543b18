"""GPU: results must not depend on the relative timing of the library's streams (main, look-ahead,
side stream of the block inverses, copy stream; per rank in a device group).  gpx_debug_set_delay
puts bounded spin kernels in front of a random third of all launches; every fit / predict /
gradient below must come out BIT-identical to the undisturbed run — a dependency that is only
satisfied by luck shows up as a changed bit."""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP, _abi
from oracle.gp_oracle import synthetic_problem

pytestmark = pytest.mark.gpu


@pytest.fixture
def delay():
    lib = _abi.load()
    yield lambda seed: lib.gpx_debug_set_delay(seed)
    lib.gpx_debug_set_delay(0)


def run(N, M, kw, grad):
    X, y, Xs = synthetic_problem(N, 3, M, seed=N)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, **kw) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        out = [mean, var, gp.alpha_.copy(), np.float64(gp.log_det_), gp.predict(Xs, return_var=False)]
        if grad:
            lml, g = gp.lml_gradient()
            out += [np.float64(lml), g]
    return out


@pytest.mark.parametrize("N,M,kw,grad,small", [
    (5000, 700, {}, True, False),                            # 5 panels: look-ahead, block inverses, copy stream
    (3000, 300, {"block": 256}, True, False),                # 12 narrow panels
    (9000, 9000, {}, False, False),                          # 9 panels of 1024, two batches of query points
    (2600, 200, {"devices": 3, "oversubscribe": True}, False, True),    # device group, distributed solves
    (2600, 200, {"devices": 4, "oversubscribe": True}, True, True),     # device group, replicated factor + sharded gradient
    (9000, 300, {"devices": 2, "oversubscribe": True}, False, False),   # device group, library-chosen 1024-blocks
    (5200, 200, {"devices": 8, "oversubscribe": True}, False, True),    # eight ranks, 21 panels of 256: the split schedule's hand-overs
    (9000, 300, {"device": 0, "world": 1, "rank": 0, "comm": "rccl"}, False, False),   # RCCL calls on two streams
    (9000, 300, {"devices": [0], "transport": "rccl"}, False, False),   # the same through ncclCommInitAll (one-rank group)
])
def test_results_do_not_depend_on_stream_timing(monkeypatch, delay, N, M, kw, grad, small):
    run_delay_case(monkeypatch, delay, N, M, kw, grad, small)


def test_fused_trailing_update_does_not_depend_on_stream_timing(monkeypatch, delay):
    """GPX_FUSED_STRIP=1: the look-ahead stream is released by a device counter the strip's tiles bump
    from inside the running update (no kernel boundary between producer and consumer): 11 panels,
    random delays in front of a third of all launches, five seeds, bit-identical."""
    monkeypatch.setenv("GPX_FUSED_STRIP", "1")
    run_delay_case(monkeypatch, delay, 12288, 300, {}, False, False)


def test_unsplit_strip_does_not_depend_on_stream_timing(monkeypatch, delay):
    """GPX_SPLIT_STRIP=0: the round-2 schedule (the whole strip on the main stream, handed to the look-ahead
    stream by an event) stays selectable for A/B measurements; the default (split) is what every other case runs."""
    monkeypatch.setenv("GPX_SPLIT_STRIP", "0")
    run_delay_case(monkeypatch, delay, 9000, 300, {}, False, False)


def test_one_pass_fit_predict_does_not_depend_on_stream_timing(delay):
    """gpx_fit_predict: the query rows are more work on the main stream between the chain's events (their panel solves,
    their updates; the split panel solve writes one panel buffer from two streams): 9 panels, M = 700 query points,
    random delays in front of a third of all launches, five seeds, bit-identical — and equal to the two calls."""
    X, y, Xs = synthetic_problem(9000, 3, 700, seed=77)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0) as gp:
        m2, v2 = gp.fit(X, y).predict(Xs)
        base = gp.fit_predict(X, y, Xs)
        a0, ld0 = gp.alpha_.copy(), gp.log_det_
        assert np.max(np.abs(base[0] - m2)) <= 1e-9 * np.max(np.abs(m2)) and np.max(np.abs(base[1] - v2)) <= 1e-9 * 1.5
        for seed in (1, 7, 123456789, 2024, 99):
            delay(seed)
            mean, var = gp.fit_predict(X, y, Xs)
            delay(0)
            assert np.array_equal(mean, base[0]) and np.array_equal(var, base[1]), f"seed {seed}"
            assert np.array_equal(gp.alpha_, a0) and gp.log_det_ == ld0, f"seed {seed}"


@pytest.mark.parametrize("ndev,repl", [(3, 0), (4, 1), (8, 0)])
def test_group_one_pass_does_not_depend_on_stream_timing(monkeypatch, delay, ndev, repl):
    """The same on a device group (round 4): every rank's slice of the query points as bordered rows of the SHARDED
    factorisation — their panel solves share the panel buffers, the copy-back stream and the events of the own rows, and the
    last panel now has an exchange of its own.  256-blocks, delays in front of a third of all launches and exchanges, five
    seeds: bit-identical."""
    monkeypatch.setenv("GPX_NB_SHARD", "256")
    monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(4000, 3, 700, seed=78)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
        base = gp.fit_predict(X, y, Xs)
        a0, ld0 = gp.alpha_.copy(), gp.log_det_
        for seed in (1, 7, 123456789, 2024, 99):
            delay(seed)
            mean, var = gp.fit_predict(X, y, Xs)
            delay(0)
            assert np.array_equal(mean, base[0]) and np.array_equal(var, base[1]), f"seed {seed}"
            assert np.array_equal(gp.alpha_, a0) and gp.log_det_ == ld0, f"seed {seed}"


def run_delay_case(monkeypatch, delay, N, M, kw, grad, small):
    if small:
        monkeypatch.setenv("GPX_NB_SHARD", "256")
        monkeypatch.setenv("GPX_NB_PRED", "256")
        monkeypatch.setenv("GPX_SHARD_REPLICATE", "1" if grad else "0")
    base = run(N, M, kw, grad)
    for seed in (1, 7, 123456789, 2024, 99):
        delay(seed)
        got = run(N, M, kw, grad)
        delay(0)
        for a, b in zip(base, got):
            assert np.array_equal(a, b), f"seed {seed}: a result changed under timing perturbation"
