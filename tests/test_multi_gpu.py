"""GPU, NEEDS AT LEAST TWO DEVICES (skipped on the one-GPU development / grading boxes): the
multi-GPU data paths that a one-GPU box cannot reach — LocalComm's cross-device branch
(hipMemcpyPeerAsync + hipEvents between streams of DIFFERENT devices, gpx_shard.inc `pull`) and the
ncclCommInitAll group where one thread per rank issues collectives on both the main and the
look-ahead stream of its communicator.  Same assertions as tests/test_group_gpu.py (oracle parity at
1e-6, gradient, bit-identity under randomised stream delays), on devices=[0, 1] (and up to 4).
Until a box with two GPUs runs this file these paths are UNVERIFIED ON HARDWARE (DESIGN.md §6,
bench.py's `multi_gpu_transport_verified_on_hardware`).  Every test carries a timeout: RCCL with more
than one rank has never executed under this library."""
import ctypes as C

import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP, _abi
from oracle.gp_oracle import OracleGP, synthetic_problem

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def device_count():
    try:
        n = C.c_int(0)
        _abi.load().gpx_device_count(C.byref(n))
        return n.value
    except Exception:
        return 0


needs2 = pytest.mark.skipif(device_count() < 2, reason="needs >= 2 GPUs in this process")
needs4 = pytest.mark.skipif(device_count() < 4, reason="needs >= 4 GPUs in this process")


def check(gp, mean, var, ref, mr, vr, sf2=1.5):
    dm = np.abs(mean - mr) / np.maximum(np.abs(mr), 1e-6)
    dv = np.abs(var - vr) / np.maximum(vr, 1e-6 * sf2)
    assert dm.max() <= 1e-6 and dv.max() <= 1e-6, (dm.max(), dv.max())
    assert np.max(np.abs(gp.alpha_ - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
    assert abs(gp.log_det_ - ref.log_det_) <= 1e-9 * abs(ref.log_det_)


@needs2
@pytest.mark.parametrize("transport", ["local", "rccl"])
@pytest.mark.parametrize("repl", ["0", "1"])
def test_two_devices_fit_predict_match_the_oracle(monkeypatch, transport, repl):
    monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
    monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    X, y, Xs = synthetic_problem(9000, 3, 300, seed=21)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=[0, 1], transport=transport) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        check(gp, mean, var, ref, mr, vr)
        m2, v2 = gp.fit(X, y).predict(Xs)          # refit: buffers, events, communicator reused
        assert np.array_equal(m2, mean) and np.array_equal(v2, var)


@needs2
@pytest.mark.parametrize("transport", ["local", "rccl"])
def test_two_devices_lml_gradient_matches_the_oracle(monkeypatch, transport):
    monkeypatch.setenv("GPX_NB_PRED", "256")
    X, y, _ = synthetic_problem(2500, 3, 1, seed=22)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    g_ref = ref.lml_gradient()
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=[0, 1], transport=transport) as gp:
        lml, g = gp.fit(X, y).lml_gradient()
        assert abs(lml - ref.log_marginal_likelihood()) <= 1e-9 * abs(ref.log_marginal_likelihood())
        assert np.max(np.abs(g - g_ref)) <= 1e-7 * np.max(np.abs(g_ref))


@needs2
@pytest.mark.parametrize("transport", ["local", "rccl"])
def test_two_devices_results_do_not_depend_on_stream_timing(monkeypatch, transport):
    monkeypatch.setenv("GPX_NB_SHARD", "256")
    lib = _abi.load()
    X, y, Xs = synthetic_problem(2600, 3, 200, seed=23)

    def run():
        with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=[0, 1], transport=transport) as gp:
            mean, var = gp.fit(X, y).predict(Xs)
            return [mean, var, gp.alpha_.copy(), np.float64(gp.log_det_)]
    base = run()
    try:
        for seed in (1, 7, 2024):
            lib.gpx_debug_set_delay(seed)
            got = run()
            lib.gpx_debug_set_delay(0)
            for a, b in zip(base, got):
                assert np.array_equal(a, b), f"seed {seed}: a result changed under timing perturbation"
    finally:
        lib.gpx_debug_set_delay(0)


@needs4
@pytest.mark.parametrize("transport", ["local", "rccl"])
def test_four_devices_match_the_single_gpu_path(transport):
    X, y, Xs = synthetic_problem(20000, 3, 500, seed=24)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0) as one:
        m1, v1 = one.fit(X, y).predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=4, transport=transport) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
    assert np.max(np.abs(mean - m1) / np.maximum(np.abs(m1), 1e-6)) <= 1e-8
    assert np.max(np.abs(var - v1) / np.maximum(v1, 1.5e-6)) <= 1e-8


@needs2
@pytest.mark.parametrize("transport", ["local", "rccl"])
@pytest.mark.parametrize("dtype,repl", [("float32", "1"), ("float32", "0"), ("mixed", "1"), ("mixed", "0")])
def test_two_devices_fp32_and_mixed_shards(monkeypatch, transport, dtype, repl):
    """Round 4: the shard in the handle's element type across two REAL devices — typed RCCL reductions (ncclFloat), byte
    broadcasts / all-gathers, the in-process transport's float sum kernel over peer copies; mixed: the fp64 refinement
    replicated on both cards with local (replicated factor) or collective (distributed factor) fp32 solves."""
    monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
    monkeypatch.setenv("GPX_NB_SHARD", "256")
    X, y, Xs = synthetic_problem(4000, 3, 300, seed=25)
    ref = OracleGP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    Xi, yi, Xsi = (v.astype(np.float32) for v in (X, y, Xs)) if dtype == "float32" else (X, y, Xs)
    with GP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, dtype=dtype, devices=[0, 1], transport=transport) as gp:
        mean, var = gp.fit(Xi, yi).predict(Xsi)
        assert gp.info_ == 0
        ev = np.max(np.abs(var - vr)) / 1.5
        if dtype == "mixed":
            em = np.max(np.abs(mean - mr) / np.maximum(np.abs(mr), 1e-6))
            ea = np.max(np.abs(gp.alpha_ - ref.alpha_)) / np.max(np.abs(ref.alpha_))
            assert em <= 1e-6 and ea <= 1e-7 and ev <= 2e-3
        else:
            assert np.max(np.abs(mean - mr)) <= 2e-3 * np.max(np.abs(mr)) and ev <= 2e-3


@needs2
@pytest.mark.parametrize("transport", ["local", "rccl"])
@pytest.mark.parametrize("repl", ["0", "1"])
def test_two_devices_one_pass_and_both_dealings(monkeypatch, transport, repl):
    """Round 4: ``fit_predict`` riding through the sharded factorisation, the snake and the cyclic dealing of the row blocks,
    the owner's chain on its own stream forced on (its default starts at 4 ranks) — on two REAL devices."""
    monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
    monkeypatch.setenv("GPX_NB_SHARD", "512")
    monkeypatch.setenv("GPX_SHARD_TWO_PIPE", "1")
    X, y, Xs = synthetic_problem(9000, 3, 700, seed=22)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    out = {}
    for deal in ("snake", "cyclic"):
        monkeypatch.setenv("GPX_SHARD_DEAL", deal)
        with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=[0, 1], transport=transport) as gp:
            mean, var = gp.fit_predict(X, y, Xs)
            check(gp, mean, var, ref, mr, vr)
            m2, v2 = gp.predict(Xs)
            assert np.max(np.abs(mean - m2)) <= 1e-9 * np.abs(m2).max() and np.max(np.abs(var - v2)) <= 1e-9 * 1.5
            out[deal] = (mean, var)
    assert np.max(np.abs(out["snake"][0] - out["cyclic"][0])) <= 1e-12 * np.abs(mr).max()


@needs4
def test_four_devices_defaults_incl_two_pipelines(monkeypatch):
    """Four real devices: the defaults of round 4 (snake dealing, the owner's chain beside the previous panel's all-gather,
    gathered panel read in place) over RCCL, distributed solves, one pass."""
    monkeypatch.setenv("GPX_SHARD_REPLICATE", "0")
    monkeypatch.setenv("GPX_NB_SHARD", "512")
    X, y, Xs = synthetic_problem(20000, 3, 1000, seed=23)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=[0, 1, 2, 3], transport="rccl") as gp:
        mean, var = gp.fit_predict(X, y, Xs)
        check(gp, mean, var, ref, mr, vr)
        m2, v2 = gp.fit(X, y).predict(Xs)
        assert np.max(np.abs(mean - m2)) <= 1e-9 * np.abs(m2).max()
