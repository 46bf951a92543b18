"""GPU: the single-process multi-GPU surface of SURVEY.md §8b — ``GP(devices=...)`` called from
this ordinary pytest process: no launcher, no torch.distributed.  The development box has ONE
GPU, so the ranks share it (``oversubscribe`` / a repeated ordinal) over the LOCAL transport:
peer copies and hipEvents between the ranks' streams, i.e. for the first time the look-ahead
stream's exchanges really run concurrently with the main stream's trailing update."""
import time

import numpy as np
import pytest

from c4_util import C4, c4_checks, c4_run, single_gpu_c4
from gaussianprocesspathmodelling_amd import GP, GpxError
from oracle.gp_oracle import OracleGP, synthetic_problem

pytestmark = pytest.mark.gpu


def check(mean, var, alpha, logdet, ref, mr, vr, sf2=1.5):
    dm = np.abs(mean - mr) / np.maximum(np.abs(mr), 1e-6)
    dv = np.abs(var - vr) / np.maximum(vr, 1e-6 * sf2)
    assert dm.max() <= 1e-6 and dv.max() <= 1e-6, (dm.max(), dv.max())
    assert np.max(np.abs(alpha - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
    assert abs(logdet - ref.log_det_) <= 1e-9 * abs(ref.log_det_)


@pytest.mark.parametrize("ndev,kernel,nb,N,M,repl", [
    (2, "rbf", 128, 700, 90, 0), (2, "rbf", 128, 700, 90, 1),
    (3, "matern52", 128, 700, 90, 0), (4, "rbf", 512, 3300, 130, 0), (4, "rbf", 512, 3300, 130, 1),
    (8, "matern52", 128, 2000, 77, 0), (8, "rbf", 256, 5000, 300, 1),
    (2, "rbf", 128, 100, 5, 0),            # one block: rank 1 owns no rows at all
    (5, "rbf", 128, 250, 3, 1),            # two blocks on five ranks, fewer query points than ranks
    (2, "rbf", 0, 9000, 200, -1),          # block height / mode chosen by the library
])
def test_devices_surface_matches_oracle(monkeypatch, ndev, kernel, nb, N, M, repl):
    if nb:
        monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    else:
        monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    if repl >= 0:
        monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(N, 3, M, seed=77)
    ls = (0.3, 0.2, 0.25)
    ref = OracleGP(kernel, ls, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
        assert abs(gp.log_marginal_likelihood(y) - ref.log_marginal_likelihood()) <= 1e-9 * abs(ref.log_marginal_likelihood())
        m2 = gp.predict(Xs, return_var=False)
        assert np.max(np.abs(m2 - mean)) <= 1e-9 * max(1.0, np.abs(mean).max())
        m3, v3 = gp.fit(X, y).predict(Xs)          # refit on the same group: buffers and events reused
        assert np.array_equal(m3, mean) and np.array_equal(v3, var)


@pytest.mark.parametrize("ndev,kernel,N,nbp,k", [
    (2, "rbf", 1500, 256, 1), (3, "matern52", 1100, 128, 2), (4, "rbf", 3300, 256, 1),
    (4, "rbf", 300, 256, 1),               # two blocks on four ranks: ranks 2, 3 own no row of L^-T
    (8, "matern52", 5000, 1024, 1),        # default block height; 5 blocks on 8 ranks
])
def test_group_lml_gradient_matches_single_gpu_and_oracle(monkeypatch, ndev, kernel, N, nbp, k):
    """gpx_lml_grad on a device group: row blocks of L^-T dealt over the ranks, one all-gather,
    the K^-1 trace pass split over the ranks — against the single-GPU gradient (same kernels, other
    summation split) and the CPU oracle's analytic gradient."""
    monkeypatch.setenv("GPX_NB_SHARD", "128")
    monkeypatch.setenv("GPX_NB_PRED", str(nbp))
    monkeypatch.setenv("GPX_SHARD_REPLICATE", "1")
    X, y, _ = synthetic_problem(N, 3, 10, seed=31)
    if k == 2:
        y = np.stack([y, np.cos(3.0 * y)], axis=1)
    ls = (0.3, 0.2, 0.25)
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0) as g1:
        lml1, grad1 = g1.fit(X, y).lml_gradient()
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
        lml, grad = gp.fit(X, y).lml_gradient()
        assert abs(lml - lml1) <= 1e-10 * abs(lml1)
        assert np.max(np.abs(grad - grad1)) <= 1e-9 * np.max(np.abs(grad1))
        lml2, grad2 = gp.lml_gradient()            # again on the same fit: buffers reused
        assert lml2 == lml and np.array_equal(grad2, grad)
        tm = gp.timings_
        assert tm["grad_total"] > 0 and tm["grad_trace"] > 0
    if k == 1:
        ref = OracleGP(kernel, ls, 1.5, 1e-2, jitter=0.0).fit(X, y)
        lml_o, grad_o = ref.log_marginal_likelihood(), ref.lml_gradient()
        assert abs(lml - lml_o) <= 1e-9 * abs(lml_o)
        assert np.max(np.abs(grad - grad_o)) <= 1e-7 * np.max(np.abs(grad_o))


@pytest.mark.parametrize("ndev,kernel,N,nbs,k,batch", [
    (3, "matern52", 2000, 128, 1, 0), (4, "rbf", 3300, 256, 2, 0), (8, "rbf", 5000, 256, 1, 1024),
    (2, "rbf", 200, 128, 1, 0),            # two blocks on two ranks
    (5, "matern52", 300, 128, 1, 0),       # three blocks on five ranks: ranks 3, 4 own no column of L^-T
])
def test_group_lml_gradient_with_a_distributed_only_factor(monkeypatch, ndev, kernel, N, nbs, k, batch):
    """GPX_SHARD_REPLICATE=0 — the C4 mode, where no rank holds L: gpx_lml_grad builds L^-T with the
    distributed forward substitution of the variance path (per-block broadcasts, row batches that skip
    their structural zeros), keeps on each rank the COLUMNS that belong to its own row blocks, contracts
    the K^-1 trace over those columns and all-reduces ntheta numbers (shard_lml_grad_dist).  Against
    the CPU oracle's analytic gradient (k = 1) and the single-GPU gradient."""
    monkeypatch.setenv("GPX_NB_SHARD", str(nbs))
    monkeypatch.setenv("GPX_SHARD_REPLICATE", "0")
    if batch:
        monkeypatch.setenv("GPX_PRED_BATCH", str(batch))     # several row batches of the identity
    X, y, _ = synthetic_problem(N, 3, 10, seed=41)
    if k == 2:
        y = np.stack([y, np.cos(3.0 * y)], axis=1)
    ls = (0.3, 0.2, 0.25)
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0) as g1:
        lml1, grad1 = g1.fit(X, y).lml_gradient()
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
        lml, grad = gp.fit(X, y).lml_gradient()
        assert abs(lml - lml1) <= 1e-10 * abs(lml1)
        assert np.max(np.abs(grad - grad1)) <= 1e-9 * np.max(np.abs(grad1))
        lml2, grad2 = gp.lml_gradient()
        assert lml2 == lml and np.array_equal(grad2, grad)
    if k == 1:
        ref = OracleGP(kernel, ls, 1.5, 1e-2, jitter=0.0).fit(X, y)
        assert abs(lml - ref.log_marginal_likelihood()) <= 1e-9 * abs(ref.log_marginal_likelihood())
        g_o = ref.lml_gradient()
        assert np.max(np.abs(grad - g_o)) <= 1e-7 * np.max(np.abs(g_o))


def test_distributed_only_gradient_at_scale(monkeypatch):
    """N = 32768 on 4 ranks with library-chosen 512-blocks (16 per rank), factor only held distributed:
    four row batches of the identity, 64 block steps with look-ahead and broadcasts — against the
    single-GPU gradient (same kernels, another contraction split)."""
    monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    monkeypatch.setenv("GPX_SHARD_REPLICATE", "0")
    X, y, _ = synthetic_problem(32768, 3, 10, seed=12345)
    with GP("matern52", 0.25, 1.5, 1e-2, jitter=0.0) as g1:
        lml1, grad1 = g1.fit(X, y).lml_gradient()
    with GP("matern52", 0.25, 1.5, 1e-2, jitter=0.0, devices=4, oversubscribe=True) as gp:
        lml, grad = gp.fit(X, y).lml_gradient()
        tm = gp.timings_
        print(f"distributed-only gradient N=32768 x 4 ranks: L^-T columns {tm['grad_trtri']:.0f} ms, trace {tm['grad_trace']:.0f} ms, total {tm['grad_total']:.0f} ms")
    assert abs(lml - lml1) <= 1e-10 * abs(lml1)
    assert np.max(np.abs(grad - grad1)) <= 1e-8 * np.max(np.abs(grad1))


def test_group_optimize_uses_the_analytic_gradient_in_both_solve_modes(monkeypatch):
    monkeypatch.setenv("GPX_NB_SHARD", "128")
    X, y, _ = synthetic_problem(500, 2, 10, seed=3)
    with GP("rbf", 0.5, 1.0, 0.1) as g1:
        r1 = g1.optimize(X, y, maxiter=25)
        for repl in ("0", "1"):
            monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
            with GP("rbf", 0.5, 1.0, 0.1, devices=2, oversubscribe=True) as gp:
                r2 = gp.optimize(X, y, maxiter=25)
                assert r2.nfev <= 2 * r1.nfev + 6          # analytic steps, not 2 p fits per gradient
                assert abs(r2.fun - r1.fun) <= 1e-6 * abs(r1.fun)
                assert np.allclose(gp.lengthscale, g1.lengthscale, rtol=1e-3) and np.isclose(gp.noise, g1.noise, rtol=1e-3)


@pytest.mark.parametrize("repl", [0, 1])
def test_group_predict_in_batches(monkeypatch, repl):
    """Sharded predict with GPX_PRED_BATCH = 128: several batches, each with its own look-ahead
    chain of per-block broadcasts in the distributed mode (repl 0)."""
    monkeypatch.setenv("GPX_NB_SHARD", "128")
    monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(1200, 3, 700, seed=5)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=[0, 0, 0]) as gp:
        m1, v1 = gp.fit(X, y).predict(Xs)
        monkeypatch.setenv("GPX_PRED_BATCH", "128")
        m2, v2 = gp.predict(Xs)
        check(m2, v2, gp.alpha_, gp.log_det_, ref, mr, vr)
        if repl == 0:      # same batches on every rank, same arithmetic per row
            assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
        else:              # the split of the query points over the ranks differs from one pass: rows still independent
            assert np.max(np.abs(m1 - m2)) <= 1e-12 and np.max(np.abs(v1 - v2)) <= 1e-12


def test_group_slab_panel_variant(monkeypatch):
    """GPX_SHARD_DENSE_PANEL=0: the sharded panel solve by the slab kernel (64-block inverses) instead
    of the dense product with the broadcast block inverse — same results."""
    monkeypatch.setenv("GPX_NB_SHARD", "256")
    X, y, Xs = synthetic_problem(2500, 3, 100, seed=12)
    ref = OracleGP("matern52", 0.3, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    out = []
    for dense in ("1", "0"):
        monkeypatch.setenv("GPX_SHARD_DENSE_PANEL", dense)
        for repl in ("0", "1"):
            monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
            with GP("matern52", 0.3, 1.5, 1e-2, jitter=0.0, devices=[0, 0, 0]) as gp:
                mean, var = gp.fit(X, y).predict(Xs)
                check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
                out.append(mean)
    assert np.max(np.abs(out[0] - out[2])) <= 1e-10


def test_devices_list_and_single_entry():
    X, y, Xs = synthetic_problem(900, 3, 40, seed=3)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for devs in ([0], 1, [0, 0, 0]):
        with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=devs, transport=None if devs != [0, 0, 0] else "local") as gp:
            mean, var = gp.fit(X, y).predict(Xs)
            check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
    with pytest.raises(GpxError, match="distinct"):
        GP(devices=[0, 0], transport="rccl")
    with pytest.raises(GpxError, match="out of range"):
        GP(devices=[0, 63])


@pytest.mark.parametrize("transport", ["rccl", "rccl-grouped-initrank", "local"])
def test_one_rank_group_runs_the_group_code_path(monkeypatch, transport):
    """devices=[0] with an EXPLICIT transport is a one-rank GROUP, not the plain handle: gpx_create ->
    create_group -> ncclCommInitAll (rccl) / LocalComm (local) -> run_group -> the sharded schedule,
    whose look-ahead stream issues collectives beside the main stream's update (N = 9000 with the
    library's 1024-blocks: 9 panels).  The only way a one-GPU box executes ncclCommInitAll and the
    group's RCCL plumbing at all (more than one rank needs distinct devices: tests/test_multi_gpu.py)."""
    monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    if transport == "rccl-grouped-initrank":      # the other way to create the group's communicators
        monkeypatch.setenv("GPX_GROUP_INITALL", "0")
        transport = "rccl"
    X, y, Xs = synthetic_problem(9000, 3, 200, seed=5)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for repl in ("0", "1"):
        monkeypatch.setenv("GPX_SHARD_REPLICATE", repl)
        with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=[0], transport=transport) as gp:
            assert gp._is_group
            mean, var = gp.fit(X, y).predict(Xs)
            check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
    # the gradient through the group entry point (replicated factor), against the oracle's
    X, y, _ = synthetic_problem(2500, 3, 1, seed=6)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    g_ref = ref.lml_gradient()
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=[0], transport=transport) as gp:
        lml, g = gp.fit(X, y).lml_gradient()
        assert abs(lml - ref.log_marginal_likelihood()) <= 1e-9 * abs(ref.log_marginal_likelihood())
        assert np.max(np.abs(g - g_ref)) <= 1e-7 * np.max(np.abs(g_ref))


def test_devices_group_device_tensor_inputs_and_multi_output():
    torch = pytest.importorskip("torch")
    X, y, Xs = synthetic_problem(1100, 3, 64, seed=9)
    Y = np.stack([y, np.cos(y)], axis=1)
    ref = OracleGP("matern52", 0.3, 1.2, 2e-2, jitter=0.0).fit(X, Y)
    mr, vr = ref.predict(Xs)
    dev = torch.device("cuda:0")
    with GP("matern52", 0.3, 1.2, 2e-2, jitter=0.0, devices=[0, 0, 0]) as gp:
        gp.fit(torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev))
        mean, var = gp.predict(torch.from_numpy(Xs).to(dev))
        assert mean.is_cuda and mean.shape == (64, 2)
        for c in range(2):
            check(mean[:, c].cpu().numpy(), var.cpu().numpy(), gp.alpha_[:, c], gp.log_det_,
                  type("R", (), {"alpha_": ref.alpha_[:, c], "log_det_": ref.log_det_}), mr[:, c], vr, 1.2)


def test_group_not_positive_definite_reports_pivot_on_every_rank():
    X = np.zeros((300, 2))
    X[:, 0] = np.repeat(np.linspace(0, 1, 150), 2)       # duplicated points, zero noise
    y = np.sin(X[:, 0])
    with GP("rbf", 0.5, 1.0, noise=0.0, jitter=0.0, max_tries=1, devices=[0, 0]) as gp:
        with pytest.raises(np.linalg.LinAlgError):
            gp.fit(X, y)
        assert gp.info_ > 0


def test_c4_shape_on_one_gpu_world8():
    """BASELINE.json configs[3] code path with the P = 8 block-cyclic maps: Matern-5/2, d = 3,
    distributed solves, library-chosen block height, N = 32768 on EIGHT ranks — threads of this
    process sharing the one GPU (8 processes would exceed the box's process limit)."""
    N, M = 32768, 1024
    import os
    os.environ["GPX_SHARD_REPLICATE"] = "0"
    os.environ.pop("GPX_NB_SHARD", None)
    try:
        with GP(jitter=0.0, devices=[0] * 8, **C4) as gp:
            r = c4_run(gp, N, M)
            tm = gp.timings_
    finally:
        os.environ.pop("GPX_SHARD_REPLICATE", None)
    resid = c4_checks([r], N, M, single_gpu_c4(N, M))
    print(f"C4 shape, 8 ranks in one process, N={N}: residual {resid:.2e}, fit {tm['fit_total']:.0f} ms "
          f"(chol {tm['chol']:.0f} ms, comm {tm['comm']:.0f} ms), predict {tm['predict_total']:.0f} ms")


def test_c4_shape_group_world4_n65536_overlapped():
    """The world-4, N = 65536 case of test_shard_gpu.py again, but through the in-process
    transport: exchanges are stream-ordered (no host synchronisation per collective), so panel
    p+1's broadcast / all-gather on the look-ahead stream overlap update p on the main stream."""
    N, M = 65536, 1024
    import os
    os.environ["GPX_SHARD_REPLICATE"] = "0"
    os.environ.pop("GPX_NB_SHARD", None)
    try:
        with GP(jitter=0.0, devices=[0] * 4, **C4) as gp:
            r = c4_run(gp, N, M)
            tm = gp.timings_
    finally:
        os.environ.pop("GPX_SHARD_REPLICATE", None)
    resid = c4_checks([r], N, M, single_gpu_c4(N, M))
    print(f"C4 shape, 4 ranks in one process, N={N}: residual {resid:.2e}, fit {tm['fit_total']:.0f} ms "
          f"(chol {tm['chol']:.0f} ms, comm {tm['comm']:.0f} ms), predict {tm['predict_total']:.0f} ms")


def test_c4_real_shape_world8_nb1024_n131072():
    """BASELINE.json configs[3] at the per-rank SHAPE it really has: P = 8 ranks x library-chosen
    nb = 1024, Matern-5/2, distributed solves, 16 panels per rank — N = 131072 (137 GB, what one card holds; C4 itself is
    N = 262144 on eight cards with 32 panels per rank), the eight ranks threads of this process sharing the one GPU.
    Checks of SURVEY.md §8(d) for C4 incl. agreement with the unsharded path at 1e-10 (run after the group is gone:
    137 GB each).  No hardware with eight cards has run this: the transport here is same-device copies."""
    N, M = 131072, 1024
    import os
    os.environ["GPX_SHARD_REPLICATE"] = "0"
    os.environ.pop("GPX_NB_SHARD", None)
    try:
        with GP(jitter=0.0, devices=[0] * 8, **C4) as gp:
            r = c4_run(gp, N, M)
            tm = gp.timings_
            # the same step as ONE pass (round 4: 128 query rows per rank ride through the sharded factorisation)
            from oracle.gp_oracle import synthetic_problem as sp
            X, y, Xs = sp(N, 3, M, seed=12345)
            t0 = time.perf_counter()
            m1p, v1p = gp.fit_predict(X, y, Xs)
            t_one = (time.perf_counter() - t0) * 1e3
            assert np.array_equal(gp.alpha_, r["alpha"]) and gp.log_det_ == r["logdet"]
            assert np.max(np.abs(m1p - r["mean"])) <= 1e-9 * np.abs(r["mean"]).max()
            assert np.max(np.abs(v1p - r["var"])) <= 1e-9 * 1.5
    finally:
        os.environ.pop("GPX_SHARD_REPLICATE", None)
    resid = c4_checks([r], N, M, single_gpu_c4(N, M))
    print(f"C4 real shape (P = 8 x nb = 1024), N={N} on one card: residual {resid:.2e}, fit {tm['fit_total']:.0f} ms "
          f"(chol {tm['chol']:.0f} ms, comm {tm['comm']:.0f} ms), predict {tm['predict_total']:.0f} ms; one pass incl. host copies {t_one:.0f} ms")


@pytest.mark.parametrize("ndev,N,nb,repl", [(3, 4000, 256, 0), (4, 4000, 256, 1), (2, 9000, 0, 1), (8, 6000, 128, 0)])
def test_group_split_schedule_is_bit_identical_to_the_round3_schedule(monkeypatch, ndev, N, nb, repl):
    """The sharded factorisation's round-4 schedule (only the next diagonal block on the chain, the rows below it and
    the REST on the main stream, early hand-over of the REST's first column block) against the round-3 one
    (GPX_SPLIT_STRIP=0: whole strip, then the chain) and against a REST that is never / always split
    (GPX_REST_SPLIT): the same arithmetic per element in the same order — every output bit-identical."""
    if nb:
        monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    else:
        monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(N, 3, 300, seed=N + ndev)

    def run():
        with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
            mean, var = gp.fit(X, y).predict(Xs)
            return mean, var, gp.alpha_.copy(), gp.log_det_

    base = run()
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    check(base[0], base[1], base[2], base[3], ref, mr, vr)
    # (round 4 end: also the owner's diagonal chain on a stream of its own beside the previous panel's all-gather — default
    #  from 4 ranks on — forced on and off, and the replicated factor's panel copy on the main stream)
    for env in ({"GPX_SPLIT_STRIP": "0"}, {"GPX_REST_SPLIT": "0"}, {"GPX_REST_SPLIT": "1"}, {"GPX_REST_SPLIT": "1000"},
                {"GPX_SHARD_TWO_PIPE": "0"}, {"GPX_SHARD_TWO_PIPE": "1"}, {"GPX_REPL_COPY_SIDE": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        got = run()
        for k_ in env:
            monkeypatch.delenv(k_)
        for a, b in zip(base, got):
            assert np.array_equal(a, b), env


def test_group_peer_copy_call_between_ranks_sharing_the_device(tmp_path):
    """GPX_LOCAL_FORCE_PEER=1 (test hook): the in-process transport pulls with hipMemcpyPeerAsync even between ranks that
    share a device — the call a multi-GPU group makes between DISTINCT devices and a one-GPU box otherwise never executes
    (with equal ordinals it is an ordinary copy: same results, bit for bit).  Child process: the switch is read once."""
    import subprocess, sys, os
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import synthetic_problem
X, y, Xs = synthetic_problem(3000, 3, 200, seed=3)
out = []
for repl in ("0", "1"):
    import os
    os.environ["GPX_SHARD_REPLICATE"] = repl
    os.environ["GPX_NB_SHARD"] = "256"
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=3, oversubscribe=True) as gp:
        m, v = gp.fit(X, y).predict(Xs)
        out += [m, v, gp.alpha_.copy()]
np.save(sys.argv[1], np.concatenate([o.ravel() for o in out]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for force in ("0", "1"):
        out = str(tmp_path / f"peer{force}.npy")
        env = dict(os.environ, GPX_LOCAL_FORCE_PEER=force)
        r = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("ndev,kernel,nb,N,M,repl,k,dtype", [
    (2, "rbf", 128, 700, 90, 0, 1, "float64"), (2, "rbf", 128, 700, 90, 1, 1, "float64"),
    (3, "matern52", 128, 1500, 500, 0, 2, "float64"), (4, "rbf", 512, 3300, 130, 1, 1, "float64"),
    (8, "matern52", 128, 2000, 77, 0, 1, "float64"), (8, "rbf", 256, 5000, 1300, 1, 3, "float64"),
    (2, "rbf", 128, 100, 5, 0, 1, "float64"),       # one block: rank 1 owns no rows, only its query slice
    (5, "rbf", 128, 250, 3, 1, 1, "float64"),       # fewer query points than ranks: four ranks ride nothing
    (2, "rbf", 0, 9000, 4096, -1, 1, "float64"),    # block height / mode chosen by the library
    (4, "matern52", 256, 4000, 600, 0, 1, "float32"), (3, "rbf", 256, 4000, 600, 1, 1, "float32"),
])
def test_group_fit_predict_rides_the_query_rows_through_the_sharded_factorisation(monkeypatch, ndev, kernel, nb, N, M, repl, k, dtype):
    """``GP.fit_predict`` on a device group (round 4): every rank's slice of the query points rides through ITS part of
    the sharded factorisation as bordered rows (shard_fit with query points; gpx_fit_predict on the member handles).
    Against the oracle at the elementwise bar of the two calls, against the two calls on the same group at 1e-9 (fp64; the
    sums run in another order), the model left behind equal to a plain fit's, and a refit in the grown buffers."""
    if nb:
        monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    else:
        monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    if repl >= 0:
        monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, Y, Xs = synthetic_problem(N, 3, M, seed=N + M)
    if k > 1:
        Y = np.stack([Y * (j + 1) + 0.1 * j * np.cos(5.0 * X[:, 0]) for j in range(k)], axis=1)
    ls = (0.3, 0.2, 0.25)
    ref = OracleGP(kernel, ls, 1.5, 1e-2, jitter=0.0).fit(X, Y)
    mr, vr = ref.predict(Xs)
    f32 = dtype == "float32"
    Xc, Yc, Xsc = (a.astype(np.float32) for a in (X, Y, Xs)) if f32 else (X, Y, Xs)
    with GP(kernel, ls, 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True, dtype=dtype) as gp:
        mean, var = gp.fit_predict(Xc, Yc, Xsc)
        a1, l1 = gp.alpha_.copy(), gp.log_det_
        if f32:
            assert np.max(np.abs(mean - mr)) <= 2e-2 * np.abs(mr).max() and np.max(np.abs(var - vr)) <= 5e-2 * 1.5
        elif k > 1:
            assert np.max(np.abs(mean - mr)) <= 1e-8 * np.abs(mr).max() and np.max(np.abs(var - vr) / np.maximum(vr, 1.5e-6)) <= 1e-6
        else:
            check(mean, var, a1, l1, ref, mr, vr)
        m2, v2 = gp.fit(Xc, Yc).predict(Xsc)
        assert np.array_equal(gp.alpha_, a1) and gp.log_det_ == l1       # the factorisation itself: not a bit changes
        bar = 2e-3 if f32 else 1e-9
        assert np.max(np.abs(mean - m2)) <= bar * max(1.0, np.abs(m2).max())
        assert np.max(np.abs(var - v2)) <= bar * 1.5
        m3, v3 = gp.fit_predict(Xc, Yc, Xsc)                             # again, buffers and events reused
        assert np.array_equal(m3, mean) and np.array_equal(v3, var)
        m4, v4 = gp.predict(Xsc)                                         # the model it leaves behind predicts as a fit's
        assert np.array_equal(m4, m2) and np.array_equal(v4, v2)


def test_group_fit_predict_reports_a_bad_pivot_and_honours_the_switches(monkeypatch):
    X, y, Xs = synthetic_problem(1500, 3, 200, seed=5)
    Xd = np.vstack([X[:700], X[:700], X[700:800]])                      # duplicate rows, no noise: not positive definite
    with GP("rbf", 0.3, 1.5, 0.0, jitter=0.0, devices=3, oversubscribe=True, max_tries=1) as gp:
        with pytest.raises(np.linalg.LinAlgError):
            gp.fit_predict(Xd, y, Xs)
    monkeypatch.setenv("GPX_NB_SHARD", "128")
    with GP("rbf", 0.3, 1.5, 1e-2, jitter=0.0, devices=3, oversubscribe=True) as gp:
        base = gp.fit_predict(X, y, Xs)
        for env in ({"GPX_SHARD_FUSED": "0"}, {"GPX_SPLIT_STRIP": "0"}):  # both: the two calls inside the library
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            two = gp.fit_predict(X, y, Xs)
            ref2 = gp.predict(Xs)
            for k_ in env:
                monkeypatch.delenv(k_)
            assert np.array_equal(two[0], ref2[0]) and np.array_equal(two[1], ref2[1])
            assert np.max(np.abs(two[0] - base[0])) <= 1e-9 and np.max(np.abs(two[1] - base[1])) <= 1e-9
        for env in ({"GPX_REST_SPLIT": "0"}, {"GPX_REST_SPLIT": "1000"}, {"GPX_SHARD_DENSE_PANEL": "0"}):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            got = gp.fit_predict(X, y, Xs)
            for k_ in env:
                monkeypatch.delenv(k_)
            if "GPX_REST_SPLIT" in env:                                   # the same arithmetic in the same order
                assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), env
            else:
                assert np.max(np.abs(got[0] - base[0])) <= 1e-8 and np.max(np.abs(got[1] - base[1])) <= 1e-8


@pytest.mark.parametrize("ndev,N,nb,repl", [(3, 4000, 256, 0), (4, 4000, 256, 1), (8, 6000, 128, 0), (5, 3000, 128, 1), (2, 700, 128, 0)])
def test_group_snake_and_cyclic_dealing_agree(monkeypatch, ndev, N, nb, repl):
    """Round 4: the row blocks are dealt over the ranks as a snake (rounds of 2 P blocks: 0 .. P-1, P-1 .. 0 — balanced row
    work) instead of cyclically (GPX_SHARD_DEAL=cyclic keeps rounds 1-3's dealing).  Who owns a row block changes no
    operand and no order of any element's arithmetic in the factorisation: replicated factor -> every output bit-identical
    (the log-determinant's all-reduce adds the ranks' partial sums, so it may differ in the last bits); distributed solves
    reduce partial products over the ranks -> 1e-12.  Both against the oracle; fit_predict and the gradient on both."""
    monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(N, 3, 333, seed=N + ndev)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    out = {}
    for deal in ("snake", "cyclic"):
        monkeypatch.setenv("GPX_SHARD_DEAL", deal)
        with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=ndev, oversubscribe=True) as gp:
            mean, var = gp.fit(X, y).predict(Xs)
            check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
            m1, v1 = gp.fit_predict(X, y, Xs)
            lml, grad = gp.lml_gradient()
            out[deal] = (mean, var, gp.alpha_.copy(), gp.log_det_, m1, v1, lml, grad)
    a, b = out["snake"], out["cyclic"]
    if repl:
        for i in (0, 1, 2, 4, 5):
            assert np.array_equal(a[i], b[i]), i
    else:
        for i in (0, 1, 2, 4, 5):
            assert np.max(np.abs(a[i] - b[i])) <= 1e-12 * max(1.0, np.max(np.abs(b[i]))), i
    assert abs(a[3] - b[3]) <= 1e-13 * abs(b[3]) and abs(a[6] - b[6]) <= 1e-12 * abs(b[6])
    assert np.max(np.abs(a[7] - b[7])) <= 1e-9 * np.max(np.abs(b[7]))
    go = ref.lml_gradient()
    assert np.max(np.abs(a[7] - go)) <= 1e-8 * np.max(np.abs(go))
