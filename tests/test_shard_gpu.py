"""GPU: the sharded schedule with the real HIP kernels.
  - 1-rank RCCL communicator: every RCCL entry point the path uses, on one GPU;
  - 2 and 3 ranks sharing the GPU through the host transport (gloo): the block-cyclic
    exchange schedule itself (RCCL refuses two ranks on one device)."""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, synthetic_problem
from c4_util import c4_checks, single_gpu_c4
from shard_util import run_ranks

pytestmark = pytest.mark.gpu


def check(mean, var, alpha, logdet, ref, mr, vr, sf2=1.5):
    dm = np.abs(mean - mr) / np.maximum(np.abs(mr), 1e-6)
    dv = np.abs(var - vr) / np.maximum(vr, 1e-6 * sf2)
    assert dm.max() <= 1e-6 and dv.max() <= 1e-6, (dm.max(), dv.max())
    assert np.max(np.abs(alpha - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
    assert abs(logdet - ref.log_det_) <= 1e-9 * abs(ref.log_det_)


@pytest.mark.parametrize("N,M,nb,repl", [(1500, 130, 256, 0), (640, 64, 128, 0), (1500, 130, 256, 1),
                                          (900, 1, 128, 1), (100, 5, 128, 0), (100, 5, 128, 1),  # last two: ONE block
                                          # library-chosen 1024-wide blocks at a size where the two streams (look-ahead
                                          # panel + RCCL calls on st2, trailing update on st) really overlap on the card
                                          (9000, 300, 0, 0), (9000, 300, 0, 1)])
def test_sharded_schedule_single_rank_rccl(N, M, nb, repl, monkeypatch):
    """repl = 0: distributed solves (the C4-sized path); 1: whole factor kept on every rank."""
    if nb:
        monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    else:
        monkeypatch.delenv("GPX_NB_SHARD", raising=False)
    monkeypatch.setenv("GPX_SHARD_REPLICATE", str(repl))
    X, y, Xs = synthetic_problem(N, 3, M, seed=N)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, world=1, rank=0, comm="rccl") as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
        assert np.max(np.abs(gp.predict(Xs, return_var=False) - mean)) <= 1e-9 * max(1.0, np.abs(mean).max())


def test_single_rank_rccl_lml_gradient(monkeypatch):
    """The sharded gradient path with P = 1 over a real RCCL communicator (broadcast-free, the
    all-reduce skipped): the plain single-GPU handle's numbers (the factor comes from the sharded
    schedule, so agreement is to rounding, not bitwise)."""
    monkeypatch.setenv("GPX_SHARD_REPLICATE", "1")
    X, y, _ = synthetic_problem(2000, 3, 10, seed=4)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as g1:
        lml1, grad1 = g1.fit(X, y).lml_gradient()
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, world=1, rank=0, comm="rccl") as gp:
        lml, grad = gp.fit(X, y).lml_gradient()
    assert abs(lml - lml1) <= 1e-10 * abs(lml1) and np.max(np.abs(grad - grad1)) <= 1e-9 * np.max(np.abs(grad1))


@pytest.mark.parametrize("world,kernel,nb,N,repl", [
    (2, "rbf", 128, 700, 0), (3, "matern52", 128, 700, 0), (2, "rbf", 256, 700, 0), (4, "rbf", 512, 3300, 0),
    (2, "rbf", 128, 700, 1), (3, "matern52", 128, 700, 1), (4, "rbf", 512, 3300, 1),
    (2, "rbf", 128, 100, 0), (2, "rbf", 128, 100, 1),      # one block: rank 1 owns no rows at all
    (3, "rbf", 128, 250, 0), (3, "rbf", 128, 250, 1),      # two blocks on three ranks
    (2, "rbf", 0, 9000, -1)])   # nb 0 / repl -1 = chosen by the library (replicated at this size)
def test_sharded_ranks_share_one_gpu(tmp_path, world, kernel, nb, N, repl):
    """Several ranks of the sharded schedule on ONE GPU through the host transport, in both
    solve modes: distributed (broadcast / reduce per panel) and replicated factor (query
    points split over the ranks; M = 90 leaves the last ranks an empty slice)."""
    env = {"SHARD_KERNEL": kernel, "SHARD_NB": str(nb), "SHARD_N": str(N)}
    if repl >= 0:
        env["GPX_SHARD_REPLICATE"] = str(repl)
    res = run_ranks("gpu", world, tmp_path, env, timeout=600)
    X, y, Xs = synthetic_problem(N, 3, 90, seed=77)
    ref = OracleGP(kernel, (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for r in res:
        assert int(r["info"]) == 0
        check(r["mean"], r["var"], r["alpha"], float(r["logdet"]), ref, mr, vr)
    # every rank returns the same replicated result
    assert np.array_equal(res[0]["mean"], res[-1]["mean"]) and np.array_equal(res[0]["var"], res[-1]["var"])


def test_ranks_that_disagree_on_the_dealing_fail_together(tmp_path):
    """Round 4: block height and dealing come from each rank's ENVIRONMENT; ranks that disagree would exchange differently
    shaped pieces and return wrong numbers silently.  One min-all-reduce before anything is exchanged: every rank raises."""
    res = run_ranks("gpu", 2, tmp_path, {"SHARD_DISAGREE": "1", "SHARD_N": "700", "SHARD_NB": "128"}, timeout=300)
    for r in res:
        assert "disagree" in str(r["error"]), r["error"]


@pytest.mark.parametrize("world,kernel,nb,N,M,repl,dtype", [
    (2, "rbf", 128, 700, 90, 0, "float64"), (3, "matern52", 128, 700, 300, 1, "float64"), (4, "rbf", 512, 3300, 1000, 0, "float64"),
    (2, "rbf", 128, 100, 5, 0, "float64"), (3, "rbf", 128, 250, 2, 1, "float64"), (2, "rbf", 0, 9000, 600, -1, "float64"),
    (3, "rbf", 256, 2000, 400, 0, "float32")])
def test_one_pass_fit_predict_over_the_host_transport(tmp_path, world, kernel, nb, N, M, repl, dtype):
    """Round 4: ``GP.fit_predict`` on a process-per-rank shard — every rank's slice of the query points rides through ITS
    part of the factorisation as bordered rows (collective decisions depend on M alone: ranks with an empty slice make the
    same broadcasts).  Oracle bar of the two calls; equal to the two calls on the factor it leaves behind at 1e-9 (fp64)."""
    env = {"SHARD_KERNEL": kernel, "SHARD_NB": str(nb), "SHARD_N": str(N), "SHARD_M": str(M), "SHARD_ONE_PASS": "1",
           "SHARD_DTYPE": dtype}
    if repl >= 0:
        env["GPX_SHARD_REPLICATE"] = str(repl)
    res = run_ranks("gpu", world, tmp_path, env, timeout=600)
    X, y, Xs = synthetic_problem(N, 3, M, seed=77)
    ref = OracleGP(kernel, (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for r in res:
        assert int(r["info"]) == 0
        if dtype == "float64":
            check(r["mean"], r["var"], r["alpha"], float(r["logdet"]), ref, mr, vr)
            assert np.max(np.abs(r["mean"] - r["mean_two_calls"])) <= 1e-9 * max(1.0, np.abs(mr).max())
            assert np.max(np.abs(r["var"] - r["var_two_calls"])) <= 1e-9 * 1.5
        else:
            assert np.max(np.abs(r["mean"] - mr)) <= 2e-3 * np.abs(mr).max() and np.max(np.abs(r["var"] - vr)) <= 2e-3 * 1.5
    assert np.array_equal(res[0]["mean"], res[-1]["mean"]) and np.array_equal(res[0]["var"], res[-1]["var"])


@pytest.mark.parametrize("world,dtype,nb,N,repl", [(2, "float32", 128, 700, 0), (3, "float32", 256, 2000, 1),
                                                   (2, "mixed", 128, 1500, 0), (3, "mixed", 256, 2000, 1)])
def test_fp32_and_mixed_shards_over_the_host_transport(tmp_path, world, dtype, nb, N, repl):
    """Round 4: the shard in the handle's element type, one PROCESS per rank over the host transport — whose callbacks
    reduce doubles (include/gpx.h: gpx_host_comm), so fp32 reductions are widened and narrowed on the host.  fp32 at the
    precision study's level, mixed at 1e-6 on the mean / 1e-7 on alpha against the fp64 oracle; identical on every rank."""
    env = {"SHARD_KERNEL": "rbf", "SHARD_NB": str(nb), "SHARD_N": str(N), "SHARD_DTYPE": dtype, "GPX_SHARD_REPLICATE": str(repl)}
    res = run_ranks("gpu", world, tmp_path, env, timeout=600)
    X, y, Xs = synthetic_problem(N, 3, 90, seed=77)
    ref = OracleGP("rbf", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for r in res:
        assert int(r["info"]) == 0
        em = np.max(np.abs(r["mean"] - mr) / np.maximum(np.abs(mr), 1e-6))
        es = np.max(np.abs(r["mean"] - mr)) / np.max(np.abs(mr))
        ea = np.max(np.abs(r["alpha"] - ref.alpha_)) / np.max(np.abs(ref.alpha_))
        ev = np.max(np.abs(r["var"] - vr)) / 1.5
        if dtype == "mixed":
            assert em <= 1e-6 and ea <= 1e-7 and ev <= 2e-3, (em, ea, ev)
        else:
            assert es <= 2e-3 and ev <= 2e-3 and ea <= 5e-2, (es, ea, ev)
    assert np.array_equal(res[0]["mean"], res[-1]["mean"]) and np.array_equal(res[0]["var"], res[-1]["var"])


@pytest.mark.parametrize("world,N,nbp", [(2, 1500, 256), (3, 1100, 128)])
def test_sharded_lml_gradient_over_host_transport(tmp_path, world, N, nbp):
    """gpx_lml_grad on a row-block shard (one process per rank, host transport): L^-T in row blocks
    dealt over the ranks, all-gather, trace pass split over the ranks — against the CPU oracle's
    analytic gradient; identical on every rank.  Then the same with the factor only held distributed."""
    env = {"SHARD_KERNEL": "matern52", "SHARD_NB": "128", "SHARD_N": str(N), "SHARD_GRAD": "1",
           "GPX_NB_PRED": str(nbp), "GPX_SHARD_REPLICATE": "1"}
    res = run_ranks("gpu", world, tmp_path, env, timeout=600)
    X, y, _ = synthetic_problem(N, 3, 90, seed=77)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    lml_o, grad_o = ref.log_marginal_likelihood(), ref.lml_gradient()
    for r in res:
        assert str(r["grad_err"]) == ""
        assert abs(float(r["lml"]) - lml_o) <= 1e-9 * abs(lml_o)
        assert np.max(np.abs(r["grad"] - grad_o)) <= 1e-7 * np.max(np.abs(grad_o))
        assert np.array_equal(r["grad"], res[0]["grad"])
    # factor ONLY held distributed (the C4 mode): L^-T built distributed, its columns on the owners of the
    # matching row blocks, the trace contracted over the own columns, ntheta numbers all-reduced
    env["GPX_SHARD_REPLICATE"] = "0"
    res = run_ranks("gpu", world, tmp_path, env, timeout=600)
    for r in res:
        assert str(r["grad_err"]) == ""
        assert abs(float(r["lml"]) - lml_o) <= 1e-9 * abs(lml_o)
        assert np.max(np.abs(r["grad"] - grad_o)) <= 1e-7 * np.max(np.abs(grad_o))
        assert np.array_equal(r["grad"], res[0]["grad"])


def test_c4_shape_on_one_gpu_world4(tmp_path):
    """BASELINE.json configs[3] code path — Matern-5/2, d = 3, distributed solves
    (GPX_SHARD_REPLICATE=0), library-chosen block height (1024) — at N = 65536 on 4 ranks
    (4 x 8.6 GB) sharing ONE GPU through the host transport: 16 panels per rank, staircase tile
    maps, un-permute and both panel-buffer sets at the per-rank shapes of the real run."""
    N, M = 65536, 1024
    res = run_ranks("c4", 4, tmp_path, {"SHARD_N": str(N), "SHARD_M": str(M), "SHARD_NB": "0",
                                        "GPX_SHARD_REPLICATE": "0"}, timeout=800)
    resid = c4_checks(res, N, M, single_gpu_c4(N, M))
    print(f"C4 shape, world 4, N={N}: residual {resid:.2e}, fit {float(res[0]['fit_s']):.1f} s "
          f"(chol {float(res[0]['chol_ms']):.0f} ms, comm {float(res[0]['comm_ms']):.0f} ms), "
          f"predict {float(res[0]['predict_s']):.1f} s")
