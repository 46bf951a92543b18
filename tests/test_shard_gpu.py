"""GPU: the sharded schedule with the real HIP kernels.
  - 1-rank RCCL communicator: every RCCL entry point the path uses, on one GPU;
  - 2 and 3 ranks sharing the GPU through the host transport (gloo): the block-cyclic
    exchange schedule itself (RCCL refuses two ranks on one device)."""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, synthetic_problem
from shard_util import run_ranks

pytestmark = pytest.mark.gpu


def check(mean, var, alpha, logdet, ref, mr, vr, sf2=1.5):
    dm = np.abs(mean - mr) / np.maximum(np.abs(mr), 1e-6)
    dv = np.abs(var - vr) / np.maximum(vr, 1e-6 * sf2)
    assert dm.max() <= 1e-6 and dv.max() <= 1e-6, (dm.max(), dv.max())
    assert np.max(np.abs(alpha - ref.alpha_)) <= 1e-7 * np.abs(ref.alpha_).max()
    assert abs(logdet - ref.log_det_) <= 1e-9 * abs(ref.log_det_)


@pytest.mark.parametrize("N,M,nb", [(1500, 130, 256), (640, 64, 128)])
def test_sharded_schedule_single_rank_rccl(N, M, nb, monkeypatch):
    monkeypatch.setenv("GPX_NB_SHARD", str(nb))
    X, y, Xs = synthetic_problem(N, 3, M, seed=N)
    ref = OracleGP("rbf", 0.25, 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, world=1, rank=0, comm="rccl") as gp:
        mean, var = gp.fit(X, y).predict(Xs)
        check(mean, var, gp.alpha_, gp.log_det_, ref, mr, vr)
        assert np.max(np.abs(gp.predict(Xs, return_var=False) - mean)) <= 1e-9 * max(1.0, np.abs(mean).max())


@pytest.mark.parametrize("world,kernel,nb,N", [(2, "rbf", 128, 700), (3, "matern52", 128, 700),
                                                 (2, "rbf", 256, 700), (4, "rbf", 512, 3300),
                                                 (2, "rbf", 0, 9000)])   # 0 = block height chosen by the library
def test_sharded_ranks_share_one_gpu(tmp_path, world, kernel, nb, N):
    res = run_ranks("gpu", world, tmp_path, {"SHARD_KERNEL": kernel, "SHARD_NB": str(nb), "SHARD_N": str(N)},
                    timeout=600)
    X, y, Xs = synthetic_problem(N, 3, 90, seed=77)
    ref = OracleGP(kernel, (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for r in res:
        assert int(r["info"]) == 0
        check(r["mean"], r["var"], r["alpha"], float(r["logdet"]), ref, mr, vr)
    # every rank returns the same replicated result
    assert np.array_equal(res[0]["mean"], res[-1]["mean"]) and np.array_equal(res[0]["var"], res[-1]["var"])
