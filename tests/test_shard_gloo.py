"""CPU, world_size 2 and 3 over gloo: (i) the host-collective callbacks the library calls,
(ii) the NumPy restatement of the sharded schedule against the single-process oracle,
(iii) the block-cyclic bookkeeping."""
import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import dist as gdist
from oracle.gp_oracle import OracleGP, synthetic_problem
from shard_util import run_ranks


@pytest.mark.parametrize("snake", [True, False])
def test_dealing_bookkeeping(snake):
    """The dealing of row blocks over the ranks (cyclic / snake), Python mirror against the library's Deal struct
    (gpx_debug_deal: host only), and the balance the snake is for: per panel, the heaviest rank's share of the trailing
    update against the mean."""
    import ctypes as C
    from gaussianprocesspathmodelling_amd import _abi
    lib = _abi.load()
    for P in (1, 2, 3, 8):
        for nblk in (1, 2, 5, 16, 17, 64):
            owned = [gdist.blocks_owned(r, nblk, P, snake) for r in range(P)]
            assert sorted(sum(owned, [])) == list(range(nblk))
            own = (C.c_int32 * nblk)()
            loc = (C.c_int64 * nblk)()
            upto = (C.c_int64 * (nblk * P))()
            assert lib.gpx_debug_deal(P, int(snake), nblk, own, loc, upto) == 0
            for p in range(nblk):
                assert own[p] == gdist.owner(p, P, snake) and loc[p] == gdist.local_index(p, P, snake)
                assert owned[own[p]][loc[p]] == p
                for r in range(P):
                    assert gdist.lb0(p, r, P, snake) == sum(1 for g in owned[r] if g <= p) == upto[p * P + r]

    def imbalance(nblk, P):
        worst = mean = 0.0
        for p in range(nblk - 1):
            w = [0.0] * P
            for g in range(p + 1, nblk):
                w[gdist.owner(g, P, snake)] += g - p - 0.5
            worst += max(w)
            mean += sum(w) / P
        return worst / mean
    # N = 65536: P = 8 x nb = 512 (128 blocks) and the C4 shape (256 blocks on 8 ranks)
    if snake:
        assert imbalance(128, 8) < 1.01 and imbalance(256, 8) < 1.005 and imbalance(64, 4) < 1.01
    else:
        assert 1.08 < imbalance(128, 8) < 1.09 and 1.04 < imbalance(256, 8) < 1.05


def test_host_collective_callbacks_world2(tmp_path):
    res = run_ranks("callbacks", 2, tmp_path)
    for rank, r in enumerate(res):
        assert r["err"] == "None"
        assert np.array_equal(r["bcast"], np.arange(6.0) + 10)           # root 1's data everywhere
        assert np.array_equal(r["allgather"], np.concatenate([np.arange(6.0), np.arange(6.0) + 100]))
        assert np.array_equal(r["armin"], [0.0, 4.0]) and r["arsum"][0] == 3.0
    assert np.array_equal(res[0]["reduce"], 2 * np.arange(6.0) + 100)
    assert np.array_equal(res[1]["reduce"], np.full(6, -1.0))            # non-root untouched


@pytest.mark.parametrize("world,kernel,one_pass,M,deal", [(2, "rbf", 0, 90, "snake"), (3, "matern52", 0, 90, "snake"),
                                                          (2, "rbf", 1, 90, "snake"),        # one pass: rank 1's slice is empty
                                                          (3, "matern52", 1, 300, "snake"),
                                                          (3, "matern52", 0, 90, "cyclic"), (2, "rbf", 1, 90, "cyclic")])
def test_sharded_schedule_matches_oracle(tmp_path, world, kernel, one_pass, M, deal):
    """one_pass: the query points as bordered rows of the sharded factorisation (gpx_fit_predict on a shard); deal: the
    dealing of row blocks over the ranks (snake = the library's default since round 4)."""
    res = run_ranks("oracle", world, tmp_path, {"SHARD_KERNEL": kernel, "SHARD_NB": "128", "SHARD_ONE_PASS": str(one_pass),
                                                "SHARD_M": str(M), "GPX_SHARD_DEAL": deal})
    X, y, Xs = synthetic_problem(700, 3, M, seed=77)
    ref = OracleGP(kernel, (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    for r in res:
        assert np.max(np.abs(r["mean"] - mr)) <= 1e-9 * max(1.0, np.abs(mr).max())
        assert np.max(np.abs(r["var"] - vr) / np.maximum(vr, 1e-6 * 1.5)) <= 1e-7
        assert np.max(np.abs(r["alpha"] - ref.alpha_)) <= 1e-8 * np.abs(ref.alpha_).max()
        assert abs(float(r["logdet"]) - ref.log_det_) <= 1e-10 * abs(ref.log_det_)
