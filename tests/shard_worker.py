"""Worker process of the shard tests: `python shard_worker.py MODE OUT.npz` with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.
  MODE = oracle   NumPy restatement of the schedule (CPU, gloo)
  MODE = callbacks  the ctypes host-collective callbacks on numpy buffers (CPU, gloo)
  MODE = gpu      the HIP path, ranks sharing one GPU, host transport over gloo
  MODE = c4       BASELINE.json configs[3] shape (Matern-5/2, d=3, SURVEY.md §8d hyper-parameters,
                  distributed solves) at a size that fits one GPU; saves what the §8d checks need
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    nb = int(os.environ.get("SHARD_NB", "128"))
    N = int(os.environ.get("SHARD_N", "700"))
    M = int(os.environ.get("SHARD_M", "90"))
    kernel = os.environ.get("SHARD_KERNEL", "rbf")
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.gp_oracle import synthetic_problem
    X, y, Xs = synthetic_problem(N, 3, M, seed=77)
    ls, sf2, sn2 = (0.3, 0.2, 0.25), 1.5, 1e-2
    res = {}
    if mode == "oracle":
        import oracle.dist_oracle as dist_oracle
        from oracle.dist_oracle import NumpyCollectives, sharded_fit_predict
        dist_oracle.SNAKE = os.environ.get("GPX_SHARD_DEAL", "snake") != "cyclic"
        mean, var, alpha, logdet = sharded_fit_predict(NumpyCollectives(), X, y, Xs, kernel, ls, sf2, sn2, nb,
                                                       one_pass=os.environ.get("SHARD_ONE_PASS") == "1")
        res = dict(mean=mean, var=var, alpha=alpha, logdet=logdet)
    elif mode == "callbacks":
        from gaussianprocesspathmodelling_amd.dist import HostCollectives
        hc = HostCollectives()
        vt = hc.vtable
        a = np.arange(6, dtype=np.float64) + 10 * rank
        vt.bcast(None, a.ctypes.data, a.nbytes, 1)
        g = np.zeros(6 * world)
        s = np.arange(6, dtype=np.float64) + 100 * rank
        vt.allgather(None, s.ctypes.data, g.ctypes.data, s.nbytes)
        r_out = np.full(6, -1.0)
        vt.reduce(None, s.ctypes.data, r_out.ctypes.data, 6, 0, 0)
        ar = np.array([float(rank), 5.0 - rank])
        vt.allreduce(None, ar.ctypes.data, 2, 1)       # min
        ar2 = np.array([float(rank + 1)])
        vt.allreduce(None, ar2.ctypes.data, 1, 0)      # sum
        res = dict(bcast=a, allgather=g, reduce=r_out, armin=ar, arsum=ar2, err=str(hc.last_error))
    elif mode == "gpu":
        if nb > 0:
            os.environ["GPX_NB_SHARD"] = str(nb)
        else:                                   # 0: let the library choose from N and world
            os.environ.pop("GPX_NB_SHARD", None)
        from gaussianprocesspathmodelling_amd import GP
        if os.environ.get("SHARD_DISAGREE") == "1":      # rank 1's environment differs: the library must notice, on every rank
            if rank == 1:
                os.environ["GPX_SHARD_DEAL"] = "cyclic"
            from gaussianprocesspathmodelling_amd import GpxError
            with GP(kernel, ls, sf2, sn2, jitter=0.0, device=0, world=world, rank=rank, comm="host") as gp:
                try:
                    gp.fit(X, y)
                    res = dict(error="")
                except GpxError as e:
                    res = dict(error=str(e))
            np.savez(out + f".rank{rank}.npz", **res)
            dist.barrier()
            dist.destroy_process_group()
            return
        dtype = os.environ.get("SHARD_DTYPE", "float64")
        if dtype == "float32":
            X, y, Xs = (v.astype(np.float32) for v in (X, y, Xs))
        with GP(kernel, ls, sf2, sn2, jitter=0.0, device=0, world=world, rank=rank, comm="host", dtype=dtype) as gp:
            if os.environ.get("SHARD_ONE_PASS") == "1":   # every rank's slice of Xs rides through its part of the factorisation
                mean, var = gp.fit_predict(X, y, Xs)
                m2, v2 = gp.predict(Xs)                   # what the model it leaves behind predicts
                res.update(mean_two_calls=m2, var_two_calls=v2)
            else:
                gp.fit(X, y)
                mean, var = gp.predict(Xs)
            res.update(mean=mean, var=var, alpha=gp.alpha_, logdet=gp.log_det_, info=gp.info_,
                       comm_ms=gp.timings_["comm"])
            if os.environ.get("SHARD_GRAD") == "1":
                from gaussianprocesspathmodelling_amd import GpxError
                try:
                    lml, grad = gp.lml_gradient()
                    res.update(lml=lml, grad=grad, grad_err="")
                except GpxError as e:
                    res.update(lml=np.nan, grad=np.zeros(0), grad_err=str(e))
    elif mode == "c4":
        if nb > 0:
            os.environ["GPX_NB_SHARD"] = str(nb)
        else:
            os.environ.pop("GPX_NB_SHARD", None)
        import time
        from gaussianprocesspathmodelling_amd import GP
        X, y, Xs = synthetic_problem(N, 3, M, seed=12345)          # the SURVEY.md §8(d) generator
        rows = np.random.default_rng(1).choice(N, min(N, 1024), replace=False)
        with GP("matern52", 0.25, 1.5, 1e-2, jitter=0.0, device=0, world=world, rank=rank, comm="host") as gp:
            t0 = time.perf_counter()
            gp.fit(X, y)
            t1 = time.perf_counter()
            mean, var = gp.predict(Xs)
            t2 = time.perf_counter()
            mean_t, var_t = gp.predict(X[rows])
            tm = gp.timings_
            res = dict(mean=mean, var=var, alpha=gp.alpha_, logdet=gp.log_det_, info=gp.info_, rows=rows,
                       mean_t=mean_t, var_t=var_t, fit_s=t1 - t0, predict_s=t2 - t1, comm_ms=tm["comm"],
                       chol_ms=tm["chol"])
    np.savez(out + f".rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
