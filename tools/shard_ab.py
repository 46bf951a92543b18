#!/usr/bin/env python3
"""A/B of the sharded factorisation's schedule on ONE card (VERDICT r3 item 1 "done" criteria): per variant
(GPX_SPLIT_STRIP = 1: round-4 split schedule, 0: round-3 schedule) the step of
  * the unsharded handle (the comparator),
  * the sharded schedule on ONE rank (1-rank RCCL communicator, replicated factor: its own overhead),
  * a device group of 4 (and 8) ranks sharing the card over the in-process transport (fit only counts: the ranks
    compete for one GPU, so this is total work + exposed chain, not a scaling number).
C3 workload: N = 65536, d = 3, RBF, M = 4096.   python tools/shard_ab.py [--n 65536] [--reps 3] [--ranks 4,8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (its bundled RCCL first: INTEGRATION.md §7)
from gaussianprocesspathmodelling_amd import GP  # noqa: E402
from oracle.gp_oracle import synthetic_problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ranks", default="4")
a = ap.parse_args()
X, y, Xs = synthetic_problem(a.n, 3, a.m, seed=12345)
dev = torch.device("cuda", 0)
Xd, yd, Xsd = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
out = {"N": a.n, "M": a.m, "reps": a.reps}


def steps(gp, tag, variants=("1", "0")):
    gp.fit(Xd, yd).predict(Xsd)  # warm-up: buffers
    res = {}
    for rep in range(a.reps):
        for v in variants:
            os.environ["GPX_SPLIT_STRIP"] = v
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gp.fit(Xd, yd)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mean, var = gp.predict(Xsd)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            tm = gp.timings_
            r = res.setdefault(v, {"fit_ms": [], "predict_ms": [], "chol_ms": [], "solve_ms": [], "comm_ms": []})
            r["fit_ms"].append((t1 - t0) * 1e3)
            r["predict_ms"].append((t2 - t1) * 1e3)
            r["chol_ms"].append(tm["chol"])
            r["solve_ms"].append(tm["solve"])
            r["comm_ms"].append(tm["comm"])
            res.setdefault("mean", {})[v] = mean.cpu().numpy()
    os.environ.pop("GPX_SPLIT_STRIP", None)
    bit = all(np.array_equal(res["mean"][variants[0]], res["mean"][v]) for v in variants)
    out[tag] = {("split" if v == "1" else "round3"): {k: round(float(np.median(x)), 2) for k, x in res[v].items()}
                for v in variants}
    out[tag]["bit_identical_between_schedules"] = bool(bit)
    print(tag, json.dumps(out[tag]), flush=True)
    return res["mean"][variants[0]]


with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, profile=True) as gp:
    m_one = steps(gp, "unsharded")
with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, profile=True, world=1, rank=0, comm="rccl") as gp:
    m_sh = steps(gp, "sharded_schedule_one_rank")
out["one_rank_vs_unsharded_max_rel"] = float(np.max(np.abs(m_sh - m_one) / np.maximum(np.abs(m_one), 1e-6)))
for P in [int(v) for v in a.ranks.split(",") if v]:
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, devices=[0] * P, transport="local", profile=True) as gp:
        m_g = steps(gp, f"group_{P}_ranks_one_card")
    out[f"group_{P}_vs_unsharded_max_rel"] = float(np.max(np.abs(m_g - m_one) / np.maximum(np.abs(m_one), 1e-6)))
print(json.dumps(out))
