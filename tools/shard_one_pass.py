#!/usr/bin/env python3
"""The one-pass step on the SHARDED schedule (round 4: every rank's slice of the query points rides through its part of
the factorisation as bordered rows) against the two calls, on ONE card: the sharded schedule on one rank (1-rank RCCL
communicator), and device groups sharing the card over the in-process transport, in both solve modes (replicated factor /
distributed solves).  The ranks of a group compete for one GPU: total work + exposed chains, not a scaling number — what
the comparison shows is the predict phase the pass absorbs (replicated: M / P rows against the whole factor; distributed:
a sweep of N / nb broadcasts).   python tools/shard_one_pass.py [--n 65536] [--m 4096] [--reps 3] [--ranks 4,8]"""
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
ap.add_argument("--ranks", default="4,8")
ap.add_argument("--kernel", default="rbf")
a = ap.parse_args()
X, y, Xs = synthetic_problem(a.n, 3, a.m, seed=12345)
dev = torch.device("cuda", 0)
Xd, yd, Xsd = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
out = {"N": a.n, "M": a.m, "reps": a.reps, "kernel": a.kernel}


def clock(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


def measure(gp, tag):
    gp.fit(Xd, yd).predict(Xsd)
    gp.fit_predict(Xd, yd, Xsd)
    two, one, fit, pred = [], [], [], []
    for _ in range(a.reps):
        t_f, _ = clock(lambda: gp.fit(Xd, yd))
        t_p, (m2, v2) = clock(lambda: gp.predict(Xsd))
        fit.append(t_f)
        pred.append(t_p)
        two.append(t_f + t_p)
        t_1, (m1, v1) = clock(lambda: gp.fit_predict(Xd, yd, Xsd))
        one.append(t_1)
    md = lambda v: round(float(np.median(v)), 2)  # noqa: E731
    out[tag] = {"two_calls_ms": md(two), "fit_ms": md(fit), "predict_ms": md(pred), "one_pass_ms": md(one),
                "gain_ms": round(md(two) - md(one), 2),
                "mean_max_abs_diff": float((m1 - m2).abs().max()), "var_max_abs_diff": float((v1 - v2).abs().max())}
    print(tag, json.dumps(out[tag]), flush=True)


with GP(a.kernel, 0.25, 1.5, 1e-2, jitter=0.0, device=0) as gp:
    measure(gp, "unsharded")
with GP(a.kernel, 0.25, 1.5, 1e-2, jitter=0.0, device=0, world=1, rank=0, comm="rccl") as gp:
    measure(gp, "sharded_schedule_one_rank")
for P in [int(v) for v in a.ranks.split(",") if v]:
    for repl in (1, 0):
        if repl and P * (a.n * a.n * 8.0 + a.n * a.n * 8.0 / P) > 250e9:  # P whole factors beside the shards on one card
            continue
        os.environ["GPX_SHARD_REPLICATE"] = str(repl)
        with GP(a.kernel, 0.25, 1.5, 1e-2, jitter=0.0, devices=[0] * P, transport="local") as gp:
            measure(gp, f"group_{P}_ranks_one_card_{'replicated' if repl else 'distributed'}")
os.environ.pop("GPX_SHARD_REPLICATE", None)
print(json.dumps(out))
