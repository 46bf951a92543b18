#!/usr/bin/env python3
"""Seeded random sweep of the HIP path against the CPU oracle over shapes the fixed tests do not
enumerate: N, M around the 64 / 128 / 1024 boundaries, d up to 32, k up to 64, both kernels, scalar
and ARD lengthscales, panel widths, device groups of 2..6 ranks (both solve modes), mean-only
predict, the analytic gradient; round 4: every fifth case in dtype "mixed" or "float32" (also on the groups: the shard
runs in the handle's element type), the self-reserving update and the sharded schedule switches at random.  Prints one line per case and a JSON summary; exit code 1 on any
miss of the bars below.   python tools/fuzz_parity.py [--cases 60] [--seed 1]
tests/test_fuzz_gpu.py runs a short sweep of it in the -m gpu suite."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP

EDGES = [1, 2, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 1023, 1024, 1025, 1500, 2047, 2049, 2600]
BARS = (("mean", 1e-6), ("var", 1e-6), ("alpha", 1e-6), ("logdet", 1e-9), ("mean_only", 1e-8), ("grad", 1e-6),
        ("one_pass", 1e-6))


def one_case(rng, c):
    N = int(rng.choice(EDGES[2:])) + int(rng.integers(0, 3)) * int(rng.integers(0, 40))
    M = int(rng.choice(EDGES[:12]))
    d = int(rng.choice([1, 2, 3, 5, 8, 17, 32]))
    k = int(rng.choice([1, 1, 1, 2, 3, 7, 64]))
    kernel = str(rng.choice(["rbf", "matern52"]))
    ard = bool(rng.integers(0, 2)) and d > 1
    base = 0.6 * np.sqrt(d)
    ls = (base * rng.uniform(0.7, 1.4, d)) if ard else float(base * rng.uniform(0.7, 1.4))
    sf2, sn2 = float(rng.uniform(0.5, 2.0)), float(10 ** rng.uniform(-3, -1))
    block = int(rng.choice([0, 0, 128, 256, 512, 2048]))
    ndev = int(rng.choice([1, 1, 1, 2, 3, 4, 6]))
    repl = int(rng.integers(0, 2))
    nbs = int(rng.choice([128, 256]))
    dtype = "float64" if c % 5 != 4 else str(rng.choice(["mixed", "float32"]))
    if dtype != "float64":          # an fp32 factorisation wants cond(K) eps_32 << 1; the mixed mode takes <= 8 targets
        k = min(k, int(rng.choice([1, 2, 8])))
        sn2 = float(10 ** rng.uniform(-1.5, -0.5))
    bars = dict(BARS)
    if dtype == "mixed":            # fp64-grade mean and alpha; variance and log-determinant through the fp32 factor
        bars.update(var=5e-3, logdet=1e-3, one_pass=5e-3)   # (the one-pass figure includes the variance)
    elif dtype == "float32":
        bars.update(mean=5e-3, var=5e-3, alpha=5e-2, logdet=1e-3, mean_only=5e-3, one_pass=5e-3)
    X = rng.uniform(0, 1, (N, d))
    W = rng.standard_normal((d, k))
    y = np.sin(3.0 * X @ W) + 0.1 * rng.standard_normal((N, k))
    y = y[:, 0] if k == 1 else y
    Xs = rng.uniform(-0.1, 1.1, (M, d))
    tag = (f"case {c}: N={N} M={M} d={d} k={k} {kernel} ard={ard} block={block} ndev={ndev} "
           f"repl={repl if ndev > 1 else '-'}")
    ref = OracleGP(kernel, ls, sf2, sn2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    kw = {}
    # round-3 schedule switches, at random: 64-wide diagonal steps, hipEvent hand-over of the diagonal chain
    variant = {"GPX_DIAG_STEP": str(rng.choice(["128", "64"])), "GPX_CHAIN_FLAG": str(rng.choice(["1", "0"])),
               "GPX_SPLIT_STRIP": str(rng.choice(["1", "0"])),
               "GPX_REST_SPLIT": str(rng.choice(["16", "0", "2"])), "GPX_SOLVE_TOP": str(rng.choice(["1", "0"])),
               "GPX_CU_SELF_RESERVE": str(rng.choice(["0", "0", "1", "2"]))}
    os.environ.update(variant)
    tag += f" {dtype} step={variant['GPX_DIAG_STEP']} flag={variant['GPX_CHAIN_FLAG']} split={variant['GPX_SPLIT_STRIP']} resv={variant['GPX_CU_SELF_RESERVE']}"
    if ndev > 1:
        os.environ["GPX_SHARD_REPLICATE"] = str(repl)
        os.environ["GPX_NB_SHARD"] = str(nbs)
        kw = dict(devices=ndev, oversubscribe=True)
    try:
        if dtype == "float32":
            Xf, yf, Xsf = X.astype(np.float32), y.astype(np.float32), Xs.astype(np.float32)
        else:
            Xf, yf, Xsf = X, y, Xs
        with GP(kernel, ls, sf2, sn2, jitter=0.0, block=block, dtype=dtype, **kw) as gp:
            mean, var = gp.fit(Xf, yf).predict(Xsf)
            Xs_in, X_in, y_in = Xsf, Xf, yf
            m2 = gp.predict(Xs_in, return_var=False)
            e = {"mean": float(np.max(np.abs(mean - mr)) / max(np.max(np.abs(mr)), 1e-30)),
                 "var": float(np.max(np.abs(var - vr)) / sf2),
                 "alpha": float(np.max(np.abs(gp.alpha_ - ref.alpha_)) / np.max(np.abs(ref.alpha_))),
                 "logdet": float(abs(gp.log_det_ - ref.log_det_) / abs(ref.log_det_)),
                 "mean_only": float(np.max(np.abs(m2 - mean)) / max(np.max(np.abs(mean)), 1e-30)), "grad": 0.0, "one_pass": 0.0}
            if c % 2 == 0:       # every other case: fit + predict as one pass (groups: through the sharded factorisation unless split=0), then
                m1, v1 = gp.fit_predict(X_in, y_in, Xs_in)    # everything below runs on the handle it leaves behind
                e["one_pass"] = float(max(np.max(np.abs(m1 - mr)) / max(np.max(np.abs(mr)), 1e-30), np.max(np.abs(v1 - vr)) / sf2))
            if N <= 1600 and dtype == "float64":   # round 3: also when the factor is only held distributed (ndev > 1, repl 0)
                lml, grad = gp.lml_gradient()
                go, lo = ref.lml_gradient(), ref.log_marginal_likelihood()
                e["grad"] = float(max(np.max(np.abs(grad - go)) / np.max(np.abs(go)), abs(lml - lo) / abs(lo)))
    finally:
        os.environ.pop("GPX_DIAG_STEP", None)
        os.environ.pop("GPX_CHAIN_FLAG", None)
        os.environ.pop("GPX_SPLIT_STRIP", None)
        os.environ.pop("GPX_REST_SPLIT", None)
        os.environ.pop("GPX_SOLVE_TOP", None)
        os.environ.pop("GPX_CU_SELF_RESERVE", None)
        os.environ.pop("GPX_SHARD_REPLICATE", None)
        os.environ.pop("GPX_NB_SHARD", None)
    return tag, e, bars


def sweep(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst = {kk: 0.0 for kk, _ in BARS}
    fails = []
    t_start = time.time()
    for c in range(cases):
        try:
            tag, e, bars = one_case(rng, c)
        except Exception as ex:                                   # noqa: BLE001 — a sweep reports, it does not stop
            print(f"case {c}: EXCEPTION {ex!r}", flush=True)
            fails.append(f"case {c}: {ex!r}")
            continue
        bad = [kk for kk, lim in bars.items() if not e[kk] <= lim]
        if bars == dict(BARS):       # the worst-error summary is the fp64 cases'; fp32 / mixed cases are judged by their own bars
            for kk in worst:
                worst[kk] = max(worst[kk], e[kk])
        if verbose or bad:
            print(tag, {kk: f"{v:.1e}" for kk, v in e.items()}, "FAIL " + ",".join(bad) if bad else "ok", flush=True)
        if bad:
            fails.append(tag)
    return {"cases": cases, "seed": seed, "failed": fails, "worst_relative_errors": worst,
            "seconds": round(time.time() - t_start, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    res = sweep(a.cases, a.seed)
    print(json.dumps(res))
    sys.exit(1 if res["failed"] else 0)
