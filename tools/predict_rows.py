#!/usr/bin/env python3
"""Predict (mean + variance) against a fitted N = 65536 factor for few query points — the per-rank share of a replicated shard
(M / P rows: 512 at P = 8) — two calls and the one-pass form.   python tools/predict_rows.py [--n 65536]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=65536); a = ap.parse_args()
N = a.n
dev = torch.device("cuda", 0)
X, y, Xs = (torch.from_numpy(v).to(dev) for v in synthetic(N, 3, 8192, 12345))
out = {"N": N}
with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0) as gp:
    gp.fit(X, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gp.fit(X, y); torch.cuda.synchronize(); fit_ms = (time.perf_counter() - t0) * 1e3
    out["fit_ms"] = round(fit_ms, 1)
    for M in (128, 256, 512, 1024, 2048, 4096, 8192):
        q = Xs[:M].contiguous()
        gp.predict(q); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); gp.predict(q); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ms = min(ts)
        t0 = time.perf_counter(); gp.fit_predict(X, y, q); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3
        gp.fit(X, y)
        out[f"M={M}"] = {"predict_ms": round(ms, 2), "tflops": round(N * N * M / (ms * 1e-3) / 1e12, 1),
                          "one_pass_minus_fit_ms": round(one - fit_ms, 1)}
        print(M, out[f"M={M}"], flush=True)
print(json.dumps(out))
