#!/usr/bin/env python3
"""One-off: the sharded schedule at a larger, ragged size (ranks share one GPU through the
host transport) against the single-GPU path.   python tools/shard_selfcheck.py [N] [world]"""
import os, sys, tempfile, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from shard_util import run_ranks
from oracle.gp_oracle import synthetic_problem
from gaussianprocesspathmodelling_amd import GP

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M, kernel, ls, sf2, sn2 = 500, "rbf", (0.3, 0.2, 0.25), 1.5, 1e-2
with tempfile.TemporaryDirectory() as tmp:
    res = run_ranks("gpu", world, tmp, {"SHARD_KERNEL": kernel, "SHARD_NB": "512", "SHARD_N": str(N),
                                         "SHARD_M": str(M)}, timeout=900)
X, y, Xs = synthetic_problem(N, 3, M, seed=77)
with GP(kernel, ls, sf2, sn2, jitter=0.0) as gp:
    mean, var = gp.fit(X, y).predict(Xs)
    alpha, logdet = gp.alpha_, gp.log_det_
out = {"N": N, "world": world}
for r in res:
    out.setdefault("mean_rel", []).append(float(np.max(np.abs(r["mean"] - mean) / np.maximum(np.abs(mean), 1e-6))))
    out.setdefault("var_rel", []).append(float(np.max(np.abs(r["var"] - var) / np.maximum(var, 1e-6 * sf2))))
    out.setdefault("alpha_rel", []).append(float(np.max(np.abs(r["alpha"] - alpha)) / np.max(np.abs(alpha))))
    out.setdefault("logdet_rel", []).append(float(abs(float(r["logdet"]) - logdet) / abs(logdet)))
    out.setdefault("comm_ms", []).append(float(r["comm_ms"]))
print(json.dumps(out))
