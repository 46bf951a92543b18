#!/usr/bin/env python3
"""One-off: the timing-perturbation check of tests/test_delay_gpu.py at larger sizes (N = 32768 on
one GPU; N = 16384 on a 4-rank device group with distributed solves; N = 32768 on 8 ranks).
Every output must be bit-identical with and without the random spin kernels."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import GP, _abi
from oracle.gp_oracle import synthetic_problem
lib = _abi.load()


def run(N, M, kw):
    kw = dict(kw)
    one_pass = kw.pop("one_pass", False)
    X, y, Xs = synthetic_problem(N, 3, M, seed=N)
    with GP("matern52", 0.25, 1.5, 1e-2, jitter=0.0, **kw) as gp:
        mean, var = gp.fit_predict(X, y, Xs) if one_pass else gp.fit(X, y).predict(Xs)
        return [mean, var, gp.alpha_.copy(), np.float64(gp.log_det_)]


out = []
for N, M, kw, env in ((32768, 2048, {}, {}),
                      (49152, 1024, {}, {}),          # round 4: 2048-wide panels (the library's choice from N = 40960 on)
                      (32768, 1024, {"devices": 4, "oversubscribe": True, "dtype": "mixed"}, {"GPX_SHARD_REPLICATE": "0"}),   # round 4: mixed shard, distributed refinement
                      (16384, 1024, {"devices": 4, "oversubscribe": True}, {"GPX_SHARD_REPLICATE": "0"}),
                      (32768, 1024, {"devices": 8, "oversubscribe": True}, {"GPX_SHARD_REPLICATE": "0"}),
                      # end of round 4: the one pass on shards (query rows riding), snake dealing, the owner's chain on its own
                      # stream (from 4 ranks on), the replicated factor's panel copy on the copy stream
                      (32768, 2048, {"devices": 8, "oversubscribe": True, "one_pass": True}, {"GPX_SHARD_REPLICATE": "0"}),
                      (24576, 2048, {"devices": 4, "oversubscribe": True, "one_pass": True}, {"GPX_SHARD_REPLICATE": "1"})):
    os.environ.update(env)
    base = run(N, M, kw)
    same = True
    t = []
    for seed in (3, 41):
        lib.gpx_debug_set_delay(seed)
        t0 = time.time(); got = run(N, M, kw); t.append(round(time.time() - t0, 2))
        lib.gpx_debug_set_delay(0)
        same = same and all(np.array_equal(a, b) for a, b in zip(base, got))
    for k in env:
        os.environ.pop(k)
    out.append({"N": N, "M": M, "config": {k: v for k, v in kw.items()}, "bit_identical_under_delays": bool(same), "seconds_delayed": t})
    print(out[-1], file=sys.stderr, flush=True)
print(json.dumps(out))
sys.exit(0 if all(o["bit_identical_under_delays"] for o in out) else 1)
