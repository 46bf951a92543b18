#!/bin/bash
set -e -o pipefail
timeout -k 10 600 python -m pytest tests/test_fit_predict_gpu.py -m gpu -x -q > gpurun_out/fp_tests.log 2>&1 || { tail -40 gpurun_out/fp_tests.log; exit 1; }; tail -3 gpurun_out/fp_tests.log
timeout -k 10 900 python -m pytest tests/test_gp_parity_gpu.py tests/test_delay_gpu.py tests/test_fp32_gpu.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import time, json, torch, numpy as np
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
for (N, M, steps) in ((8192, 4096, 20), (65536, 4096, 3)):
    dev = torch.device("cuda", 0)
    X, y, Xs = (torch.from_numpy(v).to(dev) for v in synthetic(N, 3, M, 12345))
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0) as gp:
        out = {}
        for name, fn in (("two_calls", lambda: gp.fit(X, y).predict(Xs)), ("fit_predict", lambda: gp.fit_predict(X, y, Xs))):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): r = fn()
            torch.cuda.synchronize(); out[name] = (time.perf_counter() - t0) / steps * 1e3
            out[name + "_tm"] = {k: round(v, 3) for k, v in gp.timings_.items() if k in ("fit_total", "chol", "predict_total", "kbuild")}
        print(json.dumps({"N": N, "M": M, **out}))
PY
