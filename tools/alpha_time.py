import os, sys, time, torch
sys.path.insert(0, '.')
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
X, y, Xs = (torch.from_numpy(v).to("cuda:0") for v in synthetic(65536, 3, 4096, 12345))
for few in ("1", "0"):
    os.environ["GPX_FEW_SOLVE"] = few
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0) as gp:
        gp.fit(X, y); gp.alpha_
        gp.fit(X, y); torch.cuda.synchronize(); t0 = time.perf_counter(); a = gp.alpha_; t1 = time.perf_counter()
        m = gp.predict(Xs, return_var=False); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"GPX_FEW_SOLVE={few}: alpha on demand {(t1 - t0) * 1e3:.2f} ms (incl. D2H of 0.5 MB), device solve {gp.timings_['solve']:.2f} ms; mean-only predict {(t2 - t1) * 1e3:.2f} ms")
