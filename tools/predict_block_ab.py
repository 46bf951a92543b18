import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
N = 65536
dev = torch.device("cuda", 0)
X, y, Xs = (torch.from_numpy(v).to(dev) for v in synthetic(N, 3, 4096, 12345))
out = {}
for blk, nbp in ((512, "512"), (512, "1024"), (1024, "1024"), (1024, "512"), (2048, "2048"), (2048, "1024")):
    os.environ["GPX_NB_PRED"] = nbp
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, device=0, block=blk) as gp:
        gp.fit(X, y)
        for M in (512, 1024, 4096):
            q = Xs[:M].contiguous()
            gp.predict(q); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); gp.predict(q); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            out[f"fit block {blk}, predict block {nbp}, M={M}"] = round(min(ts), 2)
            print(f"fit block {blk}, predict block {nbp}, M={M}: {min(ts):.2f} ms", flush=True)
print(json.dumps(out))
