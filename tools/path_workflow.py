#!/usr/bin/env python3
"""The reference's workflow end to end on the GPU, at a size where it matters: a synthetic CSV in
the reference's wire format (header row with the id, `t,_,x,y` rows, `###`; GPmap.py:178-204) ->
read_csv (validity rules) -> k-means over whole paths (distances on the GPU) -> one two-target GP
per cluster (x(t), y(t) on one factor) -> posterior path + band per cluster.  Prints one JSON line
with the stage times.   python tools/path_workflow.py [--paths 3000] [--clusters 6] [--optimize]"""
import argparse, io, json, os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import paths as gpaths

ap = argparse.ArgumentParser()
ap.add_argument("--paths", type=int, default=3000)
ap.add_argument("--clusters", type=int, default=6)
ap.add_argument("--optimize", action="store_true")
ap.add_argument("--devices", type=int, default=1)
a = ap.parse_args()

rng = np.random.default_rng(7)
buf = io.StringIO()
t = np.arange(33) * 0.5
for p in range(a.paths):
    g = p % a.clusters
    ang = 2 * np.pi * g / a.clusters
    s = np.linspace(0.0, 1.0, 33)
    r = 8000.0 + 22000.0 * s                       # outward from the origin (the reference's signed-sum
    x = r * np.cos(ang + 0.35 * s) + rng.normal(0, 120) + rng.normal(0, 40, 33)   # validity rule keeps those)
    y = r * np.sin(ang + 0.35 * s) + rng.normal(0, 120) + rng.normal(0, 40, 33)
    buf.write(f"hdr,P{p:05d},x,y\n")
    for i in range(33):
        buf.write(f"{t[i]},0,{int(x[i])},{int(y[i])}\n")
    buf.write("###\n")
text = buf.getvalue()

times = {}
t0 = time.perf_counter(); trajs = gpaths.read_csv(text); times["read_csv_s"] = time.perf_counter() - t0
keys = trajs.keys()
random.seed(3)
init = [keys[g] for g in range(a.clusters)]         # one path of every group: the reference draws at random and re-draws
t0 = time.perf_counter(); clusters = gpaths.kmeans(trajs, a.clusters, init_keys=init); times["kmeans_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
models = gpaths.fit_path_models(trajs, clusters, devices=a.devices, kernel="matern52", lengthscale=0.3, noise=0.02,
                                optimize=a.optimize)
times["fit_models_s"] = time.perf_counter() - t0
q = np.linspace(0.0, 16.0, 129)
t0 = time.perf_counter()
out = {}
for cid, m in models.items():
    mean, var = m.predict(q, include_noise=True)
    arr = trajs.as_array(clusters[cid])
    inside = np.mean(np.abs(arr[:, :, 1:3] - m.predict(arr[0, :, 0], return_var=False)[None]) <= 3.0 * np.sqrt(m.predict(arr[0, :, 0], include_noise=True)[1])[None])
    out[str(cid)] = {"paths": len(clusters[cid]), "n_train": 33 * len(clusters[cid]), "end_point": mean[-1].round(0).tolist(),
                     "band_3sigma_coverage": float(inside), "lengthscale": m.gp.lengthscale.round(4).tolist(), "noise": round(m.gp.noise, 5)}
times["predict_s"] = time.perf_counter() - t0
for m in models.values():
    m.close()
print(json.dumps({"config": f"{a.paths} synthetic paths of 33 points in the reference CSV format, k = {a.clusters}, devices = {a.devices}",
                  "kept_paths": len(keys), **{k: round(v, 3) for k, v in times.items()}, "clusters": out}))
