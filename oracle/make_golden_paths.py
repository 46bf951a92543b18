"""Generate tests/golden/G4.json — reference-behaviour goldens for the §8(f) rows.

TEST INFRASTRUCTURE, run in the BUILD container only (the reference does not travel):
imports /root/reference/GPmap.py with a SYNTHETIC testfile.csv in the working directory
(the reference reads 'testfile.csv' from CWD at import, GPmap.py:181,212; its own CSV is
git-ignored and not shipped) and records VALUES only:
  - what readcsvfile kept (ids, parsed arrays)                      GPmap.py:178-204
  - calc_distance for a set of pairs                                GPmap.py:114-121
  - calc_mean_traj of a list of ids                                 GPmap.py:95-112
  - check_if_valid_trajectory booleans (incl. the inward-path quirk) GPmap.py:165-175
  - the partition kmeansclustering(3) returns for recorded initial keys  GPmap.py:36-93
No reference source text is copied; the CSV below is this build's own synthetic data.

    MPLBACKEND=Agg python oracle/make_golden_paths.py
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/GPmap.py"


def synthetic_csv(seed=2024):
    """12 valid outward-moving paths in 3 separated groups, one short path (20 points) and
    one inward-moving path (rejected by the reference's signed travel sum)."""
    rng = np.random.default_rng(seed)
    blocks, meta = [], []
    centers = [(3000, 2000), (-25000, 12000), (9000, -30000)]
    pid = 0
    order = []
    for g, (cx, cy) in enumerate(centers):
        for j in range(4):
            order.append((g, j))
    rng.shuffle(order)
    for g, j in order:
        cx, cy = centers[g]
        sx, sy = (1 if cx >= 0 else -1), (1 if cy >= 0 else -1)
        x0, y0 = cx + 400 * j * sx, cy + 300 * j * sy
        steps = rng.integers(150, 400, size=(33, 2))
        xs = x0 + sx * np.cumsum(steps[:, 0])
        ys = y0 + sy * np.cumsum(steps[:, 1])
        ts = 1000.0 * pid + 0.5 * np.arange(33)
        blocks.append((f"P{pid:02d}", ts, xs, ys))
        meta.append(g)
        pid += 1
    # a short path and an inward path, inserted early so that readcsvfile(10) meets them
    xs = 40000 - 500 * np.arange(33)
    blocks.insert(2, ("INWARD", 0.5 * np.arange(33), xs, xs.copy()))
    xs = 1000 + 300 * np.arange(20)
    blocks.insert(5, ("SHORT", 0.5 * np.arange(20), xs, xs.copy()))
    lines = []
    for name, ts, xs, ys in blocks:
        lines.append(f"hdr,{name},x,y")
        for t, x, y in zip(ts, xs, ys):
            lines.append(f"{t},0,{int(x)},{int(y)}")
        lines.append("###,,,")
    return "\n".join(lines) + "\n"


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    csv_text = synthetic_csv()
    out = {"csv": csv_text}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "testfile.csv"), "w") as f:
            f.write(csv_text)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            import random
            random.seed(7)                    # the module runs kmeansclustering(3) at import
            spec = importlib.util.spec_from_file_location("GPmap_ref", REF)
            ref = importlib.util.module_from_spec(spec)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                spec.loader.exec_module(ref)  # reads 10 paths, clusters, "plots" (Agg)
            T = ref.trajs
            keys = list(T.pathdict.keys())
            out["kept_ids"] = keys
            out["paths"] = {k: {"t": T.pathdict[k].timestamp.tolist(), "x": T.pathdict[k].xs.tolist(),
                                "y": T.pathdict[k].ys.tolist()} for k in keys}
            pairs = [(keys[i], keys[j]) for i in range(len(keys)) for j in range(i + 1, len(keys))][:20]
            out["distance_pairs"] = [[a, b, float(T.calc_distance(T.pathdict[a], T.pathdict[b]))] for a, b in pairs]
            sub = keys[:4]
            mt = T.calc_mean_traj(sub)
            out["mean_of"] = sub
            out["mean_traj"] = {"t": mt.timestamp.tolist(), "x": mt.xs.tolist(), "y": mt.ys.tolist()}
            # validity incl. the quirk: an inward-moving 33-point path is rejected
            inward = ref.trajectory()
            for i in range(33):
                inward.add_point(float(i), 30000 - 500 * i, 30000 - 500 * i)
            out["valid"] = {"inward_1000": bool(ref.check_if_valid_trajectory(inward, 1000)),
                            "first_kept_1000": bool(ref.check_if_valid_trajectory(T.pathdict[keys[0]], 1000)),
                            "first_kept_default": bool(ref.check_if_valid_trajectory(T.pathdict[keys[0]]))}
            # k-means: record the initial keys the seeded run draws, then the partition
            random.seed(123)
            init = random.sample(list(T.pathdict.keys()), 3)
            random.seed(123)
            clusters = T.kmeansclustering(3)
            out["kmeans_init_keys"] = init
            out["kmeans_partition"] = sorted(sorted(v) for v in clusters.values())
        finally:
            os.chdir(cwd)
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "G4.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print("kept", out["kept_ids"])
    print("valid", out["valid"])
    print("partition", out["kmeans_partition"], "init", out["kmeans_init_keys"])
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
