"""Path data model, CSV reader and path k-means around the GP engine.

Mirrors the reference's own world so its user can switch over:

=====================================  ==========================================
reference (``GPmap.py``)               here
=====================================  ==========================================
``trajectory`` (``:12-23``)            :class:`Trajectory` (``timestamp, xs, ys``)
``trajectories.pathdict`` (``:30-34``) :class:`Trajectories` (``pathdict``, ``add_trajectory``)
``check_if_valid_trajectory``          :func:`check_if_valid_trajectory` — same *signed* travel
(``:165-175``)                         sum, so inward-moving paths are rejected exactly as there
``readcsvfile`` (``:178-204``)         :func:`read_csv` — same block rules (header row carries the
                                       id in column 1, ``t,_,x,y`` rows with x, y parsed as int,
                                       ``###`` ends a block, keep only valid paths of exactly 33
                                       points, stop after ``numoftrajstoread`` kept paths)
``calc_distance`` (``:114-121``)       :func:`path_distance_matrix` — ONE HIP kernel for all
                                       (path, centroid) pairs (``gpx_path_distance``)
``calc_mean_traj`` (``:95-112``)       :func:`mean_path` (empty cluster = ``ZeroDivisionError``)
``kmeansclustering`` (``:36-93``)      :func:`kmeans` — same Lloyd loop, first-minimum ties, stop
                                       when the centroids moved < 5 in total
=====================================  ==========================================

Deliberate divergences (documented, not reproduced): the reference's "too close" re-draw
loop (``:39-53``) never terminates once a close pair is drawn and compares only
``j in range(i+1, k-1)``; here every pair is checked and re-draws are bounded.  Plotting
(``:125-161``) is out of scope.  The file name is an argument, not the hard-coded
``'testfile.csv'`` (``:181``), and nothing runs at import.

The distance matrix — the reference's hot loop — runs on the GPU through the C ABI; the
O(P*k) bookkeeping around it (arg-min, cluster means) is host NumPy.
"""
from __future__ import annotations

import csv
import ctypes as C
import io
import os
import random as _random
from dataclasses import dataclass, field

import numpy as np

from . import _abi

PATH_LENGTH = 33          # GPmap.py:189
MIN_TRAVEL = 1000         # GPmap.py:36,189
MOVEMENT_STOP = 5.0       # GPmap.py:90


@dataclass
class Trajectory:
    """Growable (timestamp, x, y) path — ``class trajectory`` (GPmap.py:12-23)."""
    xs: np.ndarray = field(default_factory=lambda: np.array([], dtype=float))
    ys: np.ndarray = field(default_factory=lambda: np.array([], dtype=float))
    timestamp: np.ndarray = field(default_factory=lambda: np.array([], dtype=float))

    def add_point(self, time, x, y):
        self.xs = np.append(self.xs, x)
        self.ys = np.append(self.ys, y)
        self.timestamp = np.append(self.timestamp, time)

    def __len__(self):
        return len(self.xs)


class Trajectories:
    """id -> Trajectory — ``class trajectories`` (GPmap.py:28-34)."""

    def __init__(self):
        self.pathdict = {}

    def add_trajectory(self, id, trajectory):
        self.pathdict[id] = trajectory

    def __len__(self):
        return len(self.pathdict)

    def keys(self):
        return list(self.pathdict.keys())

    def as_array(self, keys=None):
        """(P, L, 3) array of (t, x, y) for equal-length paths."""
        keys = self.keys() if keys is None else list(keys)
        return np.stack([np.stack([self.pathdict[k].timestamp, self.pathdict[k].xs, self.pathdict[k].ys], 1)
                         for k in keys]) if keys else np.zeros((0, PATH_LENGTH, 3))

    def kmeansclustering(self, k, treshold=MIN_TRAVEL, **kw):
        """Same name and argument spelling as the reference (GPmap.py:36)."""
        return kmeans(self, k, too_close=treshold, **kw)


def travel_score(traj) -> float:
    """sum_{i<j} (|x_j|-|x_i| + |y_j|-|y_i|), the signed sum of GPmap.py:165-175."""
    a = np.abs(np.asarray(traj.xs, float)) + np.abs(np.asarray(traj.ys, float))
    n = len(a)
    return float(np.sum(a * (2.0 * np.arange(n) - n + 1.0)))


def check_if_valid_trajectory(traj, minimumtraveldistance=1) -> bool:
    return not (travel_score(traj) < minimumtraveldistance)


def read_csv(source, numoftrajstoread=0, minimum_travel=MIN_TRAVEL, length=PATH_LENGTH, into=None):
    """Parse the reference's CSV wire format (GPmap.py:178-204) from a path, file object or
    text; returns a :class:`Trajectories`.  A final block without a closing ``###`` row is
    dropped, as in the reference; a file that starts with ``###`` raises (there: an
    ``UnboundLocalError`` at :189)."""
    trajs = Trajectories() if into is None else into
    if isinstance(source, str) and "\n" not in source and os.path.exists(source):
        fh, close = open(source, newline=""), True
    elif isinstance(source, str):
        fh, close = io.StringIO(source), False          # CSV text
    else:
        fh, close = source, False                       # file object
    try:
        kept = 0
        new = None
        is_new = True
        pid = 0
        for row in csv.reader(fh, delimiter=","):
            if not row:
                continue
            if row[0] == "###":
                if new is None:
                    raise ValueError("CSV starts with a '###' row: no trajectory to close")
                if check_if_valid_trajectory(new, minimum_travel) and len(new.timestamp) == length:
                    trajs.add_trajectory(pid, new)
                    kept += 1
                if numoftrajstoread != 0 and kept >= numoftrajstoread:
                    break
                new = Trajectory()
                is_new = True
            elif not is_new:
                new.add_point(float(row[0]), int(row[2]), int(row[3]))
            else:
                pid = row[1]
                new = Trajectory()
                is_new = False
    finally:
        if close:
            fh.close()
    return trajs


def to_gp_inputs(trajs, keys=None, inputs=("t",), targets=("x", "y"), normalise=True):
    """Flatten paths into GP training data: X (33*P, d) from the chosen input columns, Y
    (33*P, k) from the target columns (k targets share one Cholesky factor).  With
    ``normalise`` every input column is mapped to [0, 1] (returns the (lo, span) used)."""
    col = {"t": 0, "x": 1, "y": 2}
    arr = trajs.as_array(keys).reshape(-1, 3)
    X = arr[:, [col[c] for c in inputs]].astype(np.float64)
    Y = arr[:, [col[c] for c in targets]].astype(np.float64)
    lo, span = np.zeros(X.shape[1]), np.ones(X.shape[1])
    if normalise and len(X):
        lo = X.min(axis=0)
        span = np.where(X.max(axis=0) > lo, X.max(axis=0) - lo, 1.0)
        X = (X - lo) / span
    return np.ascontiguousarray(X), np.ascontiguousarray(Y), (lo, span)


# ---- the hot loop: all path-to-centroid distances in one launch ---------------------------------
def path_distance_matrix(paths_xy, centroids_xy):
    """D[p, c] = sum_i ||paths[p, i] - centroids[c, i]||_2 on the GPU (``gpx_path_distance``).
    paths (P, L, 2), centroids (C, L, 2): NumPy arrays, or torch CUDA tensors (no copies)."""
    lib = _abi.load()
    if type(paths_xy).__module__.split(".")[0] == "torch" and paths_xy.is_cuda:
        import torch
        p = paths_xy.detach().to(torch.float64).contiguous()
        c = centroids_xy.detach().to(device=p.device, dtype=torch.float64).contiguous()
        if p.ndim != 3 or c.ndim != 3 or p.shape[1:] != c.shape[1:] or p.shape[2] != 2:
            raise ValueError("need paths (P, L, 2) and centroids (C, L, 2)")
        D = torch.empty((p.shape[0], c.shape[0]), dtype=torch.float64, device=p.device)
        torch.cuda.current_stream(p.device).synchronize()
        with torch.cuda.device(p.device):
            rc = lib.gpx_path_distance(C.c_void_p(p.data_ptr()), p.shape[0], C.c_void_p(c.data_ptr()),
                                       c.shape[0], p.shape[1], C.c_void_p(D.data_ptr()), _abi.MEM_DEVICE)
        if rc != 0:
            raise _abi.GpxError(rc, lib.gpx_last_error(None).decode())
        return D
    p = np.ascontiguousarray(paths_xy, dtype=np.float64)
    c = np.ascontiguousarray(centroids_xy, dtype=np.float64)
    if p.ndim != 3 or c.ndim != 3 or p.shape[1:] != c.shape[1:] or p.shape[2] != 2:
        raise ValueError("need paths (P, L, 2) and centroids (C, L, 2)")
    D = np.empty((p.shape[0], c.shape[0]), dtype=np.float64)
    rc = lib.gpx_path_distance(C.c_void_p(p.ctypes.data), p.shape[0], C.c_void_p(c.ctypes.data), c.shape[0],
                               p.shape[1], C.c_void_p(D.ctypes.data), _abi.MEM_HOST)
    if rc != 0:
        raise _abi.GpxError(rc, lib.gpx_last_error(None).decode())
    return D


def mean_path(paths_txy):
    """Point-wise mean of (n, L, 3) paths — ``calc_mean_traj`` (GPmap.py:95-112)."""
    paths_txy = np.asarray(paths_txy, dtype=np.float64)
    if paths_txy.shape[0] == 0:
        raise ZeroDivisionError("empty cluster (GPmap.py:111 divides by the cluster size)")
    return paths_txy.sum(axis=0) / paths_txy.shape[0]


def kmeans(trajs, k, too_close=MIN_TRAVEL, init_keys=None, rng=None, movement_stop=MOVEMENT_STOP,
           max_iter=10_000, max_redraws=1000):
    """Lloyd's k-means over whole paths (GPmap.py:36-93) -> {centroid index: [path ids]}.

    Initial centroids: ``init_keys`` or ``rng.sample(keys, k)`` (``rng`` defaults to Python's
    ``random`` module, which is what the reference draws from, so an identical seed draws the
    identical keys).  Each iteration: one GPU launch for all distances, first-minimum
    assignment (the reference's strict ``<`` in dict order), centroid = point-wise mean of its
    cluster, stop when the centroids moved less than ``movement_stop`` in total."""
    keys = trajs.keys()
    if k <= 0 or k > len(keys):
        raise ValueError("need 1 <= k <= number of paths")
    arr = trajs.as_array(keys)                      # (P, L, 3)
    xy = np.ascontiguousarray(arr[:, :, 1:3])
    rng = _random if rng is None else rng
    if init_keys is None:
        init_keys = rng.sample(keys, k)
        for _ in range(max_redraws):                # bounded; every pair checked
            idx = [keys.index(q) for q in init_keys]
            Dc = path_distance_matrix(xy[idx], xy[idx])
            if not np.any(Dc[np.triu_indices(k, 1)] < too_close):
                break
            init_keys = rng.sample(keys, k)
    idx = [keys.index(q) for q in init_keys]
    cents = arr[idx].copy()                         # (k, L, 3): deep copies of the chosen paths
    for _ in range(max_iter):
        D = path_distance_matrix(xy, np.ascontiguousarray(cents[:, :, 1:3]))
        assign = np.argmin(D, axis=1)               # first minimum = the reference's tie rule
        clusters = {c: [keys[p] for p in np.nonzero(assign == c)[0]] for c in range(k)}
        new = np.stack([mean_path(arr[assign == c]) for c in range(k)])
        d = new[:, :, 1:3] - cents[:, :, 1:3]
        moved = float(np.sum(np.sqrt(np.sum(d * d, axis=-1))))
        cents = new
        if moved < movement_stop:
            return clusters
    raise RuntimeError("k-means did not converge")


# ---- one GP per path cluster (independent replicas) ---------------------------------------------
class PathModel:
    """GP model of one cluster of paths: inputs (by default the time stamp, mapped to [0, 1]) ->
    (x, y) as two targets sharing ONE Cholesky factor.  Targets are
    standardised per column before the fit (the GP prior has zero mean) and mapped back by
    :meth:`predict`."""

    def __init__(self, gp, keys, in_lo, in_span, y_mean, y_std, inputs, targets):
        self.gp, self.keys = gp, list(keys)
        self.in_lo, self.in_span, self.y_mean, self.y_std = in_lo, in_span, y_mean, y_std
        self.inputs, self.targets = tuple(inputs), tuple(targets)

    def predict(self, q, return_var=True, include_noise=False):
        """Posterior at raw (un-normalised) inputs ``q`` (M,) or (M, d): mean (M, k) in the
        targets' own units and, with ``return_var``, the variance (M, k) per target."""
        q = np.asarray(q, dtype=np.float64)
        q = q.reshape(-1, 1) if q.ndim == 1 else q
        if q.ndim != 2 or q.shape[1] != len(self.inputs):
            raise ValueError(f"queries must be (M,) or (M, {len(self.inputs)})")
        qn = np.ascontiguousarray((q - self.in_lo) / self.in_span)
        out = self.gp.predict(qn, return_var=return_var, include_noise=include_noise)
        mean = (out[0] if return_var else out).reshape(len(qn), -1) * self.y_std + self.y_mean
        if not return_var:
            return mean
        return mean, out[1][:, None] * (self.y_std ** 2)[None, :]

    def close(self):
        self.gp.close()


def fit_path_models(trajs, clusters, inputs=("t",), targets=("x", "y"), devices=None, optimize=False,
                    lengthscale=0.25, variance=1.0, noise=0.05, **gp_kwargs):
    """One exact GP per cluster of ``clusters`` = {cluster id: [path ids]} — what
    :func:`kmeans` (``kmeansclustering``, GPmap.py:36-93) returns — modelling the cluster's paths
    as (x(t), y(t)); the modelling step the reference's title names and its clustering prepares
    (the reference itself stops after the clustering, GPmap.py:220).

    The clusters are independent problems, so they run as REPLICAS (SURVEY.md §8e "Replicas-only
    parts"): cluster i is fitted on ``devices[i % len(devices)]`` (an int n = devices 0..n-1;
    default: the current device), one host thread per device, no data-path collective.  With
    ``optimize`` the hyper-parameters of each model are fitted by :meth:`GP.optimize` (analytic
    gradient) first.  Returns {cluster id: :class:`PathModel`}; empty clusters are skipped."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from .gp import GP
    if devices is None:
        devs = [None]
    elif isinstance(devices, (int, np.integer)):
        if devices < 1:
            raise ValueError("devices must be >= 1")
        devs = list(range(int(devices)))
    else:
        devs = [int(v) for v in devices]
        if not devs:
            raise ValueError("devices must not be empty")
    known = set(trajs.keys())
    jobs = []
    for cid, keys in clusters.items():
        keys = list(keys)
        if not keys:
            continue
        missing = [q for q in keys if q not in known]
        if missing:
            raise KeyError(f"cluster {cid!r} names unknown paths {missing[:3]}")
        jobs.append((cid, keys))

    def fit_one(slot, cid, keys):
        X, Y, (lo, span) = to_gp_inputs(trajs, keys, inputs=inputs, targets=targets, normalise=True)
        mu = Y.mean(axis=0)
        sd = Y.std(axis=0)
        sd = np.where(sd > 0, sd, 1.0)
        Yn = np.ascontiguousarray((Y - mu) / sd)
        gp = GP(lengthscale=lengthscale, variance=variance, noise=noise, device=devs[slot % len(devs)],
                **gp_kwargs)
        try:
            if optimize:
                gp.optimize(X, Yn)                 # leaves the model fitted at the best point
            else:
                gp.fit(X, Yn)
        except Exception:
            gp.close()
            raise
        return cid, PathModel(gp, keys, lo, span, mu, sd, inputs, targets)

    by_dev = [[] for _ in devs]                    # one worker per device, its clusters in order
    for i, (cid, keys) in enumerate(jobs):
        by_dev[i % len(devs)].append((i, cid, keys))
    models = {}
    lock = threading.Lock()

    def worker(lst):
        # every model is filed the moment it exists: if a later cluster of this device raises, the
        # ones already built are still in `models` and get closed below (their factors live on the GPU)
        for i, c, k in lst:
            cid, m = fit_one(i, c, k)
            with lock:
                models[cid] = m

    with ThreadPoolExecutor(max_workers=len(devs)) as pool:
        futs = [pool.submit(worker, lst) for lst in by_dev if lst]
        err = None
        for f in futs:
            try:
                f.result()
            except Exception as e:                  # keep collecting so every handle gets closed
                err = err or e
        if err is not None:
            for m in models.values():
                m.close()
            raise err
    return {cid: models[cid] for cid, _ in jobs}
