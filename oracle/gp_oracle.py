"""CPU oracle for the exact-GP fit/predict hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package
(``gaussianprocesspathmodelling_amd``) never does and fails loudly when its HIP
library is missing.

Provenance / parity status
--------------------------
The upstream reference (``/root/reference/GPmap.py``, 219 lines) contains **no**
Gaussian-process code: no kernel matrix, no Cholesky, no ``fit``/``predict``
(its only linalg call is ``np.linalg.norm`` at ``GPmap.py:120``; the
``scipy.spatial.distance`` import at ``GPmap.py:10`` is never used).  Parity
with the reference is therefore **unpinned by the reference itself**.  This
oracle is a build-authored fp64 restatement of textbook exact GP regression
(Rasmussen & Williams, GPML, Algorithm 2.1), written with the reference's own
import stack (``numpy`` ``GPmap.py:1`` and ``scipy.spatial.distance``
``GPmap.py:10``) and pinned instead by

* scikit-learn 1.7.2 ``GaussianProcessRegressor`` cross-checks executed by
  ``oracle/make_golden.py`` in the build container (third-party, not the
  reference), and
* the committed golden fixtures ``tests/golden/G*.npz`` that script wrote.

Algorithm (R&W Alg. 2.1), all fp64::

    K     = sf2 * k(X, X) + (sn2 + jitter) * I
    L     = chol(K)                      (lower)
    alpha = L^-T (L^-1 y)
    mu    = K* alpha                     K* = sf2 * k(Xs, X)
    V     = L^-1 K*^T
    var   = sf2 - colsum(V*V)            (latent; + sn2 if include_noise)
    logdet= 2 * sum(log(diag L))

Kernels (r = || (x - x') / ell ||, ell scalar or per-dimension "ARD")::

    rbf      : exp(-r^2 / 2)
    matern52 : (1 + sqrt5 r + 5 r^2 / 3) * exp(-sqrt5 r)

The trajectory-side helpers at the bottom restate the reference's only
floating-point routines (``calc_distance`` ``GPmap.py:114-121``,
``calc_mean_traj`` ``GPmap.py:95-112``, ``check_if_valid_trajectory``
``GPmap.py:165-175``) in vectorised numpy for the §8(f) "next" rows.
"""
from __future__ import annotations

import math
import time

import numpy as np
from scipy.linalg import cholesky, solve_triangular
from scipy.spatial import distance as dst  # same alias as GPmap.py:10

KERNELS = ("rbf", "matern52")
SQRT5 = math.sqrt(5.0)


def _scaled(A, lengthscale):
    ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("inputs must be 2-D (n, d)")
    if ls.size not in (1, A.shape[1]):
        raise ValueError("lengthscale must be scalar or have d entries")
    return A / ls


def kernel_matrix(A, B, kernel="rbf", lengthscale=1.0, variance=1.0, out=None):
    """sf2 * k(A, B): (na, d), (nb, d) -> (na, nb) fp64.

    SURVEY.md §8 rows a1/a2.  ``cdist`` is the call the reference imports as
    ``dst`` (``GPmap.py:10``)."""
    As, Bs = _scaled(A, lengthscale), _scaled(B, lengthscale)
    if kernel == "rbf":
        K = dst.cdist(As, Bs, "sqeuclidean", out=out)
        K *= -0.5
        np.exp(K, out=K)
    elif kernel == "matern52":
        K = dst.cdist(As, Bs, "euclidean", out=out)
        K *= SQRT5                       # s = sqrt5 * r
        # (1 + s + s^2/3) * exp(-s), done blockwise to bound temporaries
        step = max(1, (1 << 22) // max(1, K.shape[1]))
        for i in range(0, K.shape[0], step):
            s = K[i:i + step]
            e = np.exp(-s)
            s[...] = (1.0 + s + s * s / 3.0) * e
    else:
        raise ValueError(f"unknown kernel {kernel!r}; expected one of {KERNELS}")
    K *= variance
    return K


class OracleGP:
    """NumPy/SciPy exact GP with the same surface as the product ``GP`` class."""

    def __init__(self, kernel="rbf", lengthscale=1.0, variance=1.0, noise=1e-2,
                 jitter=None, max_tries=1, chol="lapack"):
        if kernel not in KERNELS:
            raise ValueError(f"unknown kernel {kernel!r}")
        self.kernel = kernel
        self.lengthscale = lengthscale
        self.variance = float(variance)
        self.noise = float(noise)
        self.jitter = 1e-10 * self.variance if jitter is None else float(jitter)
        self.max_tries = int(max_tries)
        if chol not in ("lapack", "blocked"):
            raise ValueError("chol must be 'lapack' (scipy.linalg.cholesky) or 'blocked' (chol_lower_blocked)")
        self.chol = chol
        self.timings_ = {}

    # -- fit -----------------------------------------------------------------
    def fit(self, X, y, keep_K_corner=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        self._y1d = y.ndim == 1
        Y = y.reshape(len(X), -1)
        if Y.shape[0] != X.shape[0]:
            raise ValueError("X and y disagree on N")
        self.X_ = X
        jitter = self.jitter
        for attempt in range(self.max_tries):
            t0 = time.perf_counter()
            K = kernel_matrix(X, X, self.kernel, self.lengthscale, self.variance)
            K[np.diag_indices_from(K)] += self.noise + jitter
            if keep_K_corner:
                self.K_corner_ = K[:keep_K_corner, :keep_K_corner].copy()
            t1 = time.perf_counter()
            try:
                if self.chol == "blocked":      # level-3 blocked variant: scales with the BLAS threads
                    L = chol_lower_blocked(K)   # (strictly upper triangle keeps K: never read below)
                else:
                    L = cholesky(K, lower=True, overwrite_a=True, check_finite=False)
                self.info_ = 0
                break
            except np.linalg.LinAlgError:
                self.info_ = 1
                jitter = max(jitter, 1e-12 * self.variance) * 10.0
        else:
            raise np.linalg.LinAlgError("kernel matrix not positive definite")
        t2 = time.perf_counter()
        z = solve_triangular(L, Y, lower=True, check_finite=False)
        alpha = solve_triangular(L, z, lower=True, trans="T", check_finite=False)
        t3 = time.perf_counter()
        self.L_ = L
        self.jitter_used_ = jitter
        self.alpha_ = alpha[:, 0] if self._y1d else alpha
        self.log_det_ = 2.0 * float(np.sum(np.log(np.diag(L))))
        self._Y = Y
        self.timings_.update(kbuild=(t1 - t0) * 1e3, chol=(t2 - t1) * 1e3,
                             solve=(t3 - t2) * 1e3)
        return self

    # -- predict -------------------------------------------------------------
    def predict(self, Xs, return_var=True, include_noise=False):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        t0 = time.perf_counter()
        Ks = kernel_matrix(Xs, self.X_, self.kernel, self.lengthscale, self.variance)
        t1 = time.perf_counter()
        A = self.alpha_.reshape(len(self.X_), -1)
        mean = Ks @ A
        mean = mean[:, 0] if self._y1d else mean
        t2 = time.perf_counter()
        self.timings_.update(kstar=(t1 - t0) * 1e3, mean=(t2 - t1) * 1e3)
        if not return_var:
            return mean
        V = solve_triangular(self.L_, Ks.T, lower=True, check_finite=False,
                             overwrite_b=True)
        t3 = time.perf_counter()
        var = self.variance - np.einsum("ij,ij->j", V, V)
        if include_noise:
            var = var + self.noise
        t4 = time.perf_counter()
        self.timings_.update(trsm=(t3 - t2) * 1e3, var=(t4 - t3) * 1e3)
        return mean, var

    def log_marginal_likelihood(self):
        """-1/2 y^T alpha - 1/2 logdet - N/2 log(2 pi), summed over targets."""
        A = self.alpha_.reshape(len(self.X_), -1)
        n, k = self._Y.shape
        return float(-0.5 * np.sum(self._Y * A) - 0.5 * k * self.log_det_
                     - 0.5 * n * k * math.log(2.0 * math.pi))


    def lml_gradient(self):
        """d LML / d log(theta), theta = (lengthscale[0..n_ls), variance, noise): R&W eq. 5.9,
        1/2 tr((alpha alpha^T - K^-1) dK/dtheta) summed over the target columns.  SURVEY.md
        §8(f) row 1 ("hyper-parameter gradient hooks"); no counterpart in the reference."""
        X = self.X_
        n = len(X)
        A = self.alpha_.reshape(n, -1)
        k = A.shape[1]
        Linv = solve_triangular(self.L_, np.eye(n), lower=True, check_finite=False)
        W = A @ A.T - k * (Linv.T @ Linv)
        ls = np.atleast_1d(np.asarray(self.lengthscale, dtype=np.float64))
        Xs = X / ls
        Kf = kernel_matrix(X, X, self.kernel, self.lengthscale, self.variance)
        if self.kernel == "rbf":
            Kd = Kf                                       # dK/dlog l_c = Kf * d_c^2
        else:
            s = SQRT5 * dst.cdist(Xs, Xs, "euclidean")    # dK/dlog l_c = sf2 (5/3)(1+s) e^-s d_c^2
            Kd = self.variance * (5.0 / 3.0) * (1.0 + s) * np.exp(-s)
        WKd = W * Kd
        g_ls = np.empty(Xs.shape[1])
        for c in range(Xs.shape[1]):
            dc = Xs[:, c][:, None] - Xs[:, c][None, :]
            g_ls[c] = 0.5 * np.sum(WKd * dc * dc)
        if ls.size == 1:
            g_ls = np.array([g_ls.sum()])
        g_sf2 = 0.5 * np.sum(W * Kf)
        g_sn2 = 0.5 * self.noise * np.trace(W)
        return np.concatenate([g_ls, [g_sf2, g_sn2]])


# -- synthetic workload of SURVEY.md §8(d) --------------------------------------
def synthetic_problem(N, d, M, seed=12345):
    """Draw order X, Xs, noise — fixed by SURVEY.md §8(d)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.0, 1.0, (N, d))
    Xs = rng.uniform(0.0, 1.0, (M, d))
    y = (np.sin(2.0 * np.pi * X[:, 0]) + 0.5 * np.cos(3.0 * X[:, 1:].sum(axis=1))
         + 0.1 * rng.standard_normal(N))
    return X, y, Xs


# -- small linear-algebra pieces used by the kernel unit tests -------------------
def chol_lower(K):
    return cholesky(np.array(K, dtype=np.float64), lower=True, check_finite=False)


def chol_lower_blocked(K, nb=2048, rb=4096):
    """In-place lower Cholesky of the C-contiguous SPD matrix K by the same right-looking blocked
    algorithm the HIP path uses (R&W Alg. 2.1 line 2), expressed with level-3 NumPy/SciPy calls:
    LAPACK potrf only on nb x nb diagonal blocks, triangular solve for the panel, matrix products
    for the trailing update (row blocks of rb rows, lower part only).  Why it exists: the bundled
    OpenBLAS `potrf` does not parallelise below N ~ 10^4 (26-32 GF/s at any thread count at
    N = 8192, profiles/r02_cpu_baseline_host.json) and crashed on the N = 65536 matrix with a
    16-thread pool, while `dgemm` scales — so this is both the faster CPU baseline and the
    full-size parity oracle (tools/full_oracle_c3.py).  Returns K (lower triangle = L; the strictly
    upper triangle is left as it was)."""
    n = K.shape[0]
    assert K.shape == (n, n) and K.flags.c_contiguous and K.dtype == np.float64
    for o in range(0, n, nb):
        e = min(o + nb, n)
        Ld = cholesky(K[o:e, o:e], lower=True, check_finite=False)
        K[o:e, o:e] = Ld
        if e == n:
            break
        # panel: P = A[e:, o:e] Ld^-T, by row blocks (bounded temporaries)
        for i in range(e, n, rb):
            j = min(i + rb, n)
            K[i:j, o:e] = solve_triangular(Ld, K[i:j, o:e].T, lower=True, check_finite=False).T
        # trailing update, lower part: A[i:j, e:j] -= P[i:j] P[e:j]^T
        for i in range(e, n, rb):
            j = min(i + rb, n)
            K[i:j, e:j] -= K[i:j, o:e] @ K[e:j, o:e].T
    return K


def trsm_right_lower_trans(A, L):
    """X = A @ inv(L).T  (the panel solve of a right-looking blocked Cholesky)."""
    return solve_triangular(L, np.asarray(A, dtype=np.float64).T, lower=True,
                            check_finite=False).T


# -- reference trajectory numerics (for the §8(f) rows) ---------------------------
def path_distance(xs1, ys1, xs2, ys2):
    """Sum_i ||p_i - q_i||_2 — ``trajectories.calc_distance`` (GPmap.py:114-121)."""
    return float(np.sum(np.hypot(np.asarray(xs1, float) - np.asarray(xs2, float),
                                 np.asarray(ys1, float) - np.asarray(ys2, float))))


def path_distance_matrix(paths_xy, centroids_xy):
    """D[p, c] = sum_i ||paths[p, i, :] - centroids[c, i, :]||_2 (batched
    ``calc_distance``, GPmap.py:72-80,114-121).  paths (P, L, 2), centroids (C, L, 2)."""
    diff = np.asarray(paths_xy, float)[:, None, :, :] - np.asarray(centroids_xy, float)[None]
    return np.sqrt((diff * diff).sum(axis=-1)).sum(axis=-1)


def mean_path(paths_txy):
    """Point-wise mean of (P, L, 3) paths — ``calc_mean_traj`` (GPmap.py:95-112);
    like the reference, an empty cluster is a ZeroDivisionError."""
    paths_txy = np.asarray(paths_txy, float)
    if paths_txy.shape[0] == 0:
        raise ZeroDivisionError("empty cluster (GPmap.py:111)")
    return paths_txy.sum(axis=0) / paths_txy.shape[0]


def travel_score(xs, ys):
    """The signed sum of ``check_if_valid_trajectory`` (GPmap.py:165-175):
    sum_{i<j} (|x_j|-|x_i| + |y_j|-|y_i|) = sum_k (|x_k|+|y_k|) (2k - L + 1)."""
    a = np.abs(np.asarray(xs, float)) + np.abs(np.asarray(ys, float))
    L = a.shape[-1]
    return float(np.sum(a * (2.0 * np.arange(L) - L + 1.0)))


def is_valid_path(xs, ys, minimum_travel=1.0):
    return not (travel_score(xs, ys) < minimum_travel)
