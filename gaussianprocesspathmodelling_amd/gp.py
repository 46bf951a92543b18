"""Exact Gaussian-process regression with a ``fit()`` / ``predict()`` surface.

This is the Python host side of the hot path.  The upstream reference
(``GPmap.py``) has no such class — its public names are ``trajectory``,
``trajectories``, ``readcsvfile`` (``GPmap.py:12,28,178``) — so the surface is the one
BASELINE.json's north_star names: hyper-parameters are fixed inputs, ``fit(X, y)``
factorises ``K = sf2 k(X,X) + (sn2 + jitter) I`` and solves for ``alpha``,
``predict(Xs)`` returns the posterior mean and (latent) variance.

All arithmetic runs in ``csrc/libgpx.so`` (hand-written HIP for gfx950) through the
ctypes C ABI of ``include/gpx.h``.  There is no CPU code path: without the library or
without a GPU, construction raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class GP:
    """Exact GP regressor on one MI355X (or a row-block shard of several).

    Parameters
    ----------
    kernel : "rbf" | "matern52"
    lengthscale : float or array of d floats (ARD)
    variance : signal variance sf2
    noise : observation-noise variance sn2 (added to the diagonal)
    jitter : extra diagonal term; default 1e-10 * variance
    dtype : "float64" | "float32" (everything, including the factorisation, in fp32:
        the precision study of BASELINE.json configs[4]; not a 1e-6 path) | "mixed" (float64 in and
        out; the factorisation in fp32 at twice the MFMA rate, alpha refined in fp64 against the
        matrix-free fp64 kernel: **fp64-grade posterior mean** (1e-6 elementwise at N = 65536,
        tests/test_full_size_gpu.py) but an **fp32-grade variance** — sf2 - ||L^-1 k*||^2 through
        the fp32 factor is never refined, so it carries absolute errors of ~3e-6 sf2, which is
        percents of a variance of 1e-4 sf2; use "float64" when the variance matters.  At most 8
        targets).  Every dtype also runs sharded (``devices=`` / ``world=``: the shard takes the handle's
        element type; round 4)
    refine : "mixed" only — 0 (default): refine until ||y - K alpha|| <= 1e-10 ||y|| or the residual
        stops contracting (at most 12 iterations; ``timings_["refine_iters"]`` says how many ran);
        n > 0: exactly n iterations
    device : HIP device ordinal (default: LOCAL_RANK or 0)
    devices : several GPUs from ONE ordinary Python process: an int n (devices
        0..n-1) or a list of HIP ordinals.  The Gram matrix is sharded in row blocks dealt over the devices (snake dealing: balanced row work)
        over them; one worker thread per device lives inside ``fit`` / ``predict``, which stay
        plain blocking calls — no launcher, no ``torch.distributed``.  ``devices=1`` / ``[i]`` is
        the single-GPU path on that device — unless ``transport`` is given explicitly: then it is a
        ONE-rank group (the group / RCCL code path on a single GPU; tests).
    transport : how the devices of a ``devices=`` group exchange panels: "rccl"
        (``ncclCommInitAll`` inside the process; distinct devices only), "local" (peer copies
        and hipEvents between the ranks' streams, nothing but HIP), None/"auto" = rccl when the
        devices are distinct, else local.
    oversubscribe : allow ``devices=n`` with fewer than n GPUs visible: ordinals wrap around, so
        several ranks share a GPU over the local transport (how the one-GPU tests run 2..8 ranks)
    block : Cholesky panel width nb (multiple of 128, at most 4096; 0 = the library's choice: 1024, and 2048 from
        N = 40960 on, where nothing of the serial chain is exposed any more)
    max_tries : jitter escalations (x10 each) before ``LinAlgError``
    profile : record per-launch timings of the Cholesky sub-phases
    world, rank : row-block shard of ONE Gram matrix over ``world`` processes (one per GPU).
        Every rank passes the same full X, y, Xs and receives the full mean / var.
    comm : "rccl" (RCCL on the library's stream; unique id shipped by torch.distributed),
        "host" (collectives on host buffers through torch.distributed, e.g. gloo), or
        None = pick from the initialised torch.distributed backend when world > 1.
        ``comm`` with world == 1 runs the sharded schedule on one rank (tests).
    """

    def __init__(self, kernel="rbf", lengthscale=1.0, variance=1.0, noise=1e-2, jitter=None,
                 dtype="float64", device=None, block=0, max_tries=3, profile=False,
                 world=1, rank=0, comm=None, group=None, devices=None, transport=None,
                 oversubscribe=False, refine=0):
        if kernel not in _abi.KERNEL_IDS:
            raise ValueError(f"unknown kernel {kernel!r}; expected one of {sorted(_abi.KERNEL_IDS)}")
        if dtype not in _abi.DTYPE_IDS:
            raise ValueError(f"unknown dtype {dtype!r}")
        self.kernel = kernel
        self.lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)).copy()
        if self.lengthscale.ndim != 1 or not np.all(self.lengthscale > 0):
            raise ValueError("lengthscale must be a positive scalar or 1-D array")
        self.variance = float(variance)
        self.noise = float(noise)
        if not self.variance > 0 or self.noise < 0:
            raise ValueError("need variance > 0 and noise >= 0")
        self.jitter = 1e-10 * self.variance if jitter is None else float(jitter)
        self.dtype = dtype
        self._np_dtype = np.float32 if dtype == "float32" else np.float64
        self.block = int(block)
        self.max_tries = int(max_tries)
        self.refine = int(refine)
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = int(device)
        self._lib = _abi.load()
        self.world, self.rank = int(world), int(rank)
        if transport not in _abi.TRANSPORT_IDS:
            raise ValueError(f"unknown transport {transport!r}; expected rccl / local / auto")
        self.devices = self._resolve_devices(devices, oversubscribe)
        self._is_group = len(self.devices) > 1 or (len(self.devices) == 1 and _abi.TRANSPORT_IDS[transport] != 0)
        if self._is_group and (self.world != 1 or comm is not None):
            raise ValueError("devices= (one process, several GPUs) and world=/comm= (one process per GPU) "
                             "are two different process models: pick one")
        if self.devices:
            self.device = self.devices[0]
            if self._is_group and transport != "local":
                _abi.prefer_torch_rccl()     # the group will load RCCL: torch's copy, after torch (see _abi)
        cfg = _abi.GpxConfig(kernel=_abi.KERNEL_IDS[kernel], dtype=_abi.DTYPE_IDS[dtype],
                             device=self.device, block=self.block, rank=self.rank, world=self.world,
                             flags=_abi.FLAG_PROFILE if profile else 0, ndev=len(self.devices),
                             devices=(C.c_int32 * _abi.MAX_GROUP)(*self.devices),
                             transport=_abi.TRANSPORT_IDS[transport], refine=int(refine))
        h = C.c_void_p()
        rc = self._lib.gpx_create(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise _abi.GpxError(rc, self._lib.gpx_last_error(None).decode())
        self._h = h
        self._fitted = False
        self._alpha = None
        self.info_ = 0
        self.jitter_used_ = self.jitter
        self._host_comm = None
        self._has_comm = self.world > 1 or comm is not None
        if self._has_comm:
            self._init_comm(comm, group)

    def _resolve_devices(self, devices, oversubscribe):
        if devices is None:
            return []
        if isinstance(devices, (int, np.integer)):
            n = int(devices)
            if n < 1:
                raise ValueError("devices must be >= 1")
            devs = list(range(n))
        else:
            devs = [int(v) for v in devices]
            if not devs:
                raise ValueError("devices must not be empty")
        if len(devs) > _abi.MAX_GROUP:
            raise ValueError(f"at most {_abi.MAX_GROUP} devices per group")
        if oversubscribe:
            cnt = C.c_int(0)
            self._lib.gpx_device_count(C.byref(cnt))
            if cnt.value > 0:
                devs = [v % cnt.value for v in devs]
        return devs

    def _init_comm(self, comm, group):
        from . import dist as gdist
        if comm is None:
            import torch.distributed as tdist
            if not tdist.is_initialized():
                raise RuntimeError("world > 1 needs torch.distributed.init_process_group() first")
            comm = "rccl" if tdist.get_backend(group) == "nccl" else "host"
        try:
            if comm == "rccl":
                gdist.init_rccl(self._lib, self._h, self.rank, self.world, group)
            elif comm == "host":
                self._host_comm = gdist.HostCollectives(group)
                self._host_comm.attach(self._lib, self._h)
            else:
                raise ValueError(f"unknown comm {comm!r}")
        except Exception:
            self.close()
            raise

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise _abi.GpxError(rc, self._lib.gpx_last_error(self._h).decode())

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gpx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _as_input(self, a, name):
        """-> (pointer, mem_kind, keepalive, torch_device_or_None, shape)"""
        if _is_torch(a):
            import torch
            t = a.detach()
            tdt = torch.float32 if self.dtype == "float32" else torch.float64
            if t.dtype != tdt:
                t = t.to(tdt)
            t = t.contiguous()
            if t.is_cuda:
                if t.device.index != self.device:
                    raise ValueError(f"{name} lives on cuda:{t.device.index}, handle on {self.device}")
                torch.cuda.current_stream(t.device).synchronize()
                return C.c_void_p(t.data_ptr()), _abi.MEM_DEVICE, t, t.device, tuple(t.shape)
            a = t.numpy()
        arr = np.ascontiguousarray(a, dtype=self._np_dtype)
        return C.c_void_p(arr.ctypes.data), _abi.MEM_HOST, arr, None, arr.shape

    # -- API ----------------------------------------------------------------------------
    def fit(self, X, y):
        px, kx, keepx, devx, sx = self._as_input(X, "X")
        py, ky, keepy, devy, sy = self._as_input(y, "y")
        if len(sx) != 2:
            raise ValueError("X must be (N, d)")
        N, d = sx
        if len(sy) not in (1, 2) or sy[0] != N:
            raise ValueError("y must be (N,) or (N, k) with the same N as X")
        if kx != ky:
            raise ValueError("X and y must both be host arrays or both be device tensors")
        k = 1 if len(sy) == 1 else sy[1]
        if self.lengthscale.size not in (1, d):
            raise ValueError("lengthscale must be scalar or have d entries")
        self._y1d = len(sy) == 1
        self._N, self._d, self._k = N, d, k
        self._alpha = None
        self._fitted = False
        ls = self.lengthscale
        jitter = self.jitter
        info = C.c_int64(0)
        for _ in range(max(1, self.max_tries)):
            rc = self._lib.gpx_fit(self._h, px, py, N, d, k, _abi.dptr(ls), ls.size, self.variance,
                                   self.noise, jitter, kx, C.byref(info))
            self._check(rc)
            self.info_ = int(info.value)
            if self.info_ == 0:
                break
            jitter = max(jitter, 1e-12 * self.variance) * 10.0
        else:
            raise np.linalg.LinAlgError(
                f"kernel matrix not positive definite (first bad pivot {self.info_}) after "
                f"{self.max_tries} jitter escalations")
        self.jitter_used_ = jitter
        self._fitted = True
        ld = C.c_double(0.0)
        self._check(self._lib.gpx_logdet(self._h, C.byref(ld)))
        self.log_det_ = float(ld.value)
        return self

    def fit_predict(self, X, y, Xs, include_noise=False):
        """``fit(X, y)`` and ``predict(Xs)`` (mean and variance) as ONE factorisation pass: the cross-kernel rows
        of the query points ride through the blocked Cholesky as bordered rows (``gpx_fit_predict``), so the
        variance solve is part of the trailing updates instead of a pass of its own — the small-N schedule
        (N = 8192: the updates' idle CUs take the work).  The model is fitted afterwards as after ``fit``.
        On a shard or a device group every rank's slice of the query points rides through ITS part of the sharded
        factorisation (collective: every rank of a shard makes the call).  ``dtype="mixed"`` takes the two calls."""
        fused = self.dtype in ("float64", "float32")
        pq, kq, keepq, devq, sq = self._as_input(Xs, "Xs")
        px, kx, keepx, devx, sx = self._as_input(X, "X")
        py, ky, keepy, devy, sy = self._as_input(y, "y")
        if not fused or len(sq) != 2 or not (kx == ky == kq):
            mean, var = self.fit(X, y).predict(Xs, include_noise=include_noise)
            return mean, var
        if len(sx) != 2:
            raise ValueError("X must be (N, d)")
        N, d = sx
        if len(sy) not in (1, 2) or sy[0] != N:
            raise ValueError("y must be (N,) or (N, k) with the same N as X")
        if sq[1] != d:
            raise ValueError(f"Xs must be (M, {d})")
        k = 1 if len(sy) == 1 else sy[1]
        if self.lengthscale.size not in (1, d):
            raise ValueError("lengthscale must be scalar or have d entries")
        M = sq[0]
        self._y1d = len(sy) == 1
        self._N, self._d, self._k = N, d, k
        self._alpha = None
        self._fitted = False
        mshape = (M,) if self._y1d else (M, k)
        if devq is not None:
            import torch
            tdt = torch.float32 if self.dtype == "float32" else torch.float64
            mean = torch.empty(mshape, dtype=tdt, device=devq)
            var = torch.empty((M,), dtype=tdt, device=devq)
            pm, pv = C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr())
        else:
            mean = np.empty(mshape, dtype=self._np_dtype)
            var = np.empty((M,), dtype=self._np_dtype)
            pm, pv = C.c_void_p(mean.ctypes.data), C.c_void_p(var.ctypes.data)
        ls = self.lengthscale
        jitter = self.jitter
        info = C.c_int64(0)
        for _ in range(max(1, self.max_tries)):
            rc = self._lib.gpx_fit_predict(self._h, px, py, N, d, k, _abi.dptr(ls), ls.size, self.variance, self.noise,
                                           jitter, pq, M, pm, pv, kx, C.byref(info))
            if rc in (_abi.E_UNSUPPORTED, _abi.E_NOMEM):
                # the library's own limits decide (its batch cap honours GPX_PRED_BATCH; M more bordered rows of K and
                # of the panel buffers may not fit beside the factor): the documented fallback is the two calls
                return self.fit(X, y).predict(Xs, include_noise=include_noise)
            self._check(rc)
            self.info_ = int(info.value)
            if self.info_ == 0:
                break
            jitter = max(jitter, 1e-12 * self.variance) * 10.0
        else:
            raise np.linalg.LinAlgError(
                f"kernel matrix not positive definite (first bad pivot {self.info_}) after "
                f"{self.max_tries} jitter escalations")
        self.jitter_used_ = jitter
        self._fitted = True
        ld = C.c_double(0.0)
        self._check(self._lib.gpx_logdet(self._h, C.byref(ld)))
        self.log_det_ = float(ld.value)
        if include_noise:
            var += self.noise
        return mean, var

    def predict(self, Xs, return_var=True, include_noise=False):
        if not self._fitted:
            raise RuntimeError("predict() before a successful fit()")
        pq, kq, keepq, devq, sq = self._as_input(Xs, "Xs")
        if len(sq) != 2 or sq[1] != self._d:
            raise ValueError(f"Xs must be (M, {self._d})")
        M = sq[0]
        mshape = (M,) if self._y1d else (M, self._k)
        if devq is not None:
            import torch
            tdt = torch.float32 if self.dtype == "float32" else torch.float64
            mean = torch.empty(mshape, dtype=tdt, device=devq)
            var = torch.empty((M,), dtype=tdt, device=devq) if return_var else None
            pm = C.c_void_p(mean.data_ptr())
            pv = C.c_void_p(var.data_ptr()) if return_var else None
        else:
            mean = np.empty(mshape, dtype=self._np_dtype)
            var = np.empty((M,), dtype=self._np_dtype) if return_var else None
            pm = C.c_void_p(mean.ctypes.data)
            pv = C.c_void_p(var.ctypes.data) if return_var else None
        self._check(self._lib.gpx_predict(self._h, pq, M, pm, pv, kq))
        if not return_var:
            return mean
        if include_noise:
            var += self.noise
        return mean, var

    # -- checkpoint / resume (optional get_state) ------------------------------
    def get_state(self):
        """Plain-data description of the model (JSON-serialisable): kernel, hyper-parameters,
        dtype, panel width and, once fitted, the jitter that was needed and the log-determinant.
        The training data and the factor are NOT included — the factor is N^2 numbers the GPU
        rebuilds faster than storage returns them, and the path is deterministic:
        ``GP.from_state(s).fit(X, y)`` reproduces alpha, mean and variance bit for bit (tested),
        so a hyper-parameter search is resumed by saving this after every ``optimize`` step."""
        st = {"format": 1, "kernel": self.kernel, "lengthscale": [float(v) for v in self.lengthscale],
              "variance": self.variance, "noise": self.noise, "jitter": self.jitter, "dtype": self.dtype,
              "block": self.block, "max_tries": self.max_tries, "refine": self.refine}
        if self._fitted:
            st["fitted"] = {"N": int(self._N), "d": int(self._d), "k": int(self._k),
                            "jitter_used": float(self.jitter_used_), "log_det": float(self.log_det_)}
        return st

    @classmethod
    def from_state(cls, state, **overrides):
        """A fresh, unfitted model from :meth:`get_state` output; ``overrides`` are passed to the
        constructor (``device=``, ``devices=``, ``profile=`` ... are not part of the state)."""
        if state.get("format") != 1:
            raise ValueError("not a GP state of this library (format 1)")
        kw = {k: state[k] for k in ("kernel", "variance", "noise", "jitter", "dtype", "block", "max_tries")}
        kw["refine"] = state.get("refine", 0)
        ls = state["lengthscale"]
        kw["lengthscale"] = ls[0] if len(ls) == 1 else ls
        kw.update(overrides)
        return cls(**kw)

    def release_scratch(self):
        """Free the device buffers only the next ``predict`` / ``lml_gradient`` would use (V^T batch,
        L^-T, partial sums); the fit stays valid.  Buffers otherwise stay allocated for reuse."""
        self._check(self._lib.gpx_release_scratch(self._h))

    @property
    def alpha_(self):
        if not self._fitted:
            raise RuntimeError("no fit")
        if self._alpha is None:
            out = np.empty((self._N, self._k), dtype=self._np_dtype)
            self._check(self._lib.gpx_get_alpha(self._h, C.c_void_p(out.ctypes.data)))
            self._alpha = out[:, 0].copy() if self._y1d else out
        return self._alpha

    def set_profile(self, on):
        """Switch the per-launch timing of the Cholesky sub-phases (``profile=``) on an existing model."""
        self._check(self._lib.gpx_set_flags(self._h, _abi.FLAG_PROFILE if on else 0))

    @property
    def timings_(self):
        t = _abi.GpxTimings()
        self._check(self._lib.gpx_get_timings(self._h, C.byref(t)))
        return t.as_dict()

    def log_marginal_likelihood(self, y):
        """-1/2 y^T alpha - 1/2 logdet - N/2 log(2 pi), summed over target columns."""
        if _is_torch(y):
            y = y.detach().cpu().numpy()
        Y = np.asarray(y, dtype=np.float64).reshape(self._N, -1)
        A = self.alpha_.reshape(self._N, -1).astype(np.float64)
        n, k = Y.shape
        return float(-0.5 * np.sum(Y * A) - 0.5 * k * self.log_det_
                     - 0.5 * n * k * np.log(2.0 * np.pi))

    def lml_gradient(self):
        """``(lml, grad)`` of the last ``fit``: the log marginal likelihood and its analytic
        gradient w.r.t. the LOG hyper-parameters, ordered (lengthscale[0..n_ls), variance, noise)
        — R&W eq. 5.9, 1/2 tr((alpha alpha^T - K^-1) dK/dtheta), computed on the GPU by
        ``gpx_lml_grad`` (about two more factorisations' worth of MFMA work: L^-T, then K^-1
        formed and consumed tile by tile, never stored).  fp64 models.  Sharded ones (``devices=`` or
        ``world=``): with the replicated factor L^-T is built in row blocks dealt over the GPUs,
        all-gathered once, and the trace pass is split over the GPUs; when the factor is only held
        distributed (C4-sized problems) L^-T is built distributed, every GPU keeps the columns of its
        own row blocks and contracts the trace over them — either way every rank returns the same
        numbers."""
        if not self._fitted:
            raise RuntimeError("lml_gradient() before a successful fit()")
        lml = C.c_double(0.0)
        grad = np.empty(self.lengthscale.size + 2, dtype=np.float64)
        self._check(self._lib.gpx_lml_grad(self._h, C.byref(lml), _abi.dptr(grad)))
        return float(lml.value), grad

    def optimize(self, X, y, params=("lengthscale", "variance", "noise"), bounds=(1e-4, 1e4), maxiter=40,
                 rel_step=1e-4, jac="analytic"):
        """Fit the hyper-parameters by maximising the log marginal likelihood (SURVEY.md §8f
        rank 1: the natural step after ``fit``; the reference has no counterpart).

        ``params`` chooses what moves ("lengthscale" moves every ARD entry); the search runs in
        log-space with L-BFGS-B.  ``jac="analytic"`` (default): every evaluation is one ``fit()``
        plus one ``lml_gradient()`` on the GPU, about three factorisations' worth of work whatever
        the number of parameters.  Models without the analytic gradient (float32 / mixed) fall back
        to ``jac="3-point"`` central differences: 2 p extra fits per gradient.
        Non-positive-definite trial points count as very bad, they do not raise.  Leaves the
        model fitted at the best point found and returns scipy's result (``.fun`` = minus the
        log marginal likelihood there)."""
        from scipy.optimize import minimize
        names = [p for p in ("lengthscale", "variance", "noise") if p in params]
        if not names or len(names) != len(tuple(params)):
            raise ValueError("params must be a non-empty subset of lengthscale / variance / noise")
        n_ls = self.lengthscale.size

        def pack():
            v = []
            for p in names:
                v.extend(np.log(self.lengthscale) if p == "lengthscale" else
                         [np.log(max(getattr(self, p), bounds[0]))])
            return np.asarray(v, dtype=np.float64)

        def unpack(v):
            i = 0
            for p in names:
                if p == "lengthscale":
                    self.lengthscale = np.exp(v[i:i + n_ls])
                    i += n_ls
                else:
                    setattr(self, p, float(np.exp(v[i])))
                    i += 1

        best = {"f": np.inf, "v": pack()}
        analytic = jac == "analytic" and self.dtype == "float64"
        if analytic and (self.world > 1 or self._is_group or self._host_comm is not None):
            # sharded: the gradient needs the replicated-factor mode, which the library picks from N
            # and the card's memory at fit time — ask it once (every rank gets the same answer)
            try:
                self.fit(X, y)
                self.lml_gradient()
            except _abi.GpxError:
                analytic = False
            except np.linalg.LinAlgError:
                pass
        # columns of the full gradient (lengthscale.., variance, noise) that move
        cols = []
        for p in names:
            cols.extend(range(n_ls) if p == "lengthscale" else [n_ls + (0 if p == "variance" else 1)])

        class _NoAnalyticGradient(Exception):
            pass

        def objective(v, analytic):
            unpack(v)
            g = np.zeros(len(cols))
            try:
                self.fit(X, y)
                if analytic:
                    try:
                        lml, full = self.lml_gradient()
                    except _abi.GpxError as e:
                        # no gradient for this model after all (factor only held distributed, or no
                        # room for the N x N L^-T buffer): free what the attempt allocated and let the
                        # caller restart the search with central differences
                        if e.code not in (_abi.E_UNSUPPORTED, _abi.E_NOMEM):
                            raise
                        self.release_scratch()
                        raise _NoAnalyticGradient() from e
                    f, g = -lml, -full[cols]
                else:
                    f = -self.log_marginal_likelihood(y)
            except np.linalg.LinAlgError:
                f = 1e300
            if not np.isfinite(f) or not np.all(np.isfinite(g)):
                f, g = 1e300, np.zeros(len(cols))
            if f < best["f"]:
                best["f"], best["v"] = f, np.array(v, copy=True)
            return (f, g) if analytic else f

        v0 = pack()
        lo, hi = np.log(bounds[0]), np.log(bounds[1])

        def search(start, analytic):
            return minimize(objective, start, args=(analytic,), method="L-BFGS-B",
                            jac=True if analytic else "3-point", bounds=[(lo, hi)] * v0.size,
                            options={"maxiter": int(maxiter)} if analytic else
                            {"maxiter": int(maxiter), "eps": float(rel_step)})
        try:
            res = search(v0, analytic)
        except _NoAnalyticGradient:
            res = search(best["v"], False)   # from the best point the analytic steps reached
        unpack(best["v"])
        self.fit(X, y)
        res.x, res.fun = best["v"], best["f"]
        return res
