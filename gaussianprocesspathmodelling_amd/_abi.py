"""ctypes binding of include/gpx.h — the only way the Python host reaches the GPU.

There is deliberately no fallback: if ``csrc/libgpx.so`` is missing or does not load,
``load()`` raises.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgpx.so")
ABI_VERSION = 5

KERNEL_IDS = {"rbf": 0, "matern52": 1}
DTYPE_IDS = {"float64": 0, "float32": 1, "mixed": 2}
MEM_HOST, MEM_DEVICE = 0, 1
E_ARG, E_HIP, E_COMM, E_UNSUPPORTED, E_NOMEM = -1, -2, -3, -4, -5
FLAG_PROFILE = 1
TRANSPORT_IDS = {None: 0, "auto": 0, "rccl": 1, "local": 2}
MAX_GROUP = 8


class GpxConfig(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("dtype", C.c_int32), ("device", C.c_int32),
                ("block", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("flags", C.c_int32), ("ndev", C.c_int32), ("devices", C.c_int32 * MAX_GROUP),
                ("transport", C.c_int32), ("refine", C.c_int32), ("reserved", C.c_int32 * 2)]


class GpxTimings(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("h2d", "kbuild", "chol", "solve", "logdet", "fit_total",
                 "kstar", "mean", "trsm", "var", "d2h", "predict_total",
                 "comm", "chol_diag", "chol_trsm", "chol_strip", "chol_syrk", "syrk_flops")] + \
               [("syrk_launches", C.c_int64), ("kbuild_bytes", C.c_double)] + \
               [(n, C.c_double) for n in ("grad_trtri", "grad_trace", "grad_total", "refine", "refine_resid0",
                                          "refine_resid", "refine_iters", "handover_flags", "handover_retries")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# gpx_host_comm: host-buffer collectives supplied by the caller (tests / non-RCCL fabrics)
BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class GpxHostComm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("bcast", BCAST_FN), ("allgather", ALLGATHER_FN),
                ("reduce", REDUCE_FN), ("allreduce", ALLREDUCE_FN)]


# every symbol include/gpx.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PD = C.POINTER(C.c_double)
SIGNATURES = {
    "gpx_abi_version": (C.c_int, []),
    "gpx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gpx_create": (C.c_int, [C.POINTER(_P), C.POINTER(GpxConfig)]),
    "gpx_destroy": (None, [_P]),
    "gpx_last_error": (C.c_char_p, [_P]),
    "gpx_fit": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _PD, C.c_int32,
                          C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_int64)]),
    "gpx_predict": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32]),
    "gpx_fit_predict": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _PD, C.c_int32, C.c_double, C.c_double,
                                  C.c_double, _P, C.c_int64, _P, _P, C.c_int32, C.POINTER(C.c_int64)]),
    "gpx_get_alpha": (C.c_int, [_P, _P]),
    "gpx_lml_grad": (C.c_int, [_P, _PD, _PD]),
    "gpx_logdet": (C.c_int, [_P, _PD]),
    "gpx_release_scratch": (C.c_int, [_P]),
    "gpx_get_timings": (C.c_int, [_P, C.POINTER(GpxTimings)]),
    "gpx_set_flags": (C.c_int, [_P, C.c_int32]),
    "gpx_comm_unique_id": (C.c_int, [_P]),
    "gpx_comm_init": (C.c_int, [_P, _P]),
    "gpx_comm_init_host": (C.c_int, [_P, C.POINTER(GpxHostComm)]),
    "gpx_path_distance": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, _P, C.c_int32]),
    "gpx_kernel_matrix": (C.c_int, [C.c_int32, _PD, C.c_int64, _PD, C.c_int64, C.c_int32, _PD,
                                    C.c_int32, C.c_double, C.c_double, _PD]),
    "gpx_potrf": (C.c_int, [_PD, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "gpx_trsm": (C.c_int, [_PD, C.c_int64, _PD, C.c_int64]),
    "gpx_gemm_nt": (C.c_int, [_PD, C.c_int64, C.c_int64, _PD, _PD, C.c_int64, C.c_int32]),
    "gpx_debug_tile_map": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]),
    "gpx_debug_stair_map": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]),
    "gpx_debug_deal": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64)]),
    "gpx_debug_local_hub": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "gpx_debug_set_delay": (C.c_int, [C.c_uint64]),
    "gpx_debug_gemm_bench": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_double)]),
    "gpx_mfma_probe": (C.c_int, [_PD, _PD, _PD]),
    "gpx_mfma_probe_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpx_microbench": (C.c_int, [_PD, _PD]),
}

_lib = None


class GpxError(RuntimeError):
    """Non-zero return code from libgpx (API misuse, HIP or RCCL failure)."""

    def __init__(self, code, message):
        super().__init__(f"libgpx error {code}: {message}")
        self.code = code


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so / libhsa-runtime64.so (SONAME libamdhip64.so.7, loaded by file
    name); a second copy from /opt/rocm in the same process cannot see the GPU
    ("No HIP GPUs are available").  When torch is installed, map ITS runtime first:
    libgpx's DT_NEEDED libamdhip64.so.7 then binds to that object by SONAME, and a
    later (or earlier) `import torch` finds the same file already mapped.  torch itself
    is not imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def prefer_torch_rccl():
    """One RCCL per process, loaded in torch's own order.  PyTorch-ROCm wheels bundle their own
    librccl.so (and the runtime libraries it needs) and map them with ``import torch``.  Measured on
    this image (round 3, tools/rccl_exit_probe.py): a process in which libgpx dlopen()s an RCCL —
    /opt/rocm's or torch's own file — BEFORE torch is imported aborts at exit ("double free or
    corruption" / "free(): invalid pointer" in the static destructors); torch first, then libgpx's
    RCCL use, is clean.  So before libgpx's first RCCL use: import torch if it is installed (it is the
    declared tensor container of this package anyway) and point GPX_RCCL_PATH at its copy, which
    libgpx then finds already mapped.  Without torch there is nothing to collide with."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return os.environ.get("GPX_RCCL_PATH")
    import torch  # noqa: F401  (maps torch's RCCL and its dependencies in torch's own order)
    if "GPX_RCCL_PATH" not in os.environ:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(p):
            os.environ["GPX_RCCL_PATH"] = p
    return os.environ.get("GPX_RCCL_PATH")


def load():
    """dlopen csrc/libgpx.so and attach signatures.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gaussianprocesspathmodelling_amd.build` "
            "(hipcc, gfx950). The GP engine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    v = lib.gpx_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"libgpx ABI version {v} != expected {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def dptr(arr):
    """numpy float64 array -> POINTER(c_double)"""
    return arr.ctypes.data_as(_PD)
