"""CPU: the C-ABI library builds, loads and exports every symbol include/gpx.h declares;
the Python host validates its arguments; no compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_abi.SIGNATURES)


def test_library_exports_every_declared_symbol(gpx):
    for name in declared_symbols():
        assert hasattr(gpx, name), f"libgpx.so does not export {name}"
    assert gpx.gpx_abi_version() == _abi.ABI_VERSION


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.GpxConfig) == 80
    assert C.sizeof(_abi.GpxTimings) == 8 * 29
    text = open(os.path.join(ROOT, "include", "gpx.h")).read()
    body = text[text.index("typedef struct gpx_timings {"):text.index("} gpx_timings;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:double|int64_t)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in _abi.GpxTimings._fields_]


def test_no_product_import_of_oracle():
    pkg = os.path.join(ROOT, "gaussianprocesspathmodelling_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_create_without_gpu_fails_loudly(gpx):
    n = C.c_int(-1)
    gpx.gpx_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible; the no-device error path is for the CPU container")
    from gaussianprocesspathmodelling_amd import GP, GpxError
    with pytest.raises(GpxError, match="no HIP device"):
        GP()


def test_gp_argument_validation():
    from gaussianprocesspathmodelling_amd import GP
    with pytest.raises(ValueError):
        GP(kernel="linear")
    with pytest.raises(ValueError):
        GP(lengthscale=-1.0)
    with pytest.raises(ValueError):
        GP(variance=0.0)
    with pytest.raises(ValueError):
        GP(dtype="float16")


def test_null_and_bad_arguments_return_codes(gpx):
    assert gpx.gpx_create(None, None) == -1
    assert b"null" in gpx.gpx_last_error(None)
    assert gpx.gpx_device_count(None) == -1
    assert gpx.gpx_gemm_nt(None, 128, 128, None, None, 16, 0) == -1
    a = np.zeros((64, 64))
    info = C.c_int64(0)
    assert gpx.gpx_potrf(_abi.dptr(a), 63, 0, C.byref(info)) == -1   # n not a multiple of 64
    assert gpx.gpx_trsm(_abi.dptr(a), 64, _abi.dptr(a), 60) == -1
