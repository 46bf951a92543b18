"""CPU: sanitizer builds of the library's HOST side (SURVEY.md §5 "race detection / sanitizers").

libgpx.so is compiled twice more with the host half instrumented (`hipcc -Xarch_host -fsanitize=...`; the device code
is compiled as always — GPU AddressSanitizer is not available on this pool, sanitizers run on the CPU build only):

  * address + undefined behaviour: tests/host_san/driver.cpp walks everything that runs without a GPU — the rank
    threads' rendezvous (LocalHub barrier rounds, a rank that aborts), the host replay of the kernels' tile maps, the
    argument validation of every entry point, gpx_create's failure paths;
  * thread: the rendezvous rounds again (the LocalHub is what the rank threads of a device group meet at).

A finding aborts the driver (non-zero exit); its own expectations count failed calls.  The instrumented objects live
under build/ (git-ignored) and are rebuilt only when a source is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianprocesspathmodelling_amd import build as B  # noqa: E402

DRIVER = os.path.join(ROOT, "tests", "host_san", "driver.cpp")


def build_variant(name, san):
    out = os.path.join(ROOT, "build", f"san_{name}")
    os.makedirs(out, exist_ok=True)
    hipcc = B._hipcc()
    deps = [os.path.join(B.CSRC, f) for f in B.SOURCES + B.HEADERS] + [DRIVER, os.path.abspath(__file__)]
    exe = os.path.join(out, "driver")
    if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(d) for d in deps):
        return exe
    host = []
    for f in (f"-fsanitize={san}", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all"):
        host += ["-Xarch_host", f]

    def cc(src):
        obj = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.run([hipcc, "-O1", "-g", f"--offload-arch={B.ARCH}", "-std=c++17", "-fPIC", *host, "-c",
                        os.path.join(B.CSRC, src), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(cc, B.SOURCES))
    lib = os.path.join(out, "libgpx.so")
    subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", f"-fsanitize={san}", "-o", lib] + objs, check=True)
    subprocess.run([hipcc, "-O1", "-g", "-std=c++17", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-x", "c++", DRIVER,
                    "-x", "none", "-o", exe, f"-L{out}", "-lgpx", f"-Wl,-rpath,{out}", "-lpthread"], check=True)
    return exe


def run_driver(exe, arg, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([exe, arg], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "0 failed expectation(s)" in r.stdout, r.stdout
    return r


def test_host_side_under_address_and_undefined_behaviour_sanitizers():
    exe = build_variant("asan", "address,undefined")
    # leaks: the HIP runtime keeps process-lifetime allocations of its own; everything else is checked
    run_driver(exe, "asan", {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1"})


def test_rank_thread_rendezvous_under_thread_sanitizer():
    exe = build_variant("tsan", "thread")
    run_driver(exe, "tsan", {"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"})
