import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gaussianprocesspathmodelling_amd import _abi
lib = _abi.load()
a, b = C.c_double(0), C.c_double(0)
lib.gpx_microbench(C.byref(a), C.byref(b))
print("blocks", os.environ.get("GPX_MB_BLOCKS"), "mfma TF", round(a.value, 2), "copy GB/s", round(b.value))
