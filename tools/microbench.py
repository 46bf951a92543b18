import ctypes as C, sys
sys.path.insert(0,'.')
from gaussianprocesspathmodelling_amd import _abi
lib=_abi.load()
a,b=C.c_double(0),C.c_double(0)
print(lib.gpx_microbench(C.byref(a),C.byref(b)), a.value, b.value)
