#!/usr/bin/env python3
"""GPU: the 128-wide diagonal step alone (gpx_potrf with n = block = 128 is exactly one potf2_128_kernel
launch): L11 / L21 / L22 against numpy, poisoned upper triangle."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianprocesspathmodelling_amd import _abi
lib = _abi.load()
rng = np.random.default_rng(1)
for n, block in ((128, 128), (256, 256), (192, 256), (1024, 0)):
    B = rng.standard_normal((n, n))
    K = B @ B.T + n * np.eye(n)
    A = np.tril(K) + np.triu(np.full((n, n), 777.0), 1)
    info = C.c_int64(-1)
    rc = lib.gpx_potrf(_abi.dptr(A), n, block, C.byref(info))
    L = np.linalg.cholesky(K)
    E = np.abs(np.tril(A) - L)
    print(f"n={n} block={block} rc={rc} info={info.value} max err {E.max():.2e}", flush=True)
    if info.value != 0 or not E.max() < 1e-9:
        for bi in range(n // 64):
            print("  row block", bi, " ".join(f"{E[bi*64:(bi+1)*64, bj*64:(bj+1)*64].max():.1e}" for bj in range(bi + 1)))
        sys.exit(1)
