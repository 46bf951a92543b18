#!/usr/bin/env python3
"""A/B of the kernel build's store policy at N = 65536 (non-temporal vs plain 16-byte stores): the shipped
library against tools/_bw/libgpx_kbplain.so (same sources, -DGPX_KBUILD_PLAIN), alternating processes."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json
sys.path.insert(0, %r)
from gaussianprocesspathmodelling_amd import _abi
if %r: _abi.LIB_PATH = %r
import torch
from bench import synthetic
from gaussianprocesspathmodelling_amd import GP
X, y, Xs = (torch.from_numpy(v).cuda() for v in synthetic(65536, 3, 4096, 12345))
with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0) as gp:
    ks = []
    for _ in range(4):
        gp.fit(X, y); ks.append(gp.timings_["kbuild"])
    print(json.dumps({"kbuild_ms": ks[1:], "bytes": gp.timings_["kbuild_bytes"]}))
'''
out = {}
for rnd in range(2):
    for name, lib in (("nt", ""), ("plain", os.path.join(ROOT, "tools", "_bw", "libgpx_kbplain.so"))):
        r = subprocess.run([sys.executable, "-c", CODE % (ROOT, bool(lib), lib)], capture_output=True, text=True)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        out.setdefault(name, []).extend(rec["kbuild_ms"])
        gbs = rec["bytes"] / (min(rec["kbuild_ms"]) * 1e-3) / 1e9
        print(name, [round(v, 3) for v in rec["kbuild_ms"]], f"best {gbs:.0f} GB/s", flush=True)
print(json.dumps({k: {"min_ms": min(v), "mean_ms": sum(v) / len(v)} for k, v in out.items()}))
