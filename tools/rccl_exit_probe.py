#!/usr/bin/env python3
"""Which combination of communicator styles makes the process abort at exit ("double free or corruption")?
    python tools/rccl_exit_probe.py <steps>     steps: comma list of A (InitAll one-rank group), R (InitRank
    one-rank shard), L (LOCAL 2-rank group on one GPU), D (delay hook on around an A step)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussianprocesspathmodelling_amd import GP, _abi
from oracle.gp_oracle import synthetic_problem
X, y, Xs = synthetic_problem(1500, 3, 50, seed=5)
for s in sys.argv[1].split(","):
    if s == "T":
        import torch
        print("T ok", torch.__version__, flush=True)
        continue
    if s == "TD":
        import torch.distributed
        print("TD ok", flush=True)
        continue
    if s == "TC":
        import torch
        print("TC ok", torch.cuda.is_available(), flush=True)
        continue
    if s == "A":
        kw = {"devices": [0], "transport": "rccl"}
    elif s == "R":
        kw = {"device": 0, "world": 1, "rank": 0, "comm": "rccl"}
    elif s == "L":
        kw = {"devices": [0, 0], "transport": "local"}
    elif s == "D":
        _abi.load().gpx_debug_set_delay(7)
        kw = {"devices": [0], "transport": "rccl"}
    with GP("rbf", 0.25, 1.5, 1e-2, jitter=0.0, **kw) as gp:
        m, v = gp.fit(X, y).predict(Xs)
    _abi.load().gpx_debug_set_delay(0)
    print(s, "ok", float(m[0]), flush=True)
print("exiting", flush=True)
