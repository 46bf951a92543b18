import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(mode, world, tmp_path, env_extra=None, timeout=300):
    """Launch `world` worker processes on 127.0.0.1 and return their result dicts.  Worker output
    goes to files (a full pipe would block a rank inside a collective); a rank that dies takes the
    others down at once instead of leaving them waiting for it in a collective until the timeout.
    The rendezvous port is probed, then used: if somebody took it in between (EADDRINUSE in a rank's log — seen once
    on a shared box) the launch is repeated on another port, at most three times."""
    last = None
    for attempt in range(3):
        try:
            return _run_ranks_once(mode, world, tmp_path, env_extra, timeout, attempt)
        except AssertionError as e:
            last = e
            if "EADDRINUSE" not in str(e):
                raise
    raise last


def _run_ranks_once(mode, world, tmp_path, env_extra, timeout, attempt):
    import time
    port = free_port()
    out = os.path.join(str(tmp_path), f"{mode}_w{world}" + (f"_try{attempt}" if attempt else ""))
    procs, logf = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **(env_extra or {}))
        logf.append(open(out + f".rank{rank}.log", "wb"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), mode, out],
                                      env=env, stdout=logf[-1], stderr=subprocess.STDOUT))
    deadline = time.monotonic() + timeout
    why = ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes):
                why = "a rank failed; the others were stopped"
                break
            if time.monotonic() > deadline:
                why = f"timeout after {timeout} s"
                break
            time.sleep(0.05)
    finally:
        for p in procs:          # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.kill()
                p.wait()
        for f in logf:
            f.close()
    logs = [open(out + f".rank{r}.log", "rb").read().decode(errors="replace") for r in range(world)]
    for rank, p in enumerate(procs):
        assert p.returncode == 0 and not why, (f"rank {rank} exit code {p.returncode} ({why})\n" +
                                               "\n".join(f"--- rank {r} ---\n{logs[r][-3000:]}" for r in range(world)))
    return [dict(np.load(out + f".rank{r}.npz", allow_pickle=False)) for r in range(world)]
