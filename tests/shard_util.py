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
    """Launch `world` worker processes on 127.0.0.1 and return their result dicts."""
    port = free_port()
    out = os.path.join(str(tmp_path), f"{mode}_w{world}")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), mode, out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o.decode(errors="replace"))
    finally:
        for p in procs:          # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.kill()
    for rank, p in enumerate(procs):
        assert p.returncode == 0, f"rank {rank} failed:\n{logs[rank][-3000:]}"
    return [dict(np.load(out + f".rank{r}.npz", allow_pickle=False)) for r in range(world)]
