"""CPU: bench.py's launch logic without a GPU.  `python bench.py --gpus N` with no WORLD_SIZE in the
environment must start the one-process-per-GPU launcher itself, as a child, instead of refusing; on this GPU-less box the
ranks then stop at the first line that needs the card ("no HIP device visible": there is no CPU path) and the parent leaves
with a non-zero code.  The success flow of the same launch is rehearsed on a one-GPU box by tools/bench_rehearsal.sh."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_launch_starts_the_launcher():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the bare launch is rehearsed by tools/bench_rehearsal.sh")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--mode", "shard", "--backend", "gloo", "--device", "0", "--ntrain", "1024"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "without a launcher: starting -m torch.distributed.run" in r.stderr
    assert "--nproc-per-node=2" in r.stderr
    assert "no HIP device visible" in r.stderr or "needs an MI355X" in r.stderr


def test_launcher_and_gpus_must_agree():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "--nproc-per-node must equal --gpus" in (r.stderr + r.stdout)
