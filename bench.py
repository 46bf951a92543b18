#!/usr/bin/env python3
"""bench.py — GP fit+predict throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ``fit(X, y)``
(kernel build, blocked Cholesky, alpha solves) followed by ``predict(Xs)`` (cross
kernel, mean, variance TRSM) on BASELINE.json configs[2] — N=65536, d=3, RBF, fp64,
M=4096 (M fixed by SURVEY.md §8) — with X, y, Xs already resident in HBM (torch CUDA
tensors are only the containers; all arithmetic is libgpx.so).

N > 1: one process per GPU, launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
(the driver's contract) — or by ``python bench.py --gpus N`` alone: without WORLD_SIZE in the environment this
file starts that launcher itself, as a child (``launch_ranks``; the parent touches no GPU and exits with its code).
  --mode auto (default) = the graded multi-GPU path of SURVEY.md §8(e): ONE N=65536 Gram
      matrix in row-block-cyclic shards over the ranks, panel broadcast / all-gather over
      RCCL with one-panel look-ahead, value = (N+M) / max-over-ranks time, ``"scaling":
      "strong"``.  The sharded run lives in a CHILD process per rank (this process touches
      no GPU meanwhile) that writes a heartbeat and checks its posterior against the
      single-GPU path on rank 0 (``shard_check``).  A child that fails, stalls or disagrees fails
      that attempt.  The sequence is shard -> group -> fail: after a failed ``shard`` attempt the SAME
      schedule (same kernels, panels, metric) is tried once more over its other transport
      (``--mode group``: rank 0's child drives all N GPUs as one in-process device group over peer
      copies and hipEvents — no RCCL, no inter-process IPC; the other ranks only keep time, so that
      line is clocked on rank 0 alone and says so: ``timed_on``, ``fallback_from``).  If that fails
      too, rank 0 prints the JSON line with ``"value": null`` and both reasons, and every rank exits
      non-zero.  Nothing else (no replicas number) is ever substituted for the graded number.
      NOTE: RCCL with more than one rank, and peer copies between distinct devices, have only ever
      run where the driver runs this file on a multi-GPU node (the development boxes have one GPU);
      the line carries ``"multi_gpu_transport_verified_on_hardware": false`` until a SCALE record exists.
  --mode shard: the sharded run in-process (what the child executes).
  --mode group: rank 0 drives all N GPUs from one process (see above).
      ``--workload C4`` = N=262144, d=3, Matern-5/2 (needs 8 GPUs; no single-GPU check).
  --mode replicas: every rank runs its own replica of the workload (independent GPs, e.g.
      one per path cluster: no data-path collective), value = points of all ranks /
      max-over-ranks time, ``"scaling": "weak"``.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     — the dominant kernel (trailing SYRK of the blocked Cholesky, fp64 MFMA):
                 algorithmic flops n(n+1)nb per launch / hipEvent time per launch, both
                 summed over the launches of that kernel in the timed steps (library events
                 on the library's stream, GPX_FLAG_PROFILE; the last few, under-filled
                 updates of a fit run as 64-tiles — another kernel — and are not counted)
  fit_predict_one_pass — the same step as ONE call (``gp.fit_predict`` -> gpx_fit_predict, ABI v4+: the query rows ride
                 through the factorisation), a few steps after the timed region; reported beside the headline,
                 never ``value``
  cpu_baseline — the NumPy/SciPy oracle (oracle/gp_oracle.py) on the GPU box's host cores.
                 ``value`` is the rate AT THE WORKLOAD (N=65536), measured IN THIS RUN after the GPU timed
                 region (oracle/make_golden_full.py in a child: ~2 min with 16 threads, 35 GB of host memory;
                 ``measured_in_this_run``, ``cores``, ``seconds``).  Hosts with < 40 GB available, or a child that
                 fails / runs out of time, fall back to the committed record of such a run (profiles/, ``source:
                 "committed record"``) beside a bounded live sample (N=24576, ~20 s) with its phase-wise
                 extrapolation.  BLAS threads = the CPUs the process may use (affinity and cgroup quota) = ``cores``
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs[2]; hyper-parameters of SURVEY.md §8(d)
N_TRAIN, DIM, M_TEST = 65536, 3, 4096
KERNEL, LENGTHSCALE, SF2, SN2 = "rbf", 0.25, 1.5, 1e-2
PEAK_FP64_MFMA_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 FLOP/clk (v_mfma_f64_16x16x4: 2048 FLOP / 64 clk) x 2.4 GHz
PEAK_FP32_MFMA_TFLOPS = 157.3  # v_mfma_f32_16x16x4: 2048 FLOP / 32 clk
PEAK_HBM_GBS = 8000.0


def synthetic(N, d, M, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.0, 1.0, (N, d))
    Xs = rng.uniform(0.0, 1.0, (M, d))
    y = (np.sin(2.0 * np.pi * X[:, 0]) + 0.5 * np.cos(3.0 * X[:, 1:].sum(axis=1))
         + 0.1 * rng.standard_normal(N))
    return X, y, Xs


DEALING = "cyclic" if os.environ.get("GPX_SHARD_DEAL", "snake") in ("cyclic", "0") else "snake"   # gpx_internal.h: Deal


def shard_block(N, world):
    """Row-block height the library picks for the shard (gpx_shard.inc: >= 8 blocks per rank under the snake dealing of
    round 4, >= 16 under GPX_SHARD_DEAL=cyclic)."""
    nb = 1024
    per_rank = 16 if DEALING == "cyclic" else 8
    while nb > 256 and nb * per_rank * world > N:
        nb //= 2
    return int(os.environ.get("GPX_NB_SHARD", nb))


def pmc_traffic():
    """HBM bytes per SYRK launch from the committed rocprofv3 PMC passes of this same command
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs; KiB units; FETCH_SIZE x2 on gfx950 for the
    16-B/lane streaming reads — MI355X_MICROARCH.md §HBM).  None if no summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_kernel.csv")))
    if not files:
        return None, None
    fetch = write = None
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if "gemm_nt_kernel<double, 128, true, 0" in row["Kernel_Name"]:
                if row["Counter_Name"] == "FETCH_SIZE":
                    fetch = float(row["mean"])
                elif row["Counter_Name"] == "WRITE_SIZE":
                    write = float(row["mean"])
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.basename(files[-1])


def cpu_budget():
    """CPUs this process may actually use: min(scheduler affinity, cgroup CPU quota)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:                                        # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    return aff, quota, int(max(1, min(aff, quota if quota else aff)))


def full_size_oracle_record():
    """The measured FULL-size oracle run on a GPU box's host (oracle/make_golden_full.py --config C3
    prints it; committed under profiles/ next to the golden fixture it produced)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_full_oracle_c3.json")), reverse=True) + \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_full_oracle_parity.json")), reverse=True)
    for f in cands:
        try:
            with open(f) as fh:
                rec = json.load(fh)
            if "oracle_points_per_s" in rec:
                return {"file": "profiles/" + os.path.basename(f), "points_per_s": rec["oracle_points_per_s"],
                        "seconds": rec.get("oracle_fit_predict_s"), "threads": rec.get("blas_threads"),
                        "cholesky_s": rec.get("oracle_cholesky_s")}
        except Exception:
            pass
    return None


def host_memory_available_gb():
    """What this process may still allocate on the host: min(MemAvailable, cgroup limit - usage), GB."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = float(line.split()[1]) * 1024.0 / 1e9
    except OSError:
        pass
    try:                                            # cgroup v2
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            room = (float(lim) - float(open("/sys/fs/cgroup/memory.current").read())) / 1e9
            avail = room if avail is None else min(avail, room)
    except (OSError, ValueError):
        pass
    return avail


def cpu_full_size_live(timeout_s):
    """The oracle at the workload's own size, now, on this host (34.4 GB in place, about 2 minutes with 16 threads),
    in a child process (exact PID kept; killed at `timeout_s`).  Returns (record | None, why-not | None)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"gpx_g5_{os.getpid()}.npz")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden_full.py"), "--config", "C3",
                            "--out", out], capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return None, f"full-size oracle child exited with code {r.returncode}: {r.stderr.strip()[-200:]}"
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        return {"file": "measured in this run", "points_per_s": rec["oracle_points_per_s"],
                "seconds": rec["oracle_fit_predict_s"], "threads": rec["blas_threads"],
                "cholesky_s": rec["oracle_cholesky_s"], "kbuild_s": rec["oracle_kbuild_s"],
                "predict_s": rec["oracle_predict_s"], "wall_s_incl_start": time.perf_counter() - t0}, None
    except subprocess.TimeoutExpired:
        return None, f"full-size oracle child did not finish within {timeout_s:.0f} s (killed)"
    except Exception as e:                           # no JSON line, ...
        return None, f"full-size oracle child: {type(e).__name__}: {e}"
    finally:
        if os.path.exists(out):
            os.remove(out)


def cpu_sample(n_sample, threads):
    """The bounded live sample: the oracle at N = n_sample (same generator), phase-wise extrapolated to the workload."""
    from threadpoolctl import threadpool_limits
    from oracle.gp_oracle import OracleGP
    X, y, Xs = synthetic(n_sample, DIM, M_TEST, 12345)
    with threadpool_limits(limits=threads, user_api="blas"):
        t0 = time.perf_counter()
        gp = OracleGP(KERNEL, LENGTHSCALE, SF2, SN2, jitter=0.0, chol="blocked").fit(X, y)
        t1 = time.perf_counter()
        gp.predict(Xs)
        t2 = time.perf_counter()
    tm = gp.timings_
    r = N_TRAIN / n_sample
    # extrapolate each phase to N=65536 by its algorithmic work
    est = (tm["kbuild"] * r ** 2 + tm["chol"] * r ** 3 + tm["solve"] * r ** 2
           + (tm["kstar"] + tm["mean"]) * r + (tm["trsm"] + tm["var"]) * r ** 2) * 1e-3
    return {"N": n_sample, "points_per_s": (n_sample + M_TEST) / (t2 - t0), "fit_s": t1 - t0, "predict_s": t2 - t1,
            "seconds": t2 - t0, "cholesky_gflops": n_sample ** 3 / 3.0 / (tm["chol"] * 1e-3) / 1e9, "threads": threads,
            "extrapolated_seconds_at_workload": est,
            "extrapolated_points_per_s_at_workload": (N_TRAIN + M_TEST) / est}


def cpu_baseline(mode="auto", n_sample=24576, full_timeout_s=330.0):
    """`value` = the oracle's fit+predict rate AT THE WORKLOAD (N=65536) on THIS host's cores.

    mode "auto" (default): measured IN THIS RUN at full size (the GPU side is finished by then; ~2 min with 16
      threads, 35 GB of host memory) whenever the host has >= 40 GB available — `measured_in_this_run: true`;
      otherwise, or if the child fails / exceeds `full_timeout_s`, the committed record of an earlier full-size run
      on a GPU box (profiles/r*_cpu_full_oracle_c3.json; `source: "committed record"`) beside a bounded live
      sample (N=24576, ~20 s) of this run.  "full": the same without the memory check.  "record": the committed
      record + the bounded live sample (the round-3 behaviour).  "sample": the bounded sample alone, extrapolated.

    The BLAS pools are limited to the CPUs this process may really use (affinity, cgroup
    quota): a pool larger than the quota — the default is 64 threads on this pool's hosts
    whatever the box's share — leaves most workers spinning inside a throttled cgroup and the
    LAPACK Cholesky crawls (round 1: 30 GF/s with 64 threads against 27 GF/s with one)."""
    from threadpoolctl import threadpool_info
    aff, quota, usable = cpu_budget()
    pools = threadpool_info()
    blas_max = max((int(p.get("num_threads") or 1) for p in pools if p.get("user_api") == "blas"), default=1)
    threads = max(1, min(usable, blas_max))
    mem_gb = host_memory_available_gb()
    full = why = None
    if mode in ("auto", "full"):
        if mode == "auto" and mem_gb is not None and mem_gb < 40.0:
            why = f"only {mem_gb:.0f} GB of host memory available (the in-place oracle needs 35 GB)"
        else:
            full, why = cpu_full_size_live(full_timeout_s)
    measured_now = full is not None
    live = None
    if not measured_now:                             # the bounded sample: cross-check of a committed record, or the only number
        live = cpu_sample(n_sample, threads)
        if mode != "sample":
            full = full_size_oracle_record()
    if full:
        value, cores = full["points_per_s"], full["threads"] or threads
        where = ("measured in this run on this host, after the GPU timed region" if measured_now else
                 f"measured once at full size on a GPU box's host ({full['file']})")
        how = (f"FULL workload N={N_TRAIN} d={DIM} M={M_TEST} RBF fp64, same generator, {where}, {cores} BLAS threads: "
               f"{full['seconds']:.1f} s per fit+predict, Cholesky {full['cholesky_s']:.1f} s")
        if live:
            how += (f"; this run's bounded live sample (N={n_sample}, {live['seconds']:.1f} s) extrapolates phase-wise to "
                    f"{live['extrapolated_points_per_s_at_workload']:.0f} points/s")
    else:
        value, cores = live["extrapolated_points_per_s_at_workload"], threads
        how = (f"phase-wise EXTRAPOLATION to N={N_TRAIN} of a live N={n_sample} sample ({live['seconds']:.1f} s)")
    return {
        "value": value, "unit": "points/s", "cores": cores, "kind": "port",
        "sample": "oracle/gp_oracle.py (NumPy/SciPy; level-3 blocked Cholesky — LAPACK potrf of the bundled OpenBLAS "
                  "does not parallelise): " + how,
        "at_workload": bool(full), "measured_in_this_run": measured_now,
        "source": "this run" if measured_now else ("committed record" if full else "extrapolated live sample"),
        "seconds": full["seconds"] if full else live["extrapolated_seconds_at_workload"],
        "not_measured_live_because": why,
        "measured_full_size": full, "live_sample": live,
        "host": {"os_cpu_count": os.cpu_count(), "affinity_cpus": aff, "cgroup_cpu_quota": quota,
                 "memory_available_gb": mem_gb,
                 "blas_pools": [f"{p.get('internal_api')} {p.get('version')}: {p.get('num_threads')} threads by default"
                                for p in pools if p.get("user_api") == "blas"],
                 "blas_threads_used": threads},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ntrain", dest="n", type=int, default=N_TRAIN, help="override N (debug only; invalidates the metric)")
    ap.add_argument("--mtest", dest="m", type=int, default=M_TEST)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--mode", choices=["auto", "replicas", "shard", "group"], default="auto",
                    help="auto: N > 1 runs 'shard' (one process per GPU over RCCL) supervised, and if that fails 'group' "
                         "(rank 0 drives all N GPUs as one in-process device group over peer copies: same schedule, no RCCL)")
    ap.add_argument("--heartbeat", default=None, help="(internal) progress file of a supervised sharded child")
    ap.add_argument("--stall-timeout", type=float, default=300.0,
                    help="auto mode: seconds without child progress before the run counts as failed")
    ap.add_argument("--workload", choices=["C2", "C3", "C4", "C5"], default="C3",
                    help="BASELINE.json configs[1..4]: C2 N=8192; C3 N=65536 (the metric's config, default); C4 N=262144 Matern on "
                         "8 GPUs; C5 N=65536 with ARD lengthscales in --dtype float32 (default there) or mixed")
    ap.add_argument("--dtype", choices=["float64", "float32", "mixed"], default=None,
                    help="arithmetic of the path (default float64; C5: float32).  Not float64 = another metric config: "
                         "the line says so in dtype / config")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the multi-rank code on one GPU (host collectives)")
    ap.add_argument("--device", type=int, default=None, help="HIP device override (rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["auto", "full", "record", "sample"], default="auto",
                    help="auto (default): the full-size (N=65536) CPU oracle measured live after the GPU timed region (about 2 "
                         "minutes, 35 GB of host memory) when the host has >= 40 GB available, else the committed record + a "
                         "bounded live sample; full: live without the memory check; record / sample: see cpu_baseline()")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="= --cpu-baseline full (kept from round 3)")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--one-pass", action="store_true",
                    help="process-per-GPU shard, several ranks: also time gp.fit_predict (collective; one rank, groups and the "
                         "unsharded handle always do)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Bare launch (`python bench.py --gpus N`, no launcher): start the one-process-per-GPU launcher ourselves,
        # as a CHILD — this process never touches a GPU — and leave with its exit code.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.mode == "auto":
        if world > 1:
            failure = supervise_sharded_child(args, "shard")
            if failure is None:
                return                      # the child printed the JSON line
            # The RCCL transport failed.  The SAME sharded schedule has a second transport that needs
            # no RCCL and no inter-process IPC: rank 0 drives all GPUs as one in-process device group
            # (peer copies + hipEvents).  It is tried once, labelled as such in the JSON line, with
            # the reason the first attempt failed.
            os.environ["GPX_BENCH_FALLBACK_REASON"] = failure[:500]
            # a fresh rendezvous for the second attempt: the launcher's agent store still holds the
            # first attempt's keys, so the children host their own store on the next port
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            failure2 = supervise_sharded_child(args, "group")
            if failure2 is None:
                return
            failure = f"shard over RCCL: {failure}; in-process device group: {failure2}"
            # Both transports of the graded multi-GPU path failed.  Say so and stop: no replicas
            # number takes the place of `value`, and the exit code is non-zero on every rank.
            if int(os.environ.get("RANK", "0")) == 0:
                print(json.dumps({
                    "metric": "gp_fit_predict_points_per_sec", "value": None, "unit": "points/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                    "data": "synthetic",
                    "config": {"workload": f"{args.workload}: exact GP fit+predict, row-block-cyclic shard over "
                                           f"{world} gpus", "parallelism": f"shard over {world} gpus (FAILED)"},
                    "failed": failure}), flush=True)
            sys.exit(1)
        args.mode = "replicas"
    run(args)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>` as a child
    (exact PID kept; this process initialises no GPU) and return its exit code.  The ranks then take the same
    path as under the driver's own launcher: the JSON line is rank 0's."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL between processes)
    rc = 1
    for attempt in range(2):
        with socket.socket() as s:             # a port nobody listens on right now
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
        t0 = time.time()
        child = subprocess.Popen(cmd, env=env)
        try:
            rc = child.wait()
        except KeyboardInterrupt:
            child.terminate()
            return child.wait()
        # a launch that dies within seconds never got to the GPUs — typically the rendezvous port was taken between the
        # probe and its use: once more on another port, then the code is the answer
        if rc == 0 or time.time() - t0 > 20.0:
            break
    return rc


def beat(args, what):
    """Progress mark of a supervised child (mtime + last line are what the parent reads)."""
    if args.heartbeat:
        with open(args.heartbeat, "a") as f:
            f.write(f"{time.time():.3f} {what}\n")


def supervise_sharded_child(args, mode):
    """Run ``--mode shard`` (or ``group``) in a child process (exact PID kept), watch its heartbeat file and
    the node-wide failure flag.  Returns None when the child finished ("done" mark), else the
    reason it failed.  Nothing here touches the GPU."""
    import subprocess
    import tempfile
    rank = int(os.environ.get("RANK", "0"))
    # All ranks of one launch share the launcher as parent: its pid AND its start time (clock
    # ticks since boot, /proc/<pid>/stat field 22) make the names unique per launch, so no rank
    # ever has to delete a file it did not create (a flag written by a faster rank survives).
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat") as f:
            born = f.read().rsplit(")", 1)[1].split()[19]
    except Exception:
        born = "0"
    tag = f"gpx_bench_{mode}_{os.environ.get('MASTER_PORT', '0')}_{ppid}_{born}"
    tmp = tempfile.gettempdir()
    hb = os.path.join(tmp, f"{tag}_hb{rank}")
    flag = os.path.join(tmp, f"{tag}_failed")
    open(hb, "w").close()
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--mode", mode, "--heartbeat", hb]
    child = subprocess.Popen(cmd)
    reason = None
    last = ""
    while True:
        rc = child.poll()
        try:
            with open(hb) as f:
                lines = f.read().splitlines()
            last = lines[-1].split(" ", 1)[1] if lines else ""
            age = time.time() - os.path.getmtime(hb)
        except OSError:
            age = 0.0
        if last == "done":
            if rc is None:
                try:
                    child.wait(timeout=60)
                except subprocess.TimeoutExpired:
                    child.kill()
            return None
        if rc is not None:
            reason = f"{mode} child of rank {rank} exited with code {rc} after '{last}'"
            break
        if os.path.exists(flag):
            reason = f"{mode} child of another rank failed"
            break
        if age > args.stall_timeout:
            reason = f"{mode} child of rank {rank} made no progress for {age:.0f} s after '{last}'"
            break
        time.sleep(0.5)
    try:
        with open(flag, "a") as f:
            f.write(reason + "\n")
    except OSError:
        pass
    if child.poll() is None:
        child.kill()                        # exact PID of the process started above
        child.wait()
    print(f"[bench] {mode} run FAILED: {reason}", file=sys.stderr, flush=True)
    time.sleep(3.0)                         # let the other ranks see the flag and reap their children
    try:
        with open(flag) as f:
            reason = f.readline().strip() or reason
    except OSError:
        pass
    return reason


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    beat(args, "imports")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Sharded runs: torch.distributed is only the control plane (RCCL unique id, barriers, the
    # max-over-ranks time) and runs on gloo; the data path is libgpx.so calling RCCL itself.
    # Replicas have no data path between ranks: their barriers go over torch's RCCL backend.
    ctrl_nccl = args.backend == "nccl" and args.mode not in ("shard", "group")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ctrl_nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from gaussianprocesspathmodelling_amd import GP, _abi
    beat(args, "process group")

    N, M = args.n, args.m
    kernel = KERNEL
    lengthscale = LENGTHSCALE
    dtype = args.dtype or ("float32" if args.workload == "C5" else "float64")
    if args.workload == "C2" and args.n == N_TRAIN:
        N = 8192
    if args.workload == "C5":
        lengthscale = (0.3, 0.2, 0.25)              # ARD
    tdt = torch.float32 if dtype == "float32" else torch.float64
    peak_mfma = PEAK_FP64_MFMA_TFLOPS if dtype == "float64" else PEAK_FP32_MFMA_TFLOPS   # the factorisation's engine
    if args.workload == "C4":
        if args.n == N_TRAIN:
            N = 262144
        kernel = "matern52"
        if not (args.mode in ("shard", "group") and world >= 2):
            raise SystemExit("--workload C4 (550 GB Gram matrix) needs --mode shard on several GPUs")
    group = args.mode == "group"     # rank 0 drives all `world` GPUs from this one process; the other ranks only keep time
    shard = args.mode == "shard" or group   # world == 1: the sharded schedule on one rank (its own overhead)
    X, y, Xs = synthetic(N, DIM, M, 12345 + (0 if shard else rank))   # replicas: own draw each
    Xd, yd, Xsd = (torch.from_numpy(a).to(dev, tdt) for a in (X, y, Xs))
    if dtype == "float32":
        X, y, Xs = (a.astype("float32") for a in (X, y, Xs))
    if group:
        gp = None
        if rank == 0:
            devs = [args.device] * world if args.device is not None else list(range(world))
            gp = GP(kernel, lengthscale, SF2, SN2, jitter=0.0, block=args.block, profile=True, devices=devs,
                    transport="local", oversubscribe=args.device is not None, dtype=dtype)
    elif shard:
        gp = GP(kernel, lengthscale, SF2, SN2, jitter=0.0, device=local, block=args.block, profile=True,
                world=world, rank=rank, comm="rccl" if args.backend == "nccl" else "host", dtype=dtype)
    else:
        gp = GP(kernel, lengthscale, SF2, SN2, jitter=0.0, device=local, block=args.block, profile=True, dtype=dtype)
    beat(args, "communicator + inputs")
    inject = os.environ.get("GPX_BENCH_INJECT", "")     # rehearsal of the failure path: fail:R / hang:R

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            if ctrl_nccl:
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
            torch.cuda.synchronize(dev)

    def step():
        if gp is None:                      # group mode, rank > 0: rank 0's process holds every GPU
            beat(args, "step")
            return None, None, {}
        gp.fit(Xd, yd)
        mean, var = gp.predict(Xsd)
        beat(args, "step")
        if group:
            return mean, var, gp.timings_
        if shard and inject == f"fail:{rank}":
            raise RuntimeError("injected failure (GPX_BENCH_INJECT)")
        if shard and inject == f"hang:{rank}":
            time.sleep(1e6)
        return mean, var, gp.timings_

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        mean, var, tm = step()
        for k_, v_ in tm.items():
            acc[k_] = acc.get(k_, 0.0) + v_
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if ctrl_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ok = True if gp is None else bool(torch.isfinite(mean).all().item() and (var > (0 if dtype == "float64" else -1e-4)).all().item())
    pcie_ms = None
    unprofiled_ms = None
    one_pass = None
    if world == 1 and not shard:
        # the same step with HOST NumPy arrays in and out (H2D of X, y, Xs and D2H of mean, var inside
        # the clock): never `value`, stated once beside it
        gp.fit(X, y).predict(Xs)
        t1 = time.perf_counter()
        gp.fit(X, y).predict(Xs)
        pcie_ms = (time.perf_counter() - t1) * 1e3
        # what GPX_FLAG_PROFILE (a hipEvent pair around every Cholesky sub-phase launch inside the timed
        # region: the roofline's live clock) costs: the same step on a handle without the flag
        nrep = max(1, min(3, args.steps))
        t_plain = t_prof = 0.0
        for _ in range(nrep):                   # interleaved on the SAME handle: same buffers, same clock drift
            for on in (False, True):
                gp.set_profile(on)
                t1 = time.perf_counter()
                gp.fit(Xd, yd).predict(Xsd)
                torch.cuda.synchronize(dev)
                if on:
                    t_prof += time.perf_counter() - t1
                else:
                    t_plain += time.perf_counter() - t1
        unprofiled_ms = (t_plain * 1e3 / nrep, t_prof * 1e3 / nrep)
        # the ONE-PASS form of the same step (GP.fit_predict -> gpx_fit_predict, ABI v4+: the query points' cross-kernel
        # rows ride through the factorisation as bordered rows): reported beside the headline, never `value`
        one_pass = None
        if dtype in ("float64", "float32") and M <= 8192:
            gp.fit_predict(Xd, yd, Xsd)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(nrep):
                m1p, v1p = gp.fit_predict(Xd, yd, Xsd)
            torch.cuda.synchronize(dev)
            ms1 = (time.perf_counter() - t1) * 1e3 / nrep
            one_pass = {"ms_per_step": ms1, "points_per_s": (N + M) / (ms1 * 1e-3),
                        "mean_max_abs_diff_vs_two_calls": float((m1p - mean).abs().max().item()),
                        "var_max_abs_diff_vs_two_calls": float((v1p - var).abs().max().item()),
                        "note": "gp.fit_predict(X, y, Xs): same inputs, same outputs, one factorisation pass"}
            del m1p, v1p
    # every rank's own clocks of the timed steps (process-per-GPU shard): the JSON line carries them next to rank 0's
    # phases, so that the first run on real GPUs can be read term by term against tools/scaling_model.py
    per_rank = None
    if shard and not group and world > 1:
        mine = {k_: round(acc.get(k_, 0.0) / max(1, args.steps), 3)
                for k_ in ("fit_total", "chol", "comm", "solve", "predict_total", "trsm")}
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box
    # the ONE-PASS form on the sharded schedule (round 4: every rank's slice of the query points rides through its part of
    # the factorisation): beside the headline, never `value`.  One rank and groups (one process makes the calls: an error
    # is an exception here and every rank still reaches the barrier); process-per-GPU shards of several ranks on request.
    if shard and dtype in ("float64", "float32") and ok and (world == 1 or group or args.one_pass):
        err1 = None
        m1p = v1p = None
        try:
            if gp is not None:
                gp.fit_predict(Xd, yd, Xsd)
        except Exception as e:  # noqa: BLE001 — the headline above is complete without it
            err1 = f"{type(e).__name__}: {e}"[:300]
        sync()                                   # every rank reaches every barrier, whatever happened on rank 0
        t1 = time.perf_counter()
        try:
            if gp is not None and err1 is None:
                m1p, v1p = gp.fit_predict(Xd, yd, Xsd)
        except Exception as e:  # noqa: BLE001
            err1 = f"{type(e).__name__}: {e}"[:300]
        sync()
        ms1 = (time.perf_counter() - t1) * 1e3
        if world > 1:
            t = torch.tensor([ms1], dtype=torch.float64, device=dev if ctrl_nccl else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms1 = float(t.item())
        if err1 is not None:
            one_pass = {"error": err1}
        elif gp is not None:
            one_pass = {"ms_per_step": ms1, "points_per_s": (N + M) / (ms1 * 1e-3),
                        "mean_max_abs_diff_vs_two_calls": float((m1p - mean).abs().max().item()),
                        "var_max_abs_diff_vs_two_calls": float((v1p - var).abs().max().item()),
                        "note": "gp.fit_predict(X, y, Xs) on the shard: one timed call after one warm-up call"}
        del m1p, v1p
        beat(args, "one pass")
    shard_check = None
    if shard and N <= 131072:
        # the sharded posterior against the single-GPU path on the same inputs (rank 0's GPU)
        verdict = [None]
        if rank == 0:
            with GP(kernel, lengthscale, SF2, SN2, jitter=0.0, device=local, block=args.block, dtype=dtype) as one:
                m1, v1 = one.fit(Xd, yd).predict(Xsd)
            # north_star's elementwise criterion (1e-6) between two different blockings of the
            # same factorisation, plus the error relative to the largest posterior mean.  fp32 / mixed shards are
            # compared with the unsharded handle of the same dtype at that dtype's level (fp32-grade quantities:
            # 5e-3 of the largest mean / of sf2; the mixed mode's fp64-grade mean: 1e-6 elementwise)
            em = float(((mean - m1).abs() / m1.abs().clamp_min(1e-6)).max().item())
            ev = float(((var - v1).abs() / v1.clamp_min(1e-6 * SF2)).max().item())
            es = float(((mean - m1).abs().max() / m1.abs().max()).item())
            evs = float(((var - v1).abs().max() / SF2))
            del m1, v1
            if dtype == "float64":
                good = em < 1e-6 and ev < 1e-6
            elif dtype == "mixed":
                good = em < 1e-6 and evs < 5e-3
            else:
                good = es < 5e-3 and evs < 5e-3
            verdict[0] = {"vs": f"single-GPU path ({dtype}), same inputs", "mean_max_rel": em, "var_max_rel": ev,
                          "mean_err_over_max_mean": es, "var_err_over_sf2": evs, "tol": 1e-6 if dtype == "float64" else 5e-3,
                          "ok": bool(good and ok)}
        if world > 1:
            dist.broadcast_object_list(verdict, src=0)
        shard_check = verdict[0]
        beat(args, "check")
        if not shard_check["ok"]:
            if rank == 0:
                print(f"[bench] sharded result disagrees with the single-GPU path: {shard_check}",
                      file=sys.stderr, flush=True)
            sys.exit(3)

    if rank == 0:
        steps = max(1, args.steps)
        syrk_ms = acc["chol_syrk"]
        syrk_tflops = acc["syrk_flops"] / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        launches = int(acc["syrk_launches"])
        phases = {k_: round(acc[k_] / steps, 3) for k_ in
                  ("comm", "h2d", "kbuild", "chol", "chol_diag", "chol_trsm", "chol_strip", "chol_syrk", "solve", "logdet",
                   "fit_total", "kstar", "mean", "trsm", "var", "d2h", "predict_total")}
        kb_ms = acc["kbuild"] / steps
        out = {
            "metric": "gp_fit_predict_points_per_sec",
            "value": (1 if shard else world) * (N + M) * steps / elapsed,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "pcie_inclusive_ms_per_step": pcie_ms,
            "profile_flag": None if unprofiled_ms is None else {
                "ms_per_step_without": unprofiled_ms[0], "ms_per_step_with": unprofiled_ms[1],
                "cost_ms_per_step": unprofiled_ms[1] - unprofiled_ms[0],
                "note": "the timed steps run with GPX_FLAG_PROFILE (hipEvent pairs around the Cholesky sub-phase launches: "
                        "the roofline's clock); measured after the timed region on the same handle, flag toggled step by "
                        "step (without, with, without, ...), so that buffers and clock drift are the same"},
            "fit_predict_one_pass": one_pass,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": {"float64": "f64", "float32": "f32", "mixed": "f32 factor + f64 refinement"}[dtype],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: exact GP fit+predict, N={N} d={DIM} {kernel} "
                                   f"{'ARD ' if args.workload == 'C5' else ''}{dtype}, M={M}, inputs resident in HBM",
                       "N": N, "d": DIM, "M": M, "lengthscale": lengthscale,
                       "kernel": kernel, "block": args.block or (shard_block(N, world) if shard else
                                                                (2048 if N >= int(os.environ.get("GPX_NB_WIDE_FROM", "40960") or 0) > 0 else 1024)),
                       "parallelism": "1 gpu" if world == 1 else
                       (f"row-block shard ({DEALING} dealing) over {world} gpus (one process, in-process peer-copy transport)" if group else
                        f"row-block shard ({DEALING} dealing) over {world} gpus ({'RCCL' if args.backend == 'nccl' else 'host collectives, rehearsal'})" if shard
                        else f"{world} independent replicas")},
            "outputs_finite": ok,
            "shard_check": shard_check,
            **({"timed_on": "rank 0 only (its process drives all GPUs; the other ranks hold no GPU context and wait)" if group
                else "max over ranks", "multi_gpu_transport_verified_on_hardware": False,
                "panel_solve": "slab" if os.environ.get("GPX_SHARD_DENSE_PANEL") == "0" else
                "dense (block inverse broadcast with the diagonal block: +nb^2 doubles per panel; GPX_SHARD_DENSE_PANEL=0 = slab)"}
               if shard and world > 1 else {}),
            **({"fallback_from": os.environ["GPX_BENCH_FALLBACK_REASON"]} if group and os.environ.get("GPX_BENCH_FALLBACK_REASON") else {}),
            "phases_ms": phases,
            **({"per_rank_phases_ms": per_rank} if per_rank else {}),
            "roofline": {
                "kernel": "gemm_nt_kernel<128,LOWER> (trailing SYRK of the blocked Cholesky)",
                "bound": "mfma", "achieved": syrk_tflops, "peak": peak_mfma,
                "unit": "TFLOP/s", "frac": syrk_tflops / peak_mfma,
                "traffic": pmc_traffic()[0] if dtype == "float64" and N == N_TRAIN else None,
                "traffic_source": pmc_traffic()[1] if dtype == "float64" and N == N_TRAIN else None,
                "launches": launches,
                "flops_per_launch": acc["syrk_flops"] / max(1, launches),
                "avg_launch_ms": syrk_ms / max(1, launches)},
            "kbuild": {"bound": "hbm", "achieved": acc["kbuild_bytes"] / steps / (kb_ms * 1e-3) / 1e9
                       if kb_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s"},
        }
        out["kbuild"]["frac"] = out["kbuild"]["achieved"] / PEAK_HBM_GBS
        chol_ms = acc["chol"] / steps
        out["cholesky_tflops"] = (N ** 3 / 3.0) / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        if shard:   # the sharded update has no per-launch flop bookkeeping: rate the whole factorisation
            out["roofline"].update(kernel="blocked Cholesky, all ranks (trailing updates: gemm_nt_stair_kernel<128>, the staircase of the dealt row blocks)",
                                   traffic=None, traffic_source=None,
                                   achieved=out["cholesky_tflops"], peak=peak_mfma * world,
                                   frac=out["cholesky_tflops"] / (peak_mfma * world))
        if world == 1 and not args.no_microbench:
            import ctypes as C
            a, b = C.c_double(0), C.c_double(0)
            if _abi.load().gpx_microbench(C.byref(a), C.byref(b)) == 0:
                out["microbench"] = {"mfma_f64_loop_tflops": a.value, "stream_copy_gbs": b.value}
                # the kernel build against the ceiling this card reaches on a plain stream copy (one 16-byte
                # element per lane, non-temporal: the ~6.3 TB/s class of MI355X_MICROARCH.md), beside the 8 TB/s spec
                out["kbuild"]["frac_of_measured_stream_copy"] = out["kbuild"]["achieved"] / b.value if b.value > 0 else None
                out["roofline"]["frac_of_measured_mfma_loop"] = (out["roofline"]["achieved"] / a.value
                                                                 if a.value > 0 and not shard and dtype == "float64" else None)
        if dtype == "mixed":
            out["refinement"] = {k_: acc[k_] / steps for k_ in ("refine", "refine_iters", "refine_resid0", "refine_resid")}
        if world == 1 and not args.no_cpu_baseline and args.workload == "C3" and N == N_TRAIN:
            out["cpu_baseline"] = cpu_baseline("full" if args.cpu_baseline_full else args.cpu_baseline)
        if world > 1:
            sync()
        print(json.dumps(out), flush=True)
    elif world > 1:
        sync()
    beat(args, "done")
    if gp is not None:
        gp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
