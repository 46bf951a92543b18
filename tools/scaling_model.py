#!/usr/bin/env python3
"""PREDICTED multi-GPU budget of the sharded fit + predict — a model, not a measurement:
no run of this library has ever had more than one physical GPU (SCALE_r01/r02/r03: skipped).  Every input
is a number measured on ONE MI355X (cited) or a stated assumption about xGMI; when a SCALE record
exists, the per-term deviations point at the cause.

    python tools/scaling_model.py            # prints the markdown tables of DESIGN.md §6
    python tools/scaling_model.py --compare FILE [FILE ...]   # measured bench.py JSON lines against the model, term by term

Schedule modelled (csrc/gpx_shard.inc, round 4): row blocks of height nb dealt block-cyclically; per panel p
  main stream      : STRIP_B(p) (own rows below block p+1, its columns) -> REST(p)          (own rows; MFMA-bound)
  look-ahead stream: STRIP_D(p) (the next diagonal block only, on its owner) -> diagonal block p+1 -> broadcast
                     [L_pp | inverses | W_p] -> every rank solves its rows of panel p+1 -> all-gather
                     (from 4 ranks on the owner runs STRIP_D and the diagonal block on a stream of its own, right behind ITS
                     solve of panel p and beside the all-gather of panel p: per panel solve + max(STRIP_D + block + broadcast,
                     all-gather + STRIP_B) instead of their sum)
                     (no un-permute behind it any more: the update kernels read the gathered panel in place; the replicated
                     factor's copy of the panel is written on the copy stream beside the REST)
  step time        = max(STRIP_B(p) + REST(p), chain(p+1))
(round 3 ran the whole STRIP first and the chain behind it: STRIP(p) + max(REST(p), chain(p+1)); the table carries that
fit time in its own column.)
"""
import json

# ---- measured on one MI355X (profiles/) ---------------------------------------------------------
RATE = {1024: 68.3e12, 512: 63.2e12, 256: 55e12}   # sharded trailing-update kernel, flop/s (one-rank schedule
                                                    # 68.3 TF at nb=1024, 63.2 at nb=512; 256: assumption)
RATE_1GPU = 69.0e12                                 # unsharded SYRK under the bench (profiles/r03_start_bench.json: 69.6)
CHAIN_US_PER_128 = 60.0                             # diagonal chain alone, per 128 columns (profiles/r03_c2_chain_gaps.txt: 57.5-60 after the POTF2 phase overlap)
CHAIN_STRETCH = 5.6                                 # the same chain beside a busy trailing update: 214 ms / 64 blocks
                                                    # (r03 bench chol_diag) against 0.6 ms per block alone
SOLVE_RATE = 45e12                                  # panel solve as a dense product with W_p (ltri kernel; DESIGN §5.0: 39 -> ~45 TF)
SOLVE_FLOOR = 0.27e-3                               # one K = 1024 walk of a 128-tile (C2 trace), scales with nb / 1024
HBM = 3.0e12                                        # un-permute / copy-back rate actually reached by the copy kernels
PRED_1GPU = {"trsm_rate": 69e12}                    # variance TRSM at M >= 2048 rows (bench: 254 ms for 1.76e13 flop)
# predict of few query points against the whole N = 65536 factor, one card (tools/predict_rows.py, profiles/r04_predict_rows.txt):
# rows -> achieved flop/s of N^2 rows / time (the per-rank share of a replicated shard: 512 rows at P = 8)
PRED_ROWS_TF = {128: 34.5e12, 256: 44.7e12, 512: 52.4e12, 1024: 63.6e12, 2048: 67.3e12, 4096: 70.2e12, 8192: 70.9e12}
ZSOLVE = 2.3e-3                                     # z = L^-1 y on the replicated factor: streaming few-right-hand-side solver (round 4; 18 ms before)
STRIP_D_FLOOR = 45e-6                               # the next diagonal block's update alone: one K = nb walk of 64-tiles (C2 trace: 41-57 us at nb = 1024)
# ---- assumptions about the fabric (task statement: 7 links x ~153 GB/s per GPU, full mesh) -------------
LINK = 153e9 * 0.70                                 # sustained per-link payload rate (70 % of peak: assumption)
LAT = 30e-6                                         # latency of one RCCL collective on the look-ahead stream (assumption)


def pick_nb(N, P, snake=True):
    """the library's balance rule (gpx_shard.inc): at least 8 blocks per rank under the snake dealing, 16 under the cyclic"""
    nb = 1024
    while nb > 256 and nb * (8 if snake else 16) * P > N:
        nb //= 2
    return nb


def owner(g, P, snake):
    """gpx_internal.h: Deal — cyclic (rounds 1-3) or snake (round 4: rounds of 2 P blocks dealt 0 .. P-1, P-1 .. 0)."""
    if not snake:
        return g % P
    pos = g % (2 * P)
    return pos if pos < P else 2 * P - 1 - pos


def heaviest_share(nblk, p, P, snake):
    """update of panel p: the heaviest rank's work over the mean (row block g > p + 1 costs g - p - 1.5 block products in
    the REST and one in STRIP_B) — the ranks meet at every panel, so the heaviest one sets the pace."""
    w = [0.0] * P
    for g in range(p + 2, nblk):
        w[owner(g, P, snake)] += (g - p - 1.5) + 1.0
    tot = sum(w)
    return max(w) * P / tot if tot > 0 else 1.0


def chain_time(idle_work, busy_for):
    """idle_work seconds of chain at idle speed, running beside a trailing update that keeps the owner busy
    for busy_for seconds (progress 1 / CHAIN_STRETCH while busy, 1 afterwards)."""
    if busy_for * (1.0 / CHAIN_STRETCH) >= idle_work:
        return idle_work * CHAIN_STRETCH
    return busy_for + idle_work - busy_for / CHAIN_STRETCH


def fit_time(N, P, nb=None, split=True, replicated=True, snake=True, two_pipe=True):
    nb = nb or pick_nb(N, P, snake)
    rate = RATE[nb] if P > 1 else RATE_1GPU
    nblk = N // nb
    diag_idle = nb / 128 * CHAIN_US_PER_128 * 1e-6
    t = diag_idle + SOLVE_FLOOR                      # head: block 0 and panel 0, nothing to hide behind
    exposed = 0.0
    first_exposed = None
    comm_bytes = 0.0
    for p in range(nblk - 1):
        n = N - (p + 1) * nb                         # trailing rows
        heavy = heaviest_share(nblk, p, P, snake) if P > 1 else 1.0
        strip = 2.0 * (n * nb - nb * (nb - 1) / 2) * nb / P / rate * heavy
        rest_n = n - nb
        rest = max(0.0, rest_n * (rest_n + 1.0) * nb / P / rate) * heavy
        # chain of panel p+1 (runs beside REST(p))
        rows = max(0, n - nb)
        bc_bytes = (2 * nb * nb + 64 * nb) * 8
        ag_bytes = rows * nb * 8
        bcast = 0.0 if P == 1 else LAT + bc_bytes / LINK
        solve = max(SOLVE_FLOOR * nb / 1024, rows / P * nb * nb / SOLVE_RATE)
        gather = 0.0 if P == 1 else LAT + ag_bytes / P / LINK       # each peer's piece over its own link (full mesh)
        unperm = 0.0 if P == 1 else 2 * ag_bytes / HBM
        comm_bytes += (bc_bytes + ag_bytes) * (P - 1) / P if P > 1 else 0
        if split:   # round 4: only the next diagonal block's update on the chain; the rest of the strip with the REST
            strip_d = max(STRIP_D_FLOOR * nb / 1024, nb * (nb + 1.0) * nb / 30e12)
            strip_b = max(0.0, 2.0 * (n - nb) * nb * nb / P / rate) * heavy
            main = strip_b + rest   # (replicated: panel p goes into the full factor on the copy stream, beside the REST)
            diag_path = strip_d + chain_time(diag_idle, main) + bcast
            if two_pipe and P >= 4:   # the owner's chain beside the previous panel's all-gather: the gather leaves the cycle
                ch = solve + max(diag_path, gather + strip_b)
            else:
                ch = diag_path + solve + gather
            step = max(main, ch)
            over = ch - main
        else:
            rest_m = rest + (unperm if replicated else 0.0)   # (the copy into the full factor was always on the main stream)
            ch = chain_time(diag_idle, rest_m) + bcast + solve + gather + unperm
            step = strip + max(rest_m, ch)
            over = ch - rest_m
        if over > 0:
            exposed += over
            if first_exposed is None:
                first_exposed = p + 1
        t += step
    return {"nb": nb, "fit_s": t, "exposed_chain_s": exposed, "first_exposed_panel": first_exposed, "panels": nblk,
            "recv_GB_per_rank": comm_bytes / 1e9}


def predict_time(N, M, P, replicated):
    if replicated:                                   # rank r: M / P query points against the whole factor
        rows = max(128, -(-M // P // 128) * 128)
        key = min(PRED_ROWS_TF, key=lambda r: abs(r - rows))   # measured rate at the nearest row count
        rate = PRED_ROWS_TF[key]
        if P > 1 and pick_nb(N, P) == 512 and key == 512:      # the shard's 512-blocks: 46.45 ms for 512 rows (r04_predict_block_ab.txt)
            rate = N * N * 512 / 46.45e-3
        return N * N * rows / rate + 1.5e-3 + (0 if P == 1 else LAT)
    nb = pick_nb(N, P)                               # distributed variance solve: one (M x nb) broadcast per block, hidden
    rate = RATE[nb]                                  # behind the rest of the previous update when it is long enough
    t = 0.0
    for p in range(N // nb):
        n = N - (p + 1) * nb
        upd = 2.0 * M * n * nb / P / rate
        bc = LAT + M * nb * 8 / LINK + SOLVE_FLOOR * nb / 1024
        t += max(upd, bc)
    return t


def table(N, M, Ps, replicated, label):
    rows = []
    base = None
    for P in Ps:
        f = fit_time(N, P, replicated=replicated)
        f["fit_s_round3_schedule"] = fit_time(N, P, split=False, replicated=replicated, snake=False)["fit_s"]
        f["fit_s_cyclic"] = fit_time(N, P, replicated=replicated, snake=False, two_pipe=False)["fit_s"]
        f["fit_s_one_pipe"] = fit_time(N, P, replicated=replicated, two_pipe=False)["fit_s"]
        pr = predict_time(N, M, P, replicated and P > 1) if P > 1 else predict_time(N, M, 1, True)
        extra = (ZSOLVE if replicated or P == 1 else 2 * (N // f["nb"]) * (LAT + 35e-6))   # alpha solves (distributed: not overlapped)
        tot = f["fit_s"] + pr + extra
        base = base or tot
        flops = N ** 3 / 3.0 + float(N) * N * M
        rows.append({"P": P, **f, "predict_s": pr, "solves_s": extra, "total_s": tot, "points_per_s": (N + M) / tot,
                     "speedup": base / tot, "efficiency": base / tot / P, "frac_peak": flops / tot / (P * 78.6e12)})
    print(f"\n**{label}** (N = {N}, M = {M})\n")
    print("| P | nb | fit ms | (one chain stream) | (cyclic dealing, one chain stream) | (round-3 schedule, cyclic) | of which exposed chain ms | chain first exposed at panel | predict ms | solves ms | total ms | points/s | speed-up | efficiency | fraction of P x 78.6 TF | received per rank GB |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        several = len(rows) > 1
        cells = [r["P"], r["nb"], f"{r['fit_s'] * 1e3:.0f}", f"{r['fit_s_one_pipe'] * 1e3:.0f}", f"{r['fit_s_cyclic'] * 1e3:.0f}", f"{r['fit_s_round3_schedule'] * 1e3:.0f}", f"{r['exposed_chain_s'] * 1e3:.0f}",
                 f"{r['first_exposed_panel']} of {r['panels']}", f"{r['predict_s'] * 1e3:.0f}", f"{r['solves_s'] * 1e3:.0f}",
                 f"{r['total_s'] * 1e3:.0f}", f"{r['points_per_s']:.0f}", f"{r['speedup']:.2f}" if several else "-",
                 f"{r['efficiency']:.2f}" if several else "-", f"{r['frac_peak']:.2f}", f"{r['recv_GB_per_rank']:.1f}"]
        print("| " + " | ".join(str(c) for c in cells) + " |")
    return rows


def compare(paths):
    """Measured bench.py lines (one JSON object per line, e.g. the driver's SCALE record or gpurun_out files) against the
    model, term by term: which prediction a first multi-GPU run confirms and which it does not."""
    import sys
    lines = []
    for path in paths:
        for raw in open(path).read().splitlines():
            raw = raw.strip()
            if not raw.startswith("{"):
                continue
            try:
                d = json.loads(raw)
            except ValueError:
                continue
            for cand in (d, d.get("parsed") if isinstance(d, dict) else None):   # bare bench line, or a driver record wrapping it
                if isinstance(cand, dict) and "n_gpus" in cand and "phases_ms" in cand:
                    lines.append(cand)
    if not lines:
        print("no bench lines with n_gpus and phases_ms found", file=sys.stderr)
        return 1
    print("| P | block | measured step ms | model | measured fit | model | measured chol | measured comm (inside fit) | measured predict | model | measured solve | model |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for d in sorted(lines, key=lambda x: x["n_gpus"]):
        P = int(d["n_gpus"])
        cfg = d.get("config", {})
        N, M = int(cfg.get("N", 65536)), int(cfg.get("M", 4096))
        nb = int(cfg.get("block") or pick_nb(N, P))
        ph = d["phases_ms"]
        repl = N <= 131072
        f = fit_time(N, P, nb=nb if P > 1 else None, replicated=repl)
        pr = predict_time(N, M, P, repl and P > 1) if P > 1 else predict_time(N, M, 1, True)
        sol = ZSOLVE if (repl or P == 1) else 2 * (N // f["nb"]) * (LAT + 35e-6)
        print(f"| {P} | {nb} | {d['ms_per_step']:.1f} | {(f['fit_s'] + pr + sol) * 1e3:.0f} | {ph.get('fit_total', 0):.1f} | {f['fit_s'] * 1e3:.0f} | "
              f"{ph.get('chol', 0):.1f} | {ph.get('comm', 0):.1f} | {ph.get('predict_total', 0):.1f} | {pr * 1e3:.0f} | {ph.get('solve', 0):.1f} | {sol * 1e3:.1f} |")
        for r, pr_ in enumerate(d.get("per_rank_phases_ms") or []):
            print(f"|   rank {r} | | | | {pr_.get('fit_total')} | | {pr_.get('chol')} | {pr_.get('comm')} | {pr_.get('predict_total')} | | {pr_.get('solve')} | |")
    return 0


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "--compare":
        raise SystemExit(compare(sys.argv[2:]))
    out = {"C3": table(65536, 4096, (1, 2, 4, 8), True, "C3 sharded, replicated factor (the `bench.py --gpus N` path)"),
           "C4": table(262144, 4096, (8,), False, "C4, distributed solves (550 GB Gram matrix: 8 GPUs only)")}
    print("\n```json\n" + json.dumps({"inputs": {"rate": {str(k): v for k, v in RATE.items()}, "rate_1gpu": RATE_1GPU,
                                                  "chain_us_per_128": CHAIN_US_PER_128, "chain_stretch": CHAIN_STRETCH,
                                                  "solve_rate": SOLVE_RATE, "link_Bps": LINK, "latency_s": LAT}}) + "\n```")
