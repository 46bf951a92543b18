#!/usr/bin/env python3
"""Per-panel timeline of the MAIN stream of the last two-call fit in a rocprofv3 kernel trace of bench.py (tools/c3_trace.sh):
between consecutive trailing updates (the triangular 128-tile launches) every kernel of that queue with its duration, and for
the first panels the rate of the rectangular strip (rows below the next diagonal block x panel width, K = panel width) and of the
panel-solve product — the work outside the roofline kernel.   python tools/c3_main_stream.py gpurun_out/c3trace/trace.csv [N nb]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ks = [i for i, r in enumerate(rows) if "kbuild_kernel<double, 0, true" in r["Kernel_Name"]]
ke = [i for i, r in enumerate(rows) if "kbuild_kernel<double, 0, false" in r["Kernel_Name"]]
a = b = None
for cand in reversed(ks):
    nxt = [i for i in ke if i > cand]
    if nxt and rows[nxt[0]]["s"] - rows[cand]["s"] > 5e8:
        a, b = cand, nxt[0]
        break
fit = rows[a:b]
short = lambda n: ("SYRK" if "gemm_nt_kernel<double, 128, true, 0" in n else "rect128" if "gemm_nt_kernel<double, 128, false" in n else
                   "ltri128" if "ltri_kernel<double, 128" in n else "ltri64" if "ltri" in n else "tri64" if "gemm_nt_kernel<double, 64, true" in n else
                   "rect64" if "gemm_nt_kernel<double, 64, false" in n else "potf2" if "potf2" in n else "trsm" if "trsm_rlt" in n else
                   "copy" if "copy" in n.lower() else "wait" if "wait_counter" in n else n.split("(")[0][-24:])
mainq = [r for r in fit if short(r["Kernel_Name"]) == "SYRK"][0]["Queue_Id"]
mq = [r for r in fit if r["Queue_Id"] == mainq]
tot = collections.Counter()
for r in mq:
    tot[short(r["Kernel_Name"])] += r["e"] - r["s"]
print("main stream totals (ms):", {k: round(v / 1e6, 1) for k, v in tot.most_common()})
allq = collections.Counter()
for r in fit:
    allq[(r["Queue_Id"], short(r["Kernel_Name"]))] += r["e"] - r["s"]
print("every queue (ms):", {f"q{q}:{k}": round(v / 1e6, 1) for (q, k), v in allq.most_common(14)})
idx = [i for i, r in enumerate(mq) if short(r["Kernel_Name"]) == "SYRK"]
print("panel: kernels of the main stream between trailing updates (name us), then that update (ms, TF)")
for p in range(min(len(idx), 40)):
    lo = idx[p - 1] + 1 if p else 0
    seg = mq[lo:idx[p]]
    u = mq[idx[p]]
    n = N - (p + 2) * nb                      # rows of the REST of panel p
    tf = n * (n + 1.0) * nb / ((u["e"] - u["s"]) * 1e-9) / 1e12
    items = []
    for r in seg:
        nm, d = short(r["Kernel_Name"]), (r["e"] - r["s"]) / 1e3
        if d < 20:
            continue
        extra = ""
        nrest = N - (p + 2) * nb
        if nm == "rect128" and d > 300:       # STRIP_B(p): nrest x nb, K = nb
            extra = f" [{2.0 * nrest * nb * nb / (d * 1e-6) / 1e12:.0f} TF if strip]"
        if nm == "ltri128" and d > 300:       # rest of panel solve p: (rows below block p+1) x nb x nb / ... half-dense
            rows_ = N - (p + 2) * nb
            extra = f" [{1.0 * rows_ * nb * nb / (d * 1e-6) / 1e12:.0f} TF if panel product]"
        items.append(f"{nm} {d:.0f}{extra}")
    if p < 12 or p % 4 == 0:
        print(f"{p:2d}: " + "  ".join(items) + f"  || SYRK {(u['e'] - u['s']) / 1e6:.2f} ms {tf:.1f} TF")
