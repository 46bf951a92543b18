#!/usr/bin/env python3
"""Where the factorisation at C3 is NOT running its big kernels: from a rocprofv3 kernel trace of
bench.py (tools/c3_trace.sh), the idle time of the main stream's STRIP / REST sequence per panel
(the part of the look-ahead chain the trailing update does not hide).
    python tools/c3_gaps.py gpurun_out/c3trace/trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ks = [i for i, r in enumerate(rows) if "kbuild_kernel<double, 0, true" in r["Kernel_Name"]]
ke = [i for i, r in enumerate(rows) if "kbuild_kernel<double, 0, false" in r["Kernel_Name"]]
# the last TWO-CALL fit: a symmetric build whose next cross build (predict) comes more than 0.5 s later (the one-pass
# fits of bench.py build the cross kernel right behind the symmetric one)
a = b = None
for cand in reversed(ks):
    nxt = [i for i in ke if i > cand]
    if nxt and rows[nxt[0]]["s"] - rows[cand]["s"] > 5e8:
        a, b = cand, nxt[0]
        break
fit = rows[a:b]
# what the main stream runs between the REST launches now (round-3 schedule): strips, the rest of the panel solve,
# bordered rows — its own work, in its own order; the chain is exposed only where this stream WAITS
main_q = [r for r in fit if "gemm_nt_kernel<double, 128, true, 0" in r["Kernel_Name"]][0]["Queue_Id"]
mq = [r for r in fit if r["Queue_Id"] == main_q]
busy = sum(r["e"] - r["s"] for r in mq)
idle = sum(max(0, mq[i + 1]["s"] - mq[i]["e"]) for i in range(len(mq) - 1))
print("main stream (queue %s): %d kernels, busy %.1f ms, idle between its kernels %.1f ms, span %.1f ms" %
      (main_q, len(mq), busy / 1e6, idle / 1e6, (mq[-1]["e"] - mq[0]["s"]) / 1e6))
tail = [max(0, mq[i + 1]["s"] - mq[i]["e"]) for i in range(len(mq) - 1)]
import itertools
big_idle = sorted(((g, i) for i, g in enumerate(tail) if g > 2e5), reverse=True)[:8]
print("largest idle gaps on it (us, after kernel #):", " ".join(f"{g/1e3:.0f}@{i}" for g, i in big_idle))
big = [r for r in fit if "gemm_nt_kernel<double, 128, true, 0" in r["Kernel_Name"] or "gemm_nt_fused_kernel<double" in r["Kernel_Name"]]   # the trailing updates (REST, or the fused strip + rest launch)
print("REST launches found:", len(big), " fit span %.1f ms" % ((fit[-1]["e"] - fit[0]["s"]) / 1e6))
tot_rest = sum(r["e"] - r["s"] for r in big)
gaps = [(big[i + 1]["s"] - big[i]["e"]) / 1e3 for i in range(len(big) - 1)]
print("sum REST %.1f ms; sum of gaps between consecutive REST launches %.1f ms (STRIP + exposed chain)" % (tot_rest / 1e6, sum(gaps) / 1e3))
for i in range(0, len(gaps), 8):
    print("panels %2d..%2d gap us:" % (i, min(i + 7, len(gaps) - 1)), " ".join(f"{g:7.0f}" for g in gaps[i:i + 8]),
          "| REST ms:", " ".join(f"{(big[k]['e'] - big[k]['s']) / 1e6:5.1f}" for k in range(i, min(i + 8, len(big)))))
print("before first REST %.2f ms, after last REST %.2f ms" % ((big[0]["s"] - fit[0]["s"]) / 1e6, (fit[-1]["e"] - big[-1]["e"]) / 1e6))
