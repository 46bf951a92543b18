#!/usr/bin/env python3
"""Gap analysis of the diagonal chain from a rocprofv3 kernel trace of tools/c2_bench.py
(tools/c2_trace.sh): for the LAST fit in the trace, the chain kernels in start order —
potf2_64 -> in-block solve -> in-block SYRK — with their durations and the idle time between the
end of one and the start of the next.   python tools/c2_gaps.py gpurun_out/c2trace/trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: ("potf2" if "potf2" in n else "wait" if "wait_counter" in n else "trsm" if "trsm_rlt" in n else "syrk64" if "gemm_nt_kernel<double, 64, true" in n
                   else "gemm128" if "gemm_nt_kernel<double, 128" in n else "ltri" if "ltri" in n else "gemm64" if "gemm_nt_kernel<double, 64" in n
                   else "kbuild" if "kbuild" in n else "copy" if "copy" in n.lower() else "fill" if "fill" in n else n[:24])
# last fit = after the last symmetric kbuild
ks = [i for i, r in enumerate(rows) if "kbuild_kernel<double, 0, true" in r["Kernel_Name"]]
fit = rows[ks[-1]:]
pot = [i for i, r in enumerate(fit) if "potf2" in r["Kernel_Name"]]
print("kernels in last fit+predict:", len(fit), " potf2 launches:", len(pot))
# per 64-step: potf2 start to next potf2 start, within the first diagonal block (16 steps)
def seg(a, b):
    out = []
    for r in fit[a:b]:
        out.append(f'{short(r["Kernel_Name"])}[q{r["Queue_Id"]}] {(r["s"]-fit[a]["s"])/1e3:7.1f}+{(r["e"]-r["s"])/1e3:5.1f}')
    return out
SPB = 16 if not any("potf2_128" in r["Kernel_Name"] for r in fit) else 8   # chain steps per 1024-block (64- or 128-wide)
for blk in (0, 3):
    print(f"--- diagonal block {blk}, steps 2..5 of {SPB} (us from the step's potf2 start: start+duration) ---")
    for st in range(2, 6):
        a, b = pot[blk * SPB + st], pot[blk * SPB + st + 1]
        print(f"step {st}: {(fit[b]['s'] - fit[a]['s'])/1e3:6.1f} us |", "  ".join(seg(a, b)))
tot = collections.Counter(); cnt = collections.Counter()
for i in range(len(pot) - 1):
    if (i + 1) % SPB == 0: continue
    tot["step"] += fit[pot[i + 1]]["s"] - fit[pot[i]]["s"]; cnt["step"] += 1
    tot["potf2"] += fit[pot[i]]["e"] - fit[pot[i]]["s"]
print("mean step of %d columns (potf2 start to next potf2 start, inside a block): %.1f us; potf2 itself %.1f us" % (1024 // SPB, tot["step"] / cnt["step"] / 1e3, tot["potf2"] / cnt["step"] / 1e3))
# block boundary: every kernel from the last potf2 of block b to the first potf2 of block b + 1 (the part of the
# chain that is not diagonal steps: last in-block work, inverse post part, panel solve, update of the next
# diagonal block), us from the end of that last potf2
for blk in (0, 1, 2, 3, 6):
    a, b = pot[blk * SPB + SPB - 1], pot[(blk + 1) * SPB]
    t0 = fit[a]["e"]
    print(f"--- boundary between diagonal blocks {blk} and {blk + 1}: {(fit[b]['s'] - t0)/1e3:.1f} us from the end of the last potf2 to the start of the next ---")
    for r in fit[a + 1:b + 1]:
        if r["e"] < t0: continue
        nm = r["Kernel_Name"]
        print(f'   {short(nm):8s}[q{r["Queue_Id"]}] {(r["s"]-t0)/1e3:8.1f} +{(r["e"]-r["s"])/1e3:7.1f}   grid {r.get("Grid_Size", "?")}')
bt = [fit[pot[(k + 1) * SPB]]["s"] - fit[pot[k * SPB + SPB - 1]]["e"] for k in range(len(pot) // SPB - 1)]
print("block boundaries (us):", " ".join(f"{v/1e3:.0f}" for v in bt), " sum %.2f ms" % (sum(bt) / 1e6))
# the main stream's kernels of the last fit (everything that is not the chain's small launches), ms from the fit's start
if "--main" in sys.argv:
    f0 = fit[0]["s"]
    qmain = fit[pot[0]]["Queue_Id"]      # the prologue's potf2 runs on the main stream
    print("--- main stream (queue %s) and the look-ahead stream's larger launches: start ms + duration us ---" % qmain)
    for r in fit:
        d = (r["e"] - r["s"]) / 1e3
        nm = short(r["Kernel_Name"])
        if (r["Queue_Id"] == qmain and nm not in ("potf2", "trsm") and d > 15) or (nm in ("ltri",) or (nm == "syrk64" and d > 25)):
            print(f'   {nm:8s}[q{r["Queue_Id"]}] {(r["s"]-f0)/1e6:8.3f} +{d:7.1f}')
    for k in range(len(pot) // SPB):
        print(f"   diag block {k}: first potf2 at {(fit[pot[k*SPB]]['s']-f0)/1e6:.3f} ms, last ends {(fit[pot[k*SPB+SPB-1]]['e']-f0)/1e6:.3f} ms")
