#!/bin/bash
# kernel trace of one C3 step per schedule variant -> main-stream gap analysis (tools/c3_gaps.py)
set -e
export GPX_FUSED_STRIP=${1:-0}
bash tools/c3_trace.sh > /dev/null 2>&1
python tools/c3_gaps.py gpurun_out/c3trace/trace.csv | tee gpurun_out/r03_c3_main_stream_gaps_fused$GPX_FUSED_STRIP.txt
python - <<'PY'
import csv, collections, os
rows = list(csv.DictReader(open("gpurun_out/c3trace/trace.csv")))
d = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-70:]
    d[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[n][1] += 1
for n, (t, c) in sorted(d.items(), key=lambda x: -x[1][0])[:14]:
    print(f"{t/1e6:10.2f} ms {c:6d} calls  avg {t/c/1e3:9.1f} us  {n}")
PY
rm -rf gpurun_out/c3trace/*/ gpurun_out/c3trace/trace.csv
