# kernel stats of one mixed-precision fit + predict at C5 (what the fp64 refinement costs, kernel by kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mixedstats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload C5 --dtype mixed --steps 1 --warmup 1 --no-cpu-baseline --no-microbench > $O/run.json 2> $O/run.err
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
