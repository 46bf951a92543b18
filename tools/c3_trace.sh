# kernel trace of one bench step at C3 for the gap analysis of the main stream (tools/c3_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-microbench > $O/run.json 2> $O/run.err
find $O -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/trace.csv
find $O -name "*kernel_trace.csv" -path "*runc*" -delete
ls -la $O | head -5
