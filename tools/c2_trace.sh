# kernel trace of the C2 bench (N = 8192): start/end stamps of every kernel, for the gap analysis of
# the diagonal chain (tools/c2_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c2trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/c2_bench.py --steps 3 $C2_ARGS > $O/run.json 2> $O/run.err
find $O -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/trace.csv
ls -la $O | head
