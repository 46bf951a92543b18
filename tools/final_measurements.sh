# Round-3 final evidence (one gpurun call): rocprofv3 kernel stats of bench.py, three separate PMC passes
# (never combined with tracing domains), the C2 stats, then the unprofiled default bench lines.
# Under the PMC passes the library hands over by hipEvents by itself (it sees ROCPROF_COUNTER_COLLECTION=1;
# GPX_CHAIN_FLAG=0 is exported as well, belt and braces): counter collection serialises kernels across queues in
# its own order, which a stream parked on a device flag does not survive (the wait kernel runs before the kernel
# it waits for until its 15 s time-out).  The kernels the counters are read for (trailing update,
# kernel build) are the same code either way.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
export GPX_CHAIN_FLAG=0
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcA.json 2> $O/pmcA.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcB -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcB.json 2> $O/pmcB.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmcC -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcC.json 2> $O/pmcC.err
unset GPX_CHAIN_FLAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2stats -- python3 $R/tools/c2_bench.py --steps 10 > $O/c2_under_rocprof.json 2> $O/c2stats.err
cd $R
python tools/pmc_summary.py $O/pmc_per_kernel.csv $O/pmcA $O/pmcB $O/pmcC
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
python bench.py > $O/bench.json 2> $O/bench.err
python tools/c2_bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python tools/c2_bench.py --no-profile > $O/c2_bench_noprofile.json 2>> $O/c2_bench.err
python tools/c2_bench.py --no-profile --fused > $O/c2_bench_one_pass.json 2>> $O/c2_bench.err
GPX_SPLIT_STRIP=0 python tools/c2_bench.py --no-profile > $O/c2_bench_unsplit.json 2>> $O/c2_bench.err
GPX_SPLIT_STRIP=0 python bench.py --no-cpu-baseline --no-microbench > $O/bench_unsplit.json 2> $O/bench_unsplit.err
export C2_ARGS=--no-profile GAPS_ARGS=--main; bash tools/r03_c2trace.sh > /dev/null 2>&1; cp gpurun_out/r03_c2_chain_gaps.txt $O/c2_chain_gaps.txt
GPX_FUSED_STRIP=1 python bench.py --no-cpu-baseline --no-microbench > $O/bench_fused.json 2> $O/bench_fused.err
ls -la $O $O/stats/* | head -40
tail -c 400 $O/bench.json
