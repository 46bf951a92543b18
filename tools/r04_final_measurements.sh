# Round-4 evidence (one gpurun call): rocprofv3 kernel stats of bench.py, three separate PMC passes (never combined with
# tracing domains; the library hands over by hipEvents by itself there: flag_handover_probe / ROCPROF_COUNTER_COLLECTION),
# a marker trace (roctx ranges around the phases), the C2 / C5 lines, then the unprofiled default bench line (with the
# full-size CPU baseline measured in the run).  Full logs in files; nothing is piped through tail.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcA.json 2> $O/pmcA.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcB -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcB.json 2> $O/pmcB.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmcC -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcC.json 2> $O/pmcC.err
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/markers -- python3 $R/tools/c2_bench.py --steps 2 --warmup 1 --ntrain 4096 --mtest 512 > $O/markers.json 2> $O/markers.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2stats -- python3 $R/tools/c2_bench.py --steps 10 > $O/c2_under_rocprof.json 2> $O/c2stats.err
cd $R
python tools/pmc_summary.py $O/pmc_per_kernel.csv $O/pmcA $O/pmcB $O/pmcC
find $O -name "*marker*" | head; find $O/markers -name "*marker_api_trace.csv" | head -1 | xargs -r -I{} sh -c 'head -40 {} > '$O'/marker_trace_head.csv'
find $O/markers -name "*marker*stats*.csv" | head -1 | xargs -r -I{} cp {} $O/marker_stats.csv
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*marker_api_trace.csv" -delete
python tools/c2_bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python tools/c2_bench.py --no-profile > $O/c2_bench_noprofile.json 2>> $O/c2_bench.err
python tools/c2_bench.py --no-profile --fused > $O/c2_bench_one_pass.json 2>> $O/c2_bench.err
python bench.py --workload C5 --no-cpu-baseline --no-microbench > $O/bench_C5_f32.json 2> $O/bench_C5_f32.err
python bench.py --workload C5 --dtype mixed --no-cpu-baseline --no-microbench > $O/bench_C5_mixed.json 2> $O/bench_C5_mixed.err
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
ls -la $O $O/stats/* | head -40
tail -c 600 $O/bench.json
