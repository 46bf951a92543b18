set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcA.json 2> $O/pmcA.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcB -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcB.json 2> $O/pmcB.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmcC -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/pmcC.json 2> $O/pmcC.err
export GPX_SYRK_TALL=1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/tallA -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/tallA.json 2> $O/tallA.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tallB -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/tallB.json 2> $O/tallB.err
unset GPX_SYRK_TALL
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2stats -- python3 $R/tools/c2_bench.py --steps 10 > $O/c2_under_rocprof.json 2> $O/c2stats.err
cd $R
python tools/pmc_summary.py $O/pmc_per_kernel.csv $O/pmcA $O/pmcB $O/pmcC
python tools/pmc_summary.py $O/tall_pmc_per_kernel.csv $O/tallA $O/tallB
# keep the merged output small: drop the raw per-dispatch counter files
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
ls -la $O $O/stats/* | head -40
