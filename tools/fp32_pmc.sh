# SQ counters of the fp32 and the fp64 trailing update side by side (why is the fp32 engine at 81 %
# of its peak where the fp64 one holds 88 %?).  Output: gpurun_out/fp32pmc/{f32,f64}_per_kernel.csv
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fp32pmc
mkdir -p $O
for dt in float32 float64; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/$dt.C -- python3 $R/tools/fit_loop.py $dt 65536 0 2 > $O/$dt.C.json 2> $O/$dt.C.err || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/$dt.D -- python3 $R/tools/fit_loop.py $dt 65536 0 2 > $O/$dt.D.json 2> $O/$dt.D.err || exit 1
  (cd $R && python tools/pmc_summary.py $O/${dt}_per_kernel.csv $O/$dt.C $O/$dt.D)
done
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
