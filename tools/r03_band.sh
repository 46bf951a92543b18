#!/bin/bash
# A/B of the banded super-tile order of the triangular trailing update (GPX_TRI_BAND), alternating runs on one card
for b in 0 4 0 8 0 2; do
  echo "== band $b"; GPX_TRI_BAND=$b timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), round(d['roofline']['achieved'],2), d['phases_ms']['chol'], d['phases_ms']['trsm'])"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 0 4; do
  O=$R/gpurun_out/band$b; rm -rf $O; mkdir -p $O
  GPX_TRI_BAND=$b GPX_CHAIN_FLAG=0 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench > $O/run.json 2> $O/run.err
  (cd $R && python tools/pmc_summary.py $O/summary.csv $O > /dev/null && grep "gemm_nt_kernel<double, 128, true, 0" $O/summary.csv | cut -c1-40,150-)
  find $O -name "*counter_collection.csv" -delete
done
