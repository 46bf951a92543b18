#!/bin/bash
# A/B of one library switch on the GPU box, alternating processes on one card (how profiles/r03_schedule_ab.txt and
# r03_cu_reserve_experiments.txt were measured):
#     tools/ab_sweep.sh GPX_SPLIT_STRIP "1 0 1 0" c2          # C2 (tools/c2_bench.py --no-profile)
#     tools/ab_sweep.sh GPX_REST_SPLIT "16 0 16 0" c3         # C3 (bench.py --steps 3 --warmup 1)
#     tools/ab_sweep.sh GPX_CU_RESERVE "0 8 32" c2onepass     # C2, gp.fit_predict
set -e -o pipefail
VAR=$1; VALS=$2; WHAT=${3:-c2}
for v in $VALS; do
  case $WHAT in
    c2) env $VAR=$v python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 $VAR=$v', round(d['ms_per_step'],2), 'chol', d['phases_ms']['chol'], 'predict', d['phases_ms']['predict_total'])";;
    c2onepass) env $VAR=$v python tools/c2_bench.py --no-profile --fused 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 one pass $VAR=$v', round(d['ms_per_step'],2), 'chol', d['phases_ms']['chol'])";;
    c3) env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 $VAR=$v', round(d['ms_per_step'],1), 'SYRK', round(d['roofline']['achieved'],2), 'chol', d['phases_ms']['chol'], 'one pass', round(d['fit_predict_one_pass']['ms_per_step'],1))";;
  esac
done
