#!/bin/bash
# how much slower is a CU-masked stream for the big trailing update?  all panels on the masked stream (GPX_REST_SPLIT=64)
set -e -o pipefail
for v in "0 64" "1 64" "8 64" "32 64"; do set -- $v; GPX_CU_RESERVE=$1 GPX_REST_SPLIT=$2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 reserve=$1 rest_split=$2', d['ms_per_step'], d['phases_ms']['chol'], d['phases_ms']['chol_syrk'], d['phases_ms']['chol_strip'], d['phases_ms']['chol_diag'])"; done
