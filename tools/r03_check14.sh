#!/bin/bash
set -e -o pipefail
for v in "0 0" "8 1" "16 1" "8 0" "0 0" "8 1"; do set -- $v; GPX_CU_RESERVE=$1 GPX_POTF2_EXCL=$2 python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 reserve=$1 excl=$2', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
export C2_ARGS=--no-profile GAPS_ARGS=--main GPX_CU_RESERVE=8 GPX_POTF2_EXCL=1; bash tools/r03_c2trace.sh | grep "step [0-9]:\|diag block\|boundaries" | cut -c1-300
