#!/bin/bash
set -e -o pipefail
for v in "0 0" "8 0" "16 0" "32 0" "64 0" "16 1" "32 1" "0 0"; do set -- $v; GPX_CU_RESERVE=$1 GPX_POTF2_EXCL=$2 python tools/c2_bench.py --no-profile --fused 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 fused reserve=$1 excl=$2', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
