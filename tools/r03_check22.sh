#!/bin/bash
set -e -o pipefail
for v in 0 1 2 4 0 2; do GPX_START_SKEW=$v python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 skew=$v', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
for v in 0 2 4 0 2; do GPX_START_SKEW=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 skew=$v', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
