#!/bin/bash
set -e -o pipefail
for v in 1 2 3; do python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
python tools/c2_bench.py --no-profile --fused 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 one pass', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['fit_predict_one_pass']['ms_per_step'])"
timeout -k 10 600 python -m pytest tests/test_delay_gpu.py tests/test_fit_predict_gpu.py -m gpu -x -q 2>&1 | tail -2
