#!/bin/bash
set -e -o pipefail
timeout -k 10 120 python tools/potf2_128_check.py
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in 1 2 3; do python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['phases_ms']['chol_diag'])"
export C2_ARGS=--no-profile GAPS_ARGS=--main; bash tools/r03_c2trace.sh | grep "step [0-9]:\|diag block\|boundaries\|mean step" | cut -c1-300
