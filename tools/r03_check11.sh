#!/bin/bash
# early hand-over of the REST's first column block: tests, then C2 / C3 A/B over GPX_REST_SPLIT
set -e -o pipefail
timeout -k 10 1000 python -m pytest tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py -m gpu -x -q 2>&1 | tail -30
for v in 16 0 16 0; do GPX_REST_SPLIT=$v python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 rest_split=$v', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
for v in 16 0 64 16 0 64; do GPX_REST_SPLIT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 rest_split=$v', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['phases_ms']['chol_syrk'])"; done
export C2_ARGS=--no-profile; bash tools/r03_c2trace.sh | tail -40
