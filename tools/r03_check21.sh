#!/bin/bash
set -e -o pipefail
timeout -k 10 900 python -m pytest tests/test_gp_parity_gpu.py tests/test_delay_gpu.py tests/test_fit_predict_gpu.py tests/test_fuzz_gpu.py tests/test_mixed_gpu.py -m gpu -x -q 2>&1 | tail -4
for v in 1 0 1 0; do GPX_SOLVE_TOP=$v python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 solve_top=$v', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
for v in 1 0; do GPX_SOLVE_TOP=$v python tools/c2_bench.py --no-profile --fused 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 one pass solve_top=$v', round(d['ms_per_step'],2), d['phases_ms']['chol'])"; done
for v in 1 0 1 0; do GPX_SOLVE_TOP=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 solve_top=$v', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['fit_predict_one_pass']['ms_per_step'])"; done
