#!/bin/bash
set -e -o pipefail
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py tests/test_group_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | tail -5
for v in 1 2 3; do python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
for v in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['phases_ms']['chol_diag'])"; done
