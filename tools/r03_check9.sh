#!/bin/bash
set -e -o pipefail
timeout -k 10 120 python tools/potf2_128_check.py
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -40
for i in 1 2 3; do python tools/c2_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['phases_ms'])"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol_diag'])"
bash tools/r03_c2trace.sh | tail -11
