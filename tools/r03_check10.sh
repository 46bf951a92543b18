#!/bin/bash
# split-strip A/B: tests, then C2 / C3 with GPX_SPLIT_STRIP=1 (default) and 0
set -e -o pipefail
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -30
for v in 1 0 1 0; do GPX_SPLIT_STRIP=$v python tools/c2_bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 split=$v', round(d['ms_per_step'],2), d['phases_ms'])"; done
for v in 1 0 1 0; do GPX_SPLIT_STRIP=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 split=$v', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms'])"; done
