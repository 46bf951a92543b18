#!/bin/bash
set -e -o pipefail
timeout -k 10 600 python -m pytest tests/test_fp32_gpu.py "tests/test_gp_parity_gpu.py::test_schedule_variants_give_the_same_factorisation" tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python tools/w8_ab.py 2>/dev/null | tee gpurun_out/r03_w8_variant.json
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['profile_flag'], d['microbench'])"
