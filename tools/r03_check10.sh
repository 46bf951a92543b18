#!/bin/bash
set -e -o pipefail
timeout -k 10 600 python -m pytest tests/test_mixed_gpu.py "tests/test_full_size_gpu.py::test_c5_mixed_default_refinement_gives_an_fp64_grade_mean" -m gpu -x -q 2>&1 | tail -2
python bench.py --workload C5 --dtype mixed --steps 3 --warmup 1 --no-microbench 2>/dev/null | tee gpurun_out/r03_bench_C5_mixed.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['ms_per_step'],1), d['refinement'], d['phases_ms']['mean'])"
