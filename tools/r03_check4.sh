#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py "tests/test_gp_parity_gpu.py::test_schedule_variants_give_the_same_factorisation" -m gpu -x -q --durations=8 > gpurun_out/r03_t4.log 2>&1 || { tail -60 gpurun_out/r03_t4.log; exit 1; }
tail -14 gpurun_out/r03_t4.log
