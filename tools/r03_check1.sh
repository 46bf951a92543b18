#!/bin/bash
# GPU: the whole -m gpu suite, then the C5 precision study and the default bench on the same box
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 840 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03_pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r03_pytest_gpu.log; exit 1; }
tail -25 gpurun_out/r03_pytest_gpu.log
python tools/precision_study.py > gpurun_out/r03_c5_precision_study.json 2> gpurun_out/r03_c5_precision_study.err
python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
tail -c 1500 gpurun_out/r03_bench_a.json
