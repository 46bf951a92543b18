#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "full suite rc=$?"; tail -18 gpurun_out/r03_pytest_gpu.log
