#!/bin/bash
set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "full suite rc=$?"; tail -12 gpurun_out/r03_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
