#!/bin/bash
# after a change of the POTF2 kernels: exactness, phase stamps, the kernel / parity / delay tests, the C2 chain
set -e -o pipefail
timeout -k 10 120 python tools/potf2_128_check.py
timeout -k 10 120 python tools/potf2_stamps.py | tail -14
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_delay_gpu.py tests/test_fit_predict_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab_sweep.sh GPX_NONE "0 0 0" c2
bash tools/ab_sweep.sh GPX_NONE "0" c2onepass
bash tools/ab_sweep.sh GPX_NONE "0" c3
export C2_ARGS=--no-profile GAPS_ARGS=--main; bash tools/r03_c2trace.sh | grep "step [0-9]:\|mean step\|boundaries" | cut -c1-260
