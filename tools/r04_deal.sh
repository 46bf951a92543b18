#!/bin/bash
# round 4: snake dealing of the row blocks — the shard / group / delay / mixed / fp32 suites under the new default
O=gpurun_out/r04_deal; mkdir -p $O
timeout -k 10 1150 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_mixed_gpu.py tests/test_fp32_gpu.py tests/test_fit_predict_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -15 $O/tests.log
