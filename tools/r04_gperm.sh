#!/bin/bash
# round 4: the update kernels read the all-gathered panel in place (no un-permute): shard / group suites, then the A/B tool
O=gpurun_out/r04_gperm; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fit_predict_gpu.py tests/test_kernels_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 500 python tools/shard_ab.py --ranks 4 > $O/shard_ab.txt 2> $O/shard_ab.err; grep -a "^unsharded\|^sharded\|^group" $O/shard_ab.txt | cut -c1-420
