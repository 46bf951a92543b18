#!/bin/bash
# round 4: the sharded one-pass — tests (group incl. C4's real shape, host transport, delays), then tools/shard_one_pass.py at C3
O=gpurun_out/r04_fq; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_fit_predict_gpu.py -m gpu -q -s -k "one_pass or fit_predict or c4_real" > $O/tests.log 2>&1; grep -a "C4 real shape\|passed\|failed" $O/tests.log | tail -5
timeout -k 10 600 python tools/shard_one_pass.py --ranks 4,8 > $O/shard_one_pass.txt 2> $O/shard_one_pass.err; grep -a "^{\|ms" $O/shard_one_pass.txt | cut -c1-330 | tail -8; tail -3 $O/shard_one_pass.err
