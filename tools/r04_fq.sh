#!/bin/bash
# round 4: the sharded one-pass — tests (group, host transport, delays), then tools/shard_one_pass.py at C3
O=gpurun_out/r04_fq; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_fit_predict_gpu.py -m gpu -q -k "one_pass or fit_predict" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 600 python tools/shard_one_pass.py --ranks 4,8 > $O/shard_one_pass.txt 2> $O/shard_one_pass.err; tail -8 $O/shard_one_pass.txt; tail -3 $O/shard_one_pass.err
