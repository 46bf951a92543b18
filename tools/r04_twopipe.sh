#!/bin/bash
# round 4 end: the owner's diagonal chain on a stream of its own beside the previous panel's all-gather (two pipelines)
O=gpurun_out/r04_twopipe; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_mixed_gpu.py tests/test_fp32_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in 1 0 1 0; do
  GPX_SHARD_TWO_PIPE=$v timeout -k 10 400 python tools/shard_ab.py --ranks 4 --reps 2 > $O/ab_$v.txt 2> $O/ab.err; echo "two_pipe=$v"; grep -a "^sharded\|^group" $O/ab_$v.txt | cut -c1-150
done
