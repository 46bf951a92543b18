#!/bin/bash
# round 4: the update kernels read the all-gathered panel in place (no un-permute); replicated copy on the copy stream (A/B)
O=gpurun_out/r04_gperm; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py tests/test_mixed_gpu.py -m gpu -q -x > $O/tests2.log 2>&1; tail -4 $O/tests2.log
for v in 1 0 1 0; do
  GPX_REPL_COPY_SIDE=$v timeout -k 10 400 python tools/shard_ab.py --ranks 4 --reps 2 > $O/shard_ab_side$v.txt 2> $O/shard_ab.err; echo "side=$v"; grep -a "^sharded\|^group" $O/shard_ab_side$v.txt | cut -c1-200
done
