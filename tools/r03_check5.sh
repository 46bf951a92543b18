#!/bin/bash
for steps in "A,T" "A,R,A,R" "A,L,R,A,L,R" "L,A,T"; do
  timeout -k 10 200 python tools/rccl_exit_probe.py $steps > gpurun_out/tmp.log 2>&1; echo "steps=$steps rc=$? $(grep -c ok gpurun_out/tmp.log) ok; $(grep -i 'free\|abort\|error' gpurun_out/tmp.log | head -2 | tr '\n' ' ')"
done
run() { echo "== $1"; shift; timeout -k 10 600 python -m pytest "$@" -m gpu -x -q > gpurun_out/tmp.log 2>&1; echo "rc=$? $(tail -2 gpurun_out/tmp.log | tr '\n' ' ')"; }
run "test_group_gpu alone" tests/test_group_gpu.py
run "group + shard" tests/test_group_gpu.py tests/test_shard_gpu.py
