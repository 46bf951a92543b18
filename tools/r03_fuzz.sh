#!/bin/bash
mkdir -p gpurun_out
for seed in 11 12; do
  timeout -k 10 520 python tools/fuzz_parity.py --cases 220 --seed $seed > gpurun_out/r03_fuzz_seed$seed.log 2>&1; echo "seed $seed rc=$?"; tail -1 gpurun_out/r03_fuzz_seed$seed.log
done
