#!/bin/bash
# Full-size oracle fixtures (G5 = C3, G6 = C5) on the GPU box's host cores + the r03 baseline numbers
# of the code as it stood at the start of the round.  Outputs under gpurun_out/.
set -e
mkdir -p gpurun_out
python oracle/make_golden_full.py --config C3 --out gpurun_out/G5.npz > gpurun_out/r03_cpu_full_oracle_c3.json 2> gpurun_out/g5.log
python oracle/make_golden_full.py --config C5 --out gpurun_out/G6.npz > gpurun_out/r03_cpu_full_oracle_c5.json 2> gpurun_out/g6.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_start_bench.json 2> gpurun_out/r03_start_bench.err
python tools/c2_bench.py > gpurun_out/r03_start_c2.json 2>&1
tail -c 600 gpurun_out/r03_cpu_full_oracle_c3.json gpurun_out/r03_cpu_full_oracle_c5.json gpurun_out/r03_start_c2.json
