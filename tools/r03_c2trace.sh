#!/bin/bash
set -e
bash tools/c2_trace.sh > /dev/null 2>&1
python tools/c2_gaps.py gpurun_out/c2trace/trace.csv $GAPS_ARGS | tee gpurun_out/r03_c2_chain_gaps.txt
rm -rf gpurun_out/c2trace/*/ gpurun_out/c2trace/trace.csv
