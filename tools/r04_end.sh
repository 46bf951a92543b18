#!/bin/bash
# end of round 4: the evidence of tools/r04_final_measurements.sh again on the final code, then the sharded bench lines
bash tools/r04_final_measurements.sh > gpurun_out/r04_end_final.log 2>&1; echo "final rc=$?"; tail -c 400 gpurun_out/r4_final/bench.json; echo
bash tools/r04_shard_lines.sh 2>&1 | tail -12
