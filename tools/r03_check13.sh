#!/bin/bash
# CU-mask experiment: where the mask bits land, then C2 / C3 with k CUs left to the look-ahead chain
set -e -o pipefail
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/cumask_probe tools/cumask_probe.hip
timeout -k 10 60 /tmp/cumask_probe 8 > gpurun_out/r03_cumask_probe.txt
timeout -k 10 60 /tmp/cumask_probe 16 >> gpurun_out/r03_cumask_probe.txt
grep -c . gpurun_out/r03_cumask_probe.txt
GPX_CU_RESERVE=8 timeout -k 10 600 python -m pytest tests/test_gp_parity_gpu.py -m gpu -x -q -k "schedule or C2 or n32768" 2>&1 | tail -3
for v in 0 8 16 32 0 8 16; do GPX_CU_RESERVE=$v python tools/c2_bench.py --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 reserve=$v', round(d['ms_per_step'],2), d['phases_ms']['chol'], d['phases_ms']['predict_total'])"; done
for v in 0 8 16 0 8; do GPX_CU_RESERVE=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 reserve=$v', d['ms_per_step'], d['roofline']['achieved'], d['phases_ms']['chol'], d['phases_ms']['chol_syrk'])"; done
export C2_ARGS=--no-profile GAPS_ARGS=--main GPX_CU_RESERVE=8; bash tools/r03_c2trace.sh | tail -12
