#!/bin/bash
# One gpurun call's worth of checking after a change: the whole -m gpu suite, smoke(), the C3 bench
# line (no CPU baseline) and the C2 latency bench.   gpurun -- 'bash tools/gpu_check.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
# progress goes to a file under gpurun_out/ (a silent pipe into tail looks hung to gpurun after 7 minutes)
timeout -k 10 900 python -m pytest tests -x -v -m gpu --timeout 300 --timeout-method=thread > gpurun_out/gpu_suite.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_suite.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('C3', round(d['ms_per_step'], 1), 'ms/step', round(d['value']), 'points/s; SYRK', round(d['roofline']['achieved'], 2), 'TF; chol', d['phases_ms']['chol'], 'predict', d['phases_ms']['predict_total'])"
python tools/c2_bench.py 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('C2', round(d['ms_per_step'], 2), 'ms/step', d['phases_ms'])"
