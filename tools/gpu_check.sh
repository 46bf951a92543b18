#!/bin/bash
# One gpurun call's worth of checking after a change: the whole -m gpu suite, smoke(), the C3 bench
# line (no CPU baseline) and the C2 latency bench.   gpurun -- 'bash tools/gpu_check.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('C3', round(d['ms_per_step'], 1), 'ms/step', round(d['value']), 'points/s; SYRK', round(d['roofline']['achieved'], 2), 'TF; chol', d['phases_ms']['chol'], 'predict', d['phases_ms']['predict_total'])"
python tools/c2_bench.py 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('C2', round(d['ms_per_step'], 2), 'ms/step', d['phases_ms'])"
