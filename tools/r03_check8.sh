#!/bin/bash
set -e -o pipefail
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '|', d['dtype'], '|', round(d['ms_per_step'],2), 'ms', round(d['value']), 'pts/s | roofline', round(d['roofline']['achieved'],1), '/', d['roofline']['peak'], '=', round(d['roofline']['frac'],3), '|', d.get('refinement'), '| cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
python bench.py --workload C2 --steps 20 --warmup 3 --no-microbench 2>/dev/null | tee gpurun_out/r03_bench_C2.json | show
python bench.py --workload C5 --steps 3 --warmup 1 --no-microbench 2>/dev/null | tee gpurun_out/r03_bench_C5_f32.json | show
python bench.py --workload C5 --dtype mixed --steps 3 --warmup 1 --no-microbench 2>/dev/null | tee gpurun_out/r03_bench_C5_mixed.json | show
python bench.py --steps 3 --warmup 1 2>/dev/null | show
