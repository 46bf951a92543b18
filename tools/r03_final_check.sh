#!/bin/bash
# Last check of the round on the final commit: the whole -m gpu suite, smoke(), a fuzz sweep, the default bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "full suite rc=$?"; tail -9 gpurun_out/r03_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python tools/fuzz_parity.py --cases 300 --seed 32 > gpurun_out/r03_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r03_fuzz.log | cut -c1-600
python bench.py > gpurun_out/r03_final_bench_head.json 2> gpurun_out/r03_final_bench_head.err; python -c "
import json; d=json.load(open('gpurun_out/r03_final_bench_head.json')); print('bench', round(d['ms_per_step'],1), 'ms', round(d['value']), 'pts/s SYRK', round(d['roofline']['achieved'],2), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'], 'cpu', d['cpu_baseline']['value'], 'one pass', d['fit_predict_one_pass'])"
python bench.py --workload C2 --no-cpu-baseline --no-microbench > gpurun_out/r03_bench_C2.json 2> gpurun_out/r03_bench_C2.err
python bench.py --workload C5 --dtype float32 --no-cpu-baseline --no-microbench > gpurun_out/r03_bench_C5_f32.json 2> gpurun_out/r03_bench_C5_f32.err
python bench.py --workload C5 --dtype mixed --no-cpu-baseline --no-microbench > gpurun_out/r03_bench_C5_mixed.json 2> gpurun_out/r03_bench_C5_mixed.err
python -c "
import json
for f in ('C2','C5_f32','C5_mixed'):
    d=json.load(open('gpurun_out/r03_bench_%s.json'%f)); print(f, round(d['ms_per_step'],2), 'ms', round(d['value']), 'pts/s', 'SYRK', round(d['roofline']['achieved'],1), d['roofline']['frac'], 'one pass', (d.get('fit_predict_one_pass') or {}).get('ms_per_step'))"
