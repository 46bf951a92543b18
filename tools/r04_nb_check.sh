# auto panel width: correctness suites that depend on it, then the bench with the new default and wider explicit panels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_nb_check; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py tests/test_mixed_gpu.py tests/test_gp_parity_gpu.py tests/test_fit_predict_gpu.py tests/test_delay_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
GPX_NB_WIDE_FROM=4096 timeout -k 10 600 python -m pytest tests/test_gp_parity_gpu.py tests/test_fit_predict_gpu.py tests/test_mixed_gpu.py tests/test_delay_gpu.py tests/test_fp32_gpu.py -m gpu -x -q > $O/pytest_wide_everywhere.log 2>&1; echo "pytest (2048 from N=4096) rc=$?"; tail -4 $O/pytest_wide_everywhere.log
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); p=d['phases_ms']
print('$2', round(d['ms_per_step'],1), 'block', d['config']['block'], 'SYRK', round(d['roofline']['achieved'],2), 'frac', round(d['roofline']['frac'],4), 'launches', d['roofline']['launches'], 'chol', p['chol'], 'predict', p['predict_total'], 'one pass', d['fit_predict_one_pass'] and round(d['fit_predict_one_pass']['ms_per_step'],1))"; }
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/auto.json 2> $O/err.log; line $O/auto.json "auto"
GPX_NB_WIDE_FROM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb1024.json 2>> $O/err.log; line $O/nb1024.json "1024"
GPX_NB_PRED=3072 python bench.py --block 3072 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb3072.json 2>> $O/err.log; line $O/nb3072.json "3072"
GPX_NB_PRED=4096 python bench.py --block 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb4096.json 2>> $O/err.log; line $O/nb4096.json "4096"
python bench.py --workload C5 --dtype mixed --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/c5mixed.json 2>> $O/err.log; line $O/c5mixed.json "C5 mixed auto"
python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/c5f32.json 2>> $O/err.log; line $O/c5f32.json "C5 f32 auto"
python tools/alpha_time.py > $O/alpha_time.txt 2>&1; tail -3 $O/alpha_time.txt
