# Round 4, first GPU check: new hand-over self-test, the suites its changes touch, bare-launch rehearsal, default bench
# (live full-size CPU baseline).  Full output goes to files under gpurun_out/ (never `| tail`: VERDICT r3 item 8).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_check1; rm -rf $O; mkdir -p $O
set -x
timeout -k 10 600 python -m pytest tests/test_handover_gpu.py tests/test_fit_predict_gpu.py tests/test_mixed_gpu.py tests/test_gp_parity_gpu.py tests/test_delay_gpu.py -m gpu -x -q > $O/pytest1.log 2>&1
echo "pytest1 rc=$?"; tail -5 $O/pytest1.log
timeout -k 10 400 bash tools/bench_rehearsal.sh > $O/rehearsal.log 2>&1
echo "rehearsal rc=$?"; cat $O/rehearsal.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_check1/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['phases_ms'])
c=d['cpu_baseline']; print({k:c[k] for k in ('value','cores','measured_in_this_run','source','seconds','not_measured_live_because')})
P
