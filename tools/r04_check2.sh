# Round 4, second GPU check: the sharded factorisation's split schedule — group / shard / delay suites, the C4 real-shape
# rehearsal, and the A/B timing on one card.  Full logs under gpurun_out/r04_check2/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_check2; rm -rf $O; mkdir -p $O
set -x
timeout -k 10 900 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_delay_gpu.py -m gpu -x -q -s --durations=8 > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest.log
[ $rc -eq 0 ] && timeout -k 10 600 python tools/shard_ab.py --ranks 4,8 > $O/shard_ab.json 2> $O/shard_ab.err
echo "shard_ab rc=$?"; cat $O/shard_ab.json; tail -5 $O/shard_ab.err
